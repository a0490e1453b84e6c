#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs (one directory per pass, any number of passes) per kernel:
    python tools/pmc_summary.py gpurun_out/pmc_mfma [more dirs ...] > profiles/rNN_pmc_mfma.json
Derived: on this chip rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (value / duration = 8 x ~2.1 GHz) and
SQ_VALU_MFMA_BUSY_CYCLES summed over all 1024 SIMDs (it equals 32 cycles x the number of v_mfma_f32_32x32x16_bf16 the kernel
issues: fc1 at M = 32768 reads 150 994 944 = 154.6 GFLOP / 1024 FLOP per SIMD-cycle, exactly).  So
  effective clock   = GRBM_GUI_ACTIVE / 8 / kernel duration
  MFMA busy         = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)        (fraction of SIMD-cycles the matrix pipe works)
  fraction of peak  = MFMA busy x effective clock / 2.4 GHz                                 (2.5 PFLOP/s is quoted at 2.4 GHz)
FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 B, MI355X_MICROARCH.md).  Counter passes serialise the kernels, so the
clock under the profiler (idle gaps between kernels) is higher than in a back-to-back run."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"osud::", "", name).replace("unsigned short", "bf16")
    name = re.sub(r"\(.*$", "", name)
    return name[:100]


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))   # kernel -> counter -> values
    dur = defaultdict(list)
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for row in csv.DictReader(open(path)):
                k = short(row["Kernel_Name"])
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                key = (path, row.get("Dispatch_Id"))
                if key not in seen and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    seen.add(key)
                    dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    out = {}
    for k, counters in acc.items():
        n = max(len(v) for v in counters.values())
        rec = {"dispatches": n, "avg_us_under_profiler": round(sum(dur[k]) / max(1, len(dur[k])), 2)}
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        rec.update({c: round(m, 1) for c, m in mean.items()})
        if "SQ_VALU_MFMA_BUSY_CYCLES" in mean and "GRBM_GUI_ACTIVE" in mean and mean["GRBM_GUI_ACTIVE"] > 0:
            rec["mfma_busy_frac"] = round(mean["SQ_VALU_MFMA_BUSY_CYCLES"] / (mean["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
        if "GRBM_GUI_ACTIVE" in mean and dur[k]:
            rec["effective_clock_GHz"] = round(mean["GRBM_GUI_ACTIVE"] / 8 / (sum(dur[k]) / len(dur[k])) / 1e3, 3)
            if "mfma_busy_frac" in rec:
                rec["frac_of_2.5PF_peak"] = round(rec["mfma_busy_frac"] * rec["effective_clock_GHz"] / 2.4, 4)
        if "FETCH_SIZE" in mean:
            rec["hbm_read_bytes"] = round(mean["FETCH_SIZE"] * 1024 * 2)
        if "WRITE_SIZE" in mean:
            rec["hbm_write_bytes"] = round(mean["WRITE_SIZE"] * 1024)
        out[k] = rec
    order = sorted(out, key=lambda k: -out[k]["avg_us_under_profiler"] * out[k]["dispatches"])
    json.dump({"_about": __doc__.strip(), "kernels": {k: out[k] for k in order}}, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1:])
