#!/bin/bash
# Everything the round's profiles/ files are made from, in one GPU call:  gpurun -- 'bash tools/collect_profiles.sh r5p'
# (kernel traces of the bench workloads, the MFMA-busy counter pass, the HBM traffic passes of the two headline kernels, cycle stamps)
O=gpurun_out/${1:-prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-family-table --no-roofline"
OSUD_OPTIONS=wgrad_side_stream=0 rocprofv3 --kernel-trace -d /tmp/tr1 -o t -- python3 bench.py --mode train $B --no-compute-floor --steps 50 --warmup 10 > $O/train_one.log 2>&1
python tools/rocpd_summary.py $(find /tmp/tr1 -name "*.db" | head -1) > $O/train_trace_one_stream.md
rocprofv3 --kernel-trace -d /tmp/tr2 -o t -- python3 bench.py --mode train $B --no-compute-floor --steps 50 --warmup 10 > $O/train_two.log 2>&1
python tools/rocpd_summary.py $(find /tmp/tr2 -name "*.db" | head -1) > $O/train_trace_two_streams.md
rocprofv3 --kernel-trace -d /tmp/ts1 -o t -- python3 bench.py --mode sample $B --no-parity-tier --steps 100 --warmup 10 > $O/sample_bf16.log 2>&1
python tools/rocpd_summary.py $(find /tmp/ts1 -name "*.db" | head -1) > $O/sample_trace_bf16.md
rocprofv3 --kernel-trace -d /tmp/ts2 -o t -- python3 bench.py --mode sample --precision fp16f8 $B --no-parity-tier --steps 100 --warmup 10 > $O/sample_h8.log 2>&1
python tools/rocpd_summary.py $(find /tmp/ts2 -name "*.db" | head -1) > $O/sample_trace_h8.md
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o run -- python3 tools/pmc_step.py 2 4 > $O/pmc_mfma.log 2>&1
python tools/pmc_summary.py /tmp/pmc_mfma > $O/pmc_mfma.json
for t in fc1_only wgrad_only; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$t -o run -- python3 tools/$t.py > $O/pmc_f_$t.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$t -o run -- python3 tools/$t.py > $O/pmc_w_$t.log 2>&1
  python tools/pmc_summary.py /tmp/pmc_f_$t /tmp/pmc_w_$t > $O/pmc_traffic_$t.json
done
if [ -f ab/libosud_tm.so ]; then
  (echo "== slab loop, kernel totals"; LOOP=0 OSUD_LIB=ab/libosud_tm.so python tools/gemm_phase_stamps.py; echo "== phased loop, kernel totals (main loop unperturbed)"; OSUD_LIB=ab/libosud_tm.so python tools/gemm_phase_stamps.py
   echo "== slab loop, zero operands"; ZERO=1 LOOP=0 OSUD_LIB=ab/libosud_tm.so python tools/gemm_phase_stamps.py; echo "== phased loop, zero operands"; ZERO=1 OSUD_LIB=ab/libosud_tm.so python tools/gemm_phase_stamps.py
   echo "== phased loop, per-segment stamps"; OSUD_LIB=ab/libosud_tm2.so python tools/gemm_phase_stamps.py) 2>&1 | grep -v amdgpu.ids > $O/stamps.log
fi
ls -la $O
# round 6: DiT-XL traces (one stream) and the multi-GPU schedule's compute floor set-ups
for p in bf16 fp8; do
  OSUD_OPTIONS=wgrad_side_stream=0 rocprofv3 --kernel-trace -d /tmp/trx_$p -o t -- python3 bench.py --mode xl --xl-precision $p --steps 12 --warmup 4 > $O/xl_$p.log 2>&1
  python tools/rocpd_summary.py $(find /tmp/trx_$p -name "*.db" | head -1) > $O/xl_trace_$p.md
done
python3 -u tools/floor_probe.py --steps 30 --rounds 2 2>&1 | grep round > $O/floor_probe.txt
python3 -u tools/interference_probe.py 0 8 16 32 2>&1 | grep -E "static|queued" > $O/interference.txt
ls -la $O
