#!/bin/bash
# Variants of the bf16 and fp16 + e4m3 GEMM translation units for same-box A/B runs: tools/build_gemm_variants.sh name "flags" [name "flags" ...]
# -> ab/libosud_<name>.so (everything but gemm_bf16.o / gemm_h8.o is taken from the regular build; run `make -C osu_diffusion_amd/csrc` first)
set -e
root=$(cd "$(dirname "$0")/.." && pwd); cs=$root/osu_diffusion_amd/csrc
mkdir -p $root/ab
build_one() {
  name=$1; flags=$2; b=$cs/build_v_$name
  rm -rf $b && mkdir -p $b && cp $cs/build/*.o $b/ && rm -f $b/gemm_bf16.o $b/gemm_h8.o $b/dit.o
  for tu in gemm_bf16 gemm_h8 dit; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-gpu-rdc $flags -mllvm -amdgpu-atomic-optimizer-strategy=None \
       -c $cs/$tu.hip -o $b/$tu.o 2> $b/log_$tu.txt || { tail $b/log_$tu.txt; exit 1; }
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/ab/libosud_$name.so $b/*.o -ldl
  echo built ab/libosud_$name.so "($flags)"
}
while [ $# -ge 2 ]; do build_one "$1" "$2" & shift 2; if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi; done
wait
