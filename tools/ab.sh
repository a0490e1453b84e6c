#!/bin/bash
# same-box A/B of the bench: tools/ab.sh <mode> <steps> <warmup> libA.so libB.so ...   (alternates twice)
mode=$1; steps=$2; warm=$3; shift 3
for rep in 1 2; do
  for lib in "$@"; do
    v=$(OSUD_LIB=$lib python bench.py --mode $mode --steps $steps --warmup $warm --no-cpu-baseline --no-family-table --no-parity-tier --no-xl 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$mode rep$rep $lib ms_per_step=$v"
  done
done
