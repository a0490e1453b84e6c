#!/bin/bash
# Sample the GPU's shader clock and socket power (rocm-smi, read-only) while a command runs: evidence for the sustained clock the
# MFMA-heavy kernels actually get (tools/clock_watch.sh out.log -- python bench.py --mode train --steps 400 ...).
out=$1; shift; shift
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" ; sleep 0.25; done ) > "$out" 2>&1 &
wpid=$!
"$@"
rc=$?
kill $wpid 2>/dev/null
exit $rc
