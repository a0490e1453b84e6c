mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_w8.py tests/test_gpu_forward.py -q --timeout 900 2>&1 | tail -3
python bench.py --mode sample --precision fp16w8 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r4f/sample_w8.json 2> gpurun_out/r4f/sample_w8.err
python bench.py --mode sample --precision fp16f8 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r4f/sample_h8.json 2> gpurun_out/r4f/sample_h8.err
python bench.py --mode sample --precision bf16 --steps 200 --warmup 20 --no-cpu-baseline --no-roofline > gpurun_out/r4f/sample_bf16.json 2> gpurun_out/r4f/sample_bf16.err
python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r4f/train.json 2> gpurun_out/r4f/train.err
for f in sample_w8 sample_h8 sample_bf16 train; do head -c 330 gpurun_out/r4f/$f.json | cut -c 60-330; echo; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tr2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode sample --precision fp16w8 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/r4f/sample_trace.log 2>&1
OSUD_OPTIONS=wgrad_side_stream=0 rocprofv3 --kernel-trace -d /tmp/tr1 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --no-cpu-baseline --no-family-table --no-roofline --steps 10 --warmup 3 > $GRAFT_REPO_ROOT/gpurun_out/r4f/bench_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_summary.py $(find /tmp/tr2 -name "*.db" | head -1) > gpurun_out/r4f/sample_w8_trace.md
python tools/rocpd_summary.py $(find /tmp/tr1 -name "*.db" | head -1) > gpurun_out/r4f/train_trace_single_stream.md
grep -E "row_reduce|cond_bwd" gpurun_out/r4f/train_trace_single_stream.md
head -14 gpurun_out/r4f/sample_w8_trace.md
