mkdir -p gpurun_out/r4final
python -m pytest tests -m gpu -q > gpurun_out/r4final/gpu1.log 2>&1; grep -E "passed|failed" gpurun_out/r4final/gpu1.log
python -m pytest tests -m gpu -q -x > gpurun_out/r4final/gpu2.log 2>&1; grep -E "passed|failed" gpurun_out/r4final/gpu2.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4final/bench_driver_args.json 2> gpurun_out/r4final/bench_driver_args.err
head -c 400 gpurun_out/r4final/bench_driver_args.json
