for rep in 1 2; do for v in 0 1; do
r=$(OSUD_ADAMW_OVERLAP=$v python bench.py --mode train --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
echo "overlap=$v rep$rep ms_per_step=$r"; done; done
python -m pytest tests/test_gpu_train.py -m gpu -q -x 2>&1 | tail -3
