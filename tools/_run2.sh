python -m pytest tests/test_e2e_gpu.py tests/test_train_gpu.py -q -x -m gpu 2>&1 | tail -2
for v in 0 1 0 1; do SF_FUSE_TIME=$v python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_time=$v', d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
