python -m pytest tests/test_gpu_split.py -q 2>&1 | tail -3
python tools/split_net_check.py w48 64 2>&1 | grep -A6 "variant -1\|output"
POSEPIPE_SPLIT_SMALL_GRID=0 python tools/split_net_check.py w48 64 2>&1 | grep -A6 "variant -1"
python tools/split_net_check.py det 16 2>&1 | grep -A6 "variant -1"
python bench.py --steps 8 --warmup 3 --cpu-frames 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['stage_ms'])"
POSEPIPE_SPLIT_SMALL_GRID=0 python bench.py --steps 8 --warmup 3 --cpu-frames 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['stage_ms'])"
