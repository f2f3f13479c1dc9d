python -m pytest tests/test_gpu_split.py -q 2>&1 | tail -2
python tools/split_net_check.py det 32 2>&1 | grep -A60 "variant -1" | grep "k1 \|variant"
python bench.py --steps 8 --warmup 3 --cpu-frames 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['stage_ms'])"
