unset POSEPIPE_LIB
timeout 600 python -m pytest tests/test_gpu_split.py tests/test_gpu_conv.py -q -x 2>&1 | tail -3
timeout 300 python tools/split_net_check.py det 32 2>&1 | grep -A40 "variant -1" | grep "k3 s1\|variant"
timeout 600 python bench.py --steps 10 --warmup 3 --cpu-frames 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['stage_ms'])"
