cd /root/repo
python -m pytest tests/test_gpu_nv12.py tests/test_gpu_cascade.py tests/test_abi.py -q 2>&1 | tail -15
python bench.py --steps 5 --warmup 2 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/bench_nv12.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_nv12.json')); print(d['value'], d['pcie_inclusive'])
PY
