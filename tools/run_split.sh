python -m pytest tests/test_gpu_split.py -q 2>&1 | tail -2
for c in 1000000 64 256; do echo MIN_C4 $c; POSEPIPE_SPLIT_GEMM4_MIN_C=$c python tools/split_net_check.py det 32 2>&1 | grep -A60 "variant -1" | grep "k1 \|variant\|  p2"; done
