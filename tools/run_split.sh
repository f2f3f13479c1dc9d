# bf16-split convolution kernels on a GPU box: accuracy against float64, the float32-reordering control, per-layer tables of the
# detector / HRNet-W48 programs (default vs bit-exact kernels), bench line
python tools/split_check.py
python -m pytest tests/test_gpu_split.py -q -s -k reordering | grep "heat-maps\|joints\|passed\|failed"
python tools/split_net_check.py det 32
python tools/split_net_check.py w48 64
python bench.py --steps 8 --warmup 3 --cpu-frames 0
