cd /root/repo
python -m pytest tests/test_gpu_vit.py tests/test_gpu_vit_topdown.py tests/test_gpu_fullsize.py -q 2>&1 | tail -8
python bench.py --workload c5 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/c5.json; cat gpurun_out/c5.json | cut -c1-600
POSEPIPE_GEMM_CFG=2 python bench.py --workload c5 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/c5_cfg2.json; cat gpurun_out/c5_cfg2.json | cut -c1-300
python bench.py --workload cascade5 --cpu-frames 0 2>&1 | tail -1 > gpurun_out/cascade5.json; cat gpurun_out/cascade5.json | cut -c1-300
