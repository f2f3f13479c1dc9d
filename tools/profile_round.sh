set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_cascade.json 2> $O/bench_cascade.err
python $R/bench.py --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err
cd $R
POSEPIPE_NET_LANES=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python bench.py --steps 5 --warmup 1 --cpu-frames 0 > $O/serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes4 -- python bench.py --steps 5 --warmup 1 --cpu-frames 0 > $O/lanes4.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2 -- python bench.py --workload c2 --steps 5 --warmup 1 --cpu-frames 0 > $O/c2.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 375 "cascade chunk 32, 1 person" > $O/pmc_summary.txt 2>&1
# keep only the small files
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O; ls -R $O | head -40; cat $O/pmc_summary.txt; tail -2 $O/bench_cascade.json | cut -c1-200
# ---- ViTPose-H (configs[4]): bench lines, kernel stats and GEMM counters -> gpurun_out/prof/c5_*, gpurun_out/pmc_c5gemm/
python $R/bench.py --workload c5 > $O/bench_c5.json 2> $O/bench_c5.err
python $R/bench.py --workload cascade5 > $O/bench_cascade5.json 2> $O/bench_cascade5.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5 -- python bench.py --workload c5 --steps 5 --warmup 1 --cpu-frames 0 > $O/c5.log 2>&1
cp $(ls -t $(find $O/c5 -name "*kernel_stats.csv") | head -1) $O/c5_kernel_stats.csv
find $O/c5 -name "*.csv" -delete
bash tools/pmc_kernel.sh "gemm_bf16_kernel<4, 4, 4, 4, 2, 64>" c5gemm python bench.py --workload c5 --steps 2 --warmup 1 --cpu-frames 0 | tail -12
