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
