R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/prof; cd $R; export TMPDIR=/tmp
mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --steps 2 --warmup 1 --cpu-frames 0 > $O/pmc_write.log 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 264 "cascade chunk 64, 1 person" conv_split > $O/pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 109 "cascade chunk 64, 1 person" conv_p3_kernel > $O/pmc_summary_fp32.txt 2>&1
python tools/pmc_traffic_update.py cascade_chunk64_persons1 $O/pmc_summary.txt "tools/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over python bench.py --steps 2 --warmup 1 (FETCH_SIZE x2, gfx950); all conv_split_* launches" $O/pmc_traffic.json
rm -rf $O/pmc_fetch $O/pmc_write; cat $O/pmc_summary.txt $O/pmc_summary_fp32.txt
