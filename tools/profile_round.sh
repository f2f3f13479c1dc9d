# Regenerates the evidence under profiles/ (run on a GPU box: bash tools/profile_round.sh; results in gpurun_out/prof/, copy the
# small files to profiles/rNN_*).  Counters are collected in their own rocprofv3 passes (--kernel-trace only).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_cascade.json 2> $O/bench_cascade.err
python $R/bench.py --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err
python $R/bench.py --mode shard --steps 6 --warmup 2 > $O/bench_shard.json 2> $O/bench_shard.err
python $R/bench.py --persons 4 --steps 6 --warmup 2 --cpu-frames 0 > $O/bench_cascade_p4.json 2> $O/bench_cascade_p4.err
python $R/bench.py --persons 8 --steps 4 --warmup 1 --cpu-frames 0 --no-secondary > $O/bench_cascade_p8.json 2> $O/bench_cascade_p8.err
cd $R
# the roofline evidence: ONE run under rocprofv3 with every launch on one stream and no detector look-ahead (--profile-serial), so that
# AverageNs of conv_split_* x launches_per_step reproduces the line's ms_per_step_serial (tools/roofline_check.py prints both)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python bench.py --profile-serial --steps 5 --warmup 1 > $O/serial.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lanes4 -- python bench.py --steps 5 --warmup 1 --cpu-frames 0 > $O/lanes4.log 2>&1
cp $(ls -t $(find $O/serial -name "*kernel_stats.csv") | head -1) $O/cascade_serial_kernel_stats.csv
cp $(ls -t $(find $O/lanes4 -name "*kernel_stats.csv") | head -1) $O/cascade_lanes_kernel_stats.csv
# which kernels co-run (detector look-ahead + lanes): interval arithmetic over the kernel trace of the default run
# (--light runs: the timed cascade only, so that the trace's second half is the timed steps of ONE cascade)
for m in 1 3 0; do
    POSEPIPE_OVERLAP_DETECTOR=$m timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/ov$m -- python bench.py --steps 8 --warmup 2 --cpu-frames 0 --light > $O/ov$m.log 2>&1
    python tools/overlap_from_trace.py $(ls -t $(find $O/ov$m -name "*kernel_trace.csv") | head -1) > $O/overlap_lookahead_mode$m.txt 2>&1
    rm -rf $O/ov$m
done
# the same line with the look-ahead off / on the cascade's own stream, for the record beside bench_cascade.json (same box)
for m in 0 3 1; do POSEPIPE_OVERLAP_DETECTOR=$m python bench.py --light --cpu-frames 0 --steps 8 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lookahead', d['config']['detector_lookahead'], round(d['value'],1), 'frames/s', round(d['ms_per_step'],2), 'ms')"; done > $O/lookahead_modes.txt 2>&1
bash tools/pmc_clock.sh det python tools/profile_net.py det 64 > /dev/null 2>&1; cp gpurun_out/pmc_clock_det/summary.txt $O/clock.txt
grep '^{' $O/serial.log | tail -1 > $O/bench_cascade_profile_serial.json
python tools/roofline_check.py $O/cascade_serial_kernel_stats.csv $O/bench_cascade_profile_serial.json > $O/roofline_check.txt 2>&1; cat $O/roofline_check.txt
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python bench.py --profile-serial --steps 2 --warmup 1 > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python bench.py --profile-serial --steps 2 --warmup 1 > $O/pmc_write.log 2>&1
NS=$(python -c "import json;r=json.load(open('$O/bench_cascade_profile_serial.json'))['roofline'];print(r['launches_per_step'], r['fp32_mfma_kernels']['launches_per_step'])")
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write ${NS% *} "cascade chunk 64, 1 person (--profile-serial)" conv_split > $O/pmc_summary.txt 2>&1
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write ${NS#* } "cascade chunk 64, 1 person (--profile-serial)" conv_p3_kernel > $O/pmc_summary_fp32.txt 2>&1
python tools/pmc_traffic_update.py cascade_chunk64_persons1 $O/pmc_summary.txt "tools/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over python bench.py --profile-serial --steps 2 --warmup 1 (FETCH_SIZE x2, gfx950); all conv_split_* launches" $O/pmc_traffic.json
bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmc_sq/summary.txt $O/cascade_pmc_sq.txt
bash tools/pmc_kernel.sh roi_align_sep_kernel roi python bench.py --profile-serial --steps 1 --warmup 1 > $O/roi_pmc.log 2>&1; cp gpurun_out/pmc_roi/summary.txt $O/roi_pmc.txt
for w in "w48 128" "det 64" "w32 128"; do set -- $w; python tools/profile_net.py $1 $2 > $O/per_op_$1_b$2.txt 2>&1; POSEPIPE_CONV_EXACT=1 python tools/profile_net.py $1 $2 > $O/per_op_$1_b$2_exact.txt 2>&1; done
# the HBM-bound kernels: duration + FETCH / WRITE counters, one shape per run
DECODE_ONE=64 bash tools/hbm_kernel_profile.sh flip_merge_decode decode_n64 60162048 python tools/hbm_kernels_bench.py decode > /dev/null 2>&1; cp gpurun_out/hbm_decode_n64.txt $O/hbm_decode_n64.txt
DECODE_ONE=256 bash tools/hbm_kernel_profile.sh flip_merge_decode decode_n256 240648192 python tools/hbm_kernels_bench.py decode > /dev/null 2>&1; cp gpurun_out/hbm_decode_n256.txt $O/hbm_decode_n256.txt
NV12_ONE=1 bash tools/hbm_kernel_profile.sh nv12_to_bgr nv12_1080p 597196800 python tools/hbm_kernels_bench.py nv12 > /dev/null 2>&1; cp gpurun_out/hbm_nv12_1080p.txt $O/hbm_nv12_1080p.txt
bash tools/hbm_kernel_profile.sh det_preprocess det_preprocess 1111228416 python bench.py --steps 3 --warmup 1 --light --cpu-frames 0 > /dev/null 2>&1; cp gpurun_out/hbm_det_preprocess.txt $O/hbm_det_preprocess.txt
python bench.py --workload c5 --cpu-frames 0 > $O/bench_c5.json 2> $O/bench_c5.err
python bench.py --workload cascade5 --cpu-frames 0 --steps 6 --warmup 2 > $O/bench_cascade5.json 2> $O/bench_cascade5.err
python tools/margin_probe.py 16 > $O/margin_probe.txt 2>&1
bash tools/pmc_kernel.sh "conv_split_kernel<9, 6, 4, 2, 4, true, true, false>" tap4 python tools/profile_net.py det 64 > /dev/null 2>&1; cp gpurun_out/pmc_tap4/summary.txt $O/tap_kernel_pmc.txt
bash tools/pmc_kernel.sh "true, true, true>" s2p python tools/profile_net.py det 64 > /dev/null 2>&1; cp gpurun_out/pmc_s2p/summary.txt $O/s2_kernel_pmc.txt
python tools/split_check.py > $O/conv_split_accuracy.txt 2>&1
python -m pytest tests/test_gpu_split.py -q -s -k reordering 2>&1 | grep "heat-maps\|joints\|passed\|failed" > $O/conv_split_control.txt
# keep only the small files
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
rm -rf $O/serial $O/lanes4 $O/pmc_fetch $O/pmc_write
du -sh $O; ls $O; cat $O/pmc_summary.txt; cut -c1-300 $O/bench_cascade.json
