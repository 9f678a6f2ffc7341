#!/bin/bash
# profile_config.sh TAG CONFIG : every rocprofv3 run behind profiles/<TAG>_<CONFIG>_* and the roofline block of bench.py:
#   1. kernel trace + stats of `bench.py --config CONFIG` (2 timed frames)          -> kernel_stats.csv, kernel_trace_summary.txt
#   2. one --pmc pass per counter group over one frame of the same command (FETCH_SIZE, WRITE_SIZE, SQ lanes, SQ issue/wait)
#   3. tools/roofline_inputs.py turns the CSVs into <TAG>_<CONFIG>_roofline_inputs.json (what bench.py reads)
#   4. the un-profiled bench line, quoting those inputs                               -> bench.json
# Run from the repo root on the GPU box; results land in gpurun_out/<TAG>_<CONFIG>/ (copy what you want judged into profiles/).
# GPU_MAX_HW_QUEUES must be in the environment BEFORE rocprofv3 starts: its tool library initialises HIP first, and
# then neither bench.py's setdefault nor the library's constructor can change the number of hardware queues.
export GPU_MAX_HW_QUEUES=8
tag=$1; cfg=${2:-c2}
root=$(pwd); out=$root/gpurun_out/${tag}_${cfg}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-latency-frame > $out/stats.log 2>&1 || echo "stats pass failed"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
python3 $root/tools/roofline_inputs.py trace "$(find $out/stats -name "*kernel_trace.csv" | head -1)" 2 > $out/kernel_trace_summary.txt
cat $out/kernel_trace_summary.txt
grep '^{' $out/stats.log > $out/bench_under_rocprof.json
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SALU" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 500 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-latency-frame > $out/p$i.log 2>&1 || echo "pmc pass $i ($grp) failed"
  f=$(find $out/p$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $root/tools/roofline_inputs.py pmc "$f" > $out/pmc_pass$i.csv
  rm -rf $out/p$i
done
rm -rf $out/stats
cd $root
# walk statistics of the same workload (diagnostic build with counters: tools/build_variant.sh stats -DHJ_WALK_STATS)
if [ -f build/variants/var_stats.so ]; then
  case $cfg in
    c2) HJ_STATS_SPP=512 timeout 300 python3 tools/walk_stats.py 0 --json $out/walk_stats.json > $out/walk_stats.txt 2>&1 ;;
    c3) HJ_STATS_SPP=1024 timeout 300 python3 tools/walk_stats.py 1 --json $out/walk_stats.json > $out/walk_stats.txt 2>&1 ;;
    c4) HJ_STATS_SPP=256 HJ_STATS_SIZE=2048 HJ_STATS_TRIS=1000000 timeout 300 python3 tools/walk_stats.py 2 --json $out/walk_stats.json > $out/walk_stats.txt 2>&1 ;;
  esac
fi
# the inputs first, in place (profiles/ of this copy of the tree) and TOGETHER with the CSVs they come from, so that the un-profiled
# bench line below quotes THESE counters and tests/test_roofline_inputs.py still regenerates the newest inputs file from its passes
python3 tools/roofline_inputs.py build $out $cfg > $out/roofline_inputs.json
bash tools/collect_profiles.sh $tag $cfg > /dev/null
timeout 600 python3 bench.py --config $cfg --steps 5 > $out/bench.json 2> $out/bench.err || echo "bench failed"
head -6 $out/kernel_stats.csv; cat $out/pmc_pass*.csv | grep -i "k_path\|^kernel" ; cat $out/roofline_inputs.json; tail -1 $out/bench.json | cut -c1-400
