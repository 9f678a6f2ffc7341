#!/bin/bash
# collect_profiles.sh TAG [CONFIG ...] : copy what tools/profile_config.sh left in gpurun_out/<TAG>_<config>/ into
# profiles/<TAG>_<config>_* - the roofline inputs TOGETHER with the CSVs they are built from (tests/test_roofline_inputs.py
# regenerates the one from the other), so that bench.py never quotes counters whose passes are not in the tree.
tag=${1:-r02}; shift
cfgs=${@:-c2 c3 c4}
for c in $cfgs; do
  d=gpurun_out/${tag}_$c; [ -d $d ] || continue
  for f in bench.json bench_under_rocprof.json kernel_stats.csv kernel_trace_summary.txt roofline_inputs.json walk_stats.json walk_stats.txt; do
    [ -s $d/$f ] && cp $d/$f profiles/${tag}_${c}_$f
  done
  [ -s $d/roofline_inputs.json ] && python3 tools/build_stamp.py --profile $tag $c > /dev/null
  ls $d/pmc_pass*.csv >/dev/null 2>&1 && ( echo "kernel,launches,counters (one block per --pmc pass)"; cat $d/pmc_pass*.csv ) > profiles/${tag}_${c}_pmc_passes.csv
done
[ -s gpurun_out/${tag}_c5_bench.json ] && cp gpurun_out/${tag}_c5_bench.json profiles/${tag}_c5_bench.json
[ -s gpurun_out/${tag}_lbvh_probe.txt ] && cp gpurun_out/${tag}_lbvh_probe.txt profiles/${tag}_lbvh_probe.txt
[ -s gpurun_out/${tag}_shard_probe.txt ] && cp gpurun_out/${tag}_shard_probe.txt profiles/${tag}_shard_probe_virtual_ranks.txt
git status --short profiles 2>/dev/null | head -40
