#!/bin/bash
# collect_profiles.sh TAG : copy what tools/profile_config.sh left in gpurun_out/<TAG>_<config>/ into profiles/<TAG>_<config>_*
tag=${1:-r02}
for c in c2 c3 c4; do
  d=gpurun_out/${tag}_$c; [ -d $d ] || continue
  for f in bench.json bench_under_rocprof.json kernel_stats.csv kernel_trace_summary.txt roofline_inputs.json walk_stats.json walk_stats.txt; do
    [ -s $d/$f ] && cp $d/$f profiles/${tag}_${c}_$f
  done
  ( echo "kernel,launches,counters (one block per --pmc pass)"; cat $d/pmc_pass*.csv ) > profiles/${tag}_${c}_pmc_passes.csv
done
[ -s gpurun_out/${tag}_c5_bench.json ] && cp gpurun_out/${tag}_c5_bench.json profiles/${tag}_c5_bench.json
[ -s gpurun_out/${tag}_lbvh_probe.txt ] && cp gpurun_out/${tag}_lbvh_probe.txt profiles/${tag}_lbvh_probe.txt
[ -s gpurun_out/${tag}_shard_probe.txt ] && cp gpurun_out/${tag}_shard_probe.txt profiles/${tag}_shard_probe_virtual_ranks.txt
git status --short profiles | head -40
