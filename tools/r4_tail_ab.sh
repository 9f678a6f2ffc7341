#!/bin/bash
# r4_tail_ab.sh OUTDIR: same-box A/B of the tail hand-off (HJ_TAIL_EXPORT / HJ_TAIL_MERGE / HJ_TAIL_PRIORITY) on c2, c3 and the 8-rank share of c2
out=$1; mkdir -p $out
V="cur:HJ_TAIL_EXPORT=0 cur cur:HJ_TAIL_PRIORITY=0 cur:HJ_SLOTS=4 cur:HJ_TAIL_MERGE=16 cur:HJ_TAIL_EXPORT=1024 cur:HJ_TAIL_EXPORT=64 cur:HJ_TAIL_EXPORT=0,HJ_SLOTS=4"
bash tools/ab_variants.sh $V > $out/c2.txt 2>&1
PROBE_ARGS="--kind 1 --spp 1024" bash tools/ab_variants.sh $V > $out/c3.txt 2>&1
for e in "HJ_TAIL_EXPORT=0" "HJ_TAIL_EXPORT=256" "HJ_TAIL_PRIORITY=0" "HJ_TAIL_EXPORT=1024"; do
  echo "== $e" >> $out/shard8.txt; env $e timeout 200 python tools/shard_probe.py 1 8 >> $out/shard8.txt 2>&1
done
bash tools/ab_summary.sh $out > $out/summary.txt
