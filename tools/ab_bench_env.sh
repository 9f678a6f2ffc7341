#!/bin/bash
# ab_bench_env.sh "ENV_A" "ENV_B" [rounds] [configs]: same-box, interleaved A/B of two environments with FRAMES BACK TO BACK - the
# bench line's `value` (bench.py --steps 10 --warmup 3, one configuration per run) - where tools/ab_env.sh times blocking frames.
A=$1; B=$2; R=${3:-2}; CFGS=${4:-"c2 c3 c4"}
for i in $(seq $R); do
  for cfg in $CFGS; do
    steps=10; [ $cfg = c4 ] && steps=4
    for v in "A:$A" "B:$B"; do
      label=${v%%:*}; envs=${v#*:}
      out=$(env $envs timeout -k 10 400 python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-secondary --no-latency-frame 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['value'])")
      echo "$cfg $label [$envs]: $out"
    done
  done
done
