#!/bin/bash
for c in c2 c3 c4; do bash tools/profile_config.sh r02 $c > gpurun_out/prof_$c.log 2>&1; tail -3 gpurun_out/prof_$c.log | cut -c1-300; done
