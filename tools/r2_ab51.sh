#!/bin/bash
# is the walk bound by VALU issue or by the vector memory pipeline?  variants vpN: N extra VALU instructions per plain box step (33 of its own: no effect up to 16); lpK: one more 16-byte load per box step (1 same address in every lane, 2 the lane's node again through FLAT, 3 the node's global copy)
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_ab51; mkdir -p $out
V="cur lp1 lp2 lp3"
echo "== C2"; PROBE_ARGS="" tools/ab_variants.sh $V 2>&1 | tee $out/c2.txt
echo "== C4"; PROBE_ARGS="--kind 2 --tris 1000000 --size 2048 --spp 64" tools/ab_variants.sh $V 2>&1 | tee $out/c4.txt
