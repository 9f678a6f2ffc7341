#!/bin/bash
export GPU_MAX_HW_QUEUES=8
out=gpurun_out/r2_shard; mkdir -p $out
timeout 600 python tools/shard_probe.py 2>&1 | tee $out/shard_c2.txt
echo "== reconstruct exp skip"; tools/ab_variants.sh noexpskip cur 2>&1 | tee $out/expskip_c2.txt
PROBE_ARGS="--kind 1 --spp 256" tools/ab_variants.sh noexpskip cur 2>&1 | tee $out/expskip_c3.txt
echo "== 8-rank share, knobs"; for env in "" "HJ_POOL=2048" "HJ_POOL=4096" "HJ_SLOTS=4" "HJ_SLOTS=2"; do echo -n "[$env] "; env $env python - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from hijiki_amd import host, device
cs = host.Scene.synthetic(host.SYNTH_CBOX).compile()
r = device.Renderer(0); r.upload_scene(cs); r.create_framebuffer(1024, 1024)
best = 1e9
for _ in range(4):
    r.clear(); t = time.time(); r.render_frame(512, 1, rank=3, world=8); best = min(best, time.time() - t)
print(f"rank 3 of 8: {best*1e3:.2f} ms")
PY
done
