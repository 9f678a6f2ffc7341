"""`bench.py --gpus N` exactly as the driver launches it (python -m torch.distributed.run, one process per rank), with
N = 2 ranks on the ONE GPU of the test box: HIJIKI_DIST_BACKEND=gloo lets the ranks share the device (RCCL wants one GPU
per rank), everything else - the torchrun environment, hijiki_amd.dist, ShardedRenderer, the barriers, the MAX over the
ranks' wall times, the framebuffer reduce to rank 0, the one JSON line - is the code path of the 1/2/4/8-GPU scaling run.
Child processes: nothing here initialises the GPU before torchrun starts.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(tmp_path, nranks, extra, tag):
    frame = str(tmp_path / f"frame_{tag}.npy")
    args = ["bench.py", "--gpus", str(nranks), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary",
            "--dump-frame", frame] + extra
    if nranks == 1:
        cmd = [sys.executable] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
    env = dict(os.environ, HIJIKI_DIST_BACKEND="gloo", GPU_MAX_HW_QUEUES="8", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"expected ONE JSON line (rank 0 only), got {len(lines)}:\n{p.stdout[-3000:]}"
    return json.loads(lines[0]), np.load(frame)


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("config,size,spp", [("c2", 1024, 8), ("c5", 4096, 8)])
def test_bench_two_ranks_under_torchrun(tmp_path, config, size, spp):
    extra = ["--config", config, "--spp", str(spp)]
    one, f1 = _bench(tmp_path, 1, extra, "n1")
    two, f2 = _bench(tmp_path, 2, extra, "n2")
    assert one["n_gpus"] == 1 and one["rccl_ranks"] == 1
    assert two["n_gpus"] == 2 and two["rccl_ranks"] == 2 and two["rccl_backend"] == "gloo"
    for out in (one, two):
        assert out["unit"] == "Mrays/s" and out["steps"] == 1 and out["warmup"] == 0 and out["scaling"] == "strong"
        assert out["value"] > 0 and abs(out["value"] - size * size * spp / (out["ms_per_step"] * 1e-3) / 1e6) < 0.01 * out["value"]
        assert out["roofline"]["launches"] >= 1 and out["roofline"]["achieved"] > 0
        assert config + ":" in out["config"]["workload"] and f"{size}x{size} {spp}spp" in out["config"]["workload"]
    assert "mod 2" in two["config"]["partition"] and two["deal"] == "rotating"
    _check_per_rank(two, 2, size * size * spp)
    assert "per_rank" not in one
    # the reduced two-rank frame is the one-rank frame: every pixel's passes are summed as two partial sums (a few ulp)
    assert f1.shape == f2.shape == (size, size, 4) and np.isfinite(f2).all() and (f2[..., 3] > 0).all()
    np.testing.assert_allclose(f2, f1, rtol=5e-5, atol=1e-5)
    assert (f1 != f2).any() or spp == 1          # (really two partial sums: not the same bits everywhere)


def _check_per_rank(out, n, paths_per_step):
    """The fields that make a multi-GPU line explain itself (VERDICT r5 task 5): per rank and per step, wall time, time inside the
    render calls and inside the reduces, the path kernels' exclusive GPU time, the work dealt; imbalance = max / mean."""
    pr = out["per_rank"]
    for key in ("wall_ms", "render_ms", "reduce_ms", "kernel_busy_ms", "paths", "rays"):
        assert len(pr[key]) == n, key
    assert all(v > 0 for v in pr["wall_ms"] + pr["render_ms"] + pr["kernel_busy_ms"]) and all(v >= 0 for v in pr["reduce_ms"])
    assert sum(pr["paths"]) == paths_per_step * out["steps"] and min(pr["paths"]) > 0.8 * max(pr["paths"])      # every rank got its share
    assert max(pr["wall_ms"]) <= out["ms_per_step"] * 1.001                     # the line's clock is the slowest rank's + the barrier
    assert out["imbalance"] >= 1.0 and abs(out["imbalance"] - max(pr["kernel_busy_ms"]) / (sum(pr["kernel_busy_ms"]) / n)) < 1e-3
    assert out["imbalance_render"] >= 1.0 and out["reduce_ms"] == max(pr["reduce_ms"])


@pytest.mark.timeout(900)
def test_bench_static_deal_two_ranks(tmp_path):
    """`--static-deal`: all passes of a block on one rank (SURVEY 8(e)'s partition).  The line says so, and the reduced frame is the
    one-rank frame BIT FOR BIT away from the 2-pixel aprons of the block borders (a pixel's passes are then summed on one rank, in
    order, and the other rank adds zeros)."""
    extra = ["--config", "c2", "--spp", "4"]
    one, f1 = _bench(tmp_path, 1, extra, "n1")
    two, f2 = _bench(tmp_path, 2, extra + ["--static-deal"], "n2s")
    assert two["deal"] == "static" and "static deal" in two["config"]["partition"] and "mod 2" in two["config"]["partition"]
    _check_per_rank(two, 2, 1024 * 1024 * 4)
    interior = np.ones((1024, 1024), bool)
    for b in range(128, 1024, 128):
        interior[b - 2:b + 2, :] = False
        interior[:, b - 2:b + 2] = False
    assert (f2[interior].view(np.uint32) == f1[interior].view(np.uint32)).all()
    np.testing.assert_allclose(f2, f1, rtol=5e-5, atol=1e-5)


@pytest.mark.timeout(1500)
def test_bench_four_ranks_under_torchrun(tmp_path):
    """The scaling run's command with as many ranks as one test box may hold: the pool's process guard allows at most SIX
    processes with the GPU open - the test runner, the torchrun launcher and FOUR ranks - so the eight-process form of this rig
    cannot run here (VERDICT r4 #7 asked for eight: a first attempt with six ranks was killed by the guard; the 8-way deal itself
    runs below, in one process).  c5's 4096 x 4096 frame at 8 of its passes: one JSON line from rank 0,
    n_gpus == rccl_ranks == 4, the partition string, and the reduced frame == the one-rank frame."""
    extra = ["--config", "c5", "--spp", "8"]
    one, f1 = _bench(tmp_path, 1, extra, "n1")
    four, f4 = _bench(tmp_path, 4, extra, "n4")
    assert four["n_gpus"] == 4 and four["rccl_ranks"] == 4 and four["rccl_backend"] == "gloo"
    assert "mod 4" in four["config"]["partition"] and four["scaling"] == "strong" and four["steps"] == 1 and four["deal"] == "rotating"
    _check_per_rank(four, 4, 4096 * 4096 * 8)
    assert abs(four["value"] - 4096 * 4096 * 8 / (four["ms_per_step"] * 1e-3) / 1e6) < 0.01 * four["value"]
    assert f4.shape == (4096, 4096, 4) and np.isfinite(f4).all() and (f4[..., 3] > 0).all()
    np.testing.assert_allclose(f4, f1, rtol=5e-5, atol=1e-5)


@pytest.mark.timeout(900)
def test_bench_eight_way_deal_in_one_process(tmp_path):
    """`bench.py --inproc --gpus 8 --config c5`: eight contexts (HJ_COMM_SHARED_GPU=1: all on this box's one GPU, the reduce a
    kernel sum instead of RCCL), each rendering rank i's blocks of the (bx + by + p) mod 8 deal of c5's frame - the partition
    the 8-GPU scaling run uses, produced by bench.py itself."""
    env = dict(os.environ, HJ_COMM_SHARED_GPU="1", GPU_MAX_HW_QUEUES="8")
    cmd = [sys.executable, "bench.py", "--inproc", "--gpus", "8", "--config", "c5", "--spp", "8", "--steps", "1", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=800)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-6000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and "mod 8" in out["config"]["partition"] and "c5:" in out["config"]["workload"]
    assert "4096x4096 8spp" in out["config"]["workload"] and out["value"] > 0 and out["scaling"] == "strong"
    pr = out["per_rank"]
    assert len(pr["render_ms"]) == len(pr["kernel_busy_ms"]) == len(pr["paths"]) == 8 and sum(pr["paths"]) == 4096 * 4096 * 8
    assert min(pr["paths"]) > 0.8 * max(pr["paths"]) and out["imbalance"] >= 1.0 and out["reduce_ms"] >= 0 and out["deal"] == "rotating"
