"""world_size-2 gloo test of the multi-GPU exchange step (hijiki_amd.dist) on CPU.

Each rank renders ITS blocks (hj_block_owner's diagonal deal, rotating with the pass or static, as hj_render_frame does) with the CPU oracle standing in for the
device renderer, then the framebuffers are sum-reduced to rank 0 through torch.distributed — the same call the
RCCL path makes.  The reduced frame must equal the single-process frame (exactly away from block aprons).
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, H, spp, seed, static, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from hijiki_amd import abi, host
    from hijiki_amd import dist as hjdist
    from oracle import hj_oracle
    r, w, _ = hjdist.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
    per = host.blocks_per_pass(W, H)
    all_blocks = host.make_blocks(W, H, spp, seed)
    mine = [all_blocks[p * per + j] for p in range(spp) for j in hjdist.owned_blocks(W, H, rank, world, 0 if static else p)]
    arr = (abi.ImageBlock * len(mine))(*mine)
    acc, _, _ = hj_oracle.render_blocks(cs, arr, W, H, nthreads=2)
    fb = torch.from_numpy(acc)
    hjdist.barrier()                                           # bench.py's bracket around the timed region
    hjdist.reduce_framebuffer(fb, root=0)
    assert hjdist.max_over_ranks(10.0 + rank) == 10.0 + world - 1      # bench.py: the slowest rank's wall time, on every rank
    if rank == 0:
        np.save(out_path, fb.numpy())
    both = torch.full((4,), float(rank + 1))
    hjdist.reduce_framebuffer(both, all_ranks=True)
    assert both.tolist() == [float(sum(range(1, world + 1)))] * 4
    hjdist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4], ids=["2-ranks", "4-ranks"])
@pytest.mark.parametrize("static", [False, True], ids=["rotating-deal", "static-deal"])
def test_two_rank_tile_sharding_gloo(tmp_path, oracle, static, world):
    import torch.multiprocessing as mp
    from hijiki_amd import host
    W, H, spp, seed = 256, 256, 2 if world == 2 else 4, 5
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(world, _free_port(), W, H, spp, seed, static, out), nprocs=world, join=True)
    reduced = np.load(out)
    cs = host.Scene.synthetic(host.SYNTH_CBOX, mesh_triangles=320).compile()
    full, _, _ = oracle.render_blocks(cs, host.make_blocks(W, H, spp, seed), W, H, nthreads=4)
    if static:      # all passes of a block on one rank: only the 2-pixel aprons change their association
        interior = np.ones((H, W), bool)
        interior[126:130, :] = False
        interior[:, 126:130] = False
        assert (reduced[interior] == full[interior]).all()
    np.testing.assert_allclose(reduced, full, rtol=3e-6, atol=1e-6)


def test_block_ownership_rule():
    from hijiki_amd import dist as hjdist
    W = H = 1024
    per = 64
    for world in (1, 2, 4, 8, 3):
        owned = [hjdist.owned_blocks(W, H, r, world) for r in range(world)]
        flat = sorted(j for o in owned for j in o)
        assert flat == list(range(per))
        assert max(len(o) for o in owned) - min(len(o) for o in owned) <= 2
        if world in (2, 4, 8):      # diagonal deal: every rank holds blocks of every column and every row
            for o in owned:
                assert {j % 8 for j in o} == set(range(8)) and {j // 8 for j in o} == set(range(8))
        # the deal moves one diagonal per pass: over `world` passes every rank renders every block position once
        for r in range(world):
            seen = sorted(j for p in range(world) for j in hjdist.owned_blocks(W, H, r, world, p))
            assert seen == list(range(per))
            assert hjdist.owned_blocks(W, H, r, world, world) == owned[r]
