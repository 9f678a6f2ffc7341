"""Scene compiler invariants (mirror of Scene::compile, reference src/main.rs:173-357) and the block generator."""
import ctypes as C

import numpy as np
import pytest

from hijiki_amd import abi, host


def check_bvh(cs):
    bvh, f = cs.bvh, cs.bvh_f32
    n_shapes = cs.num_shapes
    N = len(bvh)
    assert N == 2 * n_shapes - 1                                   # one shape per leaf
    shape, ex = bvh[:, 3], bvh[:, 7]
    leaves = shape != abi.BVH_INNER
    assert leaves.sum() == n_shapes and sorted(shape[leaves].tolist()) == list(range(n_shapes))
    assert (ex > np.arange(N)).all()                               # exits only move forward -> the walk terminates
    root_exit = abi.BVH_ROOT_EXIT if N <= abi.BVH_ROOT_EXIT else N
    assert ex[0] == root_exit and ex[N - 1] == root_exit
    # pre-order structure: left child = i+1, right child = exit of the left child; a right child inherits the exit
    stack = [(0, root_exit)]
    seen = 0
    while stack:
        i, e = stack.pop()
        seen += 1
        assert ex[i] == e
        if shape[i] == abi.BVH_INNER:
            l = i + 1
            r = ex[l]
            assert i < l < r < N
            stack.append((r, e))
            stack.append((l, r))
            # child boxes lie inside the parent's box
            for c in (l, r):
                assert (f[c, 0:3] >= f[i, 0:3] - 1e-6).all() and (f[c, 4:7] <= f[i, 4:7] + 1e-6).all()
    assert seen == N
    return leaves


def shape_box(cs, g):
    ns, nq = cs.desc.num_spheres, cs.desc.num_quads
    if g < ns:
        c = cs.spheres[g]
        return c[:3] - c[3], c[:3] + c[3]
    if g < ns + nq:
        q = cs.quads[g - ns]
        o, e1, e2 = q[0:3], q[4:7], q[8:11]
        p = np.stack([o, o + e1, o + e2, o + e1 + e2])
        return p.min(0), p.max(0)
    p = cs.vertices[cs.triangles[g - ns - nq]][:, 0:3]
    return p.min(0), p.max(0)


def test_cbox_compile_counts_and_invariants(cbox):
    d = cbox.desc
    # 6332 triangles, 12663 nodes, 2 emitter triangles
    assert (d.num_triangles, d.num_bvh_nodes, d.num_emitters, d.num_spheres, d.num_quads) == (6332, 12663, 2, 0, 0)
    leaves = check_bvh(cbox)
    f, b = cbox.bvh_f32, cbox.bvh
    for i in np.flatnonzero(leaves)[::97]:
        lo, hi = shape_box(cbox, int(b[i, 3]))
        np.testing.assert_array_equal(f[i, 0:3], lo.astype(np.float32))     # leaf box = shape AABB (src/main.rs:74-79)
        np.testing.assert_array_equal(f[i, 4:7], hi.astype(np.float32))


def test_material_words_and_emitter_table(cbox_spheres):
    cs = cbox_spheres
    d = cs.desc
    assert d.num_spheres == 2 and d.num_materials == d.num_spheres + d.num_triangles
    mats = cs.materials
    # spheres come first in the global index space (src/main.rs:278-287): mirror, then dielectric
    assert mats[0] >> 24 == abi.MAT_MIRROR and mats[1] >> 24 == abi.MAT_DIELECTRIC
    assert d.dielectric[0].eta == 1.5 and tuple(d.dielectric[0].extinction) == (0, 0, 0)   # DielectricMaterial::clear
    em = cs.emitters
    assert len(em) == 2
    for e in em:
        assert mats[e[0]] >> 24 == abi.MAT_EMISSIVE
    pdf = em[:, 1].view(np.float32)
    cdf = em[:, 2].view(np.float32)
    assert (pdf == np.float32(0.5)).all() and cdf.tolist() == [0.5, 1.0]          # src/main.rs:300-307
    check_bvh(cs)


def test_packed_buffer_follows_reference_order(cbox_spheres):
    cs = cbox_spheres
    buf = cs.packed()
    d = cs.desc
    pad = lambda n: (n + 255) & ~255
    sizes = [64, d.num_bvh_nodes * 32, d.num_spheres * 16, d.num_quads * 48, d.num_triangles * 12, d.num_vertices * 32,
             d.num_materials * 4, d.num_emitters * 16, d.num_diffuse * 16, d.num_diffusecb * 32, d.num_dielectric * 16,
             d.num_emissive * 16]                                                  # src/main.rs:314-326
    assert len(buf) == sum(pad(s) for s in sizes)
    info = abi.SceneInfo.from_buffer_copy(buf[:64].tobytes())
    assert (info.num_spheres, info.num_triangles, info.num_emitters) == (2, d.num_triangles, 2)
    assert abs(info.camera.fov - 27.7) < 1e-6 and abs(info.camera.position[2] - 5.41) < 1e-6
    off = pad(64)
    np.testing.assert_array_equal(buf[off:off + 32 * 3].view(np.uint32).reshape(3, 8), cs.bvh[:3])
    off += pad(sizes[1])
    np.testing.assert_array_equal(buf[off:off + 32].view(np.float32).reshape(2, 4), cs.spheres)


def test_scene_api_errors():
    s = host.Scene()
    m = s.add_diffuse((1, 1, 1))
    with pytest.raises(abi.HijikiError):
        s.add_sphere((0, 0, 0), 1.0, m + 5)                 # unknown material
    with pytest.raises(abi.HijikiError):
        s.add_triangle(0, 1, 2, m)                          # unknown vertices
    s.add_sphere((0, 0, 0), 1.0, m)
    with pytest.raises(abi.HijikiError) as e:
        s.compile()                                         # reference panics on a 1-shape scene (src/main.rs:230)
    assert "2 shapes" in str(e.value)
    s.add_sphere((3, 0, 0), 1.0, m)
    cs = s.compile()
    assert cs.desc.num_bvh_nodes == 3 and cs.desc.num_emitters == 0


def test_mixed_shapes_global_index_space():
    s = host.Scene()
    d = s.add_diffuse((1, 1, 1))
    e = s.add_emissive((1, 2, 3))
    v0 = s.add_vertices([[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 0, 1]] * 3)
    s.add_triangle(v0, v0 + 1, v0 + 2, d)                   # object 0
    s.add_quad((5, 0, 0), (1, 0, 0), (0, 1, 0), e)          # object 1
    s.add_sphere((9, 0, 0), 0.5, d)                         # object 2
    s.add_sphere((12, 0, 0), 0.5, e)                        # object 3
    cs = s.compile()
    # global order: spheres (obj 2, 3), quads (obj 1), triangles (obj 0)
    tags = (cs.materials >> 24).tolist()
    assert tags == [abi.MAT_DIFFUSE, abi.MAT_EMISSIVE, abi.MAT_EMISSIVE, abi.MAT_DIFFUSE]
    assert cs.emitters[:, 0].tolist() == [1, 2]
    check_bvh(cs)


def test_degenerate_inputs_still_build():
    s = host.Scene()
    m = s.add_diffuse((1, 1, 1))
    for _ in range(33):
        s.add_sphere((1, 2, 3), 0.5, m)                     # identical centroids -> median splits
    check_bvh(s.compile())


def test_block_generator_semantics():
    W, H, spp, seed = 300, 200, 3, 42
    blocks = host.make_blocks(W, H, spp, seed)
    per = host.blocks_per_pass(W, H)
    assert per == 3 * 2 and len(blocks) == per * spp
    offs = []
    for k in range(spp + 1):
        o = (C.c_float * 2)()
        host.lib().hj_pass_offset(seed, k, o)
        assert 0 <= o[0] < 1 and 0 <= o[1] < 1
        offs.append((o[0], o[1]))
    for i, b in enumerate(blocks):
        p, j = divmod(i, per)
        assert b.id == i                                                     # ids run on across passes (main.rs:660-662)
        assert (b.origin[0], b.origin[1]) == ((j % 3) * 128, (j // 3) * 128)     # raster order
        assert b.dimension[0] == min(128, W - b.origin[0]) and b.dimension[1] == min(128, H - b.origin[1])
        assert (b.original_dimension[0], b.original_dimension[1]) == (W, H)
        assert b.seed == host.lib().hj_block_seed(seed, p, j)
        # the last block of a pass already carries the NEXT pass's offset (main.rs:664-680)
        k = p + (1 if j == per - 1 else 0)
        assert (b.sample_offset[0], b.sample_offset[1]) == offs[k]
    assert len({b.seed for b in blocks}) == len(blocks)
    again = host.make_blocks(W, H, spp, seed)
    assert bytes(again) == bytes(blocks)
    assert bytes(host.make_blocks(W, H, spp, seed + 1)) != bytes(blocks)
    # pass sub-range == slice of the full list
    assert bytes(host.make_blocks(W, H, spp, seed, 1, 2)) == bytes(blocks)[per * 40:2 * per * 40]


def test_block_generator_matches_oracle_restatement(oracle):
    for (W, H, spp, seed) in ((256, 256, 4, 1), (300, 200, 2, 7), (128, 128, 3, 9)):
        assert bytes(host.make_blocks(W, H, spp, seed)) == bytes(oracle.make_blocks(W, H, spp, seed))


def test_block_size_must_be_multiple_of_64():
    with pytest.raises(abi.HijikiError):
        host.make_blocks(256, 256, 1, 1, block_size=100)       # assert!(block_size & 63 == 0), main.rs:633


def test_large_mesh_root_exit_terminates():
    cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=600000).compile()
    N = cs.desc.num_bvh_nodes
    assert N > abi.BVH_ROOT_EXIT
    ex = cs.bvh[:, 7]
    assert (ex > np.arange(N)).all() and ex[0] == N      # the reference's constant 1 000 000 would point INTO the array


def test_tree_passes_change_the_walk_not_the_image(tmp_path):
    """The passes over the finished SAH tree - rotations (HJ_BVH_ROTATE), insertion-based optimisation (HJ_BVH_REINSERT) and
    the child order (HJ_BVH_CHILD_ORDER: 3 = fewer shapes first, 4 = voted by a sample of the renderer's rays), read once per
    process by libhijiki_host.so - keep every invariant of the reference's flattened tree, lower its surface-area cost and
    the node visits of the reference walk, and leave the frame untouched bit for bit (the image depends on the tree only
    through epsilon-ties)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "tree_probe.py"
    script.write_text(
        "import sys, json, hashlib\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import numpy as np\n"
        "from hijiki_amd import host\n"
        "from oracle import hj_oracle\n"
        "from test_gpu_parity import _check_skip_link_tree, _shape_boxes, _sah_cost\n"
        "cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()\n"
        "_check_skip_link_tree(cs.bvh, _shape_boxes(cs))\n"
        "acc, ctr, _ = hj_oracle.render_blocks(cs, host.make_blocks(96, 64, 2, 5), 96, 64, nthreads=4)\n"
        "print(json.dumps({'sah': float(_sah_cost(cs.bvh)), 'nodes': ctr['nodes'], 'closest': ctr['closest_calls'],\n"
        "                  'frame': hashlib.sha256(acc.tobytes()).hexdigest()}))\n")

    def run(rotate, order, reinsert=0):
        env = dict(os.environ, HJ_BVH_ROTATE=str(rotate), HJ_BVH_CHILD_ORDER=str(order), HJ_BVH_REINSERT=str(reinsert))
        p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-3000:]
        return json.loads(p.stdout.strip().splitlines()[-1])

    plain, ordered, full, best = run(0, 0), run(0, 3), run(8, 3), run(8, 4, 3)
    assert plain["frame"] == ordered["frame"] == full["frame"] == best["frame"]
    assert plain["closest"] == ordered["closest"] == full["closest"] == best["closest"]          # the same paths
    assert best["sah"] < 0.995 * full["sah"] and best["nodes"] < 0.97 * full["nodes"], (full, best)    # (measured: -0.9 %, -4.6 %)
    assert full["sah"] < 0.99 * plain["sah"] and abs(ordered["sah"] - plain["sah"]) < 1e-6 * plain["sah"]   # (measured: -2.4 %)
    assert ordered["nodes"] < 0.97 * plain["nodes"] and full["nodes"] < 0.98 * ordered["nodes"], (plain, ordered, full)


def test_batched_reinsertion_on_a_large_tree(tmp_path):
    """Trees beyond 400 000 nodes get the insertion-based optimisation in its batched form (tree_opt.cpp
    optimize_by_reinsertion_batched: parallel searches on the tree as it stands at the start of a batch, moves applied one after
    the other): still the reference's flattened format over the same leaves, a lower surface-area cost, the frame the one of the
    tree without the pass except where two hits lie within epsilon of each other (SURVEY C-11: over a dense mesh a few paths in a
    thousand end on the other side of a shared edge) - and the SAME tree whatever the number of threads (one core against all)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "large_tree.py"
    script.write_text(
        "import sys, json, hashlib\n"
        f"sys.path.insert(0, {root!r}); sys.path.insert(0, {os.path.join(root, 'tests')!r})\n"
        "import numpy as np\n"
        "from hijiki_amd import host\n"
        "from oracle import hj_oracle\n"
        "from test_gpu_parity import _check_skip_link_tree, _shape_boxes, _sah_cost, _record_multiset\n"
        "cs = host.Scene.synthetic(host.SYNTH_CBOX_MESH, mesh_triangles=210000).compile()\n"
        "assert len(cs.bvh) > 400000\n"
        "_check_skip_link_tree(cs.bvh, _shape_boxes(cs))\n"
        "acc, ctr, _ = hj_oracle.render_blocks(cs, host.make_blocks(64, 64, 1, 5), 64, 64, nthreads=4)\n"
        "leaves = np.sort(np.asarray(cs.bvh)[:, 3][np.asarray(cs.bvh)[:, 3] != 0xFFFFFFFF])\n"
        "np.save(sys.argv[1], acc)\n"
        "print(json.dumps({'sah': float(_sah_cost(cs.bvh)), 'nodes': ctr['nodes'] + ctr['shadow_nodes'], 'tree': hashlib.sha256(np.ascontiguousarray(cs.bvh).tobytes()).hexdigest(),\n"
        "                  'leaves': hashlib.sha256(leaves.tobytes()).hexdigest(), 'closest': ctr['closest_calls']}))\n")

    def run(passes, one_core=False):
        env = dict(os.environ, HJ_BVH_REINSERT_LARGE=str(passes), HJ_BVH_CHILD_ORDER="3")       # (no ray vote: the tree passes alone)
        frame = str(tmp_path / f"frame_{passes}_{int(one_core)}.npy")
        cmd = [sys.executable, str(script), frame]
        if one_core:
            cmd = ["taskset", "-c", "0"] + cmd
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        out = json.loads(p.stdout.strip().splitlines()[-1])
        out["frame"] = np.load(frame)
        return out

    none, one = run(0), run(1)
    assert one["leaves"] == none["leaves"]
    differ = float((one["frame"].view(np.uint32) != none["frame"].view(np.uint32)).any(axis=-1).mean())
    assert differ < 0.05 and abs(one["closest"] - none["closest"]) < 0.01 * none["closest"], (differ, one["closest"], none["closest"])
    assert one["sah"] < 0.97 * none["sah"], (none["sah"], one["sah"])            # (measured at 1 M triangles: -4.9 % after the first pass)
    assert one["tree"] != none["tree"]
    import shutil
    if shutil.which("taskset"):
        assert run(1, one_core=True)["tree"] == one["tree"]                     # batch sizes decide, not threads


def test_directional_link_orderings(oracle):
    """hjh_compiled_directional_bvh + hjo_set_directional_bvh (round 6's measurement of a direction-dependent child order; CPU only,
    the kernels know nothing of it): K = hj_direction_classes(mode) link orderings of the installed tree are K valid pre-order
    skip-link trees over the same boxes and leaves; a ray walks the array of its direction class (one text in C and numpy:
    hj_ray_direction_class); the frame is the static order's except at epsilon ties, at fewer node visits; mode 0 is the installed
    tree itself."""
    from hijiki_amd import abi
    from test_gpu_parity import _check_skip_link_tree, _shape_boxes, _record_multiset
    cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES, mesh_triangles=1280).compile()
    assert [abi.direction_classes(m) for m in range(0, 9)] == [1, 2, 2, 4, 2, 4, 4, 8, 6]
    d = np.array([[1, 2, -3], [-1, 0.5, 0.25], [-0.0, 0.0, 5], [3, -3, 3]], np.float32)
    assert abi.ray_direction_class(7, d).tolist() == [4, 1, 1, 2] and abi.ray_direction_class(8, d).tolist() == [5, 1, 4, 0]
    assert abi.ray_direction_class(5, d).tolist() == [2, 1, 1, 0]
    same = cs.directional_bvh(0)
    assert same.shape == (1, len(cs.bvh), 8) and (same[0] == np.asarray(cs.bvh)).all()
    W, H = 96, 64
    blocks = host.make_blocks(W, H, 2, 9)
    L = oracle.lib()
    L.hjo_set_shadow_anyhit(1)
    try:
        want, ctr, _ = oracle.render_blocks(cs, blocks, W, H)
        for mode in (4, 7, 8):
            arrays = cs.directional_bvh(mode, vote_paths=8000 * abi.direction_classes(mode))
            assert arrays.shape[0] == abi.direction_classes(mode)
            for k in range(arrays.shape[0]):
                _check_skip_link_tree(arrays[k], _shape_boxes(cs))
                assert _record_multiset(arrays[k]) == _record_multiset(cs.bvh)
            L.hjo_set_directional_bvh(mode, arrays.ctypes.data)
            try:
                got, c2, _ = oracle.render_blocks(cs, blocks, W, H)
            finally:
                L.hjo_set_directional_bvh(0, None)
            differ = float((got.view(np.uint32) != want.view(np.uint32)).any(axis=-1).mean())
            assert differ < 0.02 and abs(c2["hits"] - ctr["hits"]) <= 0.001 * ctr["hits"], (mode, differ)
            if mode == 7:
                assert c2["nodes"] < 0.99 * ctr["nodes"], (c2["nodes"], ctr["nodes"])          # (measured on c3: -4.5 %)
    finally:
        L.hjo_set_shadow_anyhit(0)


def test_tune_bvh_on_an_installed_tree(oracle):
    """hjh_compiled_tune_bvh: compile()'s tree passes on a tree that came from elsewhere (here: compile's own tree with every
    inner node's children exchanged - a valid tree in a poor order).  The result is a valid flattened tree over the same shapes,
    the frame is the same, the reference walk visits fewer nodes."""
    import hashlib
    from test_gpu_parity import _check_skip_link_tree, _shape_boxes
    cs = host.Scene.synthetic(host.SYNTH_CBOX_SPHERES).compile()
    blocks = host.make_blocks(96, 64, 2, 5)
    want, c_best, _ = oracle.render_blocks(cs, blocks, 96, 64)
    # exchange the children of every inner node: re-flatten by hand
    nodes = cs.bvh.copy()
    N = len(nodes)
    out = np.zeros_like(nodes)
    pos = [0]

    def emit(i, exit_index):
        me = pos[0]
        pos[0] += 1
        out[me] = nodes[i]
        if nodes[i, 3] == 0xFFFFFFFF:
            l, r = i + 1, int(nodes[i + 1, 7])
            first = emit(r, None)                       # the right child first
            second_at = pos[0]
            emit(l, None)
            fix.append((first, second_at))
        return me

    fix = []
    import sys
    sys.setrecursionlimit(10000)
    emit(0, None)

    def set_exits(me, exit_index):
        out[me, 7] = exit_index
        if out[me, 3] == 0xFFFFFFFF:
            second = second_of[me + 1]
            set_exits(me + 1, second)
            set_exits(second, exit_index)

    second_of = dict(fix)
    set_exits(0, int(nodes[0, 7]))
    cs.set_bvh(out)
    _check_skip_link_tree(cs.bvh, _shape_boxes(cs))
    got, c_swapped, _ = oracle.render_blocks(cs, blocks, 96, 64)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()
    cs.tune_bvh(reinsert_passes=0, vote_paths=60000)
    _check_skip_link_tree(cs.bvh, _shape_boxes(cs))
    got, c_tuned, _ = oracle.render_blocks(cs, blocks, 96, 64)
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()
    assert c_tuned["nodes"] < 0.97 * c_swapped["nodes"] and c_tuned["nodes"] < 1.02 * c_best["nodes"], (c_best["nodes"], c_swapped["nodes"], c_tuned["nodes"])
    with pytest.raises(abi.HijikiError):
        bad = cs.bvh.copy()
        bad[1, 7] += 1                                  # not a tree any more
        cs.set_bvh(bad)
        cs.tune_bvh()
