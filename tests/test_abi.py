"""The C-ABI libraries load and export every symbol the public headers declare (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from hijiki_amd import abi, device, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"static inline [^{]*\{.*?\n\}\n", "", text, flags=re.S)      # inline definitions (hj_ray_direction_class): not exports
    return sorted(set(re.findall(r"\b(hjh?_[a-z0-9_]+)\s*\(", text)))


def test_struct_sizes_match_reference_layouts():
    # SURVEY.md Appendix A (std430 == #[repr(C, align(16))] layouts of src/main.rs, src/shape.rs)
    assert C.sizeof(abi.Camera) == 48 and C.sizeof(abi.SceneInfo) == 64
    assert C.sizeof(abi.BvhNode) == 32 and abi.BvhNode.shape_index.offset == 12 and abi.BvhNode.exit_index.offset == 28
    assert C.sizeof(abi.Sphere) == 16 and C.sizeof(abi.Quad) == 48 and abi.Quad.edge1.offset == 16
    assert C.sizeof(abi.Triangle) == 12 and C.sizeof(abi.Vertex) == 32 and abi.Vertex.normal.offset == 16
    assert C.sizeof(abi.Emitter) == 16 and C.sizeof(abi.ImageBlock) == 40 and abi.ImageBlock.sample_offset.offset == 32
    assert C.sizeof(abi.DiffuseCB) == 32 and C.sizeof(abi.Dielectric) == 16 and abi.Dielectric.eta.offset == 12
    assert abi.SceneInfo.num_spheres.offset == 48 and abi.SceneInfo.num_emitters.offset == 60


def test_material_tags_match_reference_enum():
    # enum Material order, src/main.rs:38-44
    assert (abi.MAT_DIFFUSE, abi.MAT_DIFFUSECBOARD, abi.MAT_MIRROR, abi.MAT_DIELECTRIC, abi.MAT_EMISSIVE) == (0, 1, 2, 3, 4)
    assert abi.MATERIAL_TAG_SHIFT == 24 and abi.BVH_ROOT_EXIT == 1000000 and abi.BLOCK_SIZE == 128


def test_host_library_exports_every_declared_symbol():
    L = host.lib()
    names = declared_functions("hijiki_host.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n


def test_hip_library_exports_every_declared_symbol():
    L = device.lib()
    names = declared_functions("hijiki_hip.h")
    assert set(names) == set(device.EXPORTS), set(names) ^ set(device.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    assert L.hj_version() >= 0x000100


def test_hip_library_has_gfx950_code_object():
    blob = open(device.HIP_LIB_PATH, "rb").read()
    assert b"amdgcn-amd-amdhsa--gfx950" in blob


def test_default_opts_are_the_reference_constants():
    o = device.default_opts()
    # radius 2 / stddev 0.5: src/main.rs:1284-1285; 1000 bounces, roulette from bounce > 3: render.glsl:92,137
    assert (o.use_bvh, o.recon_radius, o.max_bounces, o.rr_start) == (1, 2, 1000, 4)
    assert o.recon_stddev == 0.5


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(abi.HijikiError) as e:
        device.Renderer(0)
    assert e.value.status == abi.HJ_ERR_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "hijiki_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "hj_oracle" not in text and "oracle/" not in text and "from oracle" not in text, os.path.join(dirpath, f)


def test_headers_are_plain_c99_and_layouts_hold(tmp_path):
    """The boundary is a C ABI: both public headers compile as C99 (-pedantic) and the example host, whose static
    asserts restate SURVEY.md Appendix A's sizes/offsets, links against the two libraries."""
    import subprocess
    exe = str(tmp_path / "render_cbox")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "render_cbox.c"), "-L" + os.path.join(ROOT, "hijiki_amd", "lib"),
           "-lhijiki_hip", "-lhijiki_host", "-Wl,-rpath," + os.path.join(ROOT, "hijiki_amd", "lib"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_argument_checks_need_no_gpu():
    """Entry points added in round 2 reject bad arguments before they touch a device (status codes, no crash)."""
    import ctypes as C
    L = device.lib()
    assert L.hj_device_count() >= 0
    assert L.hj_sync(None, None) == abi.HJ_ERR_INVALID
    assert L.hj_render_frame_async(None, 1, 1, 0, 1, 0, 1, None) == abi.HJ_ERR_INVALID
    assert L.hj_reserve(None, 1, None) == abi.HJ_ERR_INVALID
    out = C.c_void_p()
    assert L.hj_comm_create(None, 1, C.byref(out)) == abi.HJ_ERR_INVALID and not out.value
    arr = (C.c_void_p * 1)(None)
    assert L.hj_comm_create(arr, 1, C.byref(out)) == abi.HJ_ERR_INVALID
    assert L.hj_comm_create(arr, 0, C.byref(out)) == abi.HJ_ERR_INVALID
    assert L.hj_comm_reduce_framebuffers(None, 0) == abi.HJ_ERR_INVALID
    L.hj_comm_destroy(None)                                   # no-ops on NULL
    L.hj_set_progress_callback(None, device.PROGRESS_FN(), None, 1)
    assert (L.hj_version() >> 8) & 0xFF >= 2                  # ABI 0.2: statistics grew, async / comm / progress entry points


def test_walk_of_the_fused_kernel_is_spill_free():
    """The fused kernel's own code is the BVH walk (top-up, hit compaction, the camera-packet walk and shade are called
    functions with their own register allocation): no scratch (spill) instruction may sit inside the walk's loops, nor
    inside the node loop of the packet stage - one reload there is a dependent memory trip per iteration (DESIGN.md
    section 4).  tools/spill_scan.py compiles the device code to gfx950 assembly and counts them."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.setdefault("TMPDIR", "/tmp")
    import spill_scan
    res = spill_scan.scan()
    walks = [e for e in res if e["name"].startswith("k_path_wavefront<USE_BVH=1")]
    packets = [e for e in res if e["name"].startswith("stage_camera_packets_call")]
    assert len(walks) == 4 and len(packets) == 2, [e["name"] for e in res]   # pair nodes x streamed path state; NT on / off
    for e in walks + packets:
        assert e["max_loop_depth"] >= e["hot_depth"], e["name"]               # (the scan saw the loops it is meant to check)
        assert e["scratch_in_hot_loops"] == 0, (e["name"], e["hot_list"])
    for e in walks:
        assert e["between_barriers"] is not None and e["between_barriers"] <= 2, (e["name"], e["between_barriers"])
