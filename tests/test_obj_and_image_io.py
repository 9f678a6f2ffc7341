"""§8(f) rows: OBJ/MTL loader (Scene::from_obj, src/main.rs:414-530) and image output (src/main.rs:1395-1419)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from hijiki_amd import abi, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_OBJ = "/root/reference/scenes/cbox/cbox.obj"

OBJ = """# test scene
mtllib test.mtl
o floor
v -1 0 -1
v 1 0 -1
v 1 0 1
v -1 0 1
vn 0 1 0
vt 0 0
vt 1 0
vt 1 1
vt 0 1
usemtl gray
f 1/1/1 2/2/1 3/3/1 4/4/1
o lamp
v -0.2 2 -0.2
v 0.2 2 -0.2
v 0.2 2 0.2
vn 0 -1 0
usemtl light_main
f -3//2 -2//2 -1//2
o shiny
v 0 0.5 0
v 0.5 0.5 0
v 0 1 0
v 0.5 1 0
vn 0 0 1
usemtl mirror_a
f 8//3 9//3 10//3
usemtl glass_b
f 9//3 11//3 10//3
o orphan
v 5 5 5
v 6 5 5
v 5 6 5
usemtl not_in_the_library
f 12//3 13//3 14//3
"""
MTL = """newmtl gray
Kd 0.5 0.25 0.125
Ke 0 0 0
newmtl light_main
Kd 0 0 0
Ke 7.5 8.5 9.5
newmtl mirror_a
Kd 1 1 1
newmtl glass_b
Kd 1 1 1
Ni 1.5
"""


@pytest.fixture()
def obj_path(tmp_path):
    (tmp_path / "test.obj").write_text(OBJ)
    (tmp_path / "test.mtl").write_text(MTL)
    return str(tmp_path / "test.obj")


def test_from_obj_semantics(obj_path):
    cs = host.Scene.from_obj(obj_path).compile()
    d = cs.desc
    # quad fan-triangulated into 2, lamp 1, mirror 1, glass 1; the `orphan` model names a material the library
    # lacks (material_id None): faces dropped, vertices kept (src/main.rs:476-479)
    assert d.num_triangles == 5
    # (v,vt,vn) re-indexing per model: 4 + 3 + (3 + 3: a material change starts a new model) + 3 orphan vertices
    assert d.num_vertices == 4 + 3 + 3 + 3 + 3
    tags = (cs.materials >> 24).tolist()
    assert tags == [abi.MAT_DIFFUSE, abi.MAT_DIFFUSE, abi.MAT_EMISSIVE, abi.MAT_MIRROR, abi.MAT_DIELECTRIC]
    assert tuple(d.diffuse[0].color) == (0.5, 0.25, 0.125)
    assert tuple(d.emissive[0].power) == (7.5, 8.5, 9.5)              # from the unknown MTL statement `Ke` (main.rs:434)
    assert d.dielectric[0].eta == 1.5 and tuple(d.dielectric[0].extinction) == (0, 0, 0)
    v = cs.vertices
    np.testing.assert_array_equal(v[0], [-1, 0, -1, 0, 0, 1, 0, 0])    # pos, u, normal, v
    np.testing.assert_array_equal(v[2], [1, 0, 1, 1, 0, 1, 0, 1])
    np.testing.assert_array_equal(v[4][[3, 7]], [0, 0])                # no vt -> uv (0, 0) (main.rs:466)
    np.testing.assert_array_equal(cs.triangles[:2], [[0, 1, 2], [0, 2, 3]])   # fan
    np.testing.assert_array_equal(cs.triangles[2], [4, 5, 6])                  # negative (relative) indices
    # hard-coded camera (main.rs:417-425)
    assert abs(d.camera.fov - 27.7) < 1e-6 and abs(d.camera.position[2] - 5.41) < 1e-6
    assert d.num_emitters == 1


def test_from_obj_errors(tmp_path):
    (tmp_path / "a.obj").write_text("o t\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl x\nf 1 2 3\n")
    with pytest.raises(abi.HijikiError) as e:
        host.Scene.from_obj(str(tmp_path / "a.obj"))                   # vertices without normals: unwrap() panic upstream
    assert "normal" in str(e.value)
    with pytest.raises(abi.HijikiError):
        host.Scene.from_obj(str(tmp_path / "missing.obj"))
    (tmp_path / "l.obj").write_text("mtllib l.mtl\no t\nv 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nusemtl light\nf 1//1 2//1 3//1\n")
    (tmp_path / "l.mtl").write_text("newmtl light\nKd 0 0 0\n")
    with pytest.raises(abi.HijikiError) as e:
        host.Scene.from_obj(str(tmp_path / "l.obj"))                   # light without Ke: unwrap() panic upstream
    assert "Ke" in str(e.value)


def test_put_cbox_spheres(obj_path):
    s = host.Scene.from_obj(obj_path)
    s.put_cbox_spheres()
    cs = s.compile()
    assert cs.desc.num_spheres == 2
    np.testing.assert_allclose(cs.spheres, [[-0.4214, 0.3321, -0.28, 0.3263], [0.4458, 0.3321, 0.3767, 0.3263]], rtol=1e-6)
    # live upstream code: mirror + checkerboard-diffuse (main.rs:1464-1482)
    assert (cs.materials[:2] >> 24).tolist() == [abi.MAT_MIRROR, abi.MAT_DIFFUSECBOARD]
    cb = cs.desc.diffusecb[0]
    assert (cb.scale_u, cb.scale_v) == (np.float32(0.1), np.float32(0.2))


@pytest.mark.skipif(not os.path.exists(REF_OBJ), reason="reference checkout not present (GPU box)")
def test_reference_cbox_counts(oracle):
    """The reference's own scene through the loader: SURVEY.md F8 / §8: 6332 triangles, 3668 vertices after
    (v,vt,vn) re-indexing, 12663 BVH nodes, 2 emitter triangles, 5 diffuse + 1 emissive materials."""
    cs = host.Scene.from_obj(REF_OBJ).compile()
    d = cs.desc
    assert (d.num_triangles, d.num_vertices, d.num_bvh_nodes, d.num_emitters) == (6332, 3668, 12663, 2)
    assert (d.num_diffuse, d.num_emissive) == (5, 1) and tuple(d.emissive[0].power) == (15, 15, 15)
    # material order of cbox.mtl = tobj index: floor, light, porcelain, wall_blue, wall_gray, wall_red (Appendix E)
    np.testing.assert_allclose(tuple(d.diffuse[0].color), (0.455928, 0.446495, 0.427629), rtol=1e-6)
    np.testing.assert_allclose(tuple(d.diffuse[4].color), (0.63, 0.065, 0.05), rtol=1e-6)
    lo, hi = cs.vertices[:, 0:3].min(0), cs.vertices[:, 0:3].max(0)
    np.testing.assert_allclose(lo, [-1, 0, -1.04], atol=1e-5)
    np.testing.assert_allclose(hi, [1, 1.59, 0.99], atol=1e-5)
    acc, ctr, _ = oracle.render_blocks(cs, host.make_blocks(128, 128, 2, 1), 128, 128, nthreads=4)
    img = oracle.resolve(acc)
    assert np.isfinite(img).all() and 0.05 < img.mean() < 1.0


def read_exr(path):
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and struct.unpack_from("<i", b, 4)[0] == 2
    p, attrs = 8, {}
    while b[p] != 0:
        e = b.index(0, p)
        name = b[p:e].decode()
        p = e + 1
        e = b.index(0, p)
        typ = b[p:e].decode()
        p = e + 1
        size = struct.unpack_from("<i", b, p)[0]
        attrs[name] = (typ, b[p + 4:p + 4 + size])
        p += 4 + size
    p += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    chans = []
    q, cl = 0, attrs["channels"][1]
    while cl[q] != 0:
        e = cl.index(0, q)
        chans.append((cl[q:e].decode(), struct.unpack_from("<i", cl, e + 1)[0]))
        q = e + 1 + 16
    offsets = struct.unpack_from(f"<{h}Q", b, p)
    img = np.zeros((h, w, len(chans)), np.float32)
    for y, off in enumerate(offsets):
        yy, size = struct.unpack_from("<2i", b, off)
        assert yy == y and size == w * len(chans) * 4
        img[y] = np.frombuffer(b, np.float32, w * len(chans), off + 8).reshape(len(chans), w).T
    return img, chans


def test_exr_and_pfm_round_trip(tmp_path):
    rgb = np.random.default_rng(3).random((7, 13, 3)).astype(np.float32) * 4
    rgb[2, 5] = (np.inf, 0, np.nan)            # rgb/w of an untouched pixel is 0/0: must survive untouched
    host.write_image(tmp_path / "a.exr", rgb)
    img, chans = read_exr(tmp_path / "a.exr")
    assert chans == [("B", 2), ("G", 2), ("R", 2)]              # three FLOAT channels, sorted (main.rs:1411-1413)
    assert (img[..., ::-1].view(np.uint32) == rgb.view(np.uint32)).all()
    host.write_image(tmp_path / "a.pfm", rgb)
    raw = open(tmp_path / "a.pfm", "rb").read()
    head = b"PF\n13 7\n-1.0\n"
    assert raw.startswith(head)
    back = np.frombuffer(raw, np.float32, offset=len(head)).reshape(7, 13, 3)[::-1]
    assert (back.view(np.uint32) == rgb.view(np.uint32)).all()


def test_exr_file_is_the_published_scanline_layout_byte_for_byte(tmp_path):
    """Independent of read_exr above: the whole file of a 2 x 2 image, assembled here by hand from the OpenEXR file-layout
    document (magic 20000630, version 2 without flag bits = single-part scan-line file with short names; header =
    attributes `name\\0 type\\0 int32 size, value`, the eight REQUIRED attributes of the format with their standard types
    and sizes, terminated by a null byte; line offset table of one uint64 per scan line for NO_COMPRESSION; each line
    chunk = int32 y, int32 byte count, then the channels in alphabetical order, each w floats).  What the reference
    writes through the openexr crate (src/main.rs:1402-1419: 3 x FLOAT "R", "G", "B" scan lines) has this layout."""
    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value
    def chan(name):          # name\0, pixel type (2 = FLOAT), pLinear + 3 reserved bytes, xSampling, ySampling
        return name.encode() + b"\0" + struct.pack("<i", 2) + b"\0\0\0\0" + struct.pack("<ii", 1, 1)
    w = h = 2
    rgb = np.arange(12, dtype=np.float32).reshape(h, w, 3) + 0.5
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    header = (bytes([0x76, 0x2F, 0x31, 0x01]) + struct.pack("<i", 2)
              + attr("channels", "chlist", chan("B") + chan("G") + chan("R") + b"\0")
              + attr("compression", "compression", b"\0")
              + attr("dataWindow", "box2i", box)
              + attr("displayWindow", "box2i", box)
              + attr("lineOrder", "lineOrder", b"\0")
              + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
              + attr("screenWindowCenter", "v2f", struct.pack("<ff", 0.0, 0.0))
              + attr("screenWindowWidth", "float", struct.pack("<f", 1.0))
              + b"\0")
    line = 8 + w * 3 * 4
    table = b"".join(struct.pack("<Q", len(header) + 8 * h + y * line) for y in range(h))
    chunks = b""
    for y in range(h):
        chunks += struct.pack("<ii", y, w * 3 * 4)
        for c in (2, 1, 0):                                       # B, G, R planes of the line
            chunks += rgb[y, :, c].astype("<f4").tobytes()
    host.write_image(tmp_path / "tiny.exr", rgb)
    got = open(tmp_path / "tiny.exr", "rb").read()
    assert got == header + table + chunks
    # the required attributes, by the names and types every OpenEXR reader looks up
    for name, typ in (("channels", "chlist"), ("compression", "compression"), ("dataWindow", "box2i"), ("displayWindow", "box2i"),
                      ("lineOrder", "lineOrder"), ("pixelAspectRatio", "float"), ("screenWindowCenter", "v2f"), ("screenWindowWidth", "float")):
        assert name.encode() + b"\0" + typ.encode() + b"\0" in got[:len(header)]


def test_png_is_read_back_by_an_independent_decoder(tmp_path):
    """Pillow (not part of this repo) decodes the preview PNG to the same 8-bit values the sRGB transfer function gives."""
    PIL = pytest.importorskip("PIL.Image")
    rgb = np.random.default_rng(9).random((33, 47, 3)).astype(np.float32)
    host.write_image(tmp_path / "b.png", rgb)
    im = PIL.open(tmp_path / "b.png")
    assert im.size == (47, 33) and im.mode == "RGB"
    got = np.asarray(im).astype(np.int32)
    v = rgb.astype(np.float64)
    want = np.where(v <= 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055) * 255
    assert np.abs(got - want).max() <= 0.51


def test_png_preview_image(tmp_path):
    """8-bit sRGB PNG (what the reference's preview window shows): decoded with zlib here, checked against the sRGB
    transfer function, every chunk CRC verified; a frame larger than one stored deflate block."""
    import struct
    import zlib
    h, w = 90, 300                                     # 81 KB of scan lines: two stored blocks
    rgb = np.random.default_rng(4).random((h, w, 3)).astype(np.float32) * 1.2
    rgb[0, 0] = (np.nan, -1.0, 7.0)
    host.write_image(tmp_path / "a.png", rgb)
    raw = open(tmp_path / "a.png", "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
    lines = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert (lines[:, 0] == 0).all()
    got = lines[:, 1:].reshape(h, w, 3).astype(np.int32)
    v = np.clip(np.nan_to_num(rgb.astype(np.float64), nan=0.0), 0.0, 1.0)
    want = np.where(v <= 0.0031308, 12.92 * v, 1.055 * v ** (1 / 2.4) - 0.055) * 255
    assert np.abs(got - want).max() <= 0.51
    assert tuple(got[0, 0]) == (0, 0, 255)


def test_cli_usage_and_flags():
    exe = os.path.join(ROOT, "hijiki_amd", "bin", "hijiki-hip")
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "<scene>" in r.stderr
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0
    for flag in ("--put-cbox-spheres", "--use-bvh", "--width", "--height", "--present-interval", "--sample-count", "--output-image"):
        assert flag in r.stderr                                   # the reference's Opt (main.rs:1426-1456)


@pytest.mark.gpu
def test_cli_renders_like_the_library(tmp_path, obj_path):
    from hijiki_amd import device
    exe = os.path.join(ROOT, "hijiki_amd", "bin", "hijiki-hip")
    out = str(tmp_path / "o.pfm")
    r = subprocess.run([exe, "--use-bvh", "--put-cbox-spheres", "-w", "160", "-h", "128", "-s", "3", "--seed", "9", "-o", out,
                        obj_path], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Integrated 61440 rays" in r.stdout and "Built BVH with" in r.stdout
    raw = open(out, "rb").read()
    head = b"PF\n160 128\n-1.0\n"
    got = np.frombuffer(raw, np.float32, offset=len(head)).reshape(128, 160, 3)[::-1]
    s = host.Scene.from_obj(obj_path)
    s.put_cbox_spheres()
    cs = s.compile()
    with device.Renderer(0) as rr:
        rr.upload_scene(cs)
        rr.create_framebuffer(160, 128)
        rr.render_frame(3, 9)
        want = rr.resolve()
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # ... and like the ORACLE: main() of the reference = from_obj + put_cbox_spheres + compile + render + rgb / w
    # (src/main.rs:414-530,1463-1483,1395-1400), restated on the CPU for the same ImageBlock list
    from oracle import hj_oracle
    acc, _, _ = hj_oracle.render_blocks(cs, host.make_blocks(160, 128, 3, 9), 160, 128)
    assert (got.view(np.uint32) == hj_oracle.resolve(acc).view(np.uint32)).all()
    # default traversal is the linear scan, as upstream (--use-bvh off)
    out2 = str(tmp_path / "o2.exr")
    r = subprocess.run([exe, "-w", "128", "-h", "128", "-s", "1", "-o", out2, obj_path], capture_output=True, text=True)
    assert r.returncode == 0 and os.path.getsize(out2) > 128 * 128 * 12
    # tree built on the GPU, preview image as PNG
    out3 = str(tmp_path / "o3.png")
    r = subprocess.run([exe, "--use-bvh", "--device-bvh", "-w", "96", "-h", "64", "-s", "2", "-o", out3, obj_path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Built BVH with" in r.stdout and open(out3, "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    # --device-bvh is the fast-start route (shapes only on the host, the tree built and left on the device, hj_scene_upload with
    # bvh == NULL): the same frame as that route through the library, bit for bit
    out4 = str(tmp_path / "o4.pfm")
    r = subprocess.run([exe, "--use-bvh", "--device-bvh", "-w", "96", "-h", "64", "-s", "2", "--seed", "4", "-o", out4, obj_path],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    head4 = b"PF\n96 64\n-1.0\n"
    got4 = np.frombuffer(open(out4, "rb").read(), np.float32, offset=len(head4)).reshape(64, 96, 3)[::-1]
    cs4 = host.Scene.from_obj(obj_path).compile(with_tree=False)
    with device.Renderer(0) as rr:
        nodes = rr.build_bvh(cs4, keep_on_device=True)
        assert f"Built BVH with {nodes} nodes" in r.stdout
        rr.upload_scene(cs4, device_tree=True)
        rr.create_framebuffer(96, 64)
        rr.render_frame(2, 4)
        assert (got4.view(np.uint32) == rr.resolve().view(np.uint32)).all()


@pytest.mark.gpu
def test_obj_loaded_scene_against_the_oracle(obj_path):
    """Scene::from_obj (+ --put-cbox-spheres) -> compile -> HIP frame, against the oracle on the same compiled scene: the
    loader's output (fan triangulation, re-indexed vertices, material mapping by name prefix, Ke light) feeds the hot path.
    Host tree and device-built tree, BVH walk and the CLI default linear scan."""
    from hijiki_amd import device
    from oracle import hj_oracle
    s = host.Scene.from_obj(obj_path)
    s.put_cbox_spheres()
    cs = s.compile()
    W, H, spp, seed = 192, 160, 4, 13
    blocks = host.make_blocks(W, H, spp, seed)
    want, ctr, _ = hj_oracle.render_blocks(cs, blocks, W, H)
    assert ctr["tri_tests"] > 0 and ctr["sphere_tests"] > 0 and ctr["nee_evals"] > 0
    with device.Renderer(0) as rr:
        rr.upload_scene(cs)
        rr.create_framebuffer(W, H)
        st = rr.render_frame(spp, seed)
        got = rr.read()
        assert (got.view(np.uint32) == want.view(np.uint32)).all()
        assert st["closest_rays"] == ctr["closest_calls"] and st["shadow_rays"] == ctr["shadow_calls"]
        o = device.default_opts()
        o.use_bvh = 0                                                   # the reference CLI's default (scene.glsl:134-158)
        lin, _, _ = hj_oracle.render_blocks(cs, blocks, W, H, opts=o)
        rr.clear()
        rr.render_frame(spp, seed, opts=o)
        assert (rr.read().view(np.uint32) == lin.view(np.uint32)).all()
        cs.set_bvh(rr.build_bvh(cs))                                    # --device-bvh
        want2, _, _ = hj_oracle.render_blocks(cs, blocks, W, H)
        rr.upload_scene(cs)
        rr.clear()
        rr.render_frame(spp, seed)
        assert (rr.read().view(np.uint32) == want2.view(np.uint32)).all()


@pytest.mark.gpu
def test_cli_progress_percentage(tmp_path):
    """The percentage the reference shows in the window title every --present-interval blocks (src/main.rs:1335-1340) goes
    to stderr, in the title's format, and ends at 100 %."""
    exe = os.path.join(ROOT, "hijiki_amd", "bin", "hijiki-hip")
    out = str(tmp_path / "p.pfm")
    r = subprocess.run([exe, "--use-bvh", "-w", "512", "-h", "512", "-s", "48", "--present-interval", "64", "-o", out,
                        "synthetic:cbox"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    marks = [m for m in r.stderr.replace("\r", "\n").split("\n") if "%" in m]
    total = 16 * 48                                                   # 4 x 4 blocks per pass
    assert len(marks) >= 2 and marks[-1].strip() == f"100.000% {total}/{total}"
    done = [int(m.split()[1].split("/")[0]) for m in marks]
    assert done == sorted(done) and all(m.split()[1].endswith(f"/{total}") for m in marks)
