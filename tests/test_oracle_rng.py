"""RNG known answers: SURVEY.md Appendix D (exact u32 arithmetic on shader/rand.glsl:2-20)."""
import ctypes as C
import os

import numpy as np

KAT = [  # seed, state after seedRng, next 4 randUint()
    (0, 0xC0A9496A, [0xD90BC8A8, 0xA3CD8C47, 0x5AE9C9C5, 0x19FA5D8D]),
    (1, 0x27922C9D, [0x22360E3D, 0x9DCA2765, 0xFDFB9536, 0x64FF4198]),
    (2, 0xC6793575, [0xFA2B46DE, 0xCCE93B66, 0x9B345A24, 0x1E6A919C]),
    (12345, 0x0DDEEC13, [0xDBC0639D, 0x21C6A0C4, 0x4E151F4B, 0x527D3F15]),
    (0xFFFFFFFF, 0x70F499D3, [0x9A1F8EB4, 0x12EE5150, 0xC0439B72, 0xD69DDE64]),
    (0xDEADBEEF, 0x572E7C2D, [0x8DD99F78, 0x78EECC03, 0x8CB16A34, 0x9F00E32F]),
    (61, 0x00000000, [0, 0, 0, 0]),
]
FLOATS = {0: (0.8478360772, 0.6398551464), 1: (0.1336373240, 0.6163658500), 12345: (0.8584043980, 0.1319370717)}


def py_wang(s):
    m = 0xFFFFFFFF
    s = ((s ^ 61) ^ (s >> 16)) & m
    s = (s * 9) & m
    s = (s ^ (s >> 4)) & m
    s = (s * 0x27D4EB2D) & m
    return (s ^ (s >> 15)) & m


def py_xorshift(x):
    m = 0xFFFFFFFF
    x ^= (x << 13) & m
    x ^= x >> 17
    x ^= (x << 5) & m
    return x & m


def test_known_answers(oracle):
    L = oracle.lib()
    for seed, state, draws in KAT:
        assert L.hjo_rng_seed(seed) == state == py_wang(seed)
        s = C.c_uint32(state)
        assert [L.hjo_rng_next(C.byref(s)) for _ in range(4)] == draws
    for seed, (f0, f1) in FLOATS.items():
        s = C.c_uint32(L.hjo_rng_seed(seed))
        assert abs(L.hjo_rng_float(C.byref(s)) - f0) < 1e-7
        assert abs(L.hjo_rng_float(C.byref(s)) - f1) < 1e-7


def test_python_model_agrees_on_random_seeds(oracle):
    L = oracle.lib()
    r = np.random.default_rng(0)
    for seed in r.integers(0, 2**32, 200, dtype=np.uint64):
        st = py_wang(int(seed))
        assert L.hjo_rng_seed(int(seed)) == st
        s = C.c_uint32(st)
        for _ in range(3):
            st = py_xorshift(st)
            assert L.hjo_rng_next(C.byref(s)) == st


def test_seed_61_is_the_fixed_point(oracle):
    L = oracle.lib()
    s = C.c_uint32(L.hjo_rng_seed(61))
    assert s.value == 0
    assert all(L.hjo_rng_float(C.byref(s)) == 0.0 for _ in range(8))


def test_uint_to_float_rounds_to_nearest_even(oracle):
    # float(0xFFFFFF7F) * 2^-32 = 0.99999994, float(0xFFFFFF80) * 2^-32 = 1.0 (Appendix D)
    L = oracle.lib()

    def inv(y):   # inverse of one xorshift32 step, to place a chosen output
        m = 0xFFFFFFFF
        # undo x ^= x << 5
        x = y
        for _ in range(7):
            x = y ^ ((x << 5) & m)
        y = x
        for _ in range(2):
            x = y ^ (x >> 17)
        y = x
        for _ in range(3):
            x = y ^ ((x << 13) & m)
        return x & m

    for target, expect in ((0xFFFFFF7F, np.float32(0.99999994)), (0xFFFFFF80, np.float32(1.0)), (0x80000000, np.float32(0.5))):
        s = C.c_uint32(inv(target))
        assert py_xorshift(s.value) == target
        assert np.float32(L.hjo_rng_float(C.byref(s))) == expect


def test_golden_kat_file_matches(oracle):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rng_kat.npz"))
    L = oracle.lib()
    for seed, st, draws in zip(g["seeds"], g["states"], g["draws"]):
        assert L.hjo_rng_seed(int(seed)) == int(st)
        s = C.c_uint32(int(st))
        assert [L.hjo_rng_next(C.byref(s)) for _ in range(4)] == [int(d) for d in draws]


def test_barycentric_reproduces_the_reference_bug(oracle):
    # rand.glsl:45-48: when u+v>1 the "flip" yields (1-v, v, 0): the sample lies on edge a-b
    L = oracle.lib()
    out = (C.c_float * 3)()
    on_edge = total = 0
    for seed in range(2000):
        s = C.c_uint32(L.hjo_rng_seed(seed))
        s2 = C.c_uint32(s.value)
        u, v = L.hjo_rng_float(C.byref(s2)), L.hjo_rng_float(C.byref(s2))
        L.hjo_barycentric(C.byref(s), out)
        assert s.value == s2.value      # exactly two draws
        if np.float32(u) + np.float32(v) > 1:
            total += 1
            assert abs(out[0] - (1 - v)) < 1e-6 and abs(out[1] - v) < 1e-6 and abs(out[2]) < 1e-6
            on_edge += 1
        else:
            assert out[0] == u and out[1] == v
    assert 800 < total < 1200 and on_edge == total


def test_cos_hemisphere_and_sphere_are_unit(oracle):
    L = oracle.lib()
    out = (C.c_float * 3)()
    zs = []
    for seed in range(500):
        s = C.c_uint32(L.hjo_rng_seed(seed * 7919 + 3))
        L.hjo_cos_hemisphere(C.byref(s), out)
        v = np.array(out[:], np.float64)
        assert abs(np.linalg.norm(v) - 1) < 1e-6 and v[2] >= 0
        zs.append(v[2])
        L.hjo_uniform_sphere(C.byref(s), out)
        assert abs(np.linalg.norm(np.array(out[:], np.float64)) - 1) < 1e-6
    assert abs(np.mean(zs) - 2 / 3) < 0.03   # E[cos] of a cosine-weighted hemisphere
