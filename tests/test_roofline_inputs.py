"""The roofline inputs bench.py quotes are regenerated from the rocprofv3 CSVs committed beside them."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("cfg", ["c2", "c3", "c4"])
def test_roofline_inputs_regenerate_from_committed_csv(cfg):
    tag = f"r02_{cfg}_"
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "roofline_inputs.py"), "build",
                                   os.path.join(ROOT, "profiles"), cfg, tag], text=True)
    fresh = json.loads(out)
    stored = json.load(open(os.path.join(ROOT, "profiles", tag + "roofline_inputs.json")))
    assert fresh == stored
    assert stored["counters"]["FETCH_SIZE"] > 0 and stored["counters"]["WRITE_SIZE"] > 0
    assert 0 < stored["limiter"]["valu_lanes_per_instruction"] <= 64
    if cfg == "c4":
        assert stored["scene_bytes_per_ray"] > 500          # nodes beyond the LDS copy + triangle records


def test_bench_byte_model_is_consistent():
    """bench.py's implemented-bytes model: coalesced reads are a part of it, and it scales linearly with the counters."""
    sys.path.insert(0, ROOT)
    import bench
    st = {"paths": 1000, "closest_rays": 3050, "shadow_rays": 1790, "hits": 2900, "unoccluded_shadow_rays": 1200}
    b = bench.implemented_bytes(st)
    assert 800 < b / st["paths"] < 1000                     # cbox: 880 B/path
    assert 0 < bench.coalesced_read_bytes(st) < b
    st2 = {k: 2 * v for k, v in st.items()}
    assert abs(bench.implemented_bytes(st2) - 2 * b) < 1e-6 * b
