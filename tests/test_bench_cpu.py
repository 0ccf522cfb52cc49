"""bench.py's host-side pieces that the config-3 runs rely on (no GPU): the block-wise cloud generator must hand every
rank exactly its slice of the ONE global cloud, and the digest tables must compare octrees node by node."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_any_slice_of_the_config3_cloud_is_reproducible(monkeypatch):
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)  # the generator ends with a device sync
    monkeypatch.setattr(bench, "CLOUD_BLOCK", 1 << 16)  # small blocks: slices that start and end inside blocks
    dev = torch.device("cpu")
    total = 700_001
    x, y, z, rgb = bench.make_cloud_slice(torch, total, 0, total, seed=2, device=dev)
    for world in (1, 3, 8):
        parts = []
        for rank in range(world):
            lo, hi = rank * total // world, (rank + 1) * total // world
            parts.append(bench.make_cloud_slice(torch, total, lo, hi - lo, seed=2, device=dev))
        for k, whole in enumerate((x, y, z, rgb)):
            assert torch.equal(torch.cat([p[k] for p in parts]), whole), (world, k)
    other = bench.make_cloud_slice(torch, total, 0, 1000, seed=3, device=dev)
    assert not torch.equal(other[0], x[:1000])  # the seed matters
    assert float(x.min()) > -200.0 and float(x.max()) < 1200.0  # 64 clusters in a 1000 m cube, sigma <= 20 m
    h = (np.arange(5, dtype=np.int64) * 2654435761) & 0xFFFFFF  # colour = hash of the GLOBAL point index
    assert rgb[:5].numpy().tolist() == np.stack([(h >> 16) & 255, (h >> 8) & 255, h & 255], axis=1).tolist()


def test_digest_tables_compare_node_by_node():
    a = {"r": (10, 3, "x0", "c0", "i0"), "r1": (4, 2, "x1", "c1", "i1"), "r17": (0, 1, "e", "e", "e")}
    same = dict(reversed(list(a.items())))
    assert bench.compare_digests(a, same)["ok"] and bench.digest_of_digests(a) == bench.digest_of_digests(same)
    b = dict(a, r1=(4, 2, "x1", "DIFFERENT", "i1"))
    cmp_ = bench.compare_digests(a, b)
    assert not cmp_["ok"] and cmp_["mismatching_nodes"] == 1 and cmp_["first_mismatches"] == ["r1"]
    assert bench.digest_of_digests(a) != bench.digest_of_digests(b)
    c = {k: v for k, v in a.items() if k != "r17"}
    c["r2"] = (1, 1, "y", "y", "y")
    cmp_ = bench.compare_digests(a, c)
    assert (cmp_["missing_nodes"], cmp_["extra_nodes"], cmp_["mismatching_nodes"]) == (1, 1, 0) and not cmp_["ok"]


def test_build_hash_follows_the_sources(tmp_path, monkeypatch):
    h = bench.build_hash()
    assert len(h) == 16 and h == bench.build_hash()
    # a copy of csrc with one byte changed hashes differently
    import shutil
    src = os.path.join(ROOT, "point_cloud_viewer_amd", "csrc")
    dst = tmp_path / "point_cloud_viewer_amd" / "csrc"
    os.makedirs(dst)
    for name in os.listdir(src):
        if name.endswith((".hip", ".h", ".cpp", ".inc")) or name == "Makefile":
            shutil.copy(os.path.join(src, name), dst / name)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.build_hash() == h
    with open(dst / "pcv_sort.hip", "a") as f:
        f.write("// one more byte\n")
    assert bench.build_hash() != h


def test_valu_issue_model_prices_classes_with_the_measured_costs():
    """bench.valu_issue_model: dynamic per-class counts x the issue costs of this round's probe, against 1 024 SIMDs x the
    clock the counter pass saw; the two bounds bracket the mid figure, and a faster launch gives a higher fraction."""
    insts = {"valu_insts_per_point": 600.0, "f64_valu_insts_per_point": 190.0, "f64_arith_insts_per_point": 190.0,
             "cvt_insts_per_point": 36.0, "rest_insts_per_point": 374.0, "rest_static_share_f64_other": 0.4,
             "rest_static_share_cmp32": 0.06, "sustained_clock_GHz": 2.07}
    r = bench.valu_issue_model(insts, 100_000_000, 2.2)
    if r.get("frac") is None:  # no probe recorded for this round yet
        assert "note" in r
        return
    lo, hi = r["frac_bounds"]
    assert 0.0 < lo < r["frac"] < hi < 1.5
    c = r["cycles_per_class"]
    assert 3.5 < c["f64_add_mul_fma"] < 5.5 and 1.8 < c["b32_plain"] < 3.2 and c["b32_compare"] > c["b32_plain"]
    assert bench.valu_issue_model(insts, 100_000_000, 1.1)["frac"] > r["frac"] * 1.9
    # counters of an older profile (no per-class fields): the model says so instead of guessing
    assert bench.valu_issue_model({"valu_insts_per_point": 600.0}, 100_000_000, 2.2)["frac"] is None


def test_query_profile_is_quoted_only_for_the_running_build():
    p = bench.query_profile("0000000000000000")
    assert p.get("profile_matches_build") in (False, None) and "cull_nodes_kernel" not in p
    q = bench.query_profile(bench.build_hash())
    assert q.get("profile_matches_build") in (True, False, None)
