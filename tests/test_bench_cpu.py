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


def _recorded_full_record():
    """The whole record of a default run of an earlier round (22 KB as one line: what the driver failed to parse in round 5)."""
    import json
    for line in open(os.path.join(ROOT, "profiles", "r05_bench_100M.json")):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in profiles/r05_bench_100M.json")


def test_final_line_is_small(tmp_path, monkeypatch, capsys):
    """the LAST stdout line is the line of record and stays under 4 KB whatever the legs carry; it holds the
    contract fields, roofline, cpu_baseline and one parity verdict per BASELINE config; the whole record goes to a file."""
    import json
    out = _recorded_full_record()
    assert len(json.dumps(out)) > 16_000
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "gpurun_out")
    bench.emit(out)
    lines = capsys.readouterr().out.strip().splitlines()
    assert len(lines) == 2
    detail, last = lines
    assert len(last) < 4096 and len(detail) < 8192 and len(detail) + len(last) < 12_000
    rec = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in rec, k
    assert rec["value"] == out["value"] and rec["ms_per_step"] == out["ms_per_step"]
    assert rec["config"]["workload"].startswith("BASELINE config 2") and "model" not in rec["config"]
    r = rec["roofline"]
    assert r["frac"] == out["roofline"]["frac"] and r["traffic"] == out["roofline"]["traffic"] and r["hbm_view"]["frac"]
    assert r["kernel"] == "chain_pass_kernel"  # the symbol rocprofv3 lists, not the library's event slot
    assert set(rec["kernel_ms_per_step"]) >= {"chain_pass_kernel", "downsweep_rec12_kernel", "promote_settle_leaf_kernel",
                                              "downsweep_settle_kernel", "aabb_partial_kernel"}
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["cores"] == out["cpu_baseline"]["cores"]
    assert rec["encode_sort"]["frac_of_8TBps"] == round(out["encode_sort"]["GB/s"] / 8000.0, 4)
    p = rec["parity"]
    assert p["config2"] == {"ok": True, "mismatching_nodes": 0, "nodes": 6073, "tree_digest": out["tree_digest"]}
    assert all(p[k]["ok"] for k in ("config1", "config4", "config5", "intensity", "sharded"))
    whole = json.loads(open(tmp_path / "bench_detail.json").read())
    assert whole == out and json.loads(open(tmp_path / "gpurun_out" / "bench_detail.json").read()) == out
    assert "bench_detail" in json.loads(detail)
    # a leg that failed is a verdict, not a crash of the line; a record without legs (N > 1, --no-legs) still prints
    out2 = dict(out, config5={"error": "RuntimeError: " + "x" * 500}, query=None, sharded=None, intensity=None, config1=None)
    rec2 = bench.final_line(out2)
    assert rec2["parity"]["config5"]["ok"] is False and rec2["parity"]["config4"] is None
    # pathological growth (every string 10 x longer) still fits: the line sheds its optional objects
    fat = json.loads(json.dumps(out))
    fat["config"]["workload"] *= 10
    fat["cpu_baseline"]["sample"] *= 10
    fat["kernel_ms_per_step"] = {f"kernel_with_a_long_name_{i}": 1.0 for i in range(200)}
    assert len(json.dumps(bench.final_line(fat))) <= bench.FINAL_LINE_LIMIT
