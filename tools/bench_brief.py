#!/usr/bin/env python
"""One screen of a bench.py JSON line: value, parity of every leg, kernel times."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
p = d.get("parity") or {}
print("config2:", d.get("value"), d.get("unit"), d.get("ms_per_step"), "ms | parity ok=%s mismatching=%s nodes=%s digest=%s" % (
    p.get("ok"), p.get("mismatching_nodes"), p.get("nodes"), d.get("tree_digest")))
print(" kernels:", {k.replace("_kernel", ""): v for k, v in (d.get("kernel_ms_per_step") or {}).items()})
print(" stages:", d.get("stage_ms"))
r = d.get("roofline") or {}
print(" roofline:", {k: r.get(k) for k in ("kernel", "bound", "achieved", "frac", "avg_launch_ms", "profile_matches_build")},
      "| largest hbm:", {k: (r.get("largest_hbm_kernel") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")})
print(" build_info:", d.get("build_info"))
print(" cpu:", d.get("cpu_baseline"))
e = d.get("end_to_end") or {}
print(" e2e:", e.get("Mpoints_per_s_incl_files"), (e.get("from_ply_file") or {}).get("Mpoints_per_s_incl_files"))
q = d.get("query")
if q:
    print("query:", q.get("error") or (q.get("value"), q.get("unit"), "parity", q.get("parity"), "visible", q.get("visible_nodes"),
                                       "K8", {k: q["roofline"].get(k) for k in ("achieved", "frac", "avg_launch_ms")}))
c = d.get("config5")
if c:
    cp = c.get("parity") or {}
    print("config5:", c.get("error") or (c.get("value"), c.get("ms_per_step"), "parity ok=%s mismatching=%s nodes=%s" % (
        cp.get("ok"), cp.get("mismatching_nodes"), cp.get("nodes")), c.get("kernel_ms_per_step"), c.get("build_info")))
sh = d.get("sharded")
if sh:
    w1 = sh.get("world1") or {}
    print("sharded:", sh.get("error") or ("ok=%s" % sh.get("ok"), "world1", w1.get("ms_per_step"), w1.get("stage_ms"), "digest_equal", w1.get("digest_equal"),
                                          "buckets", (w1.get("buckets") or {}).get("ms_per_step"),
                                          {k: (v.get("digest_equal"), v.get("nodes_built_twice"), v.get("imbalance_max_over_mean")) for k, v in (sh.get("virtual8") or {}).items()}))
it = d.get("intensity")
if it:
    ip = it.get("parity") or {}
    print("intensity:", it.get("error") or (it.get("points"), "colour", (it.get("color_only") or {}).get("ms_per_step"), "+intensity",
                                            (it.get("color_and_intensity") or {}).get("ms_per_step"), "cost", it.get("intensity_cost"),
                                            "parity ok=%s mismatching=%s" % (ip.get("ok"), ip.get("mismatching_nodes")),
                                            (it.get("color_and_intensity") or {}).get("kernel_ms_per_step")))
c1 = d.get("config1")
if c1:
    print("config1:", c1.get("error") or (c1.get("ok"), c1.get("nodes"), c1.get("exact_pipeline"), c1.get("single_chain")))
if q and not q.get("error"):
    print(" cull_nodes:", q.get("cull_nodes"))
if c and not c.get("error"):
    print(" config5 K8:", c.get("query_flags_roofline"))
print(" box:", (d.get("config") or {}).get("box"))
