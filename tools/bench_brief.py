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
