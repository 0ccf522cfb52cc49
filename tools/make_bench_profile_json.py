#!/usr/bin/env python
"""Turn the per-kernel summaries of tools/profile_bench.sh into the two small JSON files bench.py quotes:
  <tag>_bench_traffic.json  {"per_launch": {kernel: HBM bytes}}            (2 x FETCH_SIZE + WRITE_SIZE, KiB counters)
  <tag>_bench_valu.json     {"per_launch": {kernel: {valu_insts_per_point, f64_valu_insts_per_point, f64_share, ...}}}
f64 arithmetic instructions come from the SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 counters of the run; the static f64
share of the kernel text (tools/isa_mix.py) is listed next to it for comparison (compares, min/max, conversions and
truncations are f64-pipe instructions the four counters do not include)."""
import argparse
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--points", type=int, default=100_000_000)
    a, _ = ap.parse_known_args()
    sys.path.insert(0, ROOT)
    from bench import build_hash  # the same hash bench.py recomputes: ties the profile to the library it was taken from
    stamp = {"build_hash": build_hash(),
             "git_head": subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None}
    with open(a.tag + "_kernel_stats_traffic.json") as f:
        tr = json.load(f)
    with open(a.tag + "_bench_traffic.json", "w") as f:
        json.dump(dict(stamp, note=tr["note"], per_launch=tr["bytes_per_launch"]), f, indent=1)
    mix_path = a.tag + "_isa_mix.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), "-o", mix_path], check=False,
                   stdout=subprocess.DEVNULL)
    mix = json.load(open(mix_path))["kernels"] if os.path.exists(mix_path) else {}
    per = {}
    def rows(path):
        """kernel names may contain commas (template arguments): split the numeric columns off from the right"""
        lines = open(path).read().splitlines()
        head = lines[0].split(",")
        for line in lines[1:]:
            parts = line.rsplit(",", len(head) - 1)
            yield dict(zip(head, parts))

    for row in rows(a.tag + "_f64.csv"):
        name = row["kernel"].split("<")[0]
        waves = float(row["SQ_WAVES"]) or 1.0
        valu = float(row["SQ_INSTS_VALU"])
        arith = sum(float(row.get(k, 0) or 0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        trans = float(row.get("SQ_INSTS_VALU_TRANS_F64", 0) or 0)
        cvt = float(row.get("SQ_INSTS_VALU_CVT", 0) or 0)
        f64 = arith + trans
        if name in per and per[name]["_valu"] >= valu:
            continue  # several launches with different shapes: keep the big one
        static = next((v for k, v in mix.items() if k.split("<")[0] == name and k == row["kernel"].replace("unsigned int", "unsigned int")), None)
        if static is None:
            static = next((v for k, v in mix.items() if k.split("<")[0] == name), {})
        # sustained clock during THIS dispatch: GRBM_GUI_ACTIVE (summed over the 8 XCDs by the rocpd view) / its duration
        gui, ns = float(row.get("GRBM_GUI_ACTIVE", 0) or 0), float(row.get("ns_under_counters", 0) or 0)
        clock = (gui / ns) if gui and ns else None
        if clock and clock > 4.0:
            clock /= 8.0
        # the instructions that are neither f64 add / mul / fma nor conversions, split by the static mix of the kernel text into
        # f64-pipe ones (compare, min / max, trunc), 32-bit compares and plain 32-bit instructions
        rest_static = max(1, static.get("valu", 0) - static.get("f64_arith", 0) - static.get("cvt", 0)) if static else 1
        per[name] = {"_valu": valu, "kernel": row["kernel"],
                     "valu_insts_per_point": round(valu * 64.0 / a.points, 2),
                     "f64_valu_insts_per_point": round(f64 * 64.0 / a.points, 2),
                     "f64_arith_insts_per_point": round(arith * 64.0 / a.points, 2),
                     "cvt_insts_per_point": round(cvt * 64.0 / a.points, 2),
                     "rest_insts_per_point": round((valu - arith - trans - cvt) * 64.0 / a.points, 2),
                     "rest_static_share_f64_other": round(static.get("f64_other", 0) / rest_static, 4) if static else None,
                     "rest_static_share_cmp32": round(static.get("cmp32", 0) / rest_static, 4) if static else None,
                     "sustained_clock_GHz": None if clock is None else round(clock, 3),
                     "f64_share": round(f64 / valu, 4) if valu else 0.0,
                     "f64_share_static_isa": static.get("f64_share"),
                     "valu_per_wave": round(valu / waves, 1)}
    for v in per.values():
        v.pop("_valu")
    with open(a.tag + "_bench_valu.json", "w") as f:
        json.dump(dict(stamp, note="per launch: SQ_INSTS_VALU and SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of one rocprofv3 --pmc pass "
                                    "(wave instructions x 64 lanes / points); f64_share_static_isa from tools/isa_mix.py",
                       points=a.points, per_launch=per), f, indent=1)
    print(json.dumps(per, indent=1)[:1500])


if __name__ == "__main__":
    main()
