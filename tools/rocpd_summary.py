#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (sqlite) outputs into a compact CSV/markdown for profiles/.

usage: rocpd_summary.py --stats DB [--fetch DB] [--write DB] [--filter anonymous] -o profiles/rNN_name
FETCH_SIZE/WRITE_SIZE are reported in KiB by rocprofv3; per MI355X_MICROARCH.md §HBM the gfx950 FETCH_SIZE of a
wide coalesced stream reads exactly 1/2 of the real bytes, so both the raw and the x2-corrected value are listed.
"""
import argparse
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>(]*>)?)", name)
    s = m.group(1) if m else name
    s = s.replace("unsigned long", "u64").replace("unsigned int", "u32")
    return s[:60]


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    return [(short(n), c, t, a, p) for n, c, t, a, p in rows]


def counter_avg(db, counter):
    cur = sqlite3.connect(db).cursor()
    acc = defaultdict(lambda: [0, 0.0])
    for name, val in cur.execute("select kernel_name,value from counters_collection where counter_name=?", (counter,)):
        a = acc[short(name)]
        a[0] += 1
        a[1] += val
    return {k: v[1] / v[0] for k, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--keep", default="kernel", help="substring a kernel name must contain to be listed")
    ap.add_argument("--sq", help="rocpd db of an SQ counter pass: writes <o>_sq.csv (per-kernel averages per launch)")
    ap.add_argument("--command", default="python bench.py (defaults: BASELINE config 2, 100 M points, 1 GPU)")
    ap.add_argument("-o", required=True)
    a = ap.parse_args()
    stats = [r for r in kernel_stats(a.stats) if a.keep in r[0] and not r[0].startswith("at::")]
    fetch = counter_avg(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = counter_avg(a.write, "WRITE_SIZE") if a.write else {}
    with open(a.o + ".csv", "w") as f:
        f.write("kernel,calls,total_us,avg_us,pct_of_gpu_time,avg_FETCH_SIZE_KiB_raw,avg_fetch_MB_x2_corrected,avg_WRITE_SIZE_KiB_raw,avg_write_MB\n")
        for n, c, t, av, p in stats:
            fk, wk = fetch.get(n), write.get(n)
            f.write(f"{n},{c},{t:.1f},{av:.1f},{p:.2f},{'' if fk is None else f'{fk:.0f}'},"
                    f"{'' if fk is None else f'{fk * 2 * 1024 / 1e6:.1f}'},{'' if wk is None else f'{wk:.0f}'},"
                    f"{'' if wk is None else f'{wk * 1024 / 1e6:.1f}'}\n")
    print(open(a.o + ".csv").read())
    if a.sq:
        cur = sqlite3.connect(a.sq).cursor()
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for name, cn, val, dur in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
            c = acc[short(name)][cn]
            c[0] += 1
            c[1] += val
            d = acc[short(name)]["ns_under_counters"]  # the dispatch's own duration in this pass (for GRBM_GUI_ACTIVE -> clock)
            if cn == "SQ_WAVES":
                d[0] += 1
                d[1] += dur or 0.0
        names = sorted({cn for k in acc.values() for cn in k})
        rows = []
        for k, c in acc.items():
            avg = {cn: (c[cn][1] / c[cn][0] if cn in c and c[cn][0] else 0.0) for cn in names}
            if avg.get("SQ_INSTS_VALU", 0) >= 1e6 and a.keep in k and not k.startswith("at::"):
                rows.append((k, max(v[0] for v in c.values()), avg))
        rows.sort(key=lambda r: -r[2].get("SQ_INSTS_VALU", 0))
        with open(a.o + "_sq.csv", "w") as f:
            f.write("kernel,launches," + ",".join(names) + ",valu_per_wave\n")
            for k, launches, avg in rows:
                per = avg.get("SQ_INSTS_VALU", 0) / max(avg.get("SQ_WAVES", 1), 1)
                f.write(",".join([k, str(launches)] + [f"{avg[cn]:.0f}" for cn in names] + [f"{per:.1f}"]) + "\n")
        print(open(a.o + "_sq.csv").read())
    # machine-readable HBM traffic per launch (bytes; FETCH_SIZE already doubled per the gfx950 correction) for bench.py
    import json
    traffic = {}
    for n, c, t, av, p in stats:
        fk, wk = fetch.get(n), write.get(n)
        if fk is not None and wk is not None:
            traffic[n.split("<")[0]] = max(traffic.get(n.split("<")[0], 0), round(fk * 2 * 1024 + wk * 1024))
    with open(a.o + "_traffic.json", "w") as f:
        json.dump({"note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB counters, separate rocprofv3 --pmc passes); "
                           "for kernels launched with several key widths the larger launch is listed",
                   "command": a.command, "bytes_per_launch": traffic}, f, indent=1)


if __name__ == "__main__":
    main()
