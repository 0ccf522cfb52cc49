#!/usr/bin/env python
"""Per-kernel averages of every counter in one or more rocprofv3 rocpd databases (one --pmc pass each) -> one JSON:
{kernel: {counter: average per launch, "_launches": n, "_avg_us": duration}}; kernels below --min-us are dropped."""
import argparse
import json
import re
import sqlite3
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(<[^>(]*>)?)", name)
    s = m.group(1) if m else name
    return s.replace("unsigned long", "u64").replace("unsigned int", "u32")[:70]


ap = argparse.ArgumentParser()
ap.add_argument("dbs", nargs="+")
ap.add_argument("-o", required=True)
ap.add_argument("--min-us", type=float, default=20.0)
ap.add_argument("--note", default="")
a = ap.parse_args()
out = defaultdict(dict)
for db in a.dbs:
    try:
        cur = sqlite3.connect(db).cursor()
        acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))
        for name, cn, val, dur in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
            c = acc[short(name)][cn]
            c[0] += 1
            c[1] += val
            c[2] += dur or 0.0
    except Exception as e:  # noqa: BLE001
        print("skipping", db, e)
        continue
    for k, cs in acc.items():
        for cn, (n, tot, dur) in cs.items():
            if n and dur / n / 1e3 >= a.min_us:
                out[k][cn] = tot / n
                out[k].setdefault("_launches", n)
                out[k]["_avg_us_under_counters"] = round(dur / n / 1e3, 1)
res = {"note": a.note or "averages per launch of separate rocprofv3 --pmc passes (kernels serialised by the profiler; durations under "
                         "counter collection are longer than in a plain run)", "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1].get("_avg_us_under_counters", 0)))}
json.dump(res, open(a.o, "w"), indent=1)
for k, v in list(res["kernels"].items())[:8]:
    print(k, json.dumps(v))
