#!/usr/bin/env python
"""Merge tools/f64_rate.bin's timings with the GRBM_GUI_ACTIVE pass of the same binary: SIMD cycles per wave-instruction of every
class at the clock the part really ran (GUI_ACTIVE / kernel duration), not at the nominal 2.4 GHz."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

plain, db, out = sys.argv[1:4]
rows = [json.loads(l) for l in open(plain) if l.startswith("{")]
clk = defaultdict(list)
try:
    cur = sqlite3.connect(db).cursor()
    for name, val, dur in cur.execute("select kernel_name,value,duration from counters_collection where counter_name='GRBM_GUI_ACTIVE'"):
        m = re.search(r"rate_kernel<(\d+)>", name)
        if m and dur and dur > 600_000:  # the long launches only (ns; the warm-up launch is a tenth of them)
            clk[int(m.group(1))].append((val, dur))
except Exception as e:  # noqa: BLE001
    print("no counter pass:", e)
res = {"note": "wave64 VALU issue cost per instruction class on this MI355X: 2048 workgroups x 256 lanes (8 waves per SIMD), 5 000 x 32 "
               "back-to-back instructions of one kind per lane over 8 independent chains; cycles = sustained clock x time x 1024 SIMDs / "
               "wave-instructions; sustained clock = GRBM_GUI_ACTIVE / kernel duration of a rocprofv3 --pmc pass of the same binary "
               "(raw counter value / 8 XCDs where the pass reports the sum over the XCDs)", "per_instruction": [], "per_class": {}}
cls = defaultdict(list)
for r in rows:
    c = clk.get(r["op"])
    ghz = raw = None
    if c:
        raw = sum(v / d for v, d in c) / len(c)  # cycles per ns
        ghz = raw / 8 if raw > 4.0 else raw
    cyc_nom = r["cycles_at_2.4GHz"]
    cyc = cyc_nom * ghz / 2.4 if ghz else None
    res["per_instruction"].append(dict(r, sustained_clock_GHz=None if ghz is None else round(ghz, 3), gui_active_per_ns_raw=None if raw is None else round(raw, 3),
                                       cycles_at_sustained_clock=None if cyc is None else round(cyc, 3)))
    cls[r["class"]].append((cyc_nom, cyc, ghz))
for k, v in cls.items():
    res["per_class"][k] = {"cycles_at_2.4GHz": round(sum(a for a, _, _ in v) / len(v), 3),
                           "cycles_at_sustained_clock": None if any(b is None for _, b, _ in v) else round(sum(b for _, b, _ in v) / len(v), 3),
                           "sustained_clock_GHz": None if any(g is None for _, _, g in v) else round(sum(g for _, _, g in v) / len(v), 3)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res["per_class"], indent=1))
for r in res["per_instruction"]:
    print(r["inst"], r["ms"], r["cycles_at_2.4GHz"], r["sustained_clock_GHz"], r["cycles_at_sustained_clock"])
