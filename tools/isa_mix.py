#!/usr/bin/env python
"""Static instruction mix of the gfx950 kernels: share of f64 VALU instructions among all VALU instructions.

Compiles the .hip sources to ISA text with the library's own flags (hipcc -S --cuda-device-only, no GPU needed) and
counts, per kernel, VALU instructions (v_*) and the f64 ones (v_*_f64, conversions from / to f64, v_ldexp_f64,
v_cmp_*_f64, v_trunc_f64 ...). The chain kernels spend their time in one loop whose body is almost all of the kernel
text, so the static share is what bench.py multiplies the SQ_INSTS_VALU counter of a run with to price the kernel
against the vector-FP64 issue roof. Output: JSON {kernel: {valu, f64_valu, f64_share, vgprs, sgprs, lds}}.

usage: python tools/isa_mix.py [-o profiles/r02_isa_mix.json]
"""
import argparse
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "point_cloud_viewer_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "--cuda-device-only", "-S"]
F64 = re.compile(r"^v_\w*f64\w*|^v_cvt_\w*f64|^v_cvt_f64_\w+|^v_ldexp_f64|^v_frexp_\w*f64|^v_rcp_f64|^v_div_\w*f64|^v_trig_preop_f64")
# classes of the issue-cost model (tools/f64_rate.hip measures one representative of each): f64 add / mul / fma; conversions
# from / to f64; the other f64-pipe instructions (compare, min / max, trunc ...); 32-bit compares (they write a lane mask: as
# slow as an f64 instruction); everything else (32-bit ALU, moves, selects)
F64_ARITH = re.compile(r"^v_(add|mul|fma|fmac)_f64")
CVT = re.compile(r"^v_cvt_")
CMP32 = re.compile(r"^v_cmpx?_\w+_(u32|i32|u16|i16|f32|u64|i64)")


def demangle(name):
    try:
        out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        out = name
    out = re.sub(r"\(anonymous namespace\)::", "", out)
    return re.sub(r"\(.*", "", out).replace("void ", "")


def mix_of(source):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + [source, "-o", asm], check=True,
                       stderr=subprocess.DEVNULL, cwd=CSRC)
        text = open(asm).read()
    out = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = demangle(m.group(1))
            out[cur] = {"valu": 0, "f64_valu": 0, "f64_arith": 0, "cvt": 0, "f64_other": 0, "cmp32": 0}
            continue
        if cur is None:
            continue
        s = line.strip()
        if s.startswith("s_endpgm"):
            cur = None
            continue
        op = s.split()[0] if s else ""
        if op.startswith("v_"):
            out[cur]["valu"] += 1
            if F64.match(op):
                out[cur]["f64_valu"] += 1
            if F64_ARITH.match(op):
                out[cur]["f64_arith"] += 1
            elif CVT.match(op):
                out[cur]["cvt"] += 1
            elif F64.match(op):
                out[cur]["f64_other"] += 1
            elif CMP32.match(op):
                out[cur]["cmp32"] += 1
    for name, meta in re.findall(r"\.name:\s+(_Z\w+)\n((?:\s+\.\w+:.*\n)+)", text):
        k = demangle(name)
        if k in out:
            for key, field in (("vgprs", "vgpr_count"), ("sgprs", "sgpr_count"), ("lds", "group_segment_fixed_size")):
                mm = re.search(r"\.%s:\s+(\d+)" % field, meta)
                if mm:
                    out[k][key] = int(mm.group(1))
    for v in out.values():
        v["f64_share"] = round(v["f64_valu"] / v["valu"], 4) if v["valu"] else 0.0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o", default=os.path.join(ROOT, "profiles", "r02_isa_mix.json"))
    ap.add_argument("sources", nargs="*", default=["pcv_encode.hip", "pcv_chain.hip"])
    a = ap.parse_args()
    res = {}
    for src in a.sources:
        res.update(mix_of(os.path.join(CSRC, src)))
    with open(a.o, "w") as f:
        json.dump({"note": "static VALU mix of the compiled gfx950 kernels (tools/isa_mix.py)", "kernels": res}, f, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["valu"])[:12]:
        print(f"{k[:70]:70s} valu {v['valu']:5d}  f64 {v['f64_valu']:5d}  share {v['f64_share']:.3f}  vgprs {v.get('vgprs')}")


if __name__ == "__main__":
    main()
