#!/usr/bin/env python
"""bench.py — octree-build throughput on MI355X (BASELINE.json metric: octree-build Mpoints/sec + HBM GB/s).

A "step" is one full pass of the hot path over one batch: pcv_build_octree with PCV_BUILD_COMPUTE_BBOX on a
device-resident cloud — K1 bounding box, path keys / topology, leaf encode, record sort, promotion/encode — producing
the finished node table and node-contiguous .xyz/.rgb bytes in HBM. Inputs are resident in HBM when the timed region
starts; nothing is read back except the node table.

Workload at N=1: BASELINE config 2 — 100 M synthetic Gaussian-cluster points (64 clusters in a 1000 m cube,
sigma in [1, 20] m), f64 SoA xyz + u8 rgb, resolution 1 mm; after the timed region the closed-form CPU oracle builds
the same cloud and every node of one more GPU build is compared with it byte for byte (`parity`).
With --gpus N>1: BASELINE config 3 — ONE cloud of 1 B points of the same distribution (seed 2), rank r holds the
contiguous slice [r, r + 1) * 1e9 / N of it (strong scaling; --points P gives every rank P points instead), points are
routed to their owner with ONE all-to-all (point_cloud_viewer_amd/distributed.py) and each rank builds its subtrees.
--virtual-ranks V runs that same path with V thread-ranks on ONE GPU (tests/thread_dist.py stands in for
torch.distributed) and compares the merged octree with a single-GPU build of the whole cloud (and, with --verify, with
the CPU oracle): the config-3 dress rehearsal for boxes with one GPU.

Other modes (never the driver's default): --verify forces the oracle comparison wherever it is off by default
(sharded path, --ecef ...); --ecef is BASELINE config 5; --query is BASELINE config 4 (frustum path) with parity of all
frusta; --config1 is BASELINE config 1 (CPU plumbing line, no GPU work timed).

Rank 0 writes the whole record to bench_detail.json, prints a trimmed copy of it and then, as the LAST stdout line, the
line of record (<= 4 KB: contract fields + roofline + cpu_baseline + one parity verdict per BASELINE config).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_VALU_PEAK_GINST = 39300  # vector FP64 78.6 TFLOP/s = 39.3 T FMA-instructions/s (SURVEY 8d "Roofline bound")
ROUND = "r06"

# algorithmic HBM bytes per point and launch (DESIGN.md "Kernels"); kernels bound by f64 VALU issue are marked
ALGO_BYTES = {
    "downsweep_kernel<u64>": 16.0,  # read 8 B key + write 8 B key
    "upsweep_kernel<u64>": 8.0,     # read 8 B key
    "aabb_partial_kernel": 24.0,    # read xyz f64
    "chain_keys_kernel": 28.0,      # read xyz f64 + write 4 B key (8 B when more than 10 levels are keyed)
    "leaf_encode_kernel": 24.0 + 3.0 + 20.0,  # read xyz + rgb, write rank + 16-byte payload
    "downsweep_kernel<u32>": 8.0,   # keys only: read 4 B + write 4 B
    "downsweep_rec_kernel": 2 * 4.0 + 2 * 16.0,  # rank r/w + 16-byte payload r/w
    "upsweep_kernel<u32>": 4.0,
    "promote_settle_kernel": 20.0 + 7.0 / 8.0 * 9.0,  # read record; 7 of 8 points write ~6 B xyz + 3 B rgb
    "promote_climb_kernel": 4.0 + (16.0 + 9.0) / 8.0,  # read ranks; every 8th point: payload in, xyz + rgb out
    "spec_encode_kernel": 24.0 + 3.0 + 20.0,  # single-chain pass: read xyz + rgb, write rank + 16-byte payload (+ kept codes of ~10 %)
    "rank_hist_kernel": 4.0,        # read ranks
    "spec_finalize_kernel": 8.0,    # rank read + write (+ payload patch of the points that take their kept codes)
    "upsweep_map_kernel": 8.0,      # rank read + mapped rank written back (+ the rare replay marks)
    # the record sort's second pass settling the leaves' points itself: record in; 7 of 8 points leave as ~6 B xyz + 3 B rgb, every
    # eighth as a 16-byte climber record (the few leaves it leaves to `settle` move 12 B out instead)
    "downsweep_settle_kernel": 12.0 + 7.0 / 8.0 * 9.0 + 16.0 / 8.0,
}
VALU_F64_BOUND = ("leaf_encode_kernel", "chain_keys_kernel", "spec_encode_kernel")


def valu_issue_model(insts, n, launch_ms):
    """VALU issue time of one launch from DYNAMIC counters x MEASURED issue costs: per instruction class the
    wave-instruction count of the rocprofv3 pass (f64 add / mul / fma; conversions; the rest, split by the static mix of the
    kernel text into f64-pipe compares / min / max / trunc, 32-bit compares and plain 32-bit instructions) x the SIMD cycles one
    wave64 instruction of that class holds its SIMD (tools/f64_rate.hip -> profiles/<round>_f64_rate.json, measured at the clock
    the part sustained under that class), summed and divided by the SIMD cycles the launch had: 1 024 SIMDs x the clock the part
    sustained DURING THIS KERNEL (GRBM_GUI_ACTIVE / duration of the counter pass) x the launch time measured here."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_f64_rate.json")
    if not os.path.exists(path):  # the issue costs are a property of the part, not of the build: the latest probe on record
        older = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_f64_rate.json"))
        if not older:
            return {"frac": None, "note": "no f64-rate probe on record (tools/f64_rate.sh)"}
        path = os.path.join(ROOT, "profiles", older[-1])
    with open(path) as f:
        rate = json.load(f)
    by = {r["inst"]: r["cycles_at_sustained_clock"] or r["cycles_at_2.4GHz"] for r in rate["per_instruction"]}
    mean = lambda names: sum(by[k] for k in names) / len(names)
    cyc = {"f64_add_mul_fma": mean(["v_fma_f64", "v_add_f64", "v_mul_f64"]),
           "f64_convert": mean(["v_cvt_f32_f64", "v_cvt_f64_f32", "v_cvt_u32_f64", "v_cvt_f64_u32"]),
           "f64_compare_minmax_trunc": mean(["v_trunc_f64", "v_max_f64", "v_cmp_gt_f64"]),
           "b32_compare": by["v_cmp_lt_u32"],
           "b32_plain": mean(["v_add_u32", "v_and_b32", "v_mov_b32", "v_fma_f32"])}
    rest = insts.get("rest_insts_per_point")
    if rest is None:
        return {"frac": None, "note": "the profile predates the per-class counters"}
    s64, s32 = insts.get("rest_static_share_f64_other") or 0.0, insts.get("rest_static_share_cmp32") or 0.0
    per_point = {"f64_add_mul_fma": insts["f64_arith_insts_per_point"], "f64_convert": insts["cvt_insts_per_point"],
                 "f64_compare_minmax_trunc": rest * s64, "b32_compare": rest * s32, "b32_plain": rest * (1.0 - s64 - s32)}
    clock = insts.get("sustained_clock_GHz") or 2.4
    cycles_per_point = sum(per_point[k] * cyc[k] for k in cyc)  # lane-instructions x cycles per wave64 instruction
    simd_cycles = cycles_per_point * n / 64.0                    # wave instructions = lane instructions / 64
    have = 1024 * clock * 1e9 * launch_ms * 1e-3
    # the static split counts the kernel's cold code too (the out-of-line IEEE divisions are all f64-pipe instructions), so it
    # prices the rest too high; pricing ALL of the rest as plain 32-bit instructions prices it too low: the truth lies between
    low_per_point = (per_point["f64_add_mul_fma"] * cyc["f64_add_mul_fma"] + per_point["f64_convert"] * cyc["f64_convert"] +
                     rest * cyc["b32_plain"])
    hi, lo = simd_cycles / have, low_per_point * n / 64.0 / have
    return {"frac": round(0.5 * (hi + lo), 4), "frac_bounds": [round(lo, 4), round(hi, 4)], "sustained_clock_GHz": clock,
            "cycles_per_class": {k: round(v, 3) for k, v in cyc.items()},
            "lane_insts_per_point_per_class": {k: round(v, 2) for k, v in per_point.items()},
            "simd_cycles_per_wave": round(cycles_per_point, 1), "simds": 1024,
            "source": os.path.relpath(path, ROOT) + " (issue costs) x the SQ_INSTS_VALU_* counters and GRBM_GUI_ACTIVE of this round's "
                      "rocprofv3 passes; the split of the instructions that are neither f64 add / mul / fma nor conversions follows the "
                      "static mix of the kernel text (tools/isa_mix.py)"}


def make_cloud(torch, n, seed, device, clusters=64, extent=1000.0, sigma=(1.0, 20.0), chunk=1 << 24,
               offset=(0.0, 0.0, 0.0)):
    """Config-2 distribution generated on the device (centres/sigmas from a host generator so every rank
    shares them)."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(12345))
    centres = torch.tensor(rng.uniform(0.0, extent, (clusters, 3)) + np.asarray(offset), dtype=torch.float64,
                           device=device)
    sigmas = torch.tensor(rng.uniform(sigma[0], sigma[1], clusters), dtype=torch.float64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.empty(n, dtype=torch.float64, device=device)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        which = torch.randint(0, clusters, (m,), generator=g, device=device)
        p = torch.randn((m, 3), generator=g, dtype=torch.float64, device=device) * sigmas[which, None] + centres[which]
        x[s:s + m], y[s:s + m], z[s:s + m] = p[:, 0], p[:, 1], p[:, 2]
        del p, which
    rgb = torch.empty((n, 3), dtype=torch.uint8, device=device)
    for s in range(0, n, chunk):  # chunked: the int64 temporaries of 1 B points would not fit beside the cloud
        m = min(chunk, n - s)
        h = (torch.arange(s, s + m, device=device, dtype=torch.int64) * 2654435761) & 0xFFFFFF
        rgb[s:s + m, 0], rgb[s:s + m, 1], rgb[s:s + m, 2] = (h >> 16) & 255, (h >> 8) & 255, h & 255
        del h
    torch.cuda.synchronize(device)  # the library runs on its own stream: the cloud must be complete before it is read
    return x, y, z, rgb


CLOUD_BLOCK = 1 << 22  # config-3 cloud: generated in blocks of 4 M points, block b from seed (seed, b)


def make_cloud_slice(torch, total, start, count, seed, device, clusters=64, extent=1000.0, sigma=(1.0, 20.0)):
    """Points [start, start + count) of the config-3 cloud of `total` points: the same 64 Gaussian clusters as config 2,
    generated block-wise (block b of CLOUD_BLOCK points from its own generator seed), so that any rank can produce any
    slice of the ONE global cloud without producing the rest; rgb = hash of the global point index."""
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(12345))
    centres = torch.tensor(rng.uniform(0.0, extent, (clusters, 3)), dtype=torch.float64, device=device)
    sigmas = torch.tensor(rng.uniform(sigma[0], sigma[1], clusters), dtype=torch.float64, device=device)
    x = torch.empty(count, dtype=torch.float64, device=device)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    rgb = torch.empty((count, 3), dtype=torch.uint8, device=device)
    end = min(start + count, total)
    g = torch.Generator(device=device)
    b = start // CLOUD_BLOCK
    while b * CLOUD_BLOCK < end:
        b0 = b * CLOUD_BLOCK
        m = min(CLOUD_BLOCK, total - b0)
        g.manual_seed(seed * 1_000_003 + b)
        which = torch.randint(0, clusters, (m,), generator=g, device=device)
        p = torch.randn((m, 3), generator=g, dtype=torch.float64, device=device) * sigmas[which, None] + centres[which]
        lo, hi = max(start, b0), min(end, b0 + m)  # the part of this block inside the slice
        dst = slice(lo - start, hi - start)
        src = slice(lo - b0, hi - b0)
        x[dst], y[dst], z[dst] = p[src, 0], p[src, 1], p[src, 2]
        h = (torch.arange(lo, hi, device=device, dtype=torch.int64) * 2654435761) & 0xFFFFFF
        rgb[dst, 0], rgb[dst, 1], rgb[dst, 2] = (h >> 16) & 255, (h >> 8) & 255, h & 255
        del p, which, h
        b += 1
    torch.cuda.synchronize(device)
    return x, y, z, rgb


def build_hash():
    """sha256 (16 hex digits) over the sources libpcv_hip.so is built from: what ties a rocprofv3 profile under
    profiles/ to the library that is running (tools/make_bench_profile_json.py stamps the same value)."""
    import hashlib
    d = os.path.join(ROOT, "point_cloud_viewer_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp", ".inc")) or name == "Makefile":
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def _digest(ptr, length):
    """blake2b-128 of `length` bytes at address `ptr` (no copy)."""
    import ctypes as C
    import hashlib
    if not length:
        return hashlib.blake2b(b"", digest_size=16).hexdigest()
    return hashlib.blake2b((C.c_uint8 * length).from_address(ptr), digest_size=16).hexdigest()


def tree_digests(tree, min_level=0):
    """{node name: (num_points, encoding, digest xyz, digest rgb, digest intensity)} of a built octree (host blobs)."""
    import ctypes as C
    import point_cloud_viewer_amd as pcv
    out = {}
    for i in range(tree.num_nodes):
        nd = tree.node(i)
        if nd.level < min_level:
            continue
        dig = []
        for which in range(3):
            ptr, ln = C.c_void_p(), C.c_uint64()
            tree.ctx._check(tree.lib.pcv_octree_node_data(tree.handle, i, which, C.byref(ptr), C.byref(ln)))
            dig.append(_digest(ptr.value or 0, ln.value))
        out[pcv.node_name(nd.id_high, nd.id_low)] = (nd.num_points, nd.encoding, dig[0], dig[1], dig[2])
    return out


def sharded_digests(result):
    """This rank's share of a sharded build in the same shape: its subtrees (level >= 2); the finished root / level-1
    nodes (every rank holds them after the top all-reduce) are added by rank 0 only."""
    import hashlib
    out = tree_digests(result.local, min_level=2)
    if result.builder.rank == 0:
        for name, nd in result.top_dict().items():
            out[name] = (nd["num_points"], nd["encoding"]) + tuple(
                hashlib.blake2b(nd[k], digest_size=16).hexdigest() for k in ("xyz", "rgb", "intensity"))
    return out


def digest_of_digests(digests):
    """One hash over a digest table (order-independent): equal iff the two octrees are byte-identical."""
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for name in sorted(digests):
        h.update(repr((name, digests[name])).encode())
    return h.hexdigest()


def compare_digests(want, got):
    missing = sorted(set(want) - set(got))
    extra = sorted(set(got) - set(want))
    bad = sorted(k for k in set(want) & set(got) if want[k] != got[k])
    return {"nodes": len(want), "nodes_gpu": len(got), "missing_nodes": len(missing), "extra_nodes": len(extra),
            "mismatching_nodes": len(bad), "first_mismatches": (missing + extra + bad)[:5],
            "ok": not missing and not extra and not bad}


def oracle_digests(resolution, bmin, bmax, x, y, z, rgb, threads=None, intensity=None):
    """Closed-form CPU oracle on the same cloud (host copies of the device tensors): digest table + stats."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    hx, hy, hz, hrgb = x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy(), rgb.cpu().numpy()
    hint = None if intensity is None else intensity.cpu().numpy()
    cores = threads or O.num_procs()
    t0 = time.perf_counter()
    want, stats = O.build_closed_digests(resolution, bmin, bmax, hx, hy, hz, hrgb, hint, threads=cores)
    stats = dict(stats, oracle_s=round(time.perf_counter() - t0, 2), oracle_threads=cores)
    import numpy as np
    n = int(x.numel())
    stats["bbox_equals_numpy_minmax"] = bool(n == 0 or (np.array_equal(bmin, [hx.min(), hy.min(), hz.min()]) and
                                                        np.array_equal(bmax, [hx.max(), hy.max(), hz.max()])))
    return want, stats


def verify_build(ctx, resolution, x, y, z, rgb, threads=None, intensity=None, bbox=None):
    """Byte-for-byte parity of one more build against the closed-form CPU oracle on the SAME cloud: node ids, point
    counts, encodings and a digest of every node file (reference bar: point_cloud_test/tests/main.rs:10-23 sum of
    num_points == N, src/octree/generation.rs:289-403 for the bytes)."""
    n = int(x.numel())
    t0 = time.perf_counter()
    tree = ctx.build(resolution, bbox, x, y, z, rgb, intensity)
    meta = tree.meta()
    got = tree_digests(tree)
    info = tree.build_info()
    tree.free()
    t1 = time.perf_counter()
    want, stats = oracle_digests(resolution, meta["bbox_min"], meta["bbox_max"], x, y, z, rgb, threads, intensity)
    out = {"oracle": "closed-form CPU restatement of the reference (oracle/pcv_oracle_build.cpp), not the Rust binary", "points": n}
    out.update(compare_digests(want, got))
    out.update({"sum_num_points_gpu": int(sum(v[0] for v in got.values())), "sum_num_points_oracle": stats["total_points"],
                "bbox_equals_numpy_minmax": stats["bbox_equals_numpy_minmax"],
                "max_abs_position_error_m": stats["max_abs_position_error"],
                "compared": "num_points, encoding, blake2b-128 of .xyz and .rgb" + (" and .intensity" if intensity is not None else "") +
                            " of every node",
                "tree_digest": digest_of_digests(got), "record_bytes": info.get("record_bytes"),
                "key_levels": info.get("key_levels"), "attempts": info.get("attempts"),
                "gpu_build_plus_d2h_s": round(t1 - t0, 2), "oracle_s": stats["oracle_s"], "oracle_threads": stats["oracle_threads"]})
    out["ok"] = bool(out["ok"] and (stats["bbox_equals_numpy_minmax"] or bbox is not None))
    return out


def box_info():
    """Clocks / power / temperature of GPU 0 as rocm-smi reports them (the same binary runs 5-7 % apart on
    different boxes — the record downsweeps 1.01 vs 1.25 ms — and nothing in a bench line said which kind of box it was)."""
    import subprocess
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--showperflevel", "--showmaxpower",
                            "--showmemuse", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout[r.stdout.index("{"):]) if "{" in r.stdout else {}
        card = next(iter(d.values())) if d else {}
        for k, v in card.items():
            kl = k.lower()
            if any(w in kl for w in ("sclk", "mclk", "fclk", "socclk", "power", "temperature", "performance level", "memory")):
                out[k] = v
    except Exception as e:  # noqa: BLE001 - the line of record must not depend on a monitoring tool
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def config1_cloud():
    """The BASELINE config-1 cloud: 1 M uniform points of a 200 x 200 x 20 m local frame placed in ECEF, rgb = point
    index, LOOSE bounding box of the transformed box corners (synthetic_data.rs:38-50)."""
    import numpy as np
    n = 1_000_000
    rng = np.random.Generator(np.random.PCG64(80293751232))
    local = np.stack([rng.uniform(-100.0, 100.0, n), rng.uniform(-100.0, 100.0, n), rng.uniform(-10.0, 10.0, n)], axis=1)
    lat, lon = math.radians(37.407204), math.radians(-122.147604)
    # ENU -> ECEF rotation (columns east, north, up) and the WGS84 position of the origin (src/math/mod.rs:167-183)
    east = np.array([-math.sin(lon), math.cos(lon), 0.0])
    north = np.array([-math.sin(lat) * math.cos(lon), -math.sin(lat) * math.sin(lon), math.cos(lat)])
    up = np.array([math.cos(lat) * math.cos(lon), math.cos(lat) * math.sin(lon), math.sin(lat)])
    a, f = 6378137.0, 1.0 / 298.257223563
    e2 = f * (2.0 - f)
    nn = a / math.sqrt(1.0 - e2 * math.sin(lat) ** 2)
    origin = np.array([nn * math.cos(lat) * math.cos(lon), nn * math.cos(lat) * math.sin(lon), nn * (1.0 - e2) * math.sin(lat)])
    rot = np.stack([east, north, up], axis=1)
    p = local @ rot.T + origin
    corners = np.array([[sx, sy, sz] for sx in (-100.0, 100.0) for sy in (-100.0, 100.0) for sz in (-10.0, 10.0)]) @ rot.T + origin
    bmin, bmax = corners.min(axis=0), corners.max(axis=0)  # loose box like synthetic_data.rs:46-50
    idx = np.arange(n, dtype=np.int64)
    rgb = np.stack([(idx >> 16) & 255, (idx >> 8) & 255, idx & 255], axis=1).astype(np.uint8)
    x, y, z = (np.ascontiguousarray(p[:, k]) for k in range(3))
    return n, x, y, z, rgb, bmin, bmax


def config1_leg(pcv, ctx):
    """Config 1 inside the default line: the CPU restatement's own timing (config 1 is the CPU plumbing line by definition)
    and — what the CPU-only line cannot say — byte parity of the GPU build of the same 1 M points and loose box."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n, x, y, z, rgb, bmin, bmax = config1_cloud()
    cores = O.num_procs()
    t0 = time.perf_counter()
    want = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=cores)
    oracle_s = time.perf_counter() - t0
    out = {"workload": "BASELINE config 1: 1 M uniform points (200 x 200 x 20 m local frame placed in ECEF), rgb = point index, loose "
                       "bbox, 1 mm", "points": n, "nodes": len(want.nodes), "oracle_closed_form_s": round(oracle_s, 2)}
    for single_chain in (False, True):  # the exact pipeline (the default at this size) and the single-chain build, forced
        t0 = time.perf_counter()
        tree = ctx.build(0.001, pcv.Aabb(bmin, bmax), x, y, z, rgb, single_chain=single_chain)
        ms = (time.perf_counter() - t0) * 1e3
        got = tree.to_dict()
        info = tree.build_info()
        tree.free()
        bad = [k for k in set(want.nodes) | set(got) if k not in got or k not in want.nodes or
               any(got[k][f] != want.nodes[k][f] for f in ("num_points", "encoding", "xyz", "rgb"))]
        out["single_chain" if single_chain else "exact_pipeline"] = {
            "mismatching_nodes": len(bad), "ok": not bad, "host_arrays_to_host_blobs_ms": round(ms, 2),
            "pipeline_ran": "single-chain" if info.get("single_chain") else "exact two-chain"}
    out["ok"] = bool(out["exact_pipeline"]["ok"] and out["single_chain"]["ok"])
    return out


def intensity_leg(args, torch, pcv, ctx, dev, points, steps=5):
    """The reference BINARY's payload (src/bin/build_octree.rs:47-52: attributes ["color", "intensity"]): the config-2
    distribution with an f32 intensity plane (raw.rs:374-392: one more node file, 4 bytes per point). Timed builds of the
    colour-only and the colour + intensity cloud back to back, byte parity of every node incl. `.intensity`."""
    x, y, z, rgb = make_cloud(torch, points, seed=1, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    inten = torch.rand(points, generator=g, device=dev, dtype=torch.float32) * 4096.0
    torch.cuda.synchronize()
    res = {}
    for plane in (None, inten, None, inten):  # both variants warm (pool blocks of both sizes exist) before either is timed
        ctx.build(args.resolution, None, x, y, z, rgb, plane).free()
    for name, plane in (("color_only", None), ("color_and_intensity", inten)):
        ctx.build(args.resolution, None, x, y, z, rgb, plane).free()
        ctx.set_profiling("major")
        ctx.reset_kernel_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        marks = []
        for _ in range(steps):
            t = ctx.build(args.resolution, None, x, y, z, rgb, plane)
            info = t.build_info()
            t.free()
            marks.append(time.perf_counter())  # host time the step was queued at (the builds run one behind the host)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        ctx.set_profiling(False)
        ks = ctx.kernel_stats()
        res[name] = {"ms_per_step": round(ms, 3), "Mpoints_per_s": round(points / (ms * 1e-3) / 1e6, 1), "record_bytes": info.get("record_bytes"),
                     "host_ms_between_steps": [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks[:-1], marks)],
                     "settled_in_sort": info.get("settled_in_sort"),
                     "kernel_ms_per_step": {k: round(v[1] / steps, 3) for k, v in ks.items() if v[0] > 0}}
    parity = verify_build(ctx, args.resolution, x, y, z, rgb, intensity=inten)
    del x, y, z, rgb, inten
    return {"workload": f"{points / 1e6:g} M Gaussian-cluster points (config-2 distribution) + f32 intensity plane, 1 mm", "points": points,
            "steps": steps, "color_only": res["color_only"], "color_and_intensity": res["color_and_intensity"],
            "intensity_cost": round(res["color_and_intensity"]["ms_per_step"] / res["color_only"]["ms_per_step"] - 1.0, 4),
            "parity": parity}


def sharded_leg(args, torch, pcv, ctx, dev, x, y, z, rgb, want_digest, steps=5, virtual=8):
    """The multi-GPU code path inside the default line on the config-2 cloud whose single-GPU octree the
    oracle has just verified: (a) ShardedOctreeBuilder over RCCL at world size 1 — routing, the (empty) exchange, the routed
    build, the top-node merge — timed; (b) `virtual` thread-ranks on this one GPU, contiguous slices, octants AND buckets:
    the merged octree must have the verified digest and no node may be built twice. One device: none of this is a scaling
    number — the path stays UNMEASURED ON REAL RANKS."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from thread_dist import run_ranks
    from point_cloud_viewer_amd import distributed as pdist
    import torch.distributed as dist
    n = int(x.numel())
    out = {"note": "one GPU: world-1 RCCL and thread-ranks sharing the device — code-path timing and parity, NOT scaling; unmeasured on real ranks",
           "points": n, "want_tree_digest": want_digest}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        w1 = {}
        for mode in ("octants", "buckets"):
            builder = pdist.ShardedOctreeBuilder(ctx, dist, dev, shard_mode=mode)
            bbox = builder.global_bbox(x, y, z)
            for _ in range(2):
                builder.build(args.resolution, bbox, x, y, z, rgb).free()
            dist.barrier()
            torch.cuda.synchronize()
            ctx.set_profiling("major")
            ctx.reset_kernel_stats()
            t0 = time.perf_counter()
            for _ in range(steps):
                r = builder.build(args.resolution, bbox, x, y, z, rgb)
                ex = r.exchange_info()
                r.free()
            dist.barrier()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            ctx.set_profiling(False)
            kms = {k: round(v[1] / steps, 3) for k, v in ctx.kernel_stats().items() if v[0] > 0}
            # the same with K1 + the 6-number all-reduce INSIDE the step: the scope of the unsharded step and of `--gpus N`
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                builder.build(args.resolution, builder.global_bbox(x, y, z), x, y, z, rgb).free()
            dist.barrier()
            torch.cuda.synchronize()
            ms_bbox = (time.perf_counter() - t0) / steps * 1e3
            r = builder.build(args.resolution, bbox, x, y, z, rgb)
            dig = digest_of_digests(sharded_digests(r))
            r.free()
            w1[mode] = {"ms_per_step": round(ms, 3), "ms_per_step_bbox_inside": round(ms_bbox, 3),
                        "Mpoints_per_s": round(n / (ms * 1e-3) / 1e6, 1), "stage_ms": ex["ms"],
                        "kernel_ms_per_step": kms, "rows_sent": ex["rows_sent"], "rows_received": ex["rows_received"], "tree_digest": dig,
                        "digest_equal": dig == want_digest}
        out["world1"] = dict(w1["octants"], rccl_ranks=1, shard_mode="octants", buckets=w1["buckets"])
    finally:
        dist.destroy_process_group()
    vout = {}
    for mode in ("octants", "buckets"):
        def rank_main(rank, vdist, mode=mode):
            torch.cuda.set_device(dev)
            rctx = pcv.Context(dev.index or 0)
            lo, hi = rank * n // virtual, (rank + 1) * n // virtual
            sx, sy, sz, srgb = x[lo:hi], y[lo:hi], z[lo:hi], rgb[lo:hi]
            builder = pdist.ShardedOctreeBuilder(rctx, vdist, dev, shard_mode=mode)
            bbox = builder.global_bbox(sx, sy, sz)
            res = builder.build(args.resolution, bbox, sx, sy, sz, srgb)
            torch.cuda.synchronize()
            dig = sharded_digests(res)
            ex = res.exchange_info()
            res.free()
            rctx.close()
            return dig, ex
        results = run_ranks(virtual, rank_main)
        merged, dup = {}, 0
        for dig, _ in results:
            for k, v in dig.items():
                dup += k in merged
                merged[k] = v
        d = digest_of_digests(merged)
        vout[mode] = {"tree_digest": d, "digest_equal": d == want_digest, "nodes": len(merged), "nodes_built_twice": dup,
                      "points_owned_per_rank": results[0][1]["points_owned_per_rank"],
                      "imbalance_max_over_mean": results[0][1]["imbalance_max_over_mean"],
                      "rows_sent_per_rank": [r[1]["rows_sent"] for r in results]}
    out[f"virtual{virtual}"] = vout
    out["ok"] = bool(out["world1"]["digest_equal"] and out["world1"]["buckets"]["digest_equal"] and
                     all(v["digest_equal"] and v["nodes_built_twice"] == 0 for v in vout.values()))
    return out


def config1(args):
    """BASELINE config 1 (plumbing): 1 M uniform points in a local frame placed in ECEF, CPU oracle only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n, x, y, z, rgb, bmin, bmax = config1_cloud()
    cores = O.num_procs()
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = tempfile.mkdtemp(prefix="pcv_config1_", dir=base)
    try:
        t0 = time.perf_counter()
        O.build_literal_dir(os.path.join(d, "octree"), 0.001, bmin, bmax, x, y, z, rgb, threads=cores)
        dt = time.perf_counter() - t0
        tree = O.load_dir(os.path.join(d, "octree"))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    closed = O.build_closed(0.001, bmin, bmax, x, y, z, rgb, threads=cores)
    same = closed.nodes == tree.nodes
    levels = max(v["level"] for v in tree.nodes.values())
    return {"metric": "octree-build Mpoints/sec", "value": round(n / dt / 1e6, 3), "unit": "Mpoints/s", "n_gpus": 0,
            "steps": 1, "warmup": 0, "ms_per_step": round(dt * 1e3, 1), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config 1: 1 M uniform points (200 x 200 x 20 m local frame placed in ECEF at "
                                   "lat 37.407204, lon -122.147604), rgb = point index, loose bbox of the box corners, "
                                   "resolution 1 mm — CPU restatement of the reference only (plumbing line)",
                       "nodes": len(tree.nodes), "deepest_level": levels, "sum_num_points": tree.total_points(),
                       "literal_equals_closed_form": bool(same)},
            "cpu_baseline": {"value": round(n / dt / 1e6, 3), "unit": "Mpoints/s", "cores": cores, "kind": "port",
                             "sample": f"the whole config-1 cloud, literal file-streaming restatement on tmpfs, {dt:.2f} s"}}


def query_leg(args, ctx, tree):
    """BASELINE config 4 on a built octree: F random camera frusta — node relations (K7), visible-node traversal (K7b),
    batched point query (K8) — with parity of the first --verify-frusta frusta (default: all) and of the points of the
    --verify-cull-frusta culled frusta against the oracle (reference bars: sat.rs:174-205 relation tests,
    point_cloud_test/tests/main.rs:104-127 points in a frustum)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    meta = tree.meta()
    bmin, bmax = meta["bbox_min"], meta["bbox_max"]
    M = tree.num_nodes
    npoints = tree.num_points

    rng = np.random.default_rng(3)
    persp = O.perspective3_new(1.0, 1.2, 0.1, 100.0)
    mats = []
    for _ in range(args.frusta):
        eye = rng.uniform(bmin, bmax)
        q = rng.normal(size=4)
        q = q / math.sqrt(float(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]))
        c, _ = O.frustum_new(eye, q, persp)
        mats.append(c)
    shapes = ctx.shapes([("frustum", m) for m in mats])
    ctx.set_profiling(True)
    qsteps = max(1, args.query_steps)
    for _ in range(max(1, min(args.warmup, 2))):
        rel, sizes = tree.cull_nodes(shapes, with_sizes=True)
    ctx.reset_kernel_stats()
    t0 = time.perf_counter()
    for _ in range(qsteps):
        rel, sizes = tree.cull_nodes(shapes, with_sizes=True)
    wall_a = (time.perf_counter() - t0) / qsteps
    ks = ctx.kernel_stats()["cull_nodes_kernel"]
    dense_ms = ks[1] / ks[0]
    pairs = args.frusta * M
    # the same relations as per-frustum LISTS of the nodes that are not Out (pcv_cull_nodes_sparse): sizes only where the
    # reference computes them, 13 bytes per listed node to the host instead of 9 bytes per pair
    sp_counts = tree.cull_nodes_sparse(shapes, 0, with_sizes=False)[0]
    sp_cap = max(1, int(sp_counts.max()))
    tree.cull_nodes_sparse(shapes, sp_cap)
    ctx.reset_kernel_stats()
    t0 = time.perf_counter()
    for _ in range(qsteps):
        sp_counts, sp_idx, sp_rel, sp_sizes = tree.cull_nodes_sparse(shapes, sp_cap)
    wall_s = (time.perf_counter() - t0) / qsteps
    ks = ctx.kernel_stats()["cull_nodes_sparse_kernel"]
    cull_ms = ks[1] / ks[0]

    ctx.reset_kernel_stats()
    vis, status = tree.visible_nodes(shapes)
    ks = ctx.kernel_stats()["visible_nodes_kernel"]
    vis_ms = ks[1] / ks[0]
    nvis = np.array([len(v) for v in vis])

    # PointCloud::nodes_in_location (octree/mod.rs:309-323) of every frustum: the hierarchical walk (round 6: one wave per shape)
    tree.nodes_in_location(shapes)
    ctx.reset_kernel_stats()
    nil = tree.nodes_in_location(shapes)
    nil_ms = ctx.kernel_stats()["nodes_in_location_kernel"][1]

    ctx.reset_kernel_stats()
    kept = 0
    kept_each = []
    for f in range(args.cull_frusta):
        kept_each.append(tree.query_points(shapes, f, capacity=1 << 22)["count"])
        kept += kept_each[-1]
    st = ctx.kernel_stats()
    q_ms = st["cull_points_kernel"][1] + st["query_compact_kernel"][1] + st["nodes_in_location_kernel"][1]
    q_split = {k.replace("_kernel", ""): round(st[k][1], 3) for k in ("cull_points_kernel", "query_compact_kernel",
                                                                      "nodes_in_location_kernel")}
    big = ctx.shapes([("aabb", bmin, bmin + (bmax - bmin) * 0.63)])
    ctx.reset_kernel_stats()
    r = tree.query_points(big, 0, capacity=1)
    st = ctx.kernel_stats()
    bpc = {1: 1, 2: 2, 3: 4, 4: 8}
    visited = tree.nodes_in_location(big)[0]
    tested = sum(tree.node(int(i)).num_points for i in visited)
    enc_bytes = sum(tree.node(int(i)).num_points * 3 * bpc[tree.node(int(i)).encoding] for i in visited)
    big_ms = st["cull_points_kernel"][1]
    gbs = (enc_bytes + tested) / (big_ms * 1e-3) / 1e9

    # parity + CPU baseline on the first V frusta: relation of every (frustum, node) pair, size on screen, visible lists
    V = min(args.verify_frusta, args.frusta)
    cubes = np.array([[*tree.node(i).cube_min, tree.node(i).cube_edge] for i in range(M)])
    names = tree.node_names()
    nodes = {names[i]: dict(id=(tree.node(i).id_high, tree.node(i).id_low), num_points=tree.node(i).num_points)
             for i in range(M)}
    rel_bad = size_bad = vis_bad = sparse_bad = 0
    t0 = time.perf_counter()
    for f in range(V):
        orel, osz = O.cull_cubes(O.SHAPE_FRUSTUM, mats[f], cubes, with_sizes=True)
        rel_bad += int((orel != rel[f]).sum())
        size_bad += int((~((osz == sizes[f]) | (np.isnan(osz) & np.isnan(sizes[f])))).sum())
        keep = np.nonzero(orel != 2)[0]
        k = int(sp_counts[f])
        sparse_bad += int(not (k == keep.size and np.array_equal(sp_idx[f, :k], keep) and np.array_equal(sp_rel[f, :k], orel[keep]) and
                               np.array_equal(sp_sizes[f, :k], osz[keep], equal_nan=True)))
    cpu_a = time.perf_counter() - t0
    # nodes_in_location: the walk prunes under every Out node, the sparse list holds every node that is not Out — the same list
    # unless a child of an Out node is itself not Out (an ulp-level tie); against the oracle's NodeIdsIterator for the first 1 000
    nil_bad = sum(int(not np.array_equal(np.asarray(nil[f], dtype=np.int64), sp_idx[f, :int(sp_counts[f])].astype(np.int64))) for f in range(V))
    nil_oracle_bad = sum(int([names[int(i)] for i in nil[f]] != O.nodes_in_location(bmin, bmax, nodes, O.SHAPE_FRUSTUM, mats[f]))
                         for f in range(min(V, 1000)))
    t0 = time.perf_counter()
    for f in range(V):
        want = O.get_visible_nodes(bmin, bmax, nodes, mats[f])
        got = None if status[f] != 0 else [names[int(i)] for i in vis[f]]
        vis_bad += int(want != got)
    cpu_b = time.perf_counter() - t0
    # K8 parity (reference bar point_cloud_test/tests/main.rs:104-127): the points pcv_query_points returns for each of
    # the culled frusta == for node in nodes_in_location: decode, contains, retain — positions, colours and order
    pts_bad = cnt_bad = 0
    pts_checked = 0
    index_of = {nm: i for i, nm in enumerate(names)}
    t0 = time.perf_counter()
    for f in range(min(args.cull_frusta, args.verify_cull_frusta)):
        got = tree.query_points(shapes, f, capacity=max(kept_each[f], 1))
        wx, wy, wz, wc = [], [], [], []
        for nm in O.nodes_in_location(bmin, bmax, nodes, O.SHAPE_FRUSTUM, mats[f]):
            i = index_of[nm]
            nd = tree.node(i)
            if nd.num_points == 0:
                continue
            px, py, pz = O.decode_positions(nd.encoding, nd.cube_min, nd.cube_edge, tree.node_data(i, 0))
            keep = O.cull_points(O.SHAPE_FRUSTUM, mats[f], px, py, pz).astype(bool)
            wx.append(px[keep]), wy.append(py[keep]), wz.append(pz[keep])
            wc.append(np.frombuffer(tree.node_data(i, 1), dtype=np.uint8).reshape(-1, 3)[keep])
        cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dtype=dt)
        ex, ey, ez, ec = cat(wx, np.float64), cat(wy, np.float64), cat(wz, np.float64), cat(wc, np.uint8).reshape(-1, 3)
        pts_checked += int(ex.size)
        if got["count"] != ex.size:
            cnt_bad += 1
        elif not (np.array_equal(got["x"], ex) and np.array_equal(got["y"], ey) and np.array_equal(got["z"], ez) and
                  np.array_equal(got["rgb"].reshape(-1, 3), ec)):
            pts_bad += 1
    cpu_c = time.perf_counter() - t0
    return {"metric": "frustum-cull node pairs/sec", "value": round(pairs / (cull_ms * 1e-3) / 1e6, 1), "unit": "Mpairs/s",
            "n_gpus": 1, "steps": qsteps, "warmup": max(1, min(args.warmup, 2)), "ms_per_step": round(cull_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: octree of {npoints / 1e6:g} M Gaussian-cluster points "
                                   f"({M} nodes), {args.frusta} random camera frusta (Perspective3 aspect 1, fovy 1.2, "
                                   "near 0.1, far 100), batched transform + cull on 1 GPU", "nodes": M, "frusta": args.frusta},
            "cull_nodes": {"kernel_ms": round(cull_ms, 3), "wall_ms_incl_D2H": round(wall_s * 1e3, 1), "pairs": pairs,
                           "api": "pcv_cull_nodes_sparse: per frustum the nodes that are not Out (node order) with relation + size on screen",
                           "listed_nodes": int(sp_counts.sum()), "capacity_per_frustum": sp_cap,
                           "dense_matrix": {"api": "pcv_cull_nodes: relation + size on screen of EVERY pair", "kernel_ms": round(dense_ms, 3),
                                            "wall_ms_incl_D2H": round(wall_a * 1e3, 1)},
                           "relation_histogram": np.bincount(rel.ravel(), minlength=3).tolist()},
            "visible_nodes": {"kernel_ms": round(vis_ms, 3), "frusta_per_s": round(args.frusta / (vis_ms * 1e-3), 1),
                              "mean_visible": float(nvis.mean()), "max_visible": int(nvis.max()),
                              "status_nonzero": int((status != 0).sum())},
            "nodes_in_location": {"kernel_ms": round(nil_ms, 3), "frusta_per_s": round(args.frusta / (nil_ms * 1e-3), 1),
                                  "listed_nodes": int(sum(len(v) for v in nil)),
                                  "api": "pcv_nodes_in_location: NodeIdsIterator's walk (children only under a node that is not Out) of every frustum"},
            "query_points": {"frusta": args.cull_frusta, "kept_points": int(kept), "kernel_ms": round(q_ms, 3),
                             "kernel_ms_split": q_split},
            "roofline": {"bound": "hbm", "kernel": "query_flags_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes_per_launch": int(enc_bytes + tested), "avg_launch_ms": round(big_ms, 4),
                         "points_tested": int(tested), "kept": r["count"],
                         "note": "pcv_query_points, one AABB over 63 % of the cube per axis: encoded node bytes in, 1 flag byte per "
                                 "point out; avg_launch_ms = query_chunks_kernel (descriptors) + query_flags_kernel, HIP events"},
            "parity": {"oracle": "CPU restatement (oracle/pcv_oracle_query.cpp), not the Rust binary", "frusta_checked": V,
                       "pairs_checked": V * M, "relation_mismatches": rel_bad, "size_on_screen_mismatches": size_bad,
                       "sparse_list_mismatches": sparse_bad, "visible_list_mismatches": vis_bad,
                       "nodes_in_location_vs_sparse_lists_mismatches": nil_bad,
                       "nodes_in_location_vs_oracle_mismatches_first_1000": nil_oracle_bad,
                       "query_points_frusta_checked": min(args.cull_frusta, args.verify_cull_frusta),
                       "query_points_points_checked": pts_checked, "query_points_count_mismatches": cnt_bad,
                       "query_points_content_mismatches": pts_bad, "query_points_check_s": round(cpu_c, 1),
                       "ok": (rel_bad == 0 and size_bad == 0 and sparse_bad == 0 and vis_bad == 0 and cnt_bad == 0 and pts_bad == 0 and
                              nil_oracle_bad == 0)},
            "cpu_baseline": None if V == 0 else {
                "value": round(V * M / cpu_a / 1e6, 3), "unit": "Mpairs/s", "cores": 1, "kind": "port",
                "sample": f"first {V} frusta x {M} nodes, SAT relation + size on screen, {cpu_a:.1f} s; "
                          f"get_visible_nodes: {V / cpu_b:.1f} frusta/s"}}


def query_profile(bhash):
    """Counters of the query kernels from this round's rocprofv3 passes of `bench.py --query` (tools/profile_query.sh ->
    profiles/<round>_query_counters.json), quoted only when they were taken from the library that is running."""
    path = os.path.join(ROOT, "profiles", f"{ROUND}_query_counters.json")
    if not os.path.exists(path):
        return {"source": None, "note": "no rocprofv3 pass of the query path recorded for this round"}
    with open(path) as f:
        prof = json.load(f)
    if prof.get("build_hash") != bhash:
        return {"source": os.path.relpath(path, ROOT), "profile_matches_build": False,
                "note": f"STALE: taken from build {prof.get('build_hash')}, running {bhash}"}
    return dict(prof.get("kernels", {}), source=os.path.relpath(path, ROOT), profile_matches_build=True)


def query_bench(args):
    """`bench.py --query`: BASELINE config 4 on its own (the default line carries the same leg as `query`)."""
    import torch
    import point_cloud_viewer_amd as pcv
    dev = torch.device("cuda", 0)
    x, y, z, rgb = make_cloud(torch, args.points, seed=1, device=dev)
    ctx = pcv.Context(0)
    tree = ctx.build(args.resolution, None, x, y, z, rgb)
    del x, y, z, rgb
    out = query_leg(args, ctx, tree)
    out["roofline"]["profile"] = query_profile(build_hash())
    return out


def config5_leg(args, torch, pcv, ctx, dev, points):
    """BASELINE config 5 inside the default line: `points` (500 M) Gaussian-cluster points at ECEF magnitudes (|p| ~ 6.4e6 m,
    f64 input; leaf levels Float32 / u16 / u8-coded, codec.rs:115-121), timed builds + byte parity of every node against
    the closed-form CPU oracle. This is the size at which the SHIPPED library's big-tree record sort (more than 16 384
    predicted nodes) runs."""
    offset = (-2.7e6, -4.3e6, 3.8e6)
    t_gen = time.perf_counter()
    x, y, z, rgb = make_cloud(torch, points, seed=1, device=dev, offset=offset)
    t_gen = time.perf_counter() - t_gen
    steps, warmup = max(1, args.config5_steps), 1
    ctx.set_profiling("major")
    info = {}
    for _ in range(warmup):
        ctx.build(args.resolution, None, x, y, z, rgb).free()
    ctx.reset_kernel_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        t = ctx.build(args.resolution, None, x, y, z, rgb)
        info["nodes"], info["stages"], info["build"] = t.num_nodes, t.stage_ms(), t.build_info()
        t.free()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    kstats = ctx.kernel_stats()
    t = ctx.build(args.resolution, None, x, y, z, rgb, stage_times=True)  # (one more build, outside the timed region, for stage_ms)
    info["stages"] = t.stage_ms()
    t.free()
    # K8 (query_flags_kernel) on ONE launch over every point of this tree: > 3 GB of encoded positions in, one flag byte per point
    # out (the config-4 figure is taken on 70 us launches where ramp-up dominates)
    k8 = None
    try:
        import numpy as np
        t = ctx.build(args.resolution, None, x, y, z, rgb)
        meta = t.meta()
        lo, hi = np.asarray(meta["bbox_min"]), np.asarray(meta["bbox_max"])
        big = ctx.shapes([("aabb", lo - 1.0, hi - (hi - lo) * 0.05)])  # nearly every node visited, most points kept
        bpc = {1: 1, 2: 2, 3: 4, 4: 8}
        tested = enc_bytes = 0
        visited = t.nodes_in_location(big)[0]  # the nodes the query walks: everything that is not Out of the box
        for i in visited:
            nd = t.node(int(i))
            tested += nd.num_points
            enc_bytes += nd.num_points * 3 * bpc[int(nd.encoding)]
        ctx.set_profiling(True)
        t.query_points(big, 0, capacity=1)  # warm: uploads nothing (device-resident blobs), sizes the scratch
        ctx.reset_kernel_stats()
        r = t.query_points(big, 0, capacity=1)
        st = ctx.kernel_stats()
        ctx.set_profiling(False)
        big_ms = st["cull_points_kernel"][1]
        gbs = (enc_bytes + tested) / (big_ms * 1e-3) / 1e9
        k8 = {"bound": "hbm", "kernel": "query_flags_kernel", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(enc_bytes + tested), "avg_launch_ms": round(big_ms, 4),
              "points_tested": int(tested), "nodes_visited": int(len(visited)), "kept": r["count"], "launches": st["cull_points_kernel"][0],
              "note": "pcv_query_points with one AABB over 95 % of the cube per axis of the 500 M-point tree: bytes of the VISITED nodes; "
                      "query_chunks_kernel + query_flags_kernel, HIP events"}
        t.free()
    except Exception as e:  # noqa: BLE001
        k8 = {"error": f"{type(e).__name__}: {e}"}
    parity = verify_build(ctx, args.resolution, x, y, z, rgb)
    del x, y, z, rgb
    ms = elapsed / steps * 1e3
    return {"metric": "octree-build Mpoints/sec", "value": round(points / (ms * 1e-3) / 1e6, 2), "unit": "Mpoints/s", "query_flags_roofline": k8,
            "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 3), "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE config 5 (ECEF-offset f64 input): {points / 1e6:g} M Gaussian-cluster points (64 clusters, "
                                   f"1000 m cube, sigma 1-20 m) offset by {offset} m, f64 SoA xyz + u8 rgb, resolution 1 mm, full "
                                   "build + LOD promotion, K1 inside the step",
                       "points": points, "nodes": info.get("nodes"), "cloud_generation_s": round(t_gen, 1)},
            "parity": parity, "tree_digest": parity.get("tree_digest"), "build_info": info.get("build"),
            "stage_ms": {k: round(v, 3) for k, v in (info.get("stages") or {}).items()},
            "kernel_ms_per_step": {k: round(v[1] / steps, 3) for k, v in kstats.items() if v[0] > 0}}


def run_virtual_ranks(args, torch, pcv, dev):
    """BASELINE config 3 on ONE GPU: V thread-ranks (tests/thread_dist.py in place of torch.distributed — RCCL refuses
    two ranks on one device), each with its own context and stream, rank r holding slice r of ONE cloud; the real
    ShardedOctreeBuilder / HipBackend in every shard mode asked for. The merged octree (every rank's subtrees + the
    all-reduced top nodes) is compared node by node with a single-GPU build of the whole cloud and, with --verify, with
    the closed-form CPU oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from thread_dist import run_ranks
    from point_cloud_viewer_amd import distributed as pdist
    V = args.virtual_ranks
    total = args.points * V if args.points else 1_000_000_000
    x, y, z, rgb = make_cloud_slice(torch, total, 0, total, seed=args.seed3, device=dev)
    out = {"metric": "octree-build Mpoints/sec", "unit": "Mpoints/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic"}
    # the single-GPU build of the whole cloud: the reference the merged shards are compared with
    ctx = pcv.Context(0)
    t0 = time.perf_counter()
    tree = ctx.build(args.resolution, None, x, y, z, rgb, stage_times=True)
    single_ms = (time.perf_counter() - t0) * 1e3
    meta = tree.meta()
    single = tree_digests(tree)
    single_info = tree.build_info()
    single_stage = tree.stage_ms()
    tree.free()
    ctx.close()  # its pool holds ~95 bytes per point: give it back before the ranks allocate theirs
    oracle = None
    if args.verify:
        want, stats = oracle_digests(args.resolution, meta["bbox_min"], meta["bbox_max"], x, y, z, rgb)
        oracle = dict(compare_digests(want, single), max_abs_position_error_m=stats["max_abs_position_error"],
                      oracle_s=stats["oracle_s"], oracle_threads=stats["oracle_threads"],
                      bbox_equals_numpy_minmax=stats["bbox_equals_numpy_minmax"])
    modes = ["octants", "buckets"] if args.shard_mode == "both" else [args.shard_mode]
    records = {}
    value = ms = None
    for mode in modes:
        def rank_main(rank, dist, mode=mode):
            torch.cuda.set_device(dev)
            rctx = pcv.Context(dev.index or 0)
            lo, hi = rank * total // V, (rank + 1) * total // V
            sx, sy, sz, srgb = x[lo:hi], y[lo:hi], z[lo:hi], rgb[lo:hi]
            builder = pdist.ShardedOctreeBuilder(rctx, dist, dev, shard_mode=mode)
            bbox = builder.global_bbox(sx, sy, sz)
            times = []
            res = None
            for it in range(args.warmup + args.steps):
                if res is not None:
                    res.free()
                dist.barrier()
                torch.cuda.synchronize()
                a = time.perf_counter()
                res = builder.build(args.resolution, bbox, sx, sy, sz, srgb)
                torch.cuda.synchronize()
                dist.barrier()
                if it >= args.warmup:
                    times.append(time.perf_counter() - a)
            dig = sharded_digests(res)
            info = dict(exchange=res.exchange_info(), build=res.local.build_info(), nodes_local=res.num_nodes_local,
                        bbox=(bbox.min.tolist(), bbox.max.tolist()))
            res.free()
            rctx.close()
            return dig, times, info

        results = run_ranks(V, rank_main)
        merged, dup = {}, 0
        for dig, _, _ in results:
            for k, v in dig.items():
                dup += k in merged
                merged[k] = v
        cmp_single = compare_digests(single, merged)
        step_s = [max(r[1][i] for r in results) for i in range(args.steps)]  # slowest rank per step
        ms = sum(step_s) / max(len(step_s), 1) * 1e3
        value = total / (ms * 1e-3) / 1e6
        records[mode] = {"vs_single_gpu_build": cmp_single, "nodes_built_twice": dup,
                         "tree_digest": digest_of_digests(merged), "ms_per_step_slowest_rank": round(ms, 3),
                         "bbox_equals_single_gpu": results[0][2]["bbox"] == (meta["bbox_min"].tolist(), meta["bbox_max"].tolist()),
                         "exchange_rank0": results[0][2]["exchange"],
                         "single_chain_per_rank": [r[2]["build"].get("single_chain") for r in results],
                         "nodes_local_per_rank": [r[2]["nodes_local"] for r in results]}
        if oracle is not None:
            records[mode]["vs_oracle"] = compare_digests(want, merged)
    ok = all(r["vs_single_gpu_build"]["ok"] and r["nodes_built_twice"] == 0 and r["bbox_equals_single_gpu"] for r in records.values())
    if oracle is not None:
        ok = ok and oracle["ok"] and all(r["vs_oracle"]["ok"] for r in records.values())
    out.update({"value": round(value, 2), "ms_per_step": round(ms, 3),
                "config": {"workload": f"BASELINE config 3 dress rehearsal: ONE cloud of {total / 1e6:g} M Gaussian-cluster points "
                                       f"(64 clusters, 1000 m cube, sigma 1-20 m, seed {args.seed3}), {V} VIRTUAL ranks (threads, one "
                                       f"context + stream each) on ONE GPU, contiguous slices of {total // V} points, shard "
                                       f"mode(s) {'+'.join(modes)}, resolution 1 mm; `value` is the last mode's, the ranks share "
                                       "one device, so it is NOT a scaling number",
                           "virtual_ranks": V, "points_total": total, "resolution": args.resolution},
                "single_gpu_build": {"nodes": len(single), "tree_digest": digest_of_digests(single), "build_info": single_info,
                                     "first_build_wall_ms": round(single_ms, 1), "stage_ms": {k: round(v, 3) for k, v in single_stage.items()},
                                     "vs_oracle": oracle},
                "shard_modes": records,
                "parity": {"ok": bool(ok), "compared": "every node: num_points, encoding, blake2b-128 of .xyz / .rgb; merged shards vs "
                                                        "the single-GPU build" + (" and vs the closed-form CPU oracle" if oracle else ""),
                           "mismatching_nodes": sum(r["vs_single_gpu_build"]["mismatching_nodes"] + r["vs_single_gpu_build"]["missing_nodes"] +
                                                    r["vs_single_gpu_build"]["extra_nodes"] for r in records.values())}})
    return out


# ---- the line of record (the driver parses the LAST stdout line; r05's 22 KB line was cut and left parsed = null) ----
FINAL_LINE_LIMIT = 4096   # bytes of the last stdout line
DETAIL_LINE_LIMIT = 7000  # bytes of the earlier, trimmed detail line (the whole record goes to bench_detail.json)

# the library's event slots -> the kernel symbols rocprofv3 lists for the single-chain build with 12-byte records (the slot of a
# stage is shared by the kernels that can fill it; kernel_ms_per_step names the one that ran)
SLOT_TO_SYMBOL_12 = {"spec_encode_kernel": "chain_pass_kernel", "downsweep_rec_kernel": "downsweep_rec12_kernel",
                     "promote_settle_kernel": "promote_settle_leaf_kernel", "promote_climb_kernel": "promote_climb_leaf_kernel",
                     "hist_from_rows_kernel": "hist12_from_rows_kernel"}


def kernel_symbol(slot, build_info):
    b = build_info or {}
    if b.get("single_chain") and b.get("record_bytes") == 12:
        return SLOT_TO_SYMBOL_12.get(slot, slot)
    if b.get("single_chain"):
        return {"spec_encode_kernel": "chain_pass_kernel", "promote_settle_kernel": "promote_settle_leaf_kernel",
                "promote_climb_kernel": "promote_climb_leaf_kernel"}.get(slot, slot)
    return slot


def _leg_parity(leg, key="parity"):
    """{ok, mismatching_nodes, tree_digest} of a leg's parity record (None when the leg did not run)."""
    if not leg:
        return None
    if "error" in leg:
        return {"ok": False, "error": str(leg["error"])[:120]}
    p = leg.get(key) if key else leg
    if not isinstance(p, dict):
        return {"ok": bool(leg.get("ok"))}
    out = {"ok": bool(p.get("ok"))}
    for k in ("mismatching_nodes", "nodes", "tree_digest"):
        if p.get(k) is not None:
            out[k] = p[k]
    return out


def _shrink(obj, max_str=160, max_list=8):
    """A copy of a record with long strings cut and long lists summarised (the earlier detail line; nothing is cut from the file)."""
    if isinstance(obj, dict):
        return {k: _shrink(v, max_str, max_list) for k, v in obj.items()
                if k not in ("box", "source", "note", "compared", "oracle", "traffic_source", "first_mismatches")}
    if isinstance(obj, (list, tuple)):
        if len(obj) > max_list and all(isinstance(v, (int, float)) for v in obj):
            s = sorted(obj)
            return {"n": len(obj), "min": s[0], "median": s[len(s) // 2], "max": s[-1]}
        return [_shrink(v, max_str, max_list) for v in obj[:max_list]]
    if isinstance(obj, str) and len(obj) > max_str:
        return obj[:max_str - 1] + "~"
    return obj


def final_line(out):
    """The last stdout line: the fields the driver's contract names + roofline + cpu_baseline + one parity verdict per BASELINE
    config, <= FINAL_LINE_LIMIT bytes. Everything else is in bench_detail.json (and, trimmed, on the line before)."""
    roof = out.get("roofline") or {}
    hv = roof.get("hbm_view") or (roof if roof.get("bound") == "hbm" else {})
    binfo = out.get("build_info") or {}
    sym = lambda k: kernel_symbol(k, binfo)
    r = None
    if roof:
        r = {"bound": roof.get("bound"), "kernel": sym(roof.get("kernel")), "achieved": roof.get("achieved"),
             "peak": roof.get("peak"), "unit": roof.get("unit"), "frac": roof.get("frac"),
             "avg_launch_ms": roof.get("avg_launch_ms"),
             "algorithmic_bytes_per_launch": hv.get("algorithmic_bytes_per_launch"), "traffic": roof.get("traffic"),
             "hbm_view": {"achieved": hv.get("achieved"), "peak": hv.get("peak"), "unit": hv.get("unit"), "frac": hv.get("frac")}}
        if roof.get("bound") == "valu_f64":
            r["valu_insts_per_point"] = roof.get("valu_insts_per_point")
            r["f64_arithmetic_insts_per_point"] = roof.get("f64_arithmetic_insts_per_point")
            r["valu_issue_frac"] = (roof.get("valu_issue") or {}).get("frac")
        big = roof.get("largest_hbm_kernel")
        if big:
            r["largest_hbm_kernel"] = {"kernel": sym(big.get("kernel")), "achieved": big.get("achieved"), "frac": big.get("frac"),
                                       "avg_launch_ms": big.get("avg_launch_ms"), "traffic": big.get("traffic"),
                                       "algorithmic_bytes_per_launch": big.get("algorithmic_bytes_per_launch")}
        r["profile_matches_build"] = roof.get("profile_matches_build")
    es = out.get("encode_sort") or None
    if es:
        rs = es.get("record_sort") or {}
        es = {"GB/s": es.get("GB/s"), "frac_of_8TBps": None if es.get("GB/s") is None else round(es["GB/s"] / HBM_PEAK_GBS, 4),
              "ms": es.get("ms"), "algorithmic_bytes_per_point": round(es.get("algorithmic_bytes_per_point") or 0.0, 2),
              "record_sort": {"GB/s": rs.get("GB/s"), "frac_of_8TBps": None if rs.get("GB/s") is None else round(rs["GB/s"] / HBM_PEAK_GBS, 4),
                              "ms": rs.get("ms")}}
    cpu = out.get("cpu_baseline")
    if cpu:
        cpu = {k: (str(v)[:200] if k == "sample" else v) for k, v in cpu.items() if k != "note"}
    cfg = out.get("config") or {}
    sh = out.get("sharded") or None
    parity = {"config2": _leg_parity(out, "parity") if out.get("parity") else None,
              "config1": _leg_parity(out.get("config1"), None), "config4": _leg_parity(out.get("query")),
              "config5": _leg_parity(out.get("config5")), "intensity": _leg_parity(out.get("intensity")),
              "sharded": None if not sh else ({"ok": False, "error": str(sh["error"])[:120]} if "error" in sh else
                                              {"ok": bool(sh.get("ok")), "world1_digest_equal": (sh.get("world1") or {}).get("digest_equal"),
                                               "virtual8_digest_equal": {m: v.get("digest_equal") for m, v in (sh.get("virtual8") or {}).items()
                                                                         if isinstance(v, dict)}})}
    e2e = out.get("end_to_end") or None
    if e2e:
        fb = e2e.get("from_batches") or {}
        e2e = {"Mpoints_per_s_incl_files": e2e.get("Mpoints_per_s_incl_files"),
               "from_ply_Mpoints_per_s_incl_files": (e2e.get("from_ply_file") or {}).get("Mpoints_per_s_incl_files"),
               "from_batches_Mpoints_per_s_incl_files": fb.get("Mpoints_per_s_incl_files")}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                    "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {"workload": str(cfg.get("workload", ""))[:260], "scope": str(cfg.get("scope", ""))[:200],
                      "points_per_gpu": cfg.get("points_per_gpu"), "points_total": cfg.get("points_total"),
                      "nodes": cfg.get("nodes"), "parallelism": str(cfg.get("parallelism", ""))[:120]}
    line.update({"roofline": r, "encode_sort": es, "cpu_baseline": cpu, "parity": parity, "tree_digest": out.get("tree_digest"),
                 "end_to_end": e2e,
                 "kernel_ms_per_step": {sym(k): v for k, v in (out.get("kernel_ms_per_step") or {}).items()}})
    for k in ("n1_same_cloud", "rccl_ranks", "sharded_stage_ms"):
        if out.get(k) is not None:
            line[k] = out[k]
    c5, q = out.get("config5") or {}, out.get("query") or {}
    legs = {}
    if c5.get("value") is not None:
        legs["config5_Mpoints_per_s"] = c5.get("value")
    if (sh or {}).get("world1"):
        legs["sharded_world1_ms"] = sh["world1"].get("ms_per_step")
    if q.get("value") is not None:
        legs["config4_" + str(q.get("unit", "value")).replace(" ", "_")] = q.get("value")
    if legs:
        line["legs"] = legs
    line["detail"] = "bench_detail.json (the whole record; a trimmed copy is the stdout line before this one)"
    # never longer than the limit: drop the least important objects first
    for drop in ("legs", "kernel_ms_per_step", "end_to_end", "encode_sort"):
        if len(json.dumps(line)) <= FINAL_LINE_LIMIT:
            break
        line.pop(drop, None)
    if len(json.dumps(line)) > FINAL_LINE_LIMIT:
        line["config"]["workload"] = line["config"]["workload"][:80]
        line["config"].pop("scope", None)
        if line.get("cpu_baseline"):
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:60]
    return line


def detail_line(out):
    """The earlier stdout line: the whole record with notes, box dumps and step lists cut, legs dropped from the back until it fits."""
    d = {"bench_detail": _shrink(out)}
    for drop in ("config1", "intensity", "end_to_end", "sharded", "query", "config5", "roofline"):
        if len(json.dumps(d)) <= DETAIL_LINE_LIMIT:
            break
        leg = d["bench_detail"].get(drop)
        if isinstance(leg, dict):
            d["bench_detail"][drop] = {k: v for k, v in leg.items() if not isinstance(v, (dict, list))}
    if len(json.dumps(d)) > DETAIL_LINE_LIMIT:
        d = {"bench_detail": {"see": "bench_detail.json", "keys": sorted(out)}}
    return d


def emit(out):
    """bench_detail.json (whole record) -> the trimmed detail line -> the line of record, in that order; the last line is the
    one the driver parses."""
    text = json.dumps(out)
    for path in (os.path.join(ROOT, "bench_detail.json"), os.path.join(ROOT, "gpurun_out", "bench_detail.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as f:
                    f.write(text + "\n")
        except OSError as e:  # a read-only tree must not cost the line of record
            print(f"bench_detail.json not written: {e}", file=sys.stderr)
    print(json.dumps(detail_line(out)), flush=True)
    print(json.dumps(final_line(out)), flush=True)


def n1_same_cloud_leg(args, torch, pcv, dev, device_index, total, steps=3, warmup=2):
    """The N = 1 reference of a `--gpus N` line: rank 0 regenerates the WHOLE config-3 cloud (the block generator makes any
    slice reproducible) and builds it unsharded — the same scope as the sharded step (K1 inside), 1 B points = 27 GB of input
    + ~80 GB of build scratch on one 288 GB part — so that value / n1_same_cloud.value is a like-for-like speed-up."""
    try:
        c1 = pcv.Context(device_index)
        x, y, z, rgb = make_cloud_slice(torch, total, 0, total, seed=args.seed3, device=dev)
        for _ in range(warmup):
            c1.build(args.resolution, None, x, y, z, rgb).free()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            c1.build(args.resolution, None, x, y, z, rgb).free()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out = {"points": total, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3),
               "value": round(total / dt / 1e6, 2), "unit": "Mpoints/s",
               "scope": "rank 0, one GPU, unsharded pcv_build_octree with K1 inside the step, the same cloud"}
        if not args.no_n1_digest:
            t = c1.build(args.resolution, None, x, y, z, rgb)
            out["nodes"], out["tree_digest"] = t.num_nodes, digest_of_digests(tree_digests(t))
            t.free()
        del x, y, z, rgb
        c1.close()
        torch.cuda.empty_cache()
        return out
    except Exception as e:  # noqa: BLE001 - the sharded measurement must not die with its reference
        torch.cuda.empty_cache()
        return {"error": f"{type(e).__name__}: {e}"[:300]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 timed builds after 5 untimed ones — the first builds of a process run 5-8 % slower (clocks ramp up,
    # the pool fills), a step is 7 ms, so the whole timed region is still 0.15 s
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--points", type=int, default=0,
                    help="points per GPU (default: 100 M on one GPU = config 2; 1e9 / N on N GPUs = config 3)")
    ap.add_argument("--resolution", type=float, default=0.001)
    ap.add_argument("--cpu-sample", type=int, default=100_000_000, help="points of the workload timed on the CPU")
    ap.add_argument("--ecef", action="store_true",
                    help="BASELINE config 5: place the cloud at ECEF magnitudes (|p| ~ 6.4e6 m)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path (owner kernel, partition, exchange) even with one rank")
    ap.add_argument("--shard-mode", choices=["buckets", "octants", "both"], default="octants",
                    help="multi-GPU ownership: root octant c -> rank c %% N (BASELINE north_star), or 64 level-2 buckets "
                         "bin-packed onto ranks (the skew remedy); both: --virtual-ranks only")
    ap.add_argument("--virtual-ranks", type=int, default=0,
                    help="config 3 with V thread-ranks on ONE GPU, merged octree compared with the single-GPU build")
    ap.add_argument("--seed3", type=int, default=2, help="seed of the config-3 cloud")
    ap.add_argument("--config3", action="store_true",
                    help="take the config-3 cloud (block-wise generator, --points per rank or 1e9 / N) even with one rank: with "
                         "--force-sharded this is the --gpus N code path at world size 1")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed region: one more build compared byte for byte with the CPU oracle (default at N=1 "
                         "on the plain config-2 run; this flag forces it elsewhere, incl. the sharded path)")
    ap.add_argument("--no-parity", action="store_true", help="skip the default oracle comparison of the N=1 run (A/B timing runs)")
    ap.add_argument("--digest", action="store_true", help="print a digest of the built octree (no oracle): A/B runs compare it")
    ap.add_argument("--query", action="store_true", help="BASELINE config 4 (frustum path) instead of the build")
    ap.add_argument("--frusta", type=int, default=10_000)
    ap.add_argument("--cull-frusta", type=int, default=100)
    ap.add_argument("--verify-frusta", type=int, default=10_000, help="frusta whose relations / sizes / visible lists are compared with the oracle")
    ap.add_argument("--verify-cull-frusta", type=int, default=100, help="culled frusta whose query_points result is compared with the oracle")
    ap.add_argument("--query-steps", type=int, default=5, help="timed repetitions of the node-cull launch sequence (config 4)")
    ap.add_argument("--config1", action="store_true", help="BASELINE config 1 (CPU plumbing line)")
    ap.add_argument("--no-legs", action="store_true",
                    help="default N=1 line only: skip the config-4 (`query`) and config-5 (`config5`) legs after the timed region")
    ap.add_argument("--intensity-points", type=int, default=20_000_000,
                    help="points of the colour + intensity leg of the default line (the reference binary's payload); 100000000 for the full-size record")
    ap.add_argument("--only-intensity", action="store_true", help="run the colour + intensity leg alone (with --intensity-points) and print its record")
    ap.add_argument("--config5-points", type=int, default=500_000_000)
    ap.add_argument("--config5-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-files end-to-end leg (N=1 only)")
    ap.add_argument("--kernel-events", choices=["major", "all", "none"], default="major",
                    help="HIP events around kernel launches inside the timed region: the kernels that pass over the whole "
                         "cloud (default), every launch (~100 per build, costs the step a few percent), or none")
    ap.add_argument("--no-kernel-events", action="store_true", help="same as --kernel-events none")
    ap.add_argument("--exact-pipeline", action="store_true",
                    help="force the exact two-chain pipeline (K2 keys + key sort + node split + K5) instead of the single-chain build")
    ap.add_argument("--n1-same-cloud", action="store_true",
                    help="config 3: also build the whole cloud unsharded on rank 0 before the timed region (default with N > 1)")
    ap.add_argument("--no-n1", action="store_true", help="N > 1: skip the unsharded reference build of the same cloud on rank 0")
    ap.add_argument("--no-n1-digest", action="store_true", help="N > 1: no digest comparison of the merged octree with the unsharded one")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole record as ONE stdout line (rounds 1-5; tools/ that post-process it) instead of "
                         "bench_detail.json + a trimmed detail line + the <= 4 KB line of record")
    ap.add_argument("--fixed-bbox", action="store_true",
                    help="take the bounding box as an argument (K1 outside the step), as round 1 measured")
    args = ap.parse_args()
    if args.no_kernel_events:
        args.kernel_events = "none"

    if args.config1:
        print(json.dumps(config1(args)), flush=True)
        return
    if args.query:
        if not args.points:
            args.points = 100_000_000
        print(json.dumps(query_bench(args)), flush=True)
        return

    import numpy as np
    import torch
    import point_cloud_viewer_amd as pcv

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.only_intensity:
        ctx = pcv.Context(local_rank)
        out = intensity_leg(args, torch, pcv, ctx, dev, args.intensity_points, steps=max(1, min(args.steps, 10)))
        out["build_hash"] = build_hash()
        print(json.dumps(out), flush=True)
        return
    if args.virtual_ranks:
        out = run_virtual_ranks(args, torch, pcv, dev)
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
        return
    if args.shard_mode == "both":
        raise SystemExit("--shard-mode both needs --virtual-ranks")
    dist = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if world == 1:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    sharded = world > 1 or args.force_sharded
    offset = (-2.7e6, -4.3e6, 3.8e6) if args.ecef else (0.0, 0.0, 0.0)
    config3 = (world > 1 or args.config3) and not args.ecef
    if config3:
        # BASELINE config 3: ONE cloud (1 B points unless --points gives every rank its share), rank r holds slice r
        total = args.points * world if args.points else 1_000_000_000
        lo, hi = rank * total // world, (rank + 1) * total // world
        n = hi - lo
        x, y, z, rgb = make_cloud_slice(torch, total, lo, n, seed=args.seed3, device=dev)
    else:
        n = args.points or 100_000_000
        total = n * world
        x, y, z, rgb = make_cloud(torch, n, seed=1 + rank, device=dev, offset=offset)  # ends with a device synchronize
    ctx = pcv.Context(local_rank)  # the library's own stream; torch work is ordered explicitly (wait_torch)
    n1 = None
    if config3 and rank == 0 and (world > 1 or args.n1_same_cloud) and not args.no_n1:
        # the SAME cloud built unsharded on this one GPU (own context, released afterwards) while the other ranks wait at the
        # warm-up's first collective: the line then carries its own N = 1 reference (VERDICT r05 #2b)
        n1 = n1_same_cloud_leg(args, torch, pcv, dev, local_rank, total)

    info = {}
    if not sharded:
        bbox = None
        if args.fixed_bbox:
            bmin, bmax = ctx.aabb_reduce(x, y, z)
            bbox = pcv.Aabb(bmin, bmax)

        def step(details=True):
            t = ctx.build(args.resolution, bbox, x, y, z, rgb,  # bbox None: K1 runs inside the step
                          single_chain=False if args.exact_pipeline else None)
            if details:  # (not inside the timed region: a dozen ctypes calls per step)
                info["nodes"], info["stages"], info["build"] = t.num_nodes, t.stage_ms(), t.build_info()
                info.setdefault("all_stages", []).append({k: round(v, 2) for k, v in info["stages"].items()})
            info.setdefault("gpu_ms", []).append(round(t.total_gpu_ms(), 3))
            t.free()
    else:
        from point_cloud_viewer_amd import distributed as pdist
        builder = pdist.ShardedOctreeBuilder(ctx, dist, dev, shard_mode=args.shard_mode)
        bbox = builder.global_bbox(x, y, z)

        def step(details=True):
            # the same scope as the unsharded step: K1 over the local slice + ONE 6-number all-reduce (min of [lo, -hi]) are
            # INSIDE the step (VERDICT r05 #2a); --fixed-bbox takes the box as an argument, as the reference's library entry does
            box = bbox if args.fixed_bbox else builder.global_bbox(x, y, z)
            r = builder.build(args.resolution, box, x, y, z, rgb)
            info["nodes"], info["stages"] = r.num_nodes_local, r.stage_ms
            info["build"] = r.local.build_info() if hasattr(r.local, "build_info") else None
            info["exchange"] = r.exchange_info()
            r.free()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    box = {"idle_before_warmup": box_info()} if rank == 0 else None
    if args.kernel_events != "none":
        ctx.set_profiling("major" if args.kernel_events == "major" else True)  # during warmup: the event pool exists before the timed region
    import gc
    gc.collect()  # before the warm-up: a collection between warm-up and timed region leaves the GPU idle for tens of
    gc.disable()  # milliseconds, its clocks drop, and the first timed steps pay for the ramp (5.4 / 5.2 ms against 4.9)
    for _ in range(args.warmup):  # (no collector pauses inside the timed region: the steps allocate no garbage to speak of)
        step()
    if args.kernel_events != "none":
        ctx.reset_kernel_stats()
    barrier()
    t0 = time.perf_counter()
    step_marks = [t0]
    for _ in range(args.steps):
        step(False)
        step_marks.append(time.perf_counter())  # every step ends with a stream sync inside the library
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if rank == 0:
        box["right_after_timed_region"] = box_info()
    per_step_ms = [round((b - a) * 1e3, 3) for a, b in zip(step_marks[:-1], step_marks[1:])]
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ctx.set_profiling(False)
    kstats = ctx.kernel_stats()
    # stage times: the 16 stage events of a build cost its stream ~0.1 ms (PCV_BUILD_STAGE_TIMES, off by default and off in
    # the timed region); `stage_ms` comes from two more builds AFTER the timed region with the stage events on
    timed_gpu_ms = list(info.get("gpu_ms") or [])
    ctx.stage_times = True
    for _ in range(2):
        step()
    ctx.stage_times = False
    info["gpu_ms"] = timed_gpu_ms
    per_rank = None
    if dist is not None:  # every rank's view of its last step: rows / bytes moved and the stage times (exchange, local build, merge)
        per_rank = [None] * world
        dist.all_gather_object(per_rank, info.get("exchange"))

    total_points = total * args.steps
    value = total_points / elapsed / 1e6  # Mpoints/s, whole job

    roofline, encode_sort = None, None
    timed = {k: v for k, v in kstats.items() if v[0] > 0}
    plain = n == 100_000_000 and world == 1 and not sharded and not args.ecef and not args.exact_pipeline
    bhash = build_hash()
    if timed:
        # packed 12-byte records (single-chain build, build_info record_bytes == 12): u32 key + uint2 payload
        rec_b = float((info.get("build") or {}).get("record_bytes") or 20)
        if rec_b == 12.0:
            ALGO_BYTES.update({"downsweep_rec_kernel": 2 * 12.0, "promote_settle_kernel": 12.0 + 7.0 / 8.0 * 9.0,
                               "spec_encode_kernel": 24.0 + 3.0 + 12.0})

        # the record sort's second pass finishes most leaves itself (build_info settled_in_sort): `settle` then only sees the points
        # of the leaves left over, and the pass moves 12 B in + 12 B out for those
        settled = float((info.get("build") or {}).get("settled_in_sort") or 0)

        def algo_bytes(name):
            if settled and name == "promote_settle_kernel":
                return ALGO_BYTES[name] * (n - settled)
            if settled and name == "downsweep_settle_kernel":
                return ALGO_BYTES[name] * settled + 24.0 * (n - settled)
            return ALGO_BYTES.get(name, 0.0) * n

        def hbm_view(name):
            launches, ms = timed[name]
            avg_ms = ms / launches
            gbs = algo_bytes(name) / (avg_ms * 1e-3) / 1e9
            return {"kernel": name, "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                    "algorithmic_bytes_per_launch": algo_bytes(name)}

        profile_state = {"matches": None}

        def pmc(name, key):
            """Per-launch figure from this round's rocprofv3 PMC passes of the same command (tools/profile_bench.sh) — only
            when the profile was taken from the library that is running (build_hash stamped into the file)."""
            path = os.path.join(ROOT, "profiles", f"{ROUND}_bench_100M_{key}.json")
            if not plain or not os.path.exists(path):
                return None, None
            with open(path) as f:
                prof = json.load(f)
            same = prof.get("build_hash") == bhash
            profile_state["matches"] = same if profile_state["matches"] is None else (profile_state["matches"] and same)
            if not same:
                return None, os.path.relpath(path, ROOT) + f" (STALE: taken from build {prof.get('build_hash')}, running {bhash})"
            per = prof.get("per_launch", {})
            # the library's event names and the rocprof kernel names differ for some kernels (the profile lists the larger
            # launch where one kernel runs with several key widths: the u32 upsweep, not the sample's u64 one)
            alias = {"promote_settle_kernel": "promote_settle_leaf_kernel", "promote_climb_kernel": "promote_climb_leaf_kernel",
                     "downsweep_rec_kernel": "downsweep_rec12_kernel", "upsweep_kernel<u32>": "upsweep_kernel",
                     "spec_encode_kernel": "chain_pass_kernel"}
            return per.get(name, per.get(alias.get(name, name))), os.path.relpath(path, ROOT)

        dom = max(timed, key=lambda k: timed[k][1])  # dominant kernel by accumulated time inside the timed region
        view = hbm_view(dom)
        traffic, traffic_src = pmc(dom, "traffic")
        if dom in VALU_F64_BOUND:
            # f64-VALU bound by construction (two correctly rounded f64 divisions per coordinate and level): priced
            # against the vector FP64 issue roof, instructions from the SQ counters of this round's profile of the same
            # command, f64 share from the disassembly (tools/isa_mix.py)
            insts, insts_src = pmc(dom, "valu")
            roofline = {"bound": "valu_f64", "kernel": dom, "peak": F64_VALU_PEAK_GINST, "unit": "G f64-inst/s",
                        "avg_launch_ms": view["avg_launch_ms"], "launches": view["launches"]}
            if insts:
                g = insts["f64_valu_insts_per_point"] * n / (view["avg_launch_ms"] * 1e-3) / 1e9
                issue = valu_issue_model(insts, n, view["avg_launch_ms"])
                roofline.update({"achieved": round(g, 1), "frac": round(g / F64_VALU_PEAK_GINST, 4),
                                 "valu_issue": issue,
                                 "valu_insts_per_point": insts["valu_insts_per_point"],
                                 "f64_arithmetic_insts_per_point": insts["f64_valu_insts_per_point"],
                                 "f64_arithmetic_share_of_valu": insts["f64_share"],
                                 "f64_pipe_share_static_isa": insts.get("f64_share_static_isa"),
                                 "source": insts_src + " (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 of a rocprofv3 --pmc pass of "
                                                       "the same command and the same build; compares / min / max / conversions "
                                                       "also issue on the f64 pipe and are only in the static share)"})
            else:
                roofline.update({"achieved": None, "frac": None,
                                 "source": insts_src or "no SQ-counter pass of this round found (tools/profile_bench.sh)"})
            if insts and (roofline.get("valu_issue") or {}).get("cycles_per_class"):
                # the same figure against what THIS part sustains: 1 024 SIMDs x 64 lanes / measured cycles per f64 instruction x
                # the clock it held during the kernel (the nominal peak assumes 4.0 cycles at 2.4 GHz)
                vi = roofline["valu_issue"]
                pk = 1024 * 64 / vi["cycles_per_class"]["f64_add_mul_fma"] * vi["sustained_clock_GHz"]
                roofline["peak_sustained_measured"] = round(pk, 1)
                roofline["frac_of_sustained_measured_peak"] = round(roofline["achieved"] / pk, 4)
            roofline["traffic"] = traffic
            roofline["traffic_source"] = traffic_src
            roofline["hbm_view"] = dict(view, traffic=traffic, traffic_source=traffic_src)
            # the largest HBM-bound kernel next to it
            rest = {k: v for k, v in timed.items() if k not in VALU_F64_BOUND and k in ALGO_BYTES}
            if rest:
                hk = max(rest, key=lambda k: rest[k][1])
                ht, hs = pmc(hk, "traffic")
                roofline["largest_hbm_kernel"] = dict(hbm_view(hk), bound="hbm", traffic=ht, traffic_source=hs)
                roofline["hbm_kernels"] = {k: dict(GBps=hbm_view(k)["achieved"], frac=hbm_view(k)["frac"], avg_launch_ms=hbm_view(k)["avg_launch_ms"],
                                                   traffic=pmc(k, "traffic")[0]) for k in rest}
        else:
            roofline = dict(view, bound="hbm", traffic=traffic, traffic_source=traffic_src)
        roofline["build_hash"] = bhash
        roofline["profile_matches_build"] = profile_state["matches"]
        # encode+sort figure the BASELINE metric names (stage events of the last step)
        st = info.get("stages") or {}
        binfo = info.get("build") or {}
        rec = timed.get("downsweep_rec_kernel")
        rec_passes = rec[0] / args.steps if rec else 0.0
        rec_ms = st.get("sort_records", 0.0)
        # the settling second pass (downsweep_settle_kernel) is queued with the promotion stage, once the node tables are up: its
        # event time joins the sort's here, its bytes are ALGO_BYTES' (it also does `settle`'s work for the leaves it finishes)
        fused = timed.get("downsweep_settle_kernel")
        fused_passes = fused[0] / args.steps if fused else 0.0
        fused_ms = fused[1] / args.steps if fused else 0.0
        pass_b = 4.0 + 2 * rec_b  # per pass: 4 B histogram read + record read + record write
        # single-chain build with the first histogram taken from the rank counts (no upsweep_map launch): that pass does not
        # read the keys an extra time; with upsweep_map it reads them and writes the mapped keys back (4 B more)
        rows_path = bool(binfo.get("single_chain")) and rec_b == 12.0 and "upsweep_map_kernel" not in timed
        # ... and with the second pass cut into whole first-pass runs, its histogram comes from the same counts: no
        # upsweep<u32> launch either
        rows_both = rows_path and "upsweep_kernel<u32>" not in timed
        sort_b = rec_passes * pass_b + (-4.0 * rec_passes if rows_both else -4.0 if rows_path else (4.0 if "upsweep_map_kernel" in timed else 0.0))
        sort_b += fused_passes * algo_bytes("downsweep_settle_kernel") / n
        rec_ms += fused_ms
        record_sort = None if not rec else {"passes": rec_passes + fused_passes, "ms": round(rec_ms, 3), "record_bytes": rec_b,
                                            "second_pass_settles_the_leaves": bool(fused),
                                            "GB/s": round(n * sort_b / (rec_ms * 1e-3) / 1e9, 1) if rec_ms else None,
                                            "algorithmic_bytes_per_point": sort_b,
                                            "histograms": "both from the rank counts" if rows_both else "first from the rank counts" if rows_path else "own passes over the keys"}
        if binfo.get("single_chain"):
            # single-chain build: the encode is the one chain pass (read xyz + rgb, write rank + payload), the sort is the
            # stable record sort by leaf rank (per pass: 4 B histogram read + record read + record write) — no key sort exists
            es_ms = st.get("leaf_encode", 0.0) + rec_ms
            es_bytes_pp = 27.0 + rec_b + sort_b
            encode_sort = {"GB/s": round(n * es_bytes_pp / (es_ms * 1e-3) / 1e9, 1) if es_ms else None, "ms": round(es_ms, 3),
                           "pipeline": "single-chain: spec_encode + record sort", "algorithmic_bytes_per_point": es_bytes_pp,
                           "record_sort": record_sort}
        else:
            es_ms = st.get("chain_keys", 0.0) + st.get("sort_keys", 0.0)
            key32 = binfo.get("key_levels", 21) <= 10
            key_bytes = 4.0 if key32 else 8.0
            down = "downsweep_kernel<u32>" if key32 else "downsweep_kernel<u64>"
            passes = timed.get(down, (0, 0))[0] / args.steps
            if not key32:
                passes = max(0.0, passes - 5)  # the depth probe's own tiny u64 sort
            sort_ms = sum(timed.get(k, (0, 0.0))[1] for k in (down, down.replace("down", "up"), "scan_kernel")) / args.steps
            es_bytes_pp = 24.0 + key_bytes + passes * 3 * key_bytes
            encode_sort = {"GB/s": round(n * es_bytes_pp / (es_ms * 1e-3) / 1e9, 1) if es_ms else None,
                           "ms": round(es_ms, 3), "pipeline": "exact: chain_keys + key sort", "key_bits": int(key_bytes * 8),
                           "sort_passes": passes, "algorithmic_bytes_per_point": es_bytes_pp,
                           "key_sort_only": {"ms": round(sort_ms, 3),
                                             "GB/s": round(n * passes * 3 * key_bytes / (sort_ms * 1e-3) / 1e9, 1) if sort_ms else None},
                           "record_sort": record_sort}

    # ---- parity: one more build compared with the CPU oracle on the same cloud (default on the plain N=1 run) ----
    parity = None
    tree_digest = None
    if not sharded and rank == 0 and (args.verify or (plain and not args.no_parity)):
        parity = verify_build(ctx, args.resolution, x, y, z, rgb)
        tree_digest = parity.get("tree_digest")
    elif sharded and (args.verify or args.digest or (n1 is not None and "tree_digest" in n1) or
                      (world > 1 and not args.no_n1 and not args.no_n1_digest)):
        # sharded path: every rank hashes its subtrees, rank 0 merges them with the all-reduced top nodes; --verify:
        # rank 0 regenerates the whole cloud (the block generator makes any slice reproducible) for the oracle
        r = builder.build(args.resolution, builder.global_bbox(x, y, z), x, y, z, rgb)
        mine = sharded_digests(r)
        r.free()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank == 0:
            merged, dup = {}, 0
            for part in gathered:
                for k, v in part.items():
                    dup += k in merged
                    merged[k] = v
            tree_digest = digest_of_digests(merged)
            if n1 is not None and "tree_digest" in n1:
                n1["digest_equal"] = bool(n1["tree_digest"] == tree_digest and dup == 0)
                n1["speedup"] = None  # filled in below, once `value` is known
            if args.verify:
                if config3:
                    wx, wy, wz, wrgb = make_cloud_slice(torch, total, 0, total, seed=args.seed3, device=dev)
                elif world == 1:
                    wx, wy, wz, wrgb = x, y, z, rgb
                else:
                    raise SystemExit("--verify on the sharded path needs the config-3 cloud (or one rank)")
                want, stats = oracle_digests(args.resolution, bbox.min, bbox.max, wx, wy, wz, wrgb)
                parity = {"oracle": "closed-form CPU restatement of the reference (oracle/pcv_oracle_build.cpp), not the Rust binary",
                          "points": total, "sharded": True, "ranks": world, "nodes_built_twice": dup}
                parity.update(compare_digests(want, merged))
                parity.update({"bbox_equals_numpy_minmax": stats["bbox_equals_numpy_minmax"], "tree_digest": tree_digest,
                               "max_abs_position_error_m": stats["max_abs_position_error"], "oracle_s": stats["oracle_s"],
                               "oracle_threads": stats["oracle_threads"],
                               "compared": "num_points, encoding, blake2b-128 of .xyz and .rgb of every node (merged over the ranks)"})
                parity["ok"] = bool(parity["ok"] and dup == 0 and stats["bbox_equals_numpy_minmax"])
                del wx, wy, wz, wrgb
    elif args.digest and rank == 0:
        t = ctx.build(args.resolution, None, x, y, z, rgb, single_chain=False if args.exact_pipeline else None)
        tree_digest = digest_of_digests(tree_digests(t))
        t.free()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not sharded:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import shutil
        import tempfile
        import oracle_lib as O
        cores = O.num_procs()
        bmin, bmax = ctx.aabb_reduce(x, y, z)
        base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        m, cdt = min(args.cpu_sample, n), None
        # the literal build keeps up to ~2 encoded copies of the cloud on the file system (<= 30 B/point) and, like the
        # reference, does not survive a full disk: size the sample to the space that is really there
        free = shutil.disk_usage(base).free
        m = int(min(m, free // 80))
        while cdt is None and m >= 1_000_000:  # the literal build streams node files: halve the sample if tmpfs is short
            hx, hy, hz = x[:m].cpu().numpy(), y[:m].cpu().numpy(), z[:m].cpu().numpy()
            hrgb = rgb[:m].cpu().numpy()
            d = tempfile.mkdtemp(prefix="pcv_cpu_baseline_", dir=base)
            try:
                c0 = time.perf_counter()
                O.build_literal_dir(os.path.join(d, "octree"), args.resolution, bmin, bmax, hx, hy, hz, hrgb,
                                    threads=cores)
                cdt = time.perf_counter() - c0
            except Exception as e:  # noqa: BLE001 - reported, then retried smaller
                print(f"cpu_baseline: {m} points failed ({e}); retrying with half", file=sys.stderr)
                m //= 2
            finally:
                shutil.rmtree(d, ignore_errors=True)
        cpu = None if cdt is None else {"value": round(m / cdt / 1e6, 3), "unit": "Mpoints/s", "cores": cores, "kind": "port",
               "sample": f"first {m} points of the same cloud, literal file-streaming restatement of the reference "
                         f"(oracle/pcv_oracle_build.cpp) incl. node files on tmpfs, {cores} OpenMP threads, {cdt:.1f} s",
               "note": "scope differs from `value` (device-resident build without file writes) — the like-for-like figure is "
                       "end_to_end.Mpoints_per_s_incl_files"}

    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e and not sharded:
        # One untimed-region pass from HOST arrays to files on tmpfs: H2D staging + build, D2H of the node blobs,
        # threaded file writes. Never part of `value` (tier rule (4)); reported so the PCIe / file-system cost is visible.
        import shutil
        import tempfile
        hx, hy, hz, hrgb = x.cpu().numpy(), y.cpu().numpy(), z.cpu().numpy(), rgb.cpu().numpy()
        base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
        d = tempfile.mkdtemp(prefix="pcv_e2e_", dir=base)
        try:
            for attempt in range(2):  # the first pass pays one-time costs (pinned host blocks, staging buffers)
                shutil.rmtree(os.path.join(d, "octree"), ignore_errors=True)
                torch.cuda.synchronize()
                # (a) host arrays -> host blobs: pinned-ring H2D + build, then one D2H of the node blobs
                a0 = time.perf_counter()
                t = ctx.build(args.resolution, None, hx, hy, hz, hrgb)
                a1 = time.perf_counter()
                t.node_data(0, 0)  # forces the D2H of all node blobs (pinned host memory)
                a2 = time.perf_counter()
                t.free()
                # (b) host arrays -> files: the node files are written while the blobs are still coming down
                b0 = time.perf_counter()
                t = ctx.build(args.resolution, None, hx, hy, hz, hrgb)
                b1 = time.perf_counter()
                t.write_dir(os.path.join(d, "octree"))
                b2 = time.perf_counter()
                files = len(os.listdir(os.path.join(d, "octree")))
                t.free()
            # (b') the reference's own input: an iterator of PointsBatch — positions AoS, 500 000 points at a time
            # (src/lib.rs:52,102-107) — streamed through pcv_ingest_*: every batch is one pinned copy + one DMA + one
            # transposition kernel, the producer is free while it goes up; box folded during the ingest; five runs
            pos = np.empty((n, 3), dtype=np.float64)
            pos[:, 0], pos[:, 1], pos[:, 2] = hx, hy, hz
            batch = 500_000
            fb_total, fb_build = [], []
            for attempt in range(6):  # the first pass is a warm-up (the device staging chunks, the pool)
                shutil.rmtree(os.path.join(d, "octree"), ignore_errors=True)
                torch.cuda.synchronize()
                f0 = time.perf_counter()
                ing = ctx.ingest(n, has_intensity=False)
                for at in range(0, n, batch):
                    ing.append(pos[at:at + batch], hrgb[at:at + batch])
                t = ing.finish(args.resolution, None)
                f1 = time.perf_counter()
                t.write_dir(os.path.join(d, "octree"))
                f2 = time.perf_counter()
                fb_nodes = t.num_nodes
                t.free()
                if attempt:
                    fb_total.append(n / (f2 - f0) / 1e6)
                    fb_build.append((f1 - f0) * 1e3)
            del pos
            fb_total.sort()
            fb_build.sort()
            # (c) the reference's own entry: build_octree_from_file on a binary PLY (float x y z + uchar r g b = 15 bytes per
            # point) — the vertex records go up as they are and are decoded on the device (pcv_build_octree_from_ply)
            rec = np.empty(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
            rec["x"], rec["y"], rec["z"] = hx, hy, hz
            rec["r"], rec["g"], rec["b"] = hrgb[:, 0], hrgb[:, 1], hrgb[:, 2]
            ply_path = os.path.join(d, "cloud.ply")
            with open(ply_path, "wb") as f:
                f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\nproperty float x\nproperty float y\n"
                         "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n").encode())
                rec.tofile(f)
            del rec
            for attempt in range(2):
                shutil.rmtree(os.path.join(d, "octree"), ignore_errors=True)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                t = ctx.build_from_ply(args.resolution, ply_path)
                c1 = time.perf_counter()
                t.write_dir(os.path.join(d, "octree"))
                c2 = time.perf_counter()
                ply_nodes = t.num_nodes
                t.free()
        finally:
            shutil.rmtree(d, ignore_errors=True)
        e2e = {"h2d_plus_build_ms": round((a1 - a0) * 1e3, 1), "d2h_blobs_ms": round((a2 - a1) * 1e3, 1),
               "d2h_overlapped_with_file_writes_tmpfs_ms": round((b2 - b1) * 1e3, 1), "files": files,
               "Mpoints_per_s_h2d_build_d2h": round(n / (a2 - a0) / 1e6, 1),
               "Mpoints_per_s_incl_files": round(n / (b2 - b0) / 1e6, 1),
               "input_bytes_per_point": 27, "h2d_GBps": round(27.0 * n / max((a1 - a0) - elapsed / args.steps, 1e-9) / 1e9, 1),
               "from_batches": {"batch_points": batch, "runs": len(fb_total), "nodes": fb_nodes,
                                "Mpoints_per_s_incl_files": round(fb_total[len(fb_total) // 2], 1),
                                "Mpoints_per_s_incl_files_min_median_max": [round(fb_total[0], 1), round(fb_total[len(fb_total) // 2], 1), round(fb_total[-1], 1)],
                                "ingest_plus_build_ms_min_median_max": [round(fb_build[0], 1), round(fb_build[len(fb_build) // 2], 1), round(fb_build[-1], 1)],
                                "host_memory": "O(batch): the ring of pinned chunks (3 x 32 MiB)",
                                "note": "pcv_ingest_begin / _append x 200 / _finish (PCV_BUILD_COMPUTE_BBOX: the box folded during the "
                                        "ingest) + pcv_octree_write_dir; batches in the reference's layout (n x 3 f64 AoS, n x 3 u8), "
                                        "generation.rs:289-295, src/lib.rs:52,102-107"},
               "from_ply_file": {"input_bytes_per_point": 15, "read_upload_decode_build_ms": round((c1 - c0) * 1e3, 1),
                                 "d2h_overlapped_with_file_writes_tmpfs_ms": round((c2 - c1) * 1e3, 1),
                                 "Mpoints_per_s_incl_files": round(n / (c2 - c0) / 1e6, 1), "nodes": ply_nodes,
                                 "note": "build_octree_from_file on a binary PLY on tmpfs (float x y z + uchar r g b): vertex records "
                                         "pread into the pinned ring and uploaded as they are, cast to f64 on the device (ply.rs:488-493); "
                                         "the f32 coordinates make this a different (coarser) cloud than the one `value` is measured on"},
               "note": "pageable numpy inputs staged through a pinned ring (one DMA per 32 MiB chunk), bounding box computed "
                       "on the device; creating the node files in ONE directory serialises on the directory lock "
                       "(reference layout); not part of `value`"}

    # ---- the other single-GPU BASELINE configs, after the timed region of the default line: config 4 (frustum path on the
    # octree of this cloud) and config 5 (500 M ECEF points), each with its own timing and oracle parity ----
    query, config5, sharded_out, config1_out, intensity_out = None, None, None, None, None
    if plain and rank == 0 and not args.no_parity and not args.no_legs:
        try:  # the multi-GPU code path on this cloud: RCCL at world size 1 + 8 thread-ranks, digests against the verified one
            sharded_out = sharded_leg(args, torch, pcv, ctx, dev, x, y, z, rgb, tree_digest)
        except Exception as e:  # noqa: BLE001 - the leg reports its failure, the line of record still prints
            sharded_out = {"error": f"{type(e).__name__}: {e}", "ok": False}
        try:
            qt = ctx.build(args.resolution, None, x, y, z, rgb)
            query = query_leg(args, ctx, qt)
            query["roofline"]["profile"] = query_profile(bhash)
            qt.free()
        except Exception as e:  # noqa: BLE001 - the leg reports its failure, the line of record still prints
            query = {"error": f"{type(e).__name__}: {e}", "parity": {"ok": False}}
        del x, y, z, rgb
        try:
            config5 = config5_leg(args, torch, pcv, ctx, dev, args.config5_points)
        except Exception as e:  # noqa: BLE001
            config5 = {"error": f"{type(e).__name__}: {e}", "parity": {"ok": False, "mismatching_nodes": None}}
        try:
            intensity_out = intensity_leg(args, torch, pcv, ctx, dev, args.intensity_points, steps=10)
        except Exception as e:  # noqa: BLE001
            intensity_out = {"error": f"{type(e).__name__}: {e}", "parity": {"ok": False}}
        try:
            config1_out = config1_leg(pcv, ctx)
        except Exception as e:  # noqa: BLE001
            config1_out = {"error": f"{type(e).__name__}: {e}", "ok": False}

    if rank == 0:
        if config3:
            workload = (f"BASELINE config 3: ONE cloud of {total / 1e6:g} M Gaussian-cluster points (64 clusters, 1000 m cube, "
                        f"sigma 1-20 m, seed {args.seed3}), rank r holds the contiguous slice r of {n} points, sharded by "
                        f"{'root octant (top-3-bit prefix): octant c -> rank c % N' if args.shard_mode == 'octants' else '64 level-2 buckets bin-packed onto the ranks'}"
                        ", ONE all-to-all(v) over RCCL, f64 SoA xyz + u8 rgb, resolution 1 mm, full build + LOD promotion")
        else:
            workload = (("BASELINE config 5 (ECEF-offset f64 input): " if args.ecef else "BASELINE config 2: ") +
                        f"{n / 1e6:g} M Gaussian-cluster points (64 clusters, 1000 m cube, "
                        "sigma 1-20 m), f64 SoA xyz + u8 rgb, resolution 1 mm, full build + LOD promotion")
        out = {
            "metric": "octree-build Mpoints/sec", "value": round(value, 2), "unit": "Mpoints/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if (config3 and not args.points) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "scope": "device-resident inputs -> bounding box (K1" + (" outside the step" if args.fixed_bbox else "") +
                                (" + one 6-number all-reduce" if sharded and not args.fixed_bbox else "") +
                                (") + routing + ONE all-to-all(v) + per-rank subtree builds + top-node all-reduce: node table" if sharded else ") + node table") +
                                " + node-contiguous .xyz/.rgb bytes in HBM; no D2H of the blobs, no file writes",
                       "points_per_gpu": n, "points_total": total, "resolution": args.resolution, "nodes": info.get("nodes"),
                       "kernel_events_in_timed_region": args.kernel_events, "box": box,
                       "parallelism": "1 GPU" if world == 1 else
                       f"{world} GPUs, one process each, shard mode {args.shard_mode}: one all-to-all(v) over RCCL"},
            "roofline": roofline, "encode_sort": encode_sort, "cpu_baseline": cpu, "parity": parity, "tree_digest": tree_digest,
            "end_to_end": e2e, "query": query, "config5": config5, "sharded": sharded_out, "intensity": intensity_out,
            "config1": config1_out,
            "build_info": info.get("build"), "exchange": info.get("exchange"),
            "exchange_per_rank": None if per_rank is None else [
                None if e is None else {"rank": r, "rows_sent": e["rows_sent"], "rows_received": e["rows_received"],
                                        "bytes_sent": e["bytes_sent"], "bytes_received": e["bytes_received"], "ms": e["ms"]}
                for r, e in enumerate(per_rank)],
            "rccl_ranks": world if dist is not None else None,
            "n1_same_cloud": n1,
            "sharded_stage_ms": None if not sharded else {k: round(v, 3) for k, v in (info.get("stages") or {}).items()
                                                          if k in ("bbox", "exchange", "local_build", "top_merge")},
            "stage_ms": {k: round(v, 3) for k, v in (info.get("stages") or {}).items()},
            "wall_ms_each_step": per_step_ms, "gpu_ms_each_step": (info.get("gpu_ms") or [])[-args.steps:],
            "kernel_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in kstats.items() if v[0] > 0},
        }
        if n1 and n1.get("value"):
            n1["speedup"] = round(value / n1["value"], 4)
            n1["note"] = (f"speed-up of {world} rank(s) over one GPU on the same {total} points, both with the bounding box inside "
                          "the step; efficiency = speedup / n_gpus")
        if os.environ.get("PCV_BENCH_DEBUG"):
            out["stages_each_step"] = info.get("all_stages") or []
    # RCCL prints a version banner on stdout when the communicator goes away: tear it down and flush the C streams
    # first, so that the JSON line is the last thing this process writes
    if dist is not None:
        dist.destroy_process_group()
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    if rank == 0:
        if world > 1:
            time.sleep(1.5)  # let the other ranks finish writing their own teardown chatter first
        if args.full_line:
            print(json.dumps(out), flush=True)
        else:
            emit(out)


if __name__ == "__main__":
    main()
