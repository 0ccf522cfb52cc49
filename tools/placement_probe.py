#!/usr/bin/env python
"""Does the speed of the scatter kernels depend on WHERE the pool's blocks land? (round 5: the same binary runs its two record
downsweeps in 1.03 or 1.25 ms — in one gpurun call, on one box, with identical clocks, power and temperatures in rocm-smi.)
One process, one cloud; between rounds the context's pool is trimmed (every scratch block goes back to the driver) and a
few torch tensors of odd sizes are allocated / freed to move the next allocations somewhere else. Prints the kernel times
of every round:  python tools/placement_probe.py [points] [rounds]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import point_cloud_viewer_amd as pcv  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    dev = torch.device("cuda", 0)
    x, y, z, rgb = bench.make_cloud(torch, n, seed=1, device=dev)
    ctx = pcv.Context(0)
    junk = []
    g = torch.Generator().manual_seed(5)
    for r in range(rounds):
        for _ in range(3):
            ctx.build(0.001, None, x, y, z, rgb).free()
        ctx.set_profiling("major")
        ctx.reset_kernel_stats()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            ctx.build(0.001, None, x, y, z, rgb).free()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        ctx.set_profiling(False)
        ks = {k.replace("_kernel", ""): round(v[1] / 8, 3) for k, v in ctx.kernel_stats().items() if v[0] > 0}
        print(json.dumps({"round": r, "pool_contiguous": os.environ.get("PCV_POOL_CONTIG", "0"), "pool_vmm_chunk_MiB": os.environ.get("PCV_POOL_VMM", "0"), "ms_per_step": round(ms, 3), "kernels": ks,
                          "junk_tensors": len(junk)}), flush=True)
        # move the next round's blocks: give the pool back, then take / release odd-sized blocks in between
        ctx.trim()
        if r % 2 == 0:
            for _ in range(4):
                sz = int(torch.randint(50, 900, (1,), generator=g).item()) << 20
                junk.append(torch.empty(sz + 4096 * int(torch.randint(1, 200, (1,), generator=g).item()), dtype=torch.uint8, device=dev))
        else:
            junk = junk[::2]
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
