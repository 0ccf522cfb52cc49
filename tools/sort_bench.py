#!/usr/bin/env python
"""Micro-benchmark of K3 on realistic keys: the 30-bit path keys of the config-2 cloud (u32, 4 passes) and the
(rank, payload) record sort stand-in (u32 pairs, 2 passes). Prints the per-kernel HIP-event times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import point_cloud_viewer_amd as pcv
    from bench import make_cloud
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    dev = torch.device("cuda", 0)
    x, y, z, rgb = make_cloud(torch, n, seed=1, device=dev)
    ctx = pcv.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    bmin, bmax = ctx.aabb_reduce(x, y, z)
    keys64 = ctx.chain_keys(0.001, pcv.Aabb(bmin, bmax), x, y, z, 10)
    keys = (keys64 >> 33).to(torch.int32).contiguous()
    del x, y, z, rgb, keys64
    ctx.set_profiling(True)
    for name, mk in (("path keys", lambda: keys.clone()),
                     ("uniform", lambda: torch.randint(0, 1 << 30, (n,), dtype=torch.int32, device=dev))):
        for _ in range(2):
            k = mk()
            ctx.reset_kernel_stats()
            ctx.sort_keys32(k, 0, 30)
        st = {a: b for a, b in ctx.kernel_stats().items() if b[0]}
        print(name, {a: (b[0], round(b[1] / b[0], 4)) for a, b in st.items()}, "total ms", round(sum(b[1] for b in st.values()), 3))


if __name__ == "__main__":
    main()
