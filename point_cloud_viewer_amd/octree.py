"""Host-side mirror of the reference's octree-build surface over the C ABI (thin; all work is in HIP).

Reference names kept: `build_octree(output_directory, resolution, bounding_box, input, attributes)`
(src/octree/generation.rs:289-295), `NodeId` Display/parse (src/octree/node.rs:59-86), `Aabb`.
Inputs may be numpy arrays (host) or torch CUDA tensors (device-resident; zero copy).
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib as L


def _is_torch(a):
    return type(a).__module__.startswith("torch")


def node_name(id_high, id_low):
    """NodeId Display (node.rs:73-86): 'r' + octal index padded to `level` digits."""
    level = id_high >> 56
    index = ((id_high & ((1 << 56) - 1)) << 64) | id_low
    return "r" + "".join(str((index >> (3 * j)) & 7) for j in range(level - 1, -1, -1))


class Aabb:
    """geometry::Aabb (src/geometry/aabb.rs:13-27): mins/maxs are the inf/sup of the two corners."""

    def __init__(self, a, b):
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        self.min = np.minimum(a, b)
        self.max = np.maximum(a, b)


class _Buf:
    """Pointer + keep-alive for one input array."""

    def __init__(self, arr, dtype, what):
        self.keep = arr
        if arr is None:
            self.ptr, self.device, self.size = None, None, 0
            return
        if _is_torch(arr):
            import torch
            want = {np.float64: torch.float64, np.uint8: torch.uint8, np.float32: torch.float32,
                    np.uint64: torch.int64, np.uint32: torch.int32}[dtype]
            if arr.dtype != want and not (dtype is np.uint64 and arr.dtype == torch.uint64) and not (
                    dtype is np.uint32 and arr.dtype == torch.uint32):
                raise TypeError(f"{what}: expected torch dtype {want}, got {arr.dtype}")
            if not arr.is_contiguous():
                raise ValueError(f"{what}: tensor must be contiguous")
            self.ptr = arr.data_ptr()
            self.device = arr.is_cuda
            self.size = arr.numel()
        else:
            a = np.ascontiguousarray(arr, dtype=dtype)
            self.keep = a
            self.ptr = a.ctypes.data
            self.device = False
            self.size = a.size


class Context:
    """One pcv_ctx: bound to one HIP device and stream, not thread-safe (include/pcv_hip.h).

    stream=None (or handle 0, which the C ABI cannot tell from NULL) gives the context its OWN non-blocking stream.
    Torch's default stream is handle 0 and therefore can never be shared: work queued by torch (or RCCL) that produces
    the context's inputs must be ordered with `wait_torch()` / `wait_stream(handle)` before the first call that reads
    it, and `signal_torch()` orders torch work after asynchronous context calls. Calls that return results to the host
    end with a stream synchronisation of their own."""

    def __init__(self, device=0, stream=None):
        self.lib = L.load_library()
        h = C.c_void_p()
        self.device = int(device)
        self.shares_stream = bool(stream)
        rc = self.lib.pcv_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != L.PCV_OK:
            raise L.PcvError(rc, f"pcv_ctx_create(device={device}) failed — is a HIP device visible?")
        self.handle = h
        self._children = weakref.WeakSet()  # octrees / shape sets that borrow this context's pool
        # every build of this context records its per-stage GPU times (PCV_BUILD_STAGE_TIMES: ~0.1 ms of stream time per
        # build); off, stage_ms() of a tree reports the total only. build(stage_times=...) overrides it per call.
        self.stage_times = False

    def trim(self):
        """Give the cached scratch blocks of the context's pool back to the driver (pcv_ctx_trim)."""
        self._check(self.lib.pcv_ctx_trim(self.handle))

    def close(self):
        for child in list(getattr(self, "_children", [])):
            child.free()
        if getattr(self, "handle", None):
            self.lib.pcv_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_profiling(self, enabled=True):
        """Bracket kernel launches with HIP events on the ctx stream (see kernel_stats): True / 1 = every launch,
        "major" / 2 = only the kernels that pass over the whole cloud, False / 0 = off."""
        level = 2 if enabled in ("major", 2) else (1 if enabled else 0)
        self._check(self.lib.pcv_ctx_set_profiling(self.handle, level))

    def reset_kernel_stats(self):
        self._check(self.lib.pcv_ctx_reset_kernel_stats(self.handle))

    def kernel_stats(self):
        """{kernel name: (launches, total_ms)} accumulated since the last reset."""
        out = {}
        count = self.lib.pcv_ctx_kernel_stats(self.handle, -1, None, None, None)
        for k in range(count):
            name, launches, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self.lib.pcv_ctx_kernel_stats(self.handle, k, C.byref(name), C.byref(launches), C.byref(ms))
            out[name.value.decode()] = (launches.value, ms.value)
        return out

    def synchronize(self):
        """Wait for the work queued on the context's stream."""
        self._check(self.lib.pcv_ctx_synchronize(self.handle))

    def wait_stream(self, stream_handle):
        """Order the context's stream after what is queued on `stream_handle` (0 = the default stream) right now."""
        self._check(self.lib.pcv_ctx_wait_stream(self.handle, C.c_void_p(stream_handle) if stream_handle else None))

    def signal_stream(self, stream_handle):
        """Order work queued on `stream_handle` from now on after the context's work queued so far."""
        self._check(self.lib.pcv_ctx_signal_stream(self.handle, C.c_void_p(stream_handle) if stream_handle else None))

    def wait_torch(self):
        """wait_stream on torch's current stream of this device: call after torch / torch.distributed produced
        tensors the next context call reads."""
        import torch
        self.wait_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def signal_torch(self):
        import torch
        self.signal_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != L.PCV_OK:
            raise L.PcvError(rc, self.lib.pcv_last_error(self.handle).decode())

    def _points(self, x, y, z, color=None, intensity=None):
        bx, by, bz = _Buf(x, np.float64, "x"), _Buf(y, np.float64, "y"), _Buf(z, np.float64, "z")
        bc = _Buf(color, np.uint8, "color")
        bi = _Buf(intensity, np.float32, "intensity")
        n = bx.size
        if by.size != n or bz.size != n:
            raise ValueError("x, y, z must have the same length")
        devs = {b.device for b in (bx, by, bz, bc, bi) if b.ptr is not None}
        if len(devs) > 1:
            raise ValueError("all point arrays must live in the same memory space")
        stride = 3
        if color is not None:
            if bc.size == 4 * n and n > 0:
                stride = 4
            elif bc.size != 3 * n:
                raise ValueError("color must hold 3 or 4 bytes per point")
        if intensity is not None and bi.size != n:
            raise ValueError("intensity must hold one f32 per point")
        p = L.Points()
        p.n = n
        p.x, p.y, p.z = bx.ptr, by.ptr, bz.ptr
        p.color = bc.ptr
        p.color_stride = stride
        p.intensity = bi.ptr
        p.mem = L.MEM_DEVICE if (devs and devs.pop()) else L.MEM_HOST
        return p, (bx, by, bz, bc, bi)

    @staticmethod
    def _params(resolution, bmin, bmax, max_points_per_node=0, flags=0):
        pr = L.BuildParams()
        pr.resolution = float(resolution)
        if bmin is not None:
            for a in range(3):
                pr.bbox_min[a] = float(bmin[a])
                pr.bbox_max[a] = float(bmax[a])
        pr.max_points_per_node = int(max_points_per_node)
        pr.flags = int(flags)
        return pr

    # ---- the build -------------------------------------------------------------------------------
    def build(self, resolution, bounding_box, x, y, z, color, intensity=None, max_points_per_node=0,
              speculate_depth=True, single_chain=None, check_resolve=False, stage_times=None):
        """build_octree up to (not including) the file writes. bounding_box=None computes it on the
        device (== build_octree_from_file's find_bounding_box pass). single_chain: None = the library decides (from
        2^22 points on), True = force the single-chain build, False = exact two-chain pipeline; speculate_depth=False
        additionally computes and sorts full-depth keys. The result is identical in every mode. check_resolve: the
        single-chain build compares the device's rank map with the host's entry by entry (PCV_BUILD_CHECK_RESOLVE).
        stage_times: record the GPU time of every stage for stage_ms() (PCV_BUILD_STAGE_TIMES: ~0.1 ms of stream time per
        build; without it stage_ms() reports the total only); None = the context's `stage_times` attribute."""
        p, keep = self._points(x, y, z, color, intensity)
        flags = 0 if speculate_depth else L.BUILD_NO_SPECULATION
        if check_resolve:
            flags |= L.BUILD_CHECK_RESOLVE
        if self.stage_times if stage_times is None else stage_times:
            flags |= L.BUILD_STAGE_TIMES
        if single_chain is True:
            flags |= L.BUILD_FORCE_SINGLE_CHAIN
        elif single_chain is False:
            flags |= L.BUILD_NO_SINGLE_CHAIN
        if bounding_box is None:
            pr = self._params(resolution, None, None, max_points_per_node, flags | L.BUILD_COMPUTE_BBOX)
        else:
            pr = self._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node, flags)
        h = C.c_void_p()
        self._check(self.lib.pcv_build_octree(self.handle, C.byref(pr), C.byref(p), C.byref(h)))
        del keep
        return OctreeResult(self, h)

    def ingest(self, num_points_hint=0, has_intensity=False):
        """Streaming batch ingest (pcv_ingest_begin): the reference's `impl Iterator<Item = PointsBatch>` input of
        build_octree (generation.rs:289-295), one batch at a time in the reference's own AoS layout."""
        h = C.c_void_p()
        self._check(self.lib.pcv_ingest_begin(self.handle, int(num_points_hint), 1 if has_intensity else 0, C.byref(h)))
        return Ingest(self, h, has_intensity)

    def build_from_ply(self, resolution, filename, with_intensity=False, max_points_per_node=0):
        """build_octree_from_file (generation.rs:272-287) with the decode on the device: the file's vertex records go up as
        they are, a HIP kernel casts x / y / z to f64 and adds the header offset (ply.rs:488-493), the bounding box is
        computed on the device."""
        pr = self._params(resolution, None, None, max_points_per_node, L.BUILD_COMPUTE_BBOX)
        h = C.c_void_p()
        self._check(self.lib.pcv_build_octree_from_ply(self.handle, C.byref(pr), str(filename).encode(), 1 if with_intensity else 0,
                                                       C.byref(h)))
        return OctreeResult(self, h)

    # ---- loading + queries -----------------------------------------------------------------------
    def open_dir(self, directory):
        """Octree::from_data_provider over a directory (octree/mod.rs:156-215)."""
        h = C.c_void_p()
        self._check(self.lib.pcv_octree_open_dir(self.handle, str(directory).encode(), C.byref(h)))
        return OctreeResult(self, h)

    def shapes(self, shapes):
        """Prepare query shapes on the device. Each entry: ("all",), ("aabb", min3, max3), ("frustum", clip_from_query16),
        ("frustum2", clip_from_query16, query_from_clip16), ("obb", translation3, quat_ijkw4, half_extent3)."""
        arr = (L.Shape * max(1, len(shapes)))()
        for i, sh in enumerate(shapes):
            kind = {"all": L.SHAPE_ALL, "aabb": L.SHAPE_AABB, "frustum": L.SHAPE_FRUSTUM, "obb": L.SHAPE_OBB,
                    "frustum2": L.SHAPE_FRUSTUM_WITH_INVERSE}[sh[0]]
            arr[i].kind = kind
            flat = [float(v) for part in sh[1:] for v in np.asarray(part, dtype=np.float64).ravel()]
            for j, v in enumerate(flat):
                arr[i].params[j] = v
        h = C.c_void_p()
        self._check(self.lib.pcv_shapes_create(self.handle, arr, len(shapes), C.byref(h)))
        return Shapes(self, h, len(shapes))

    def cull_points(self, shapes, shape_index, x, y, z, intensity=None, interval=None):
        """FilteredIterator keep mask for raw positions. Returns (keep uint8 array/tensor, kept count)."""
        p, keep_alive = self._points(x, y, z, None, intensity)
        iv = (C.c_double * 2)(*[float(v) for v in interval]) if interval is not None else None
        kept = C.c_uint64()
        if p.mem == L.MEM_DEVICE:
            import torch
            keep = torch.empty(p.n, dtype=torch.uint8, device=x.device)
            ptr = keep.data_ptr()
        else:
            keep = np.zeros(p.n, dtype=np.uint8)
            ptr = keep.ctypes.data
        self._check(self.lib.pcv_cull_points(self.handle, shapes.handle, shape_index, C.byref(p), iv, ptr, C.byref(kept)))
        return keep, kept.value

    def transform_points(self, iso7, x, y, z):
        """Isometry3 * Point3 for a batch; iso7 = translation xyz + unit quaternion ijkw."""
        p, keep_alive = self._points(x, y, z)
        iso = (C.c_double * 7)(*[float(v) for v in iso7])
        if p.mem == L.MEM_DEVICE:
            import torch
            out = [torch.empty(p.n, dtype=torch.float64, device=x.device) for _ in range(3)]
            ptrs = [o.data_ptr() for o in out]
        else:
            out = [np.zeros(p.n) for _ in range(3)]
            ptrs = [o.ctypes.data for o in out]
        self._check(self.lib.pcv_transform_points(self.handle, iso, C.byref(p), *ptrs))
        return out

    # ---- stage-level entry points ----------------------------------------------------------------
    def aabb_reduce(self, x, y, z):
        p, keep = self._points(x, y, z)
        bmin, bmax = (C.c_double * 3)(), (C.c_double * 3)()
        self._check(self.lib.pcv_aabb_reduce(self.handle, C.byref(p), bmin, bmax))
        return np.array(bmin[:]), np.array(bmax[:])

    def chain_keys(self, resolution, bounding_box, x, y, z, nlevels=0):
        p, keep = self._points(x, y, z)
        pr = self._params(resolution, bounding_box.min, bounding_box.max)
        if p.mem == L.MEM_DEVICE:
            import torch
            keys = torch.empty(p.n, dtype=torch.int64, device=x.device)
            self._check(self.lib.pcv_chain_keys(self.handle, C.byref(pr), C.byref(p), nlevels, keys.data_ptr()))
            return keys
        keys = np.zeros(p.n, dtype=np.uint64)
        self._check(self.lib.pcv_chain_keys(self.handle, C.byref(pr), C.byref(p), nlevels, keys.ctypes.data))
        return keys

    def node_split(self, resolution, bounding_box, sorted_keys, max_points_per_node=0, capacity=1 << 16):
        """K4 on sorted full-depth path keys: ctypes array of SplitNode (breadth first, children consecutive)."""
        b = _Buf(sorted_keys, np.uint64, "sorted_keys")
        pr = self._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node)
        nodes = (L.SplitNode * capacity)()
        num = C.c_uint64()
        self._check(self.lib.pcv_node_split(self.handle, C.byref(pr), b.ptr, b.size, L.MEM_DEVICE if b.device else L.MEM_HOST,
                                            nodes, capacity, C.byref(num)))
        if num.value > capacity:
            return self.node_split(resolution, bounding_box, sorted_keys, max_points_per_node, int(num.value))
        return nodes, int(num.value)

    def gather_encode(self, resolution, bounding_box, x, y, z, color, nodes, num_nodes, intensity=None, max_points_per_node=0):
        """K5 + K6 for a given topology (node_split's table): the finished octree."""
        p, keep = self._points(x, y, z, color, intensity)
        pr = self._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node)
        h = C.c_void_p()
        self._check(self.lib.pcv_gather_encode(self.handle, C.byref(pr), C.byref(p), nodes, num_nodes, C.byref(h)))
        del keep
        return OctreeResult(self, h)

    def selftest_division(self, divisors, samples_per_divisor=1 << 22):
        """Number of inputs for which the exact constant-divisor division differs from IEEE division (must be 0)."""
        d = np.ascontiguousarray(divisors, dtype=np.float64)
        bad = C.c_uint64()
        self._check(self.lib.pcv_selftest_division(self.handle, d.ctypes.data_as(C.POINTER(C.c_double)), d.size,
                                                   int(samples_per_divisor), C.byref(bad)))
        return bad.value

    def route_buckets(self, resolution, bounding_box, x, y, z, color=None, with_state=False):
        """(bucket int32 tensor, 64 counts[, state]) for device-resident points: bucket = 8 * level-1 digit + level-2
        digit. with_state (needs color) also returns the level-1 chain state dict(cx, cy, cz = Float32 bits of the
        level-1 codes, oct_rgb = digit | r << 8 | g << 16 | b << 24), all int32 tensors."""
        import torch
        p, keep = self._points(x, y, z, color if with_state else None)
        if p.mem != L.MEM_DEVICE:
            raise ValueError("route_buckets needs device tensors")
        pr = self._params(resolution, bounding_box.min, bounding_box.max)
        bucket = torch.empty(p.n, dtype=torch.int32, device=x.device)
        counts = (C.c_uint64 * 64)()
        st, state = None, None
        if with_state:
            state = {k: torch.empty(p.n, dtype=torch.int32, device=x.device) for k in ("cx", "cy", "cz", "oct_rgb")}
            st = L.RouteState(state["cx"].data_ptr(), state["cy"].data_ptr(), state["cz"].data_ptr(), state["oct_rgb"].data_ptr())
        self._check(self.lib.pcv_route_buckets(self.handle, C.byref(pr), C.byref(p), bucket.data_ptr(), counts,
                                               C.byref(st) if st is not None else None))
        counts = np.array(counts[:], dtype=np.int64)
        return (bucket, counts, state) if with_state else (bucket, counts)

    def route_plan(self, resolution, bounding_box, x, y, z, octants_only=False, out=None):
        """First pass of the two-pass routing (pcv_route_plan): (bucket uint8 tensor, per-tile bucket histograms, 64 counts)
        for device-resident points; the histograms are what pcv_route_scatter derives every owner's row offsets from.
        octants_only: ownership by root octant — the bucket is the level-1 digit alone (PCV_ROUTE_OCTANTS_ONLY). out: (bucket,
        tile_hist) tensors of an earlier call with the same number of points, reused."""
        import torch
        p, keep = self._points(x, y, z, None)
        if p.mem != L.MEM_DEVICE:
            raise ValueError("route_plan needs device tensors")
        pr = self._params(resolution, bounding_box.min, bounding_box.max, 0, L.ROUTE_OCTANTS_ONLY if octants_only else 0)
        tiles = int(self.lib.pcv_route_tiles(p.n))
        if out is not None and int(out[0].numel()) == p.n and int(out[1].shape[0]) == max(tiles, 1):
            bucket, tile_hist = out
        else:
            bucket = torch.empty(p.n, dtype=torch.uint8, device=x.device)
            tile_hist = torch.empty((max(tiles, 1), 64), dtype=torch.int16, device=x.device)
        counts = (C.c_uint64 * 64)()
        self._check(self.lib.pcv_route_plan(self.handle, C.byref(pr), C.byref(p), bucket.data_ptr(), tile_hist.data_ptr(), counts))
        return bucket, tile_hist, np.array(counts[:], dtype=np.int64)

    def route_scatter(self, resolution, bounding_box, x, y, z, color, bucket, tile_hist, rank_of_bucket, dsts, intensity=None):
        """Second pass (pcv_route_scatter): the level-1 state of every point written straight to its owner's planes.
        dsts[r] = dict(oct_rgb, cx, cy, cz[, intensity]) of int32 (float32) device tensors (views into send / receive buffers)."""
        p, keep = self._points(x, y, z, color, intensity)
        pr = self._params(resolution, bounding_box.min, bounding_box.max)
        world = len(dsts)
        arr = (L.RouteDst * world)()
        for r, d in enumerate(dsts):
            for k in ("oct_rgb", "cx", "cy", "cz"):
                setattr(arr[r], k, d[k].data_ptr() if d[k].numel() else None)
            arr[r].intensity = d["intensity"].data_ptr() if intensity is not None and d["intensity"].numel() else None
        table = (C.c_uint8 * 64)(*[int(v) for v in rank_of_bucket])
        self._check(self.lib.pcv_route_scatter(self.handle, C.byref(pr), C.byref(p), bucket.data_ptr(), tile_hist.data_ptr(), world,
                                               table, arr))
        del keep

    def partition_by_owner(self, owner, planes, dsts, rank_of_bucket=None):
        """Stable partition of row-aligned device planes by owner. planes: list of tensors with the same number of rows;
        dsts[r][p]: tensor (view into a send / receive buffer) that receives rank r's rows of plane p in input order.
        With rank_of_bucket (64 entries) `owner` holds buckets and the table maps them to ranks."""
        n = int(planes[0].shape[0])
        world, npl = len(dsts), len(planes)
        arr = (L.Plane * npl)()
        for k, t in enumerate(planes):
            if not t.is_contiguous() or int(t.shape[0]) != n:
                raise ValueError("planes must be contiguous and row-aligned")
            arr[k].src = t.data_ptr()
            arr[k].elem_bytes = t.element_size() * (int(t.numel()) // n if n else 1)
        dst = (C.c_void_p * (world * npl))()
        for r, row in enumerate(dsts):
            for k, t in enumerate(row):
                dst[r * npl + k] = t.data_ptr()
        table = (C.c_uint8 * 64)(*[int(v) for v in rank_of_bucket]) if rank_of_bucket is not None else None
        self._check(self.lib.pcv_partition_by_owner(self.handle, n, owner.data_ptr(), world, table, npl, arr, dst))

    def build_begin_routed(self, resolution, bounding_box, state, intensity=None, max_points_per_node=0,
                           force_split_level1=0):
        """build_begin for points that arrive as their level-1 chain state (route_buckets(with_state=True))."""
        rp = L.RoutedPoints()
        rp.n = int(state["oct_rgb"].shape[0])
        rp.cx, rp.cy, rp.cz, rp.oct_rgb = (state[k].data_ptr() for k in ("cx", "cy", "cz", "oct_rgb"))
        rp.intensity = intensity.data_ptr() if intensity is not None else None
        flags = ((int(force_split_level1) & 0xFF) << 8) | (L.BUILD_STAGE_TIMES if self.stage_times else 0)
        pr = self._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node, flags)
        h = C.c_void_p()
        self._check(self.lib.pcv_build_begin_routed(self.handle, C.byref(pr), C.byref(rp), C.byref(h)))
        return PendingBuild(self, h, (state, intensity))

    def build_begin(self, resolution, bounding_box, x, y, z, color, intensity=None, max_points_per_node=0,
                    force_split_level1=0):
        """First half of the two-step build (multi-GPU path): topology + stream lengths. Returns a PendingBuild; the
        input tensors must stay alive until finish()."""
        p, keep = self._points(x, y, z, color, intensity)
        flags = ((int(force_split_level1) & 0xFF) << 8) | (L.BUILD_STAGE_TIMES if self.stage_times else 0)
        pr = self._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node, flags)
        h = C.c_void_p()
        self._check(self.lib.pcv_build_begin(self.handle, C.byref(pr), C.byref(p), C.byref(h)))
        return PendingBuild(self, h, keep)

    def sort_keys64(self, keys, begin_bit=0, end_bit=64):
        b = _Buf(keys, np.uint64, "keys")
        self._check(self.lib.pcv_sort_keys64(self.handle, b.ptr, b.size, begin_bit, end_bit,
                                             L.MEM_DEVICE if b.device else L.MEM_HOST))
        return b.keep

    def sort_keys32(self, keys, begin_bit=0, end_bit=32):
        b = _Buf(keys, np.uint32, "keys")
        self._check(self.lib.pcv_sort_keys32(self.handle, b.ptr, b.size, begin_bit, end_bit,
                                             L.MEM_DEVICE if b.device else L.MEM_HOST))
        return b.keep

    def sort_pairs32(self, keys, values, begin_bit=0, end_bit=32):
        k, v = _Buf(keys, np.uint32, "keys"), _Buf(values, np.uint32, "values")
        if k.size != v.size:
            raise ValueError("keys and values must have the same length")
        self._check(self.lib.pcv_sort_pairs32(self.handle, k.ptr, v.ptr, k.size, begin_bit, end_bit,
                                              L.MEM_DEVICE if k.device else L.MEM_HOST))
        return k.keep, v.keep


def promote_assign(nodes, num_nodes, n=0, with_slots=False):
    """Closed form of the every-8th promotion on a node table: (stream_len, num_points, child_offset) arrays and, with
    with_slots, (node_of_slot, slot_in_node) for every position of the leaf-sorted order."""
    lib = L.load_library()
    per = (L.PromoteNode * max(1, num_nodes))()
    node_of = np.zeros(n if with_slots else 0, dtype=np.uint32)
    slot_in = np.zeros(n if with_slots else 0, dtype=np.uint32)
    rc = lib.pcv_promote_assign(nodes, num_nodes, per, n, node_of.ctypes.data if with_slots else None,
                                slot_in.ctypes.data if with_slots else None)
    if rc != L.PCV_OK:
        raise L.PcvError(rc, "pcv_promote_assign: inconsistent node table")
    stream = np.array([per[i].stream_len for i in range(num_nodes)], dtype=np.int64)
    kept = np.array([per[i].num_points for i in range(num_nodes)], dtype=np.int64)
    off = np.array([per[i].child_offset for i in range(num_nodes)], dtype=np.int64)
    return (stream, kept, off, node_of, slot_in) if with_slots else (stream, kept, off)


def level_table(bbox_min, bbox_max, resolution, cap=64):
    lib = L.load_library()
    bmin = (C.c_double * 3)(*[float(v) for v in bbox_min])
    bmax = (C.c_double * 3)(*[float(v) for v in bbox_max])
    edge = (C.c_double * (cap + 2))()
    enc = (C.c_int32 * (cap + 2))()
    ml = lib.pcv_level_table(bmin, bmax, float(resolution), cap, edge, enc)
    return ml, np.array(edge[:ml + 1]), np.array(enc[:ml + 1], dtype=np.int32)


def level_shortcuts(bbox_min, bbox_max, resolution):
    """(max_level, digit_mode[k], code_threshold[k]): the per-level shortcuts of the single chain pass for this cube
    (pcv_level_shortcuts; host tables, held against exact arithmetic by tests/test_oracle_kats.py)."""
    lib = L.load_library()
    bmin = (C.c_double * 3)(*[float(v) for v in bbox_min])
    bmax = (C.c_double * 3)(*[float(v) for v in bbox_max])
    mode = (C.c_uint32 * (L.MAX_KEY_LEVELS + 2))()
    thr = (C.c_double * (L.MAX_KEY_LEVELS + 2))()
    ml = lib.pcv_level_shortcuts(bmin, bmax, float(resolution), mode, thr)
    return ml, np.array(mode[:], dtype=np.uint32), np.array(thr[:])


class Ingest:
    """One pcv_ingest: batches of `PointsBatch` shape (src/lib.rs:102-107) go to the device as they are — positions (n, 3)
    f64 AoS like Vec<Point3<f64>>, colour (n, 3) u8, intensity (n,) f32 — and `finish` builds the octree."""

    def __init__(self, ctx, handle, has_intensity):
        self.ctx, self.lib, self.handle, self.has_intensity = ctx, ctx.lib, handle, bool(has_intensity)

    def append(self, position, color, intensity=None):
        pos = np.ascontiguousarray(position, dtype=np.float64)
        if pos.ndim != 2 or pos.shape[1] != 3:
            raise ValueError("position must be (n, 3): one Point3<f64> per row")
        n = pos.shape[0]
        col = np.ascontiguousarray(color, dtype=np.uint8)
        if col.shape != (n, 3):
            raise ValueError("color must be (n, 3) u8")
        inten = None
        if self.has_intensity:
            if intensity is None:
                raise ValueError("the ingest was begun with intensity: every batch must carry it")
            inten = np.ascontiguousarray(intensity, dtype=np.float32)
            if inten.shape != (n,):
                raise ValueError("intensity must be (n,) f32")
        if self.handle is None:
            raise ValueError("the ingest is finished")
        self.ctx._check(self.lib.pcv_ingest_append(self.handle, pos.ctypes.data, col.ctypes.data,
                                                   None if inten is None else inten.ctypes.data, n))

    @property
    def num_points(self):
        return int(self.lib.pcv_ingest_num_points(self.handle)) if self.handle is not None else 0

    def bbox(self):
        lo, hi = (C.c_double * 3)(), (C.c_double * 3)()
        self.ctx._check(self.lib.pcv_ingest_bbox(self.handle, lo, hi))
        return np.array(lo[:]), np.array(hi[:])

    def finish(self, resolution, bounding_box=None, max_points_per_node=0, single_chain=None, stage_times=None):
        """pcv_ingest_finish: bounding_box None = the box folded during the ingest (find_bounding_box); consumes the ingest."""
        flags = 0
        if self.ctx.stage_times if stage_times is None else stage_times:
            flags |= L.BUILD_STAGE_TIMES
        if single_chain is True:
            flags |= L.BUILD_FORCE_SINGLE_CHAIN
        elif single_chain is False:
            flags |= L.BUILD_NO_SINGLE_CHAIN
        if bounding_box is None:
            pr = self.ctx._params(resolution, None, None, max_points_per_node, flags | L.BUILD_COMPUTE_BBOX)
        else:
            pr = self.ctx._params(resolution, bounding_box.min, bounding_box.max, max_points_per_node, flags)
        h, mine = C.c_void_p(), self.handle
        self.handle = None  # consumed whatever the call returns
        self.ctx._check(self.lib.pcv_ingest_finish(mine, C.byref(pr), C.byref(h)))
        return OctreeResult(self.ctx, h)

    def abort(self):
        if self.handle is not None:
            self.lib.pcv_ingest_abort(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.abort()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


class PendingBuild:
    """A tree between pcv_build_begin and pcv_build_finish."""

    def __init__(self, ctx, handle, keep):
        self.ctx, self.handle, self._keep = ctx, handle, keep
        ctx._children.add(self)

    def top_streams(self):
        """(l1[8], l2[64], l1_split_mask): local stream lengths of the level-1 / level-2 nodes (0 = absent)."""
        ts = L.TopStreams()
        self.ctx._check(self.ctx.lib.pcv_build_top_streams(self.handle, C.byref(ts)))
        return np.array(ts.l1[:], dtype=np.int64), np.array(ts.l2[:], dtype=np.int64), int(ts.l1_split_mask)

    def finish(self, layout=None):
        """Second half: encode, record sort, promotion. layout = dict(root_points, l1_stream[8], l1_offset[8],
        l2_offset[64]) with the GLOBAL top-of-tree streams, or None for a self-contained tree."""
        tl = None
        if layout is not None:
            tl = L.TopLayout()
            tl.root_points = int(layout["root_points"])
            for c in range(8):
                tl.l1_stream[c] = int(layout["l1_stream"][c])
                tl.l1_offset[c] = int(layout["l1_offset"][c])
            for b in range(64):
                tl.l2_offset[b] = int(layout["l2_offset"][b])
        h, self.handle = self.handle, None
        rc = self.ctx.lib.pcv_build_finish(h, C.byref(tl) if tl is not None else None)
        self._keep = None
        if rc != L.PCV_OK:
            msg = self.ctx.lib.pcv_last_error(self.ctx.handle)
            self.ctx.lib.pcv_octree_free(h)
            raise L.PcvError(rc, msg.decode() if msg else "")
        return OctreeResult(self.ctx, h)

    def free(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.pcv_octree_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Shapes:
    """Prepared query shapes (device resident)."""

    def __init__(self, ctx, handle, count):
        self.ctx, self.handle, self.count = ctx, handle, count
        ctx._children.add(self)

    def get(self, i):
        corners, axes = (C.c_double * 24)(), (C.c_double * 78)()
        n, valid = C.c_uint32(), C.c_int()
        self.ctx._check(self.ctx.lib.pcv_shapes_get(self.handle, i, corners, axes, C.byref(n), C.byref(valid)))
        return np.array(corners[:]).reshape(8, 3), np.array(axes[:3 * n.value]).reshape(n.value, 3), bool(valid.value)

    def free(self):
        if self.handle and self.ctx.handle:
            self.ctx.lib.pcv_shapes_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class OctreeResult:
    """A finished octree held by the library: node table + node-contiguous file bytes."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self.lib = ctx.lib
        self.handle = handle
        ctx._children.add(self)

    def free(self):
        if self.handle and self.ctx.handle:
            self.lib.pcv_octree_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    @property
    def num_nodes(self):
        return self.lib.pcv_octree_num_nodes(self.handle)

    @property
    def num_points(self):
        return self.lib.pcv_octree_num_points(self.handle)

    def meta(self):
        res = C.c_double()
        bmin, bmax = (C.c_double * 3)(), (C.c_double * 3)()
        ver = C.c_int()
        self.lib.pcv_octree_meta(self.handle, C.byref(res), bmin, bmax, C.byref(ver))
        return dict(resolution=res.value, bbox_min=np.array(bmin[:]), bbox_max=np.array(bmax[:]), version=ver.value)

    def node(self, i):
        info = L.NodeInfo()
        self.ctx._check(self.lib.pcv_octree_node(self.handle, i, C.byref(info)))
        return info

    def node_data(self, i, which):
        ptr, ln = C.c_void_p(), C.c_uint64()
        self.ctx._check(self.lib.pcv_octree_node_data(self.handle, i, which, C.byref(ptr), C.byref(ln)))
        return C.string_at(ptr, ln.value) if ln.value else b""

    def total_gpu_ms(self):
        """GPU time of the build from its first to its last kernel (PCV_STAGE_TOTAL; measured for every build)."""
        ms = (C.c_float * L.NUM_STAGES)()
        self.lib.pcv_octree_stage_ms(self.handle, ms, L.NUM_STAGES)
        return float(ms[L.NUM_STAGES - 1])

    def stage_ms(self):
        ms = (C.c_float * L.NUM_STAGES)()
        n = self.lib.pcv_octree_stage_ms(self.handle, ms, L.NUM_STAGES)
        return {L.STAGE_NAMES[i]: ms[i] for i in range(n)}

    def build_info(self):
        """key_levels, attempts (0 = single-chain build, 1 = exact pipeline, >= 2 something was redone) and the
        single-chain statistics (predicted nodes / leaves, points whose leaf is an unsplit candidate, points whose chain
        was continued from a split candidate's codes, points that replayed the chain from their coordinates);
        record_bytes: bytes per record of the record sort (20, or 12 packed); settled_in_sort: points of the leaves the
        record sort's second pass finished itself."""
        lv, at = C.c_int(), C.c_int()
        self.lib.pcv_octree_build_info(self.handle, C.byref(lv), C.byref(at))
        st = (C.c_uint64 * 4)()
        self.lib.pcv_octree_spec_stats(self.handle, st)
        return dict(key_levels=lv.value, attempts=at.value, single_chain=at.value == 0,
                    record_bytes=int(self.lib.pcv_octree_record_bytes(self.handle)), predicted_nodes=st[0], predicted_leaves=st[1], kept_code_points=st[2], replayed_points=st[3],
                    continued_points=int(self.lib.pcv_octree_spec_continued(self.handle)),
                    wide_pool_entries=int(self.lib.pcv_octree_wide_pool_entries(self.handle)),
                    settled_in_sort=int(self.lib.pcv_octree_settled_in_sort(self.handle)))

    def write_dir(self, directory):
        self.ctx._check(self.lib.pcv_octree_write_dir(self.handle, str(directory).encode()))

    # ---- queries ----
    def node_names(self):
        return [node_name(self.node(i).id_high, self.node(i).id_low) for i in range(self.num_nodes)]

    def cull_nodes(self, shapes, with_sizes=False):
        """Relation matrix [shape][node] (0 In, 1 Cross, 2 Out) and optionally relative_size_on_screen."""
        m, f = self.num_nodes, shapes.count
        rel = np.zeros((f, m), dtype=np.uint8)
        sizes = np.zeros((f, m)) if with_sizes else None
        self.ctx._check(self.lib.pcv_cull_nodes(self.ctx.handle, shapes.handle, self.handle, rel.ctypes.data,
                                                sizes.ctypes.data if with_sizes else None))
        return (rel, sizes) if with_sizes else rel

    def cull_nodes_sparse(self, shapes, capacity, with_sizes=True):
        """Per shape the nodes that are not Out, in node order: (counts[f], node_indices[f][capacity], relation, sizes)."""
        f = shapes.count
        counts = np.zeros(f, dtype=np.uint32)
        idx = np.zeros((f, max(capacity, 1)), dtype=np.uint32)
        rel = np.full((f, max(capacity, 1)), 2, dtype=np.uint8)
        sizes = np.zeros((f, max(capacity, 1))) if with_sizes else None
        self.ctx._check(self.lib.pcv_cull_nodes_sparse(self.ctx.handle, shapes.handle, self.handle, capacity, counts.ctypes.data,
                                                       idx.ctypes.data, rel.ctypes.data, sizes.ctypes.data if with_sizes else None))
        return counts, idx, rel, sizes

    def _traverse(self, fn, shapes, with_status):
        m, f = max(1, self.num_nodes), shapes.count
        counts = np.zeros(f, dtype=np.uint32)
        idx = np.zeros((f, m), dtype=np.uint32)
        status = np.zeros(f, dtype=np.int32)
        if with_status:
            self.ctx._check(fn(self.ctx.handle, shapes.handle, self.handle, m, counts.ctypes.data, idx.ctypes.data,
                               status.ctypes.data))
        else:
            self.ctx._check(fn(self.ctx.handle, shapes.handle, self.handle, m, counts.ctypes.data, idx.ctypes.data))
        return [idx[i, :counts[i]].copy() for i in range(f)], status

    def visible_nodes(self, frusta):
        """Octree::get_visible_nodes per frustum: (list of node-index arrays in heap pop order, status array)."""
        return self._traverse(self.lib.pcv_visible_nodes, frusta, True)

    def nodes_in_location(self, shapes):
        return self._traverse(self.lib.pcv_nodes_in_location, shapes, False)[0]

    def cull_node_points(self, shapes, shape_index, node, interval=None):
        n = self.node(node).num_points
        keep = np.zeros(n, dtype=np.uint8)
        kept = C.c_uint64()
        iv = (C.c_double * 2)(*[float(v) for v in interval]) if interval is not None else None
        self.ctx._check(self.lib.pcv_cull_node_points(self.ctx.handle, shapes.handle, shape_index, self.handle, node, iv,
                                                      keep.ctypes.data, C.byref(kept)))
        return keep, kept.value

    def query_points(self, shapes, shape_index, interval=None, capacity=None, node=None):
        """All points of the octree inside shape `shape_index` (and the intensity interval): dict of numpy arrays
        x, y, z (decoded f64), rgb (n x 3), intensity (or None), in (node traversal, point) order. With `node` only
        that node's points (stream_points_for_query_in_node)."""
        cap = (self.num_points if node is None else self.node(node).num_points) if capacity is None else int(capacity)
        x, y, z = np.zeros(cap), np.zeros(cap), np.zeros(cap)
        rgb = np.zeros((cap, 3), dtype=np.uint8)
        has_int = bool(self.lib.pcv_octree_has_intensity(self.handle))
        inten = np.zeros(cap, dtype=np.float32) if has_int else None
        iv = (C.c_double * 2)(*[float(v) for v in interval]) if interval is not None else None
        count = C.c_uint64()
        outs = (cap, L.MEM_HOST, x.ctypes.data, y.ctypes.data, z.ctypes.data, rgb.ctypes.data,
                inten.ctypes.data if has_int else None, C.byref(count))
        if node is None:
            self.ctx._check(self.lib.pcv_query_points(self.ctx.handle, shapes.handle, shape_index, self.handle, iv, *outs))
        else:
            self.ctx._check(self.lib.pcv_query_node_points(self.ctx.handle, shapes.handle, shape_index, self.handle, int(node),
                                                           iv, *outs))
        n = min(count.value, cap)
        return dict(count=count.value, x=x[:n], y=y[:n], z=z[:n], rgb=rgb[:n], intensity=inten[:n] if has_int else None)

    def nodes_blob(self, node_indices):
        """octree_web_viewer's /nodes_data reply body for the given nodes."""
        idx = np.ascontiguousarray(node_indices, dtype=np.uint64)
        need = C.c_uint64()
        ip = idx.ctypes.data_as(C.POINTER(C.c_uint64))
        self.ctx._check(self.lib.pcv_octree_nodes_blob(self.handle, ip, idx.size, None, 0, C.byref(need)))
        buf = np.zeros(need.value, dtype=np.uint8)
        self.ctx._check(self.lib.pcv_octree_nodes_blob(self.handle, ip, idx.size, buf.ctypes.data, buf.size, C.byref(need)))
        return buf.tobytes()

    def write_nodes(self, directory, min_level=0):
        """Node files of the nodes at level >= min_level, without meta.pb (multi-GPU output)."""
        self.ctx._check(self.lib.pcv_octree_write_nodes(self.handle, str(directory).encode(), int(min_level)))

    def synchronize(self):
        self.ctx.synchronize()

    def copy_node_into(self, i, which, dst):
        """Copy node i's bytes (0 xyz, 1 rgb, 2 intensity) from the device blob into a uint8 tensor/array view."""
        if hasattr(dst, "data_ptr"):
            ptr, cap, mem = dst.data_ptr(), dst.numel() * dst.element_size(), (L.MEM_DEVICE if dst.is_cuda else L.MEM_HOST)
        else:
            ptr, cap, mem = dst.ctypes.data, dst.nbytes, L.MEM_HOST
        self.ctx._check(self.lib.pcv_octree_copy_node(self.handle, i, which, ptr, cap, mem))

    def copy_nodes_into(self, copies, dst):
        """Batch form of copy_node_into: copies = [(node index, (xyz offset, rgb offset, intensity offset))] with None for
        a file kind to skip; everything lands in the one uint8 tensor / array `dst` (one ABI call)."""
        arr = (L.NodeCopy * max(1, len(copies)))()
        for k, (node, offs) in enumerate(copies):
            arr[k].node = int(node)
            for w in range(3):
                arr[k].dst_offset[w] = 0xFFFFFFFFFFFFFFFF if offs[w] is None else int(offs[w])
        if hasattr(dst, "data_ptr"):
            ptr, cap, mem = dst.data_ptr(), dst.numel() * dst.element_size(), (L.MEM_DEVICE if dst.is_cuda else L.MEM_HOST)
        else:
            ptr, cap, mem = dst.ctypes.data, dst.nbytes, L.MEM_HOST
        self.ctx._check(self.lib.pcv_octree_copy_nodes(self.handle, arr, len(copies), ptr, cap, mem))

    def to_dict(self):
        """{node name: dict(id, num_points, encoding, level, xyz, rgb, intensity)} — same shape the test-side
        oracle wrapper uses, so parity tests are plain dict comparisons."""
        has_int = bool(self.lib.pcv_octree_has_intensity(self.handle))
        out = {}
        for i in range(self.num_nodes):
            nd = self.node(i)
            name = node_name(nd.id_high, nd.id_low)
            out[name] = dict(id=(nd.id_high, nd.id_low), num_points=nd.num_points, encoding=nd.encoding,
                             level=nd.level, xyz=self.node_data(i, 0), rgb=self.node_data(i, 1),
                             intensity=self.node_data(i, 2) if has_int else b"",
                             cube_min=tuple(nd.cube_min), cube_edge=nd.cube_edge)
        return out


def write_meta(directory, resolution, bbox_min, bbox_max, nodes):
    """meta.pb (version 13) for a list of (id_high, id_low, num_points, encoding) tuples."""
    lib = L.load_library()
    arr = (L.NodeInfo * max(1, len(nodes)))()
    for k, (hi, lo, npts, enc) in enumerate(nodes):
        arr[k].id_high, arr[k].id_low, arr[k].num_points, arr[k].encoding = int(hi), int(lo), int(npts), int(enc)
    bmin = (C.c_double * 3)(*[float(v) for v in bbox_min])
    bmax = (C.c_double * 3)(*[float(v) for v in bbox_max])
    rc = lib.pcv_write_meta(str(directory).encode(), float(resolution), bmin, bmax, arr, len(nodes))
    if rc != L.PCV_OK:
        raise L.PcvError(rc, f"cannot write meta.pb in {directory}")


def read_ply(path):
    """PlyIterator in one pass (src/read_write/ply.rs): dict(x, y, z float64; color (n,3) uint8 or None; intensity or None)."""
    lib = L.load_library()
    h = C.c_void_p()
    err = C.create_string_buffer(512)
    rc = lib.pcv_ply_read(str(path).encode(), C.byref(h), err, 512)
    if rc != L.PCV_OK:
        raise L.PcvError(rc, err.value.decode())
    try:
        p = L.Points()
        lib.pcv_ply_points(h, C.byref(p))
        n = p.n

        def arr(ptr, ctype, count, dtype):
            if not ptr or count == 0:
                return None if not ptr else np.zeros(0, dtype=dtype)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).astype(dtype, copy=True)

        out = dict(x=arr(p.x, C.c_double, n, np.float64), y=arr(p.y, C.c_double, n, np.float64),
                   z=arr(p.z, C.c_double, n, np.float64), color=arr(p.color, C.c_uint8, 3 * n, np.uint8),
                   intensity=arr(p.intensity, C.c_float, n, np.float32))
        if n == 0:
            out["x"] = out["y"] = out["z"] = np.zeros(0)
        if out["color"] is not None:
            out["color"] = out["color"].reshape(-1, 3)
        return out
    finally:
        lib.pcv_ply_free(h)


def build_octree_from_file(output_directory, resolution, filename, attributes=("color", "intensity"), ctx=None,
                           host_decode=False):
    """Drop-in for reference `build_octree_from_file` (generation.rs:272-287): the vertex records of the file go to the
    device as they are and are decoded there (host_decode=True: the one-pass host parser instead), bounding box on the
    device, build, directory write. Like the reference binary (src/bin/build_octree.rs:47-52) the default attribute
    list asks for intensity; a PLY without it raises (the reference panics)."""
    attributes = tuple(attributes)
    if not host_decode:
        if "color" not in attributes:
            raise ValueError("the octree format requires the 'color' attribute (on_disk.rs:20-22)")
        for a in attributes:
            if a not in ("color", "intensity"):
                raise ValueError(f"unsupported attribute {a!r} (octree/mod.rs:62-74 implies color and intensity)")
        ctx = ctx or default_context()
        try:
            tree = ctx.build_from_ply(resolution, filename, with_intensity="intensity" in attributes)
        except L.PcvError as e:
            if "requested but the PLY has none" in str(e) or "requires colour" in str(e):
                raise ValueError(str(e)) from None
            raise
        tree.write_dir(output_directory)
        return tree
    pts = read_ply(filename)
    if pts["color"] is None:
        raise ValueError("the PLY has no red/green/blue properties; the octree format requires colour")
    cloud = dict(x=pts["x"], y=pts["y"], z=pts["z"], color=pts["color"])
    if "intensity" in attributes:
        if pts["intensity"] is None:
            raise ValueError("attribute 'intensity' requested but the PLY has none")
        cloud["intensity"] = pts["intensity"]
    return build_octree(output_directory, resolution, None, cloud, attributes, ctx)


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def build_octree(output_directory, resolution, bounding_box, points, attributes=("color",), ctx=None,
                 max_points_per_node=0):
    """Drop-in for reference `build_octree` (generation.rs:289-295): builds on the GPU and writes the
    reference's directory layout. `points` = dict(x=, y=, z=, color=, intensity=optional) — or, like the reference, an
    ITERATOR OF BATCHES, each dict(position=(n, 3) f64, color=(n, 3) u8[, intensity=(n,) f32]) (PointsBatch,
    src/lib.rs:102-107): the batches are streamed to the device one at a time (pcv_ingest_*), host memory stays O(batch).
    An iterator with `num_points` (NumberOfPoints, src/lib.rs:56-58) sizes the device arrays up front."""
    attributes = tuple(attributes)
    if "color" not in attributes:
        raise ValueError("the octree format requires the 'color' attribute (on_disk.rs:20-22)")
    for a in attributes:
        if a not in ("color", "intensity"):
            raise ValueError(f"unsupported attribute {a!r} (octree/mod.rs:62-74 implies color and intensity)")
    ctx = ctx or default_context()
    if not isinstance(points, dict):
        want_intensity = "intensity" in attributes
        hint = getattr(points, "num_points", 0)
        ing = ctx.ingest(hint() if callable(hint) else int(hint or 0), want_intensity)
        try:
            for batch in points:
                if want_intensity and batch.get("intensity") is None:
                    raise ValueError("attribute 'intensity' requested but not present in the input")
                ing.append(batch["position"], batch["color"], batch.get("intensity") if want_intensity else None)
        except BaseException:
            ing.abort()
            raise
        tree = ing.finish(resolution, bounding_box, max_points_per_node)
        tree.write_dir(output_directory)
        return tree
    inten = points.get("intensity") if "intensity" in attributes else None
    if "intensity" in attributes and inten is None:
        raise ValueError("attribute 'intensity' requested but not present in the input")
    tree = ctx.build(resolution, bounding_box, points["x"], points["y"], points["z"], points["color"], inten,
                     max_points_per_node)
    tree.write_dir(output_directory)
    return tree
