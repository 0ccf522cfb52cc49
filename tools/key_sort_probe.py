"""Times the sample's key sort alone (libpcv_hip_exp.so: pcv_exp_time_key_sort): onesweep against the three-kernel passes,
and the timing-only variants of the onesweep kernel (diag bits: 1 no look-back, 2 no ranking, 4 no scatter, 8 no key load)."""
import ctypes, json, os, sys
os.environ["PCV_HIP_LIBRARY"] = "exp"
import point_cloud_viewer_amd as pcv
from point_cloud_viewer_amd import _lib
ctx = pcv.Context(0)
lib = ctx.lib
f = lib.pcv_exp_time_key_sort
f.restype = ctypes.c_int
f.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
iters = 12
out = (ctypes.c_float * iters)()
for n in (1_562_500, 15_625_000):
    for bits in (36, 39):
        for one, diag in ((0, 0), (1, 0), (1, 1), (1, 2), (1, 3), (1, 4), (1, 8), (1, 15)):
            rc = f(ctx.handle, n, bits, one, diag, iters, out)
            v = sorted(out[2:])
            print(json.dumps(dict(n=n, bits=bits, onesweep=one, diag=diag, rc=rc, min_us=round(v[0] * 1e3, 1), med_us=round(v[len(v) // 2] * 1e3, 1))), flush=True)
