"""Dev timing (GPU box): pieces of pooled_moments at 64 real parameters."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
from metropolisengine_amd.distributed import moments_size
n = 1 << 19
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=1, cov_mode="fixed", dtype=(sys.argv[1] if len(sys.argv) > 1 else "f32"))
e.step_all(10); e.sync()
size = moments_size(64, 0)
out = np.empty(size)
ptr = out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
lib, h = e._lib, e._handle
def t(f, reps=50):
    f(); e.sync(); t0 = time.perf_counter()
    for _ in range(reps): f()
    e.sync(); return (time.perf_counter() - t0) / reps * 1e6
print("me_pooled_moments (host):   %.1f us" % t(lambda: lib.me_pooled_moments(h, ptr, size)))
import torch
buf = torch.empty(size, dtype=torch.float64, device="cuda")
print("me_pooled_moments_device:   %.1f us" % t(lambda: lib.me_pooled_moments_device(h, ctypes.c_void_p(buf.data_ptr()), size)))
print("engine.pooled_moments():    %.1f us" % t(e.pooled_moments))
print("me_sync alone:              %.1f us" % t(lambda: lib.me_sync(h)))
