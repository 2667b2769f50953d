"""Dev timing (GPU box): measure() at 16 real parameters with per-chain covariance (k_measure + k_factor, or the fused
form when the library is built with -DME_MEASURE_FUSED_MAX_P=136; METROPOLIS_HIP_LIB selects the build)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
n = 1 << 20
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3)
for _ in range(55):
    e.step_all(2); e.measure()
e.sync()
t0 = time.perf_counter()
for _ in range(40): e.measure()
e.sync()
print("%s: measure %.1f us" % (os.environ.get("METROPOLIS_HIP_LIB", "default"), (time.perf_counter() - t0) / 40 * 1e6))
