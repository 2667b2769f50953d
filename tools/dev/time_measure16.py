"""Dev timing (GPU box): k_measure variants at 16 real parameters, 2^20 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
n = 1 << 20
def timeit(e, reps=30):
    e.sync(); t0 = time.perf_counter()
    for _ in range(reps): e.measure()
    e.sync(); return (time.perf_counter() - t0) / reps * 1e3
for mode in ("reference", "fixed"):
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3, cov_mode=mode)
    e.step_all(5)
    print(mode, "early measure (mean/obs only): %.4f ms" % timeit(e, 20))
    for _ in range(40): e.measure()
    print(mode, "late measure (cov%s): %.4f ms" % (" + Cholesky + factor store" if mode == "reference" else " only", timeit(e, 30)))
