"""Dev timing (GPU box): measure() at (16,0) per-chain and (64,0) means-only; METROPOLIS_HIP_LIB selects a variant."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
def timeit(e, calls=30):
    best = 1e9
    for _ in range(3):
        e.sync(); t0 = time.perf_counter()
        for _ in range(calls): e.measure()
        e.sync(); best = min(best, (time.perf_counter() - t0) / calls * 1e6)
    return best
tag = os.environ.get("METROPOLIS_HIP_LIB", "default")[-16:]
if "64" not in sys.argv[1:]:
    n = 1 << 20
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3)
    for _ in range(55):
        e.step_all(2); e.measure()
    us = timeit(e); print("%s (16,0) k_measure %.1f us %.2f of 8 TB/s; then k_step %.1f us" % (tag, us, 2084 * n / us / 8e6, e.time_steps(100, 1) * 10), flush=True)
    del e
if "16" not in sys.argv[1:]:
    n = 1 << 19
    m = np.random.default_rng(5).standard_normal((64, 64))
    e = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, cov_mode="fixed")
    e.step_all(20)
    us = timeit(e); print("%s (64,0) k_measure means/obs %.1f us %.2f of 8 TB/s; then k_step %.1f us" % (tag, us, 1796 * n / us / 8e6, e.time_steps(100, 1) * 10), flush=True)
