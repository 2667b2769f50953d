"""Dev timing (GPU box): BASELINE config 4's parameter space in cov_mode="reference" (per-chain 64 x 64 shapes, streamed)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
for dtype, n_log2 in (("f32", 19), ("f64", 19), ("f32", 16)):
    n = 1 << n_log2
    e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype, sampling_width=0.2)
    for _ in range(52):
        e.step_all(2); e.measure()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(5): e.measure()
    e.sync(); meas = (time.perf_counter() - t0) / 5 * 1e3
    step = min(e.time_steps(10, 1) for _ in range(2)) / 10
    fused = min(e.time_steps(3, 5) for _ in range(2)) / 15
    print("%s 2^%d chains, per-chain 64x64 shapes: measure %.2f ms, step %.3f ms (%.2e chain-steps/s), fused %.3f ms per sweep, acceptance %.3f" %
          (dtype, n_log2, meas, step, n / step * 1e3, fused, e.acceptance_rate()), flush=True)
    del e
