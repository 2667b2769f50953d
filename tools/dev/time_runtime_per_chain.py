"""Dev (GPU box): per-chain proposal shapes at runtime dimensions (cov_mode="reference" beyond build.MAX_COMPILED_DOF):
time per step_all and per measure() (covariance recursion + per-chain Cholesky) once the adaptive shapes are live.
    python tools/dev/time_runtime_per_chain.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
for dtype, nr, lg in (("f64", 140, 14), ("f32", 140, 14), ("f64", 200, 12), ("f32", 200, 12), ("f64", 290, 10), ("f32", 500, 10)):
    n = 1 << lg
    e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr))), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=3,
                            dtype=dtype, sampling_width=0.05)
    for k in range(51):
        e.step_all(1); e.measure()
    e.sync()
    t_identity = None
    ms_step = min(e.time_steps(5, 1) for _ in range(2)) / 5
    t0 = time.perf_counter()
    for k in range(3):
        e.measure()
    e.sync()
    ms_measure = (time.perf_counter() - t0) / 3 * 1e3
    p = nr * (nr + 1) // 2
    es = 8 if dtype == "f64" else 4
    print("%s %3d real x 2^%d chains: step_all %.3f ms (%.0f GB/s of factor reads), measure + factor refresh %.2f ms, acceptance %.2f"
          % (dtype, nr, lg, ms_step, p * es * n / ms_step / 1e6, ms_measure, e.acceptance_rate()), flush=True)
