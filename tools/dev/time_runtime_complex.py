"""Dev (GPU box): per-chain shapes of spaces with complex parameters -- the Hermitian block's factor refresh (k_factor_mixed in the
compiled sets, k_factor_runtime_complex at runtime dimensions) once the adaptive shapes are live.
    python tools/dev/time_runtime_complex.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
for dtype, nr, nc, lg in (("f64", 0, 13, 16), ("f64", 1, 13, 16), ("f64", 100, 20, 12), ("f64", 0, 70, 12), ("f32", 0, 70, 12)):
    n = 1 << lg
    e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr)), tuple(np.linspace(0.7, 1.6, nc))), None,
                            [0.0] * nr if nr else None, [0j] * nc, temp=1.0, n_chains=n, seed=3, dtype=dtype, sampling_width=0.05)
    for k in range(51):
        e.step_all(1); e.measure()
    e.sync()
    ms_step = min(e.time_steps(5, 1) for _ in range(2)) / 5
    t0 = time.perf_counter()
    for k in range(3):
        e.measure()
    e.sync()
    print("%s (%d real, %d complex) x 2^%d chains: step_all %.3f ms, measure + factor refresh %.2f ms" %
          (dtype, nr, nc, lg, ms_step, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
