"""Workload for the kernel-trace / PMC passes of the streamed-shape kernels (GPU box, under rocprofv3): BASELINE config 4's
parameter space (64 real, dense quadratic form) in cov_mode="reference" at 2^17 chains, both precisions; the per-chain
shapes are in use after the 50th measure, so summarise the LAST dispatches only.

Algorithmic bytes per chain (P = 2 080 packed entries, es = 4 / 8):
    k_factor_tile   read C, write L                                  2 P es                 = 16 640 / 33 280
    k_step (CK 3)   read L, state r/w, energy + width r/w            P es + 2 D es + 4 es   =  8 848 / 17 696
    k_measure       C r/w, mean r/w, observables r/w, x, width       2 P es + (5 D + 1) es  = 17 924 / 35 848
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import metropolisengine_amd as me  # noqa: E402

m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
n = 1 << 17
for dtype in ("f32", "f64"):
    e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype,
                            sampling_width=0.2)
    for _ in range(52):
        e.step_all(2)
        e.measure()
    for _ in range(12):
        e.step_all()
        e.step_all()
        e.measure()
    e.sync()
    del e
print("done")
