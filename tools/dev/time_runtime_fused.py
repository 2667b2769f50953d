"""Dev timing (GPU box): the LDS-resident runtime-dimension kernel, one sweep per launch against fused sweeps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
nr = 128
m = np.random.default_rng(0).standard_normal((nr, nr))
amat = m @ m.T / nr + np.identity(nr)
for dtype in ("f32", "f64"):
    for lg in (15, 17):
        n = 1 << lg
        e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=1, cov_mode="fixed", dtype=dtype, sampling_width=0.02)
        e.time_steps(10, 1)
        k1 = min(e.time_steps(20, 1) for _ in range(3)) / 20
        k10 = min(e.time_steps(4, 10) for _ in range(3)) / 40
        print("%s 2^%d chains: %.1f us per one-sweep launch, %.1f us per fused sweep" % (dtype, lg, k1 * 1e3, k10 * 1e3), flush=True)
