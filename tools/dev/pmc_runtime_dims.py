"""Dev workload (GPU box, under rocprofv3 --pmc): the LDS-resident runtime-dimension kernel, dense energy, 128 parameters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
nr, n = 128, 1 << 17
m = np.random.default_rng(0).standard_normal((nr, nr))
amat = m @ m.T / nr + np.identity(nr)
for dtype in ("f32", "f64"):
    e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=1, cov_mode="fixed", dtype=dtype, sampling_width=0.02)
    for _ in range(10):
        e.step_all(1)
    e.sync()
print("done")
