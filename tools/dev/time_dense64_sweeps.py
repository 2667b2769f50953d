"""Dev timing (GPU box): config 4 float64, launch time against the number of fused sweeps (is the memory phase hidden?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
for n_log2 in (19, 17, 15):
    e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                             n_chains=1 << n_log2, seed=2026, cov_mode="fixed", dtype="f64", sampling_width=0.2)
    e4.time_steps(20, 1)
    for k in (1, 2, 3, 4, 8):
        ms = min(e4.time_steps(40, k) for _ in range(3)) / 40
        print("2^%d chains, %d sweeps per launch: %.1f us per launch" % (n_log2, k, ms * 1e3), flush=True)
