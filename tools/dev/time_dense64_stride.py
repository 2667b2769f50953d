"""Dev timing (GPU box): does the power-of-two row stride of 2^19 chains (4 MiB rows in float64) cost bandwidth?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
for dtype in ("f64", "f32"):
    for n in (1 << 19, (1 << 19) + 64 * 17, (1 << 19) + (1 << 14), (1 << 19) - 64 * 33):
        e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                                 n_chains=n, seed=2026, cov_mode="fixed", dtype=dtype, sampling_width=0.2)
        e4.time_steps(20, 1)
        ms = min(e4.time_steps(50, 1) for _ in range(3)) / 50
        msf = min(e4.time_steps(5, 10) for _ in range(2)) / 50
        print("%s n=%d: %.1f us per one-sweep launch = %.3f ns per chain; fused %.1f us per sweep" % (dtype, n, ms * 1e3, ms * 1e6 / n, msf * 1e3), flush=True)
        del e4
