"""Dev timing (GPU box): BASELINE config 4 (64 real, dense quadratic form, 2^19 chains), float64 and float32, identity
shape and the pooled shared factor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
from metropolisengine_amd.distributed import adapt_pooled_shape
m = np.random.default_rng(5).standard_normal((64, 64))
for dtype, per in (("f64", 1056), ("f32", 528)):
    for mode in ("fixed", "pooled"):
        e4 = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0,
                                 n_chains=1 << 19, seed=2026, cov_mode=mode, dtype=dtype, sampling_width=0.2)
        e4.time_steps(20, 10)
        if mode == "pooled":
            adapt_pooled_shape(e4)
        e4.time_steps(20, 1)
        ms = min(e4.time_steps(50, 1) for _ in range(3)) / 50
        msf = min(e4.time_steps(5, 10) for _ in range(2)) / 50
        print("%s %s: %.1f us per one-sweep launch (%.2e chain-steps/s, %.0f GB/s), fused %.1f us per sweep, acceptance %.3f" %
              (dtype, mode, ms * 1e3, (1 << 19) / ms * 1e3, per * (1 << 19) / ms / 1e6, msf * 1e3, e4.acceptance_rate()), flush=True)
        del e4
