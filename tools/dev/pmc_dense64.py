"""Dev workload (GPU box, under rocprofv3 --pmc ...): k_step_dense64_f64 at 2^19 chains -- 30 one-sweep launches, then 6 launches
of 10 fused sweeps (told apart by their order in the dispatch list)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
e = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0, n_chains=1 << 19,
                        seed=2026, cov_mode="fixed", dtype="f64")
for _ in range(30):
    e.step_all(1)
e.sync()
for _ in range(6):
    e.step_all(10)
e.sync()
print("done")
