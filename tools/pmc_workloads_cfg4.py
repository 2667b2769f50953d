"""Workload for the rocprofv3 passes of BASELINE config 4's step kernels (GPU box):
    python3 tools/pmc_workloads_cfg4.py [launches]
64 real parameters, dense SPD quadratic form, 2^19 chains, identity shape, one sweep per launch, float32
(k_step_dense64_bf16x3, 528 B per chain-step) then float64 (k_step_dense64_f64, 1 056 B per chain-step)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 1 << 19
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
for dtype in ("f32", "f64"):
    e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026,
                            cov_mode="fixed", dtype=dtype)
    e.time_steps(30, 1)
    ms = e.time_steps(launches, 1) / launches
    print("%s: %.2f us per launch (HIP events), acceptance %.3f" % (dtype, ms * 1e3, e.acceptance_rate()), flush=True)
    e.close()
