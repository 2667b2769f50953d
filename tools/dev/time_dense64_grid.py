"""Dev probe (GPU box): config-4 sweep time vs grid size (ME_GRID_BLOCKS is read at engine creation)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me  # noqa: E402

n = 1 << 19
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
for grid in [int(g) for g in sys.argv[1:]] or [0]:
    os.environ["ME_GRID_BLOCKS"] = str(grid)
    eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026,
                              cov_mode="fixed")
    eng.step_all(50)
    k1 = min(eng.time_steps(200, 1) for _ in range(3)) / 200
    k8 = min(eng.time_steps(50, 8) for _ in range(3)) / 50 / 8
    k64 = min(eng.time_steps(8, 64) for _ in range(3)) / 8 / 64
    print("grid %4d: %.1f us/sweep at K=1 (%.2e chain-steps/s), %.1f at K=8, %.1f at K=64" %
          (grid, k1 * 1e3, n / (k1 * 1e-3), k8 * 1e3, k64 * 1e3), flush=True)
    del eng
