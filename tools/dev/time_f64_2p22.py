"""Dev timing (GPU box): the float64 headline kernel at 2^20 and 2^22 chains (identity shape, no per-chain fields)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
for lg in (20, 22):
    n = 1 << lg
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026, dtype="f64", cov_mode="fixed")
    e.time_steps(50, 1)
    k1 = min(e.time_steps(200, 1) for _ in range(3)) / 200
    print("f64 2^%d: %.1f us per launch (%.0f GB/s, %.3f of 8 TB/s)" % (lg, k1 * 1e3, 288 * n / k1 / 1e6, 288 * n / k1 / 1e6 / 8000), flush=True)
    del e
