"""Dev timing (GPU box): the float64 build of the headline kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
n = 1 << 20
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026, dtype="f64")
e.time_steps(50, 1)
k1 = min(e.time_steps(200, 1) for _ in range(3)) / 200
k32 = min(e.time_steps(10, 32) for _ in range(3)) / 10 / 32
print("f64 config 2: %.1f us per launch (%.2e chain-steps/s, %.0f GB/s), fused %.1f us per sweep" %
      (k1 * 1e3, n / (k1 * 1e-3), 288 * n / k1 / 1e6, k32 * 1e3))
