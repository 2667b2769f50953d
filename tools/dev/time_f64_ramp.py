"""Dev timing (GPU box): the float64 headline kernel against the chain count -- time = fixed cost (launch ramp + tail: the
first round of wavefronts loads, then computes, while memory idles) + slope x chains.  Beside tools/variants/rows_probe_f64
(the memory-only kernel of the same access pattern at the same sizes) this separates what the 6-7 us between k_step and
its memory-only floor at 2^20 chains is made of."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
for dtype in ("f64", "f32"):
    for lg in (17, 18, 19, 20, 21, 22):
        n = 1 << lg
        e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype, cov_mode="fixed")
        e.time_steps(200, 1)
        us = min(e.time_steps(300, 1) for _ in range(3)) / 300 * 1e3
        per = (288 if dtype == "f64" else 144) * n
        print("%s 2^%d chains: %.2f us per one-sweep launch, %.0f GB/s" % (dtype, lg, us, per / us / 1e3), flush=True)
        e.close()
