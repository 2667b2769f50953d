"""Dev timing (GPU box): k_step with per-chain factors and k_measure at 16 real parameters, 2^20 chains.
    python tools/dev/time_perchain16.py [f32|f64]      (METROPOLIS_HIP_LIB selects a variant build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
es = 4 if dtype == "f32" else 8
n = 1 << 20
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3, dtype=dtype)
for _ in range(55):
    e.step_all(2); e.measure()
e.sync()
ms = min(e.time_steps(100, 1) for _ in range(3)) / 100
bs = es * (2 * 16 + 4 + 136)
print("%s %s k_step per-chain factors: %.1f us -> %.0f GB/s of %d B/chain (%.2f)" % (os.environ.get("METROPOLIS_HIP_LIB", "default")[-14:], dtype, ms * 1e3, bs * n / ms / 1e6, bs, bs * n / ms / 8e9))
best = 1e9
for _ in range(3):
    t0 = time.perf_counter()
    for _ in range(30): e.measure()
    e.sync()
    best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
b = es * (16 + 1 + 2 * 16 + 2 * 136 + 2 * 32 + 136)
print("%s %s k_measure: %.1f us -> %.0f GB/s of %d B/chain (%.2f)" % (os.environ.get("METROPOLIS_HIP_LIB", "default")[-14:], dtype, best * 1e3, b * n / best / 1e6, b, b * n / best / 8e9), flush=True)
