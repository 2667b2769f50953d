"""Dev timing (GPU box): k_step with per-chain factors and k_measure at 16 real parameters, 2^20 chains."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
n = 1 << 20
e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3)
for _ in range(55):
    e.step_all(2); e.measure()
e.sync()
ms = e.time_steps(100, 1) / 100
print("k_step per-chain factors: %.4f ms  -> %.0f GB/s of %d B/chain" % (ms, 688 * n / ms / 1e6, 688))
t0 = time.perf_counter()
for _ in range(50): e.measure()
e.sync()
ms = (time.perf_counter() - t0) / 50 * 1e3
b = 4*16 + 4 + 8*16 + 8*136 + 8*32 + 4*136
print("k_measure: %.4f ms -> %.0f GB/s of %d B/chain" % (ms, b * n / ms / 1e6, b))
