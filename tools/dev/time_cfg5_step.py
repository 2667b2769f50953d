"""Dev timing (GPU box): config 5's k_step (2 real + 7 complex, user energy, wall, live per-chain factors), one sweep per
launch against fused sweeps, both precisions, 2^18 and 2^20 chains."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
for dtype, es in (("f32", 4), ("f64", 8)):
    for lg in (18, 20):
        n = 1 << lg
        e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7,
                                temp=0.1, n_chains=n, seed=2026, dtype=dtype)
        for _ in range(52):
            e.step_all(10); e.measure()
        e.time_steps(50, 1)
        k1 = min(e.time_steps(200, 1) for _ in range(3)) / 200
        k32 = min(e.time_steps(10, 32) for _ in range(3)) / 10 / 32
        b = es * (2 * 16 + 4 + 52)
        print("%s 2^%d: %.1f us per launch (%.0f GB/s, %.2f of 8 TB/s), fused %.1f us per sweep" % (dtype, lg, k1 * 1e3, b * n / k1 / 1e6, b * n / k1 / 1e6 / 8000, k32 * 1e3), flush=True)
        del e
