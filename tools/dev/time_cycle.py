"""Dev timing (GPU box): k_cycle (10 sweeps + measure per launch) with every chain's own factor live.
    python tools/dev/time_cycle.py 4 4 | 2 7        (METROPOLIS_HIP_LIB selects a variant build)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
nr, nc = int(sys.argv[1]), int(sys.argv[2])
for dtype in ("f32", "f64"):
    if (nr, nc) == (4, 4):
        a = b = (1.0, 2.0, 4.0, 8.0)
        n = 1 << 20
        e = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026, dtype=dtype)
    else:
        n = 1 << 18
        e = me.MetropolisEngine(me.CylinderSurrogate(1.0, 0.5, 1.0), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7, temp=0.1,
                                n_chains=n, seed=2026, dtype=dtype)
    for _ in range(60):
        e.cycle(10)
    e.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            e.cycle(10)
        e.sync()
        best = min(best, (time.perf_counter() - t0) / 50)
    assert e.fused_cycles() == 210
    print("%s (%d,%d) %s: %.1f us per cycle of 10 sweeps + measure -> %.2e chain-steps/s" %
          (os.environ.get("METROPOLIS_HIP_LIB", "default")[-12:], nr, nc, dtype, best * 1e6, n * 10 / best), flush=True)
    e.close()
