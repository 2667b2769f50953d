"""Dev A/B (GPU box), ONE process, interleaved rounds: k_step (per-chain factors) and k_measure with the non-temporal
policy forced on (budget 0), off (budget huge) and at two thresholds, for the BASELINE shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me

def measure_us(e, calls=20):
    e.measure(); e.sync(); t0 = time.perf_counter()
    for _ in range(calls): e.measure()
    e.sync(); return (time.perf_counter() - t0) / calls * 1e6

def make(kind, dtype):
    if kind == "16,0":
        e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=1 << 20, seed=1, sampling_width=0.3, dtype=dtype)
        w = 2
    elif kind == "4,4":
        a = (1.0, 2.0, 4.0, 8.0)
        e = me.MetropolisEngine(me.DiagQuadratic(a, a), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=1 << 20, seed=2026, dtype=dtype)
        w = 10
    elif kind == "2,7":
        src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
        e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7, temp=0.1, n_chains=1 << 18, seed=2026, dtype=dtype)
        w = 10
    else:
        m = np.random.default_rng(5).standard_normal((64, 64))
        e = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0, n_chains=1 << 19, seed=2026, cov_mode="fixed", dtype=dtype)
        w = 2
    for _ in range(52):
        e.step_all(w); e.measure()
    return e

budgets = {"never": 1 << 60, "always": 0, "224MiB": 224 << 20, "320MiB": 320 << 20}
for kind in sys.argv[1:] or ["16,0", "4,4", "2,7", "64,0"]:
    for dtype in ("f32", "f64"):
        e = make(kind, dtype)
        res = {k: [] for k in budgets}
        for rnd in range(3):
            for name, b in budgets.items():
                me.set_cache_budget(b)
                e.time_steps(10, 1); e.measure(); e.sync()
                s = e.time_steps(40, 1) / 40 * 1e3
                m = measure_us(e)
                # the pair as a protocol cycle: 10 steps + 1 measure
                e.sync(); t0 = time.perf_counter()
                for _ in range(5):
                    for _ in range(10): e.step_all()
                    e.measure()
                e.sync(); cyc = (time.perf_counter() - t0) / 5 * 1e6
                res[name].append((s, m, cyc))
        print("(%s) %s  [k_step us | k_measure us | 10 steps + measure us], best of 3:" % (kind, dtype))
        for name in budgets:
            r = res[name]
            print("   %-8s %7.1f | %7.1f | %8.1f" % (name, min(x[0] for x in r), min(x[1] for x in r), min(x[2] for x in r)), flush=True)
        del e
me.set_cache_budget(224 << 20)
