"""Workload for the rocprofv3 passes of k_cycle (GPU box):  python3 tools/pmc_workloads_cycle.py [cycles after the threshold]
BASELINE config 3 (4 real + 4 complex, 2^20 chains) and config 5 (2 real + 7 complex user plugin + wall, 2^18 chains), both
dtypes: 52 cycles of 10 sweeps to pass the 50-measure threshold (identity shape, CK = 0), then `cycles` cycles with every
chain's own factor (CK = 2) -- summarise the LAST `cycles` k_cycle dispatches per engine.
Algorithmic bytes per chain and cycle (per-chain phase), es = 4 / 8:  es x (4 D + 4 + 2 nobs + 4 P)
  config 3: D = 12, nobs = 12, P = 26 -> 720 / 1 440 B      config 5: D = 16, nobs = 11, P = 52 -> 1 192 / 2 384 B
(x, energy, width read and written once; mean, observables, covariance read and written; factor read by the sweeps and
written by the refresh) -- against 10 x B_step(per-chain) + B_measure = 2 716 B for config 3 in float32 as separate launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402

cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 12
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "user_energy_cylinder.h")
a = b = (1.0, 2.0, 4.0, 8.0)
for dtype in ("f32", "f64"):
    for name, make in (("config3", lambda: me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0,
                                                                n_chains=1 << 20, seed=2026, dtype=dtype)),
                       ("config5", lambda: me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)),
                                                                me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7, temp=0.1,
                                                                n_chains=1 << 18, seed=2026, dtype=dtype))):
        e = make()
        for _ in range(52 + cycles):
            e.cycle(10)
        e.sync()
        assert e.fused_cycles() == 52 + cycles
        print(name, dtype, "acceptance %.3f" % e.acceptance_rate(), flush=True)
        e.close()
