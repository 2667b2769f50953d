"""Workloads for the PMC passes of the covariance-carrying kernels (GPU box, under rocprofv3 --pmc ...):
    python3 tools/pmc_workloads.py            # (16,0) and (4,4) engines, per-chain factors active, a few launches each
Kernel names to summarise afterwards with tools/pmc_summary.py:
    k_step<float, 16, 0, ... 3, false, 0>   per-chain factor, non-temporal variant      688 B per chain-step
    k_measure<float, 16, 0, true, true, true, false>                                   2084 B per chain-measure
    k_step<float, 4, 4, ... 2, false, 0>    per-chain factor                             216 B
    k_measure<float, 4, 4, true, true, true, false>                                     556 B
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402

n = 1 << 20
for dtype in ("f32", "f64"):
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3, dtype=dtype)
    for _ in range(52):
        e.step_all(2)
        e.measure()
    for _ in range(12):
        e.step_all()
        e.step_all()
        e.measure()
    e.sync()
    del e
    a = b = (1.0, 2.0, 4.0, 8.0)
    e = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026, dtype=dtype)
    for _ in range(52):
        e.step_all(10)
        e.measure()
    for _ in range(12):
        e.step_all()
        e.step_all()
        e.measure()
    e.sync()
    del e
print("done")
