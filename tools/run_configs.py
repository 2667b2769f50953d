"""Dev/bench tool (GPU box): run the BASELINE.json configurations 2-5 through their protocols and
print chain-steps/s per configuration (wall clock incl. measure launches).  Use under rocprofv3 --kernel-trace
--stats for per-kernel times.

    python tools/run_configs.py [--configs 2,3,4,5] [--scale-log2 0]   (scale < 0 shrinks the chain counts)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402
from metropolisengine_amd.distributed import (moments_to_statistics, pooled_statistics,  # noqa: E402
                                              pooled_statistics_begin, pooled_statistics_end)

ap = argparse.ArgumentParser()
ap.add_argument("--configs", default="2,3,4,5")
ap.add_argument("--scale-log2", type=int, default=0)
args = ap.parse_args()
out = {}


def run(name, eng, n, steps_per_measure, n_measures, warm_measures=0, sweeps_fused=False, pooled=False, overlap=False):
    for _ in range(warm_measures):
        eng.step_all(steps_per_measure)
        eng.measure()
    if pooled:
        pooled_statistics(eng)       # first call imports torch.distributed: keep that out of the timed region
    eng.sync()
    pending = False
    t0 = time.perf_counter()
    for _ in range(n_measures):
        if sweeps_fused:
            eng.step_all(steps_per_measure)
        else:
            for _ in range(steps_per_measure):
                eng.step_all()
        if steps_per_measure and name != "cfg2":
            eng.measure()
        if pooled and not overlap:   # config 5's protocol: the pooled moments every cycle (an RCCL all-reduce when ranks > 1)
            pooled_statistics(eng)
        elif pooled:                 # same, but collected one cycle later so that the next cycle's steps hide it
            if pending:
                pooled_statistics_end(eng)
            pooled_statistics_begin(eng)
            pending = True
    if pooled and overlap and pending:
        pooled_statistics_end(eng)
    eng.sync()
    dt = time.perf_counter() - t0
    rate = n * steps_per_measure * n_measures / dt
    name = name + ("_pooled" if pooled else "") + ("_overlap" if overlap else "")
    out[name + ("_fused" if sweeps_fused else "")] = {"chain_steps_per_s": rate, "seconds": dt,
                                                        "acceptance": eng.acceptance_rate()}
    print("%-12s %.3e chain-steps/s  (%.3f s, acceptance %.3f)" % (name + ("_fused" if sweeps_fused else ""), rate, dt,
                                                                   eng.acceptance_rate()), flush=True)


for cfg in [int(c) for c in args.configs.split(",")]:
    if cfg == 2:
        n = 1 << (20 + args.scale_log2)
        eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026)
        eng.step_all(1000)
        run("cfg2", eng, n, 1000, 1)
        run("cfg2", eng, n, 1024, 1, sweeps_fused=True)
    elif cfg == 3:
        n = 1 << (20 + args.scale_log2)
        a = b = (1.0, 2.0, 4.0, 8.0)
        for fused in (False, True):
            eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026)
            run("cfg3", eng, n, 10, 200, warm_measures=60, sweeps_fused=fused)
        st = moments_to_statistics(eng.pooled_moments(), 4, 4)
        print("   cfg3 pooled var*2a:", np.round(np.diag(st["covariance"])[:4] * 2 * np.array(a), 4))
    elif cfg == 4:
        n = 1 << (19 + args.scale_log2)
        m = np.random.default_rng(5).standard_normal((64, 64))
        amat = m @ m.T / 64 + np.identity(64)
        eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026,
                                  cov_mode="fixed")
        eng.step_all(200)
        run("cfg4", eng, n, 200, 1)
        run("cfg4", eng, n, 192, 1, sweeps_fused=True)
    elif cfg == 5:
        n = 1 << (18 + args.scale_log2)
        src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "user_energy_cylinder.h")
        for fused in (False, True):
            eng = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0),
                                      [0.1, 0.0], [0.05] * 7, temp=0.1, n_chains=n, seed=2026)
            run("cfg5", eng, n, 10, 200, warm_measures=60, sweeps_fused=fused)
            run("cfg5", eng, n, 10, 200, sweeps_fused=fused, pooled=True)
            run("cfg5", eng, n, 10, 200, sweeps_fused=fused, pooled=True, overlap=True)
print(json.dumps(out))
