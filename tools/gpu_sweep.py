"""Dev tool (GPU box): time k_step for BASELINE config 2 over grid sizes and fused-sweep counts.

    python tools/gpu_sweep.py [--chains-log2 20]
Each engine is created under a different ME_GRID_BLOCKS (read at me_create).  Interleaved rounds, median reported.
"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--chains-log2", type=int, default=20)
ap.add_argument("--grids", default="0,1024,2048,4096,8192")
ap.add_argument("--sweeps", default="1,2,4,8,32")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--launches", type=int, default=200)
args = ap.parse_args()
n = 1 << args.chains_log2
engines = {}
for g in [int(v) for v in args.grids.split(",")]:
    os.environ["ME_GRID_BLOCKS"] = str(g)
    engines[g] = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026)
    engines[g].time_steps(200, 1)
print("chains=2^%d  bytes/launch=%.1f MB" % (args.chains_log2, 144 * n / 1e6))
for k in [int(v) for v in args.sweeps.split(",")]:
    res = {g: [] for g in engines}
    for _ in range(args.rounds):
        for g, e in engines.items():
            launches = max(10, args.launches // k)
            res[g].append(e.time_steps(launches, k) / launches)
    for g in engines:
        ms = statistics.median(res[g])
        print("sweeps=%3d grid=%5d  %.4f ms/launch  %.3e chain-steps/s  state %.0f GB/s (min %.4f ms)"
              % (k, g, ms, n * k / (ms * 1e-3), 144 * n / (ms * 1e-3) / 1e9, min(res[g])), flush=True)
