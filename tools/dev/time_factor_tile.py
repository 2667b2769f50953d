"""Dev timing (GPU box): measure() with streamed per-chain shapes at 2^19 (or argv[1]) chains x 64 parameters, both precisions."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
n_log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 19
for dtype in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("f32", "f64")):
    n = 1 << n_log2
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, dtype=dtype, sampling_width=0.2, cov_mode="reference")
    for _ in range(52):
        e.step_all(2); e.measure()
    e.sync()
    t0 = time.perf_counter()
    for _ in range(5): e.measure()
    e.sync(); meas = (time.perf_counter() - t0) / 5 * 1e3
    print("%s 2^%d chains: measure %.2f ms" % (dtype, n_log2, meas), flush=True)
    del e
