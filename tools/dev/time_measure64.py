"""Dev timing (GPU box): measure() with the streamed per-chain covariance at 64 real parameters (track_covariance)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
for log2n in (17, 19):
    n = 1 << log2n
    for track in (False, True):
        e = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=1,
                                cov_mode="fixed", track_covariance=track)
        for _ in range(52):
            e.measure()
        e.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            e.measure()
        e.sync()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        b = (4 * 64 + 4 + 8 * 64 + 8 * 128) + (8 * 2080 if track else 0)
        print("2^%d chains, track_covariance=%s: measure %.3f ms -> %.0f GB/s of %d B/chain" %
              (log2n, track, ms, b * n / ms / 1e6, b), flush=True)
        del e
