"""Dev (GPU box): soak of k_step_dense64_f64 (LDS-DMA prefetch, register tiles): tens of thousands of launches at 2^19 chains, then a
sub-range of the big engine against a small engine that owns the same chain ids -- bit for bit -- in both proposal shapes.
    python tools/dev/soak_dense64_f64.py [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
factor = np.linalg.cholesky(0.5 * np.linalg.inv(amat))[np.tril_indices(64)]
n, lo, cnt = 1 << 19, 64 * 5003 + 11, 333
for mode in ("fixed", "pooled"):
    kw = dict(temp=1.0, seed=99, sampling_width=0.2, cov_mode=mode, dtype="f64")
    big = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, n_chains=n, **kw)
    small = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, n_chains=cnt, chain_offset=lo, **kw)
    if mode == "pooled":
        big.set_shared_factor(factor); small.set_shared_factor(factor)
    t0 = time.time()
    for k in range(launches):
        s = 3 if k % 7 == 0 else 1
        big.step_all(s); small.step_all(s)
    big.sync(); small.sync()
    same = np.array_equal(big._get(0)[lo:lo + cnt], small._get(0)) and np.array_equal(big.energy_total[lo:lo + cnt], small.energy_total)
    x = big._get(0)
    print("%s: %d launches in %.1f s, shard bitwise equal: %s, all finite: %s, acceptance %.3f, variance of x_0 %.4f (T/2 A^-1_00 = %.4f)"
          % (mode, launches, time.time() - t0, same, bool(np.all(np.isfinite(x))), big.acceptance_rate(), x[:, 0].var(), 0.5 * np.linalg.inv(amat)[0, 0]), flush=True)
