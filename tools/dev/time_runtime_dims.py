"""Dev timing (GPU box): the runtime-dimension kernels (D > 96) against their traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
for dtype, es in (("f32", 4), ("f64", 8)):
    for nr, n in ((128, 1 << 18), (256, 1 << 17), (1024, 1 << 15)):
        e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=1, cov_mode="fixed", dtype=dtype, sampling_width=0.02)
        e.time_steps(30, 1)
        ms = min(e.time_steps(50, 1) for _ in range(3)) / 50
        rate = e.acceptance_rate()
        # bytes: read x once, accepted chains read and write it again; energy and width r/w
        b = es * (nr * (1 + 2 * rate) + 4)
        print("%s %d real x 2^%d chains: %.1f us per sweep, %.2e chain-steps/s, %.2e parameter-updates/s, ~%.0f GB/s (acceptance %.2f)" %
              (dtype, nr, n.bit_length() - 1, ms * 1e3, n / ms * 1e3, n * nr / ms * 1e3, b * n / ms / 1e6, rate), flush=True)

# LDS-resident forms: dense quadratic energy (identity shape) and one shared factor on a separable energy
import numpy as np
for dtype in ("f32", "f64"):
    for nr, n in ((128, 1 << 17),):
        rng = np.random.default_rng(0)
        m = rng.standard_normal((nr, nr))
        amat = m @ m.T / nr + np.identity(nr)
        for label, energy, mode in (("dense energy, identity shape", me.DenseQuadratic(amat), "fixed"),
                                    ("iso energy, shared factor", me.IsoQuadratic(1.0), "pooled"),
                                    ("dense energy, shared factor", me.DenseQuadratic(amat), "pooled")):
            e = me.MetropolisEngine(energy, None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=1, cov_mode=mode, dtype=dtype, sampling_width=0.02)
            if mode == "pooled":
                e.set_shared_factor(np.linalg.cholesky(np.linalg.inv(amat))[np.tril_indices(nr)])
            e.time_steps(10, 1)
            ms = min(e.time_steps(20, 1) for _ in range(3)) / 20
            flops = 2.0 * nr * (nr + 1) / 2 * (("dense" in label.split(",")[0]) + (mode == "pooled"))
            print("%s %d real x 2^%d chains, %s: %.1f us per sweep, %.2e chain-steps/s, %.2f TFLOP/s of triangle products (acceptance %.2f)" %
                  (dtype, nr, n.bit_length() - 1, label, ms * 1e3, n / ms * 1e3, flops * n / ms / 1e9, e.acceptance_rate()), flush=True)
