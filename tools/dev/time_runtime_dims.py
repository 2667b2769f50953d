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
