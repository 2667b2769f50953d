"""Per-kernel roofline table (GPU box): every hot kernel of BASELINE configs 2-5 timed in isolation at full size, in the
state the protocol puts it in, against its algorithmic bytes.

    python tools/kernel_roofline.py > profiles/rNN_kernel_roofline.md

k_step: me_time_steps (HIP events around back-to-back one-sweep launches).  k_measure / pooled moments: wall time of
back-to-back calls between two stream syncs (the launches queue up; the host is not the limit at these durations).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import metropolisengine_amd as me  # noqa: E402

PEAK = 8000.0   # GB/s


def packed(nr, nc):
    return nr * (nr + 1) // 2 + nc * nc


def step_us(eng, launches=300):
    eng.time_steps(50, 1)
    return min(eng.time_steps(launches, 1) for _ in range(3)) / launches * 1e3


def measure_us(eng, calls=40):
    eng.measure(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.measure()
    eng.sync()
    return (time.perf_counter() - t0) / calls * 1e6


def pooled_us(eng, calls=40):
    eng.pooled_moments(); eng.sync()
    t0 = time.perf_counter()
    for _ in range(calls):
        eng.pooled_moments()
    return (time.perf_counter() - t0) / calls * 1e6


rows = []
ES = 4            # bytes per value of the dtype being measured


def add(label, values_per_chain, n, us, bound="hbm"):
    bytes_per_chain = values_per_chain * ES
    gbps = bytes_per_chain * n / us / 1e3
    rows.append((label, bytes_per_chain, n, bytes_per_chain * n / 1e6, us, gbps, gbps / PEAK, bound))
    print("  " + label + ": %.1f us" % us, file=sys.stderr, flush=True)


def table(dtype):
    """Every hot kernel of configs 2-5 for one device dtype; `values` are counted in units of one stored value."""
    global ES
    ES = 4 if dtype == "f32" else 8
    del rows[:]
    kw = dict(dtype=dtype)
    # config 2
    n = 1 << 20
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=2026, **kw)
    eng.step_all(500)
    add("k_step (16,0) identity  [config 2, headline]", 36, n, step_us(eng, 1000))
    add("pooled_moments (16,0): stage 1 + k_pool_finish + copy", 16, n, pooled_us(eng), "latency")
    del eng
    # config 2 shape with per-chain covariance (the reference's default mode at 16 parameters)
    eng = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1, sampling_width=0.3, **kw)
    for _ in range(55):
        eng.step_all(2); eng.measure()
    p = packed(16, 0)
    add("k_step (16,0) per-chain factor", 36 + p, n, step_us(eng, 100))
    add("k_measure (16,0), fused Cholesky", 16 + 1 + 2 * 16 + 2 * 32 + 3 * p, n, measure_us(eng))
    del eng
    # config 3
    a = b = (1.0, 2.0, 4.0, 8.0)
    eng = me.MetropolisEngine(me.DiagQuadratic(a, b), None, [0.0] * 4, [0j] * 4, temp=1.0, n_chains=n, seed=2026, **kw)
    for _ in range(55):
        eng.step_all(10); eng.measure()
    p = packed(4, 4)
    add("k_step (4,4) per-chain factor  [config 3]", 2 * 12 + 4 + p, n, step_us(eng))
    add("k_measure (4,4), fused Cholesky  [config 3]", 12 + 1 + 2 * 12 + 2 * 12 + 3 * p, n, measure_us(eng))
    del eng
    # config 4
    n4 = 1 << 19
    m = np.random.default_rng(5).standard_normal((64, 64))
    amat = m @ m.T / 64 + np.identity(64)
    eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, [0.0] * 64, None, temp=1.0, n_chains=n4, seed=2026, cov_mode="fixed", **kw)
    eng.step_all(100)
    kernel = "k_step_dense64_bf16x3" if dtype == "f32" else "k_step_dense64_f64 (v_mfma_f64_16x16x4_f64)"
    add(kernel + " (64,0)  [config 4]", 2 * 64 + 4, n4, step_us(eng, 200), "valu+mfma")
    add("k_measure (64,0): means, observables", 64 + 1 + 2 * 64 + 2 * 128, n4, measure_us(eng))
    add("pooled_moments (64,0): stage 1 + k_pool_finish + copy", 64, n4, pooled_us(eng), "mfma")
    del eng
    # config 5
    n5 = 1 << 18
    src = os.path.join(ROOT, "examples", "user_energy_cylinder.h")
    eng = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7,
                              temp=0.1, n_chains=n5, seed=2026, **kw)
    for _ in range(55):
        eng.step_all(10); eng.measure()
    p = packed(2, 7)
    add("k_step (2,7) user energy, per-chain factor  [config 5]", 36 + p, n5, step_us(eng))
    add("k_measure (2,7), fused Cholesky  [config 5]", 16 + 1 + 2 * 16 + 2 * 11 + 3 * p, n5, measure_us(eng))
    add("pooled_moments (2,7): stage 1 + k_pool_finish + copy", 16, n5, pooled_us(eng), "latency")

    print("### %s\n" % ("float32 (production dtype)" if dtype == "f32" else "float64 (the reference's dtype)"))
    print("| kernel / call | algorithmic B per chain | chains | MB per launch | µs | GB/s | of 8 TB/s | bound |")
    print("|---|---|---|---|---|---|---|---|")
    for label, bpc, chains, mb, us, gbps, frac, bound in rows:
        print("| %s | %d | 2^%d | %.0f | %.1f | %.0f | %.2f | %s |" % (label, bpc, chains.bit_length() - 1, mb, us, gbps, frac, bound))
    print()


for dt in (sys.argv[1:] or ["f64", "f32"]):
    table(dt)
print("One MI355X; every kernel in the state the protocol leaves it in (per-chain factors active after 50 measures).  Working "
      "sets up to ~225 MB sit in the 256 MiB Infinity Cache between launches (the identity-shape kernel of config 2, config 5); "
      "larger ones stream from HBM, where plain float4 kernels reach 6.4 TB/s reading, 4.7-5.5 TB/s writing and 4.7-5.8 TB/s "
      "copying on this part (tools/dev/rows_probe2.hip -> profiles/r02_memory_probes.txt): 0.6-0.8 of the 8 TB/s peak is the practical ceiling "
      "of a read-modify-write stream.  Fields that a launch touches once are accessed non-temporally when the working set "
      "exceeds the cache (DESIGN.md section 3).")
