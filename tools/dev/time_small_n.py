"""Dev timing (GPU box): the launch-bound regime (few chains): per-launch cost of one-sweep launches enqueued from C
(me_time_steps) and from Python, against fused sweeps -- the data behind "no HIP graphs" in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
for log2n in (10, 14, 17, 20):
    n = 1 << log2n
    e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * 16, None, temp=1.0, n_chains=n, seed=1)
    e.step_all(100); e.sync()
    c_loop = min(e.time_steps(2000, 1) for _ in range(3)) / 2000 * 1e3
    e.sync(); t0 = time.perf_counter()
    for _ in range(2000): e.step_all()
    e.sync(); py_loop = (time.perf_counter() - t0) / 2000 * 1e6
    fused = min(e.time_steps(50, 64) for _ in range(3)) / 50 / 64 * 1e3
    print("2^%-2d chains: %.2f us per one-sweep launch (C loop, device time), %.2f us per step_all() from Python (wall), "
          "%.2f us per sweep fused x64" % (log2n, c_loop, py_loop, fused), flush=True)
