"""Dev A/B (GPU box): variants of the (64,0) kernel set in cov_mode="reference" (streamed per-chain shapes), 2^17 chains."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
n = 1 << 17
e = me.MetropolisEngine(me.DenseQuadratic(m @ m.T / 64 + np.identity(64)), None, [0.0] * 64, None, temp=1.0, n_chains=n, seed=2026, sampling_width=0.2, dtype=sys.argv[1])
for _ in range(52):
    e.step_all(); e.measure()
e.sync(); t0 = time.perf_counter()
for _ in range(5): e.measure()
e.sync(); meas = (time.perf_counter() - t0) / 5 * 1e3
step = min(e.time_steps(10, 1) for _ in range(2)) / 10
print("measure %%.2f ms, step %%.3f ms" %% (meas, step))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        for dt in ("f32", "f64"):
            env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
            out = subprocess.run([sys.executable, "-c", CHILD, dt], env=env, capture_output=True, text=True)
            print(rnd, name, dt, out.stdout.strip() or out.stderr[-300:], flush=True)
