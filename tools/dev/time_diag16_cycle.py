"""Dev A/B (GPU box): DiagQuadratic at (16,0) with per-chain shapes -- step_all, measure and cycle(10) once the shapes are live.
    python tools/dev/time_diag16_cycle.py d16_old d16_new"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
out = []
for dtype in ("f64", "f32"):
    e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, 16))), None, [0.0] * 16, None, temp=1.0,
                            n_chains=1 << 20, seed=2026, dtype=dtype)
    for k in range(52):
        e.cycle(2)
    e.sync()
    ms = min(e.time_steps(50, 1) for _ in range(3)) / 50
    t0 = time.perf_counter()
    for k in range(20):
        e.cycle(10)
    e.sync()
    out.append("%%s step %%.1f us, cycle(10) %%.1f us" %% (dtype, ms * 1e3, (time.perf_counter() - t0) / 20 * 1e6))
print("   ".join(out))
''' % ROOT
for rnd in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, METROPOLIS_HIP_LIB=os.path.join(ROOT, "tools", "variants", name + ".so"))
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(rnd, name, out.stdout.strip() or out.stderr[-300:], flush=True)
