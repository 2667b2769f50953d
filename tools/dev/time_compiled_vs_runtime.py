"""Dev (GPU box): cov_mode="reference" at 100 and 128 real parameters -- the space's own compiled kernel set (build.build_dims, streamed
shapes) against the runtime-dimension set (forced by lowering build.MAX_COMPILED_DOF in a child process).
    python tools/dev/time_compiled_vs_runtime.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, numpy as np
sys.path.insert(0, %r)
import metropolisengine_amd as me
from metropolisengine_amd import build
if sys.argv[1] == "runtime":
    build.MAX_COMPILED_DOF = build.MAX_REGISTER_DOF
for dtype in ("f64", "f32"):
    for nr in (100, 128):
        n = 1 << 14
        e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr))), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=3,
                                dtype=dtype, sampling_width=0.05)
        for k in range(51):
            e.step_all(1); e.measure()
        e.sync()
        ms_step = min(e.time_steps(5, 1) for _ in range(2)) / 5
        t0 = time.perf_counter()
        for k in range(3):
            e.measure()
        e.sync()
        print("%%s %%s %%3d real x 2^14 chains: step_all %%.3f ms, measure + factor refresh %%.2f ms" %% (sys.argv[1], dtype, nr, ms_step, (time.perf_counter() - t0) / 3 * 1e3), flush=True)
''' % ROOT
for mode in ("compiled", "runtime"):
    out = subprocess.run([sys.executable, "-c", CHILD, mode], capture_output=True, text=True)
    print(out.stdout.strip() or out.stderr[-400:], flush=True)
