"""Dev probe (GPU box): walk the streamed-shape path call by call with a sync after each, to localise a failure."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
for dtype, n in (("f64", 70), ("f32", 70), ("f64", 128)):
    e = me.MetropolisEngine(me.DenseQuadratic(amat), None, list(np.linspace(-0.2, 0.2, 64)), None, temp=1.0, n_chains=n, seed=41, dtype=dtype, sampling_width=0.1)
    print("created", dtype, n, flush=True)
    for k in range(58):
        e.step_all(2); e.sync()
        print("step", k, flush=True)
        e.measure(); e.sync()
        print("measure", k, flush=True)
    for sweeps in (1, 1, 3, 5):
        e.step_all(sweeps); e.sync()
        print("streamed step", sweeps, e.acceptance_rate(), flush=True)
