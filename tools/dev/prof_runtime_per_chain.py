import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import metropolisengine_amd as me
nr, n = 140, 1 << 14
e = me.MetropolisEngine(me.DiagQuadratic(tuple(np.linspace(0.5, 2.0, nr))), None, [0.0] * nr, None, temp=1.0, n_chains=n, seed=3, dtype="f64", sampling_width=0.05)
for k in range(56):
    e.step_all(1); e.measure()
e.sync()
