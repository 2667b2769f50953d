import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
from oracle import energies
from oracle.manychain import ManyChainOracle
m = np.random.default_rng(5).standard_normal((64, 64))
amat = m @ m.T / 64 + np.identity(64)
n, seed = 70, 41
x0 = list(np.linspace(-0.2, 0.2, 64))
mode = sys.argv[1]
eng = me.MetropolisEngine(me.DenseQuadratic(amat), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64", sampling_width=0.1)
ora = ManyChainOracle(64, 0, energies.dense_quadratic(64, 0, amat), n, seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.1)
print("created", flush=True)
for k in range(58):
    eng.step_all(2)
    if mode == "oracle": ora.step(2)
    eng.measure()
    if mode == "oracle": ora.measure()
    if mode == "chol": np.linalg.cholesky(np.tile(amat, (70, 1, 1)))
    print("iter", k, flush=True)
eng.sync()
print("synced", flush=True)
for sweeps in (1, 1, 3, 5):
    eng.step_all(sweeps)
eng.sync()
print("done", eng.acceptance_rate(), flush=True)
