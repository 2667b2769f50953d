"""Dev (GPU box): the compile-on-demand kernel set at a given size (default 128 = build.MAX_COMPILED_DOF) against the oracle,
cov_mode="reference" across the 50-measure threshold (same protocol as tests/test_gpu_streamed_shapes.py at 100 parameters).
    python tools/dev/check_compiled_dims.py [n_real]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import metropolisengine_amd as me
from oracle import energies
from oracle.manychain import ManyChainOracle
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n, seed = 70, 51
x0 = list(np.linspace(-0.2, 0.2, nr))
weights = tuple(np.linspace(0.5, 2.0, nr))
t0 = time.time()
eng = me.MetropolisEngine(me.DiagQuadratic(weights), None, x0, None, temp=1.0, n_chains=n, seed=seed, dtype="f64", sampling_width=0.1)
print("engine ready after %.1f s" % (time.time() - t0), flush=True)
ora = ManyChainOracle(nr, 0, energies.diag_quadratic(nr, 0, weights, ()), n, seed=seed, temp=1.0, initial_real_params=x0, sampling_width=0.1)
for k in range(54):
    eng.step_all(2); ora.step(2); eng.measure(); ora.measure()
fr, _ = eng.proposal_factors()
print("cov   max |diff|", np.max(np.abs(eng.covariance_matrix_real - ora.cov_real)))
print("factor max |diff|", np.max(np.abs(fr - ora.factor_real)))
for sweeps in (1, 3):
    eng.step_all(sweeps); ora.step(sweeps)
print("x     max |diff|", np.max(np.abs(eng._get(0) - ora.x)), "accepts equal", eng.accept_stats() == (ora.accepted, ora.proposed))
big = me.MetropolisEngine(me.DiagQuadratic(weights), None, x0, None, temp=1.0, n_chains=1 << 14, seed=seed, dtype="f64", sampling_width=0.1)
big.step_all(3)
print("2^14 chains: %.3f ms per step_all, %.3f ms per measure" % (big.time_steps(10, 1) / 10, big.time_measure(3) / 3 if hasattr(big, "time_measure") else float("nan")))
