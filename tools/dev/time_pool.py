"""Dev timing (GPU box): k_pool_reduce and the config-5 cycle with a pooled reduction every measure."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import metropolisengine_amd as me
src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "examples", "user_energy_cylinder.h")
for log2n, dims in ((18, (2, 7)), (20, (16, 0)), (19, (64, 0))):
    n = 1 << log2n
    if dims == (2, 7):
        e = me.MetropolisEngine(me.UserEnergy("cylinder", src, (1.0, 0.5, 1.0)), me.AbsReal0AtLeast(1.0), [0.1, 0.0], [0.05] * 7, temp=0.1, n_chains=n, seed=1)
    else:
        e = me.MetropolisEngine(me.IsoQuadratic(1.0), None, [0.0] * dims[0], None, temp=1.0, n_chains=n, seed=1, cov_mode="fixed")
    e.step_all(10); e.pooled_moments(); e.sync()
    t0 = time.perf_counter()
    for _ in range(20): e.pooled_moments()
    dt = (time.perf_counter() - t0) / 20
    print("dims %s n=2^%d: pooled_moments %.1f us per call" % (dims, log2n, dt * 1e6))
