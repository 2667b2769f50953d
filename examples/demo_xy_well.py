"""The reference's simplest demo (demo/toymodel_xypotentialwell.py: two real parameters in the well E = c (x^2 + y^2),
temp 0.1, 1000 x (10 steps + 1 measure)) on the GPU engine -- once as the reference runs it (one chain) and once as an
ensemble of 2^16 chains whose pooled statistics replace the single chain's time averages.

    python examples/demo_xy_well.py            (needs an MI355X and the built library)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402
from metropolisengine_amd.distributed import moments_to_statistics  # noqa: E402


def main(const=1.0, temp=0.1, n_measures=1000, steps_per_measure=10, ensemble=1 << 16, seed=12345):
    # the energy is a device-side specification in the slot where the reference takes its Python callable
    energy = me.DiagQuadratic(a=(const, const))

    # --- one chain, the reference's loop
    engine = me.MetropolisEngine(energy, initial_real_params=np.array([0.0, 0.0]), temp=temp, seed=seed)
    for _ in range(n_measures):
        for _ in range(steps_per_measure):
            engine.step_all()
        engine.measure()
    print("single chain: running mean %s, running covariance\n%s" % (engine.real_mean, engine.covariance_matrix_real))
    for name, value in zip(engine.observables_names, engine.observables_mean):
        print("   <%s> = %.4f" % (name, value))

    # --- the same protocol on an ensemble: one launch steps every chain, step_all(10) fuses the ten sweeps
    many = me.MetropolisEngine(energy, initial_real_params=[0.0, 0.0], temp=temp, n_chains=ensemble, seed=seed)
    for _ in range(n_measures):
        many.step_all(steps_per_measure)
        many.measure()
    stats = moments_to_statistics(many.pooled_moments(), 2, 0)
    print("ensemble    mean", stats["mean"], " covariance diag", np.diag(stats["covariance"]),
          " (exact: %.4f)" % (temp / (2 * const)))
    print("ensemble    acceptance %.3f, mean sampling width %.3f" % (many.acceptance_rate(),
                                                                      float(np.mean(many.real_group_sampling_width))))
    return engine, many, stats


if __name__ == "__main__":
    main()
