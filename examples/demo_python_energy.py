"""The reference's README example (README.md:26-51) with its energy left as the Python it is: the lambda is traced once on
symbolic parameters, compiled into a device function (metropolisengine_amd/pyenergy.py; ~10 s on first use, cached) and the
kernels call that -- first as the reference runs it (one chain), then as 2^16 chains.

    python examples/demo_python_energy.py            (needs an MI355X and the built library)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import metropolisengine_amd as me  # noqa: E402


def energy_function(x):
    return x**2


def main(n_steps=1000, ensemble=1 << 16, temp=.01):
    my_engine = me.MetropolisEngine(lambda real_params, complex_params: energy_function(*real_params),
                                    initial_real_params=[0.0], temp=temp, seed=12345)
    for i in range(n_steps):
        my_engine.step_all()
        my_engine.measure()
    print("one chain:", my_engine.real_mean, my_engine.covariance_matrix_real,
          list(zip(my_engine.observables_names, my_engine.observables_mean)))

    many = me.MetropolisEngine(lambda real_params, complex_params: energy_function(*real_params),
                               initial_real_params=[0.0], temp=temp, n_chains=ensemble, seed=12345)
    for i in range(n_steps // 10):
        many.cycle(10)                       # ten step_all() and one measure() per launch
    x = many.real_params[:, 0]
    print("ensemble of %d chains: mean %.5f, variance %.5f (exact T/2 = %.5f), acceptance %.3f"
          % (ensemble, x.mean(), x.var(), temp / 2, many.acceptance_rate()))
    return my_engine, many


if __name__ == "__main__":
    main()
