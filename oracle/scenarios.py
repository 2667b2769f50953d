"""Workloads behind the golden fixtures -- TEST INFRASTRUCTURE ONLY.

Each scenario is a small, fully specified run of the reference algorithm: the
energy (a Python callable ``energy(real_params, complex_params) -> float`` as the
reference expects, ``metropolis_engine.py:20``), constructor arguments and the
``step_all``/``measure`` protocol.  ``make_golden.py`` runs them through the
imported reference; ``tests/test_oracle_golden.py`` replays them through
``oracle.reference_chain``.  Workload sources: ``README.md:26-44`` (readme_1real),
``demo/toymodel_xypotentialwell.py:13-45`` (well_2real),
``demo/toymodel_complex_and_real.py:17-46`` (landau_toy).
"""
import numpy as np


def _diag(a=(), b=()):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)

    def energy(real_params, complex_params):
        e = 0.0
        if len(a):
            e += float(np.sum(a * np.asarray(real_params) ** 2))
        if len(b):
            e += float(np.sum(b * np.abs(np.asarray(complex_params)) ** 2))
        return e
    return energy


def _coupled(real_params, complex_params):
    """Anisotropic well with real-real and complex-complex couplings (off-diagonal covariances)."""
    x = np.asarray(real_params)
    z = np.asarray(complex_params)
    return float(1.0 * x[0] ** 2 + 2.0 * x[1] ** 2 + 0.6 * x[0] * x[1]
                 + 1.5 * abs(z[0]) ** 2 + 3.0 * abs(z[1]) ** 2
                 + (z[0] * np.conj(z[1]) * (0.4 - 0.7j)).real * 2.0)


def landau_terms(k=1.0, alpha=-1.0, beta=0.5):
    """The term dictionary of demo/toymodel_complex_and_real.py:17-33 (energies are real-valued here; quirk Q11)."""
    def field(r, c):
        a2 = abs(c[0]) ** 2
        return float(r[0] * r[1] * (alpha * a2 + beta * a2 * a2))

    def area(r, c):
        return float(k * (1 - r[0]) ** 2 + k * (1 - r[1]) ** 2)
    return {"complex": {"field": field}, "real": {"field": field, "area": area},
            "all": {"field": field, "area": area}}


def landau_total(k=1.0, alpha=-1.0, beta=0.5):
    terms = landau_terms(k, alpha, beta)["all"]
    return lambda r, c: terms["field"](r, c) + terms["area"](r, c)


def _wall(real_params, complex_params):
    """Hard wall in the spirit of the legacy engine's ``abs(amplitude) >= 1`` (/metropolis_engine.py:139-141)."""
    return abs(real_params[0]) >= 0.25


SCENARIOS = {
    # name: nr, nc, energy, temp, initial point, steps per measure, number of measures
    "readme_1real": dict(energy=lambda r, c: float(r[0] ** 2), real=[0.0], cplx=None, temp=0.01,
                         steps_per_measure=1, n_measures=400),
    "well_2real": dict(energy=_diag(a=(1.0, 1.0)), real=[0.0, 0.0], cplx=None, temp=0.1,
                       steps_per_measure=10, n_measures=120),
    "aniso_3real": dict(energy=_diag(a=(1.0, 2.0, 4.0)), real=[0.3, -0.2, 0.1], cplx=None, temp=1.0,
                        steps_per_measure=2, n_measures=150),
    "aniso_2complex": dict(energy=_diag(b=(1.0, 3.0)), real=None, cplx=[0.1 + 0.2j, -0.3j], temp=1.0,
                           steps_per_measure=2, n_measures=150),
    "coupled_2real_2complex": dict(energy=_coupled, real=[0.1, -0.1], cplx=[0.2 + 0.1j, -0.1 + 0.3j], temp=1.0,
                                   steps_per_measure=2, n_measures=200),
    "landau_toy": dict(energy=landau_terms(), real=[0.0, 0.0], cplx=[0j], temp=0.1,
                       steps_per_measure=10, n_measures=100),
    "wall_1real_1complex": dict(energy=_diag(a=(0.5,), b=(1.0,)), real=[0.0], cplx=[0.1j], temp=1.0,
                                steps_per_measure=3, n_measures=100, reject=_wall),
    "zero_temp_2real": dict(energy=_diag(a=(1.0, 3.0)), real=[1.0, -1.0], cplx=None, temp=0.0,
                            steps_per_measure=5, n_measures=60),
    # group-wise stepping of mixed engines (metropolis_engine.py:209-239 called directly, as the author's cylinder
    # driver does: separate real / complex widths in exampledata.csv) and the magnitude-phase sampler (:168-207)
    "groups_2real_2complex": dict(energy=_coupled, real=[0.1, -0.1], cplx=[0.2 + 0.1j, -0.1 + 0.3j], temp=1.0,
                                  ops=("real", "complex", "all", "real", "measure"), n_measures=120),
    "groups_landau_terms": dict(energy=landau_terms(), real=[0.0, 0.0], cplx=[0j], temp=0.1,
                                ops=("real", "complex", "real", "complex", "measure"), n_measures=100),
    "magphase_1real_2complex": dict(energy=_diag(a=(0.5,), b=(1.0, 3.0)), real=[0.2], cplx=[0.3 + 0.1j, 0.2j], temp=0.5,
                                    ops=("real", "complex", "measure"), n_measures=150, method="magnitude-phase"),
    "magphase_2complex": dict(energy=_diag(b=(1.0, 2.0)), real=None, cplx=[0.3 + 0.1j, 0.2j], temp=0.5,
                              ops=("complex", "complex", "all", "measure"), n_measures=120, method="magnitude-phase"),
}


def ops(spec):
    """One cycle of the scenario's protocol: step kinds ("all", "real", "complex") ending in "measure"."""
    if "ops" in spec:
        return tuple(spec["ops"])
    return ("all",) * spec["steps_per_measure"] + ("measure",)


def n_steps(spec):
    return sum(op != "measure" for op in ops(spec)) * spec["n_measures"]


def dims(spec):
    nr = 0 if spec["real"] is None else len(spec["real"])
    nc = 0 if spec["cplx"] is None else len(spec["cplx"])
    return nr, nc
