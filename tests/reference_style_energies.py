"""Energies written the way the reference's README and demos write them (README.md:26-33,
demo/toymodel_xypotentialwell.py:8-32, demo/toymodel_complex_and_real.py:10-33): plain Python on ``real_params`` /
``complex_params``.  Shared by tests/test_pyenergy_cpu.py (tracing, code generation, the hipcc build) and
tests/test_gpu_pyenergy.py (the same callables on the GPU against the oracle) -- the two must trace IDENTICAL functions so
that the plugins built on the CPU side are the ones the GPU side loads."""
import numpy as np

def energy_function(x):                                  # README.md:26-27
    return x ** 2


def readme_energy(real_params, complex_params):          # README.md:33
    return energy_function(*real_params)


class WellSystem:                                        # demo/toymodel_xypotentialwell.py:8-20
    def __init__(self, const=1):
        self.const = const

    def calc_system_energy(self, state):
        x = state[0]
        y = state[1]
        return self.const * (x ** 2 + y ** 2)


class LandauSystem:                                      # demo/toymodel_complex_and_real.py:10-27
    def __init__(self, k, alpha, beta):
        self.k, self.alpha, self.beta = k, alpha, beta

    def calc_system_energy(self, x, y, c):
        area_term = self.k * (1 - x) ** 2 + self.k * (1 - y) ** 2
        field_term = x * y * (self.alpha * c * c.conjugate() + self.beta * c ** 2 * c.conjugate() ** 2)
        return area_term + field_term

    def calc_field_energy(self, x, y, c):
        return x * y * (self.alpha * c * c.conjugate() + self.beta * c ** 2 * c.conjugate() ** 2)

    def calc_area_energy(self, x, y):
        return self.k * (1 - x) ** 2 + self.k * (1 - y) ** 2


WELL = WellSystem(const=1)
LANDAU = LandauSystem(k=1, alpha=-1, beta=.5)


def well_energy(real_params, complex_params):            # the lambda of demo/toymodel_xypotentialwell.py:32
    return WELL.calc_system_energy(real_params)


def landau_dictionary():                                 # demo/toymodel_complex_and_real.py:31-33
    field_fct = lambda r, c: LANDAU.calc_field_energy(*r, *c)                                   # noqa: E731
    area_fct = lambda real_params, complex_params: LANDAU.calc_area_energy(*real_params)       # noqa: E731
    return {"complex": {"field": field_fct}, "real": {"field": field_fct, "area": area_fct},
            "all": {"field": field_fct, "area": area_fct}}


def landau_total(real_params, complex_params):
    return LANDAU.calc_system_energy(*real_params, *complex_params)


def numpy_style(real_params, complex_params):            # reductions, ufuncs, a matrix, a non-integer power
    a = np.array([[2.0, 0.3, 0.0], [0.3, 1.0, 0.2], [0.0, 0.2, 0.5]])
    return (real_params @ a @ real_params + np.sum(np.abs(complex_params) ** 2) + 0.1 * np.exp(-np.sum(real_params ** 2))
            + np.sqrt(1.0 + real_params[0] ** 2) ** 1.5 + np.real(complex_params[0] * np.conj(complex_params[1])))


def wall(real_params, complex_params):                   # reject_condition: the legacy hard wall, /metropolis_engine.py:139-141
    return abs(real_params[0]) >= 1




# ---- beyond 96 degrees of freedom: 100 real parameters, anisotropic well with one coupling (a user energy there compiles the
# register-resident kernels for its own size instead of falling to the runtime-dimension set)
WEIGHTS_100 = np.linspace(0.5, 2.0, 100)


def hundred_parameters(real_params, complex_params):
    return np.sum(WEIGHTS_100 * real_params ** 2) + 0.1 * real_params[0] * real_params[99]


# ---- for the setters (set_energy_function / set_reject_condition, metropolis_engine.py:134-146)
def stiffer_well(real_params, complex_params):
    return 2.0 * energy_function(*real_params)


def narrow_wall(real_params, complex_params):
    return abs(real_params[0]) >= 0.05


# (energy, n_real, n_complex, reject) of every plugin the examples and tests construct: built by __graft_entry__.build()
PLUGINS = (
    (readme_energy, 1, 0, None), (well_energy, 2, 0, None), (landau_dictionary(), 2, 1, wall), (hundred_parameters, 100, 0, None),
    (stiffer_well, 1, 0, None), (readme_energy, 1, 0, narrow_wall), (landau_total, 2, 1, None), (landau_dictionary(), 2, 1, None),
)
