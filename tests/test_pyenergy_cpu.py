"""Python energy callables (metropolisengine_amd/pyenergy.py), CPU side: the reference's own ways of writing an energy --
README.md:26-33, demo/toymodel_xypotentialwell.py:13-32, demo/toymodel_complex_and_real.py:17-33 -- are traced on symbolic
parameters; the recorded graph, evaluated here with numpy, must equal the function called on numbers; what cannot be traced
fails loudly; and the generated HIP source compiles for gfx950 (hipcc cross-compiles without a GPU).  The plugins built here
are the ones tests/test_gpu_pyenergy.py runs on the GPU box."""
import math

import numpy as np
import pytest

from metropolisengine_amd import pyenergy as pe

from reference_style_energies import (landau_dictionary, landau_total, numpy_style, readme_energy, wall,   # noqa: E402
                                      well_energy)

# ---- a numpy interpreter of the recorded graph (test infrastructure only) ----------------------------------------
_OPS = {"add": np.add, "sub": np.subtract, "mul": np.multiply, "div": np.divide, "pow": np.power, "neg": np.negative,
        "abs": np.abs, "sqrt": np.sqrt, "exp": np.exp, "log": np.log, "sin": np.sin, "cos": np.cos, "tan": np.tan,
        "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh, "arctan": np.arctan, "arctan2": np.arctan2}
_CMP = {"lt": np.less, "le": np.less_equal, "gt": np.greater, "ge": np.greater_equal, "eq": np.equal, "ne": np.not_equal}


def evaluate(node, x):
    if isinstance(node, pe.SymBool):
        if node.op == "cmp":
            return _CMP[node.cmp](evaluate(node.args[0], x), evaluate(node.args[1], x))
        if node.op == "const":
            return node.cmp
        if node.op == "not":
            return not evaluate(node.args[0], x)
        a, b = evaluate(node.args[0], x), evaluate(node.args[1], x)
        return (a and b) if node.op == "and" else (a or b)
    if node.op == "x":
        return x[node.value]
    if node.op == "const":
        return node.value
    return _OPS[node.op](*[evaluate(a, x) for a in node.args])


@pytest.mark.parametrize("fn,nr,nc", [(readme_energy, 1, 0), (well_energy, 2, 0), (landau_total, 2, 1), (numpy_style, 3, 2)],
                         ids=["readme", "xy_well", "landau", "numpy_style"])
def test_traced_graph_equals_the_function_on_numbers(fn, nr, nc):
    node = pe.trace_energy(fn, nr, nc)
    rng = np.random.default_rng(3)
    for _ in range(50):
        x = rng.standard_normal(nr + 2 * nc)
        want = fn(x[:nr], x[nr:nr + nc] + 1j * x[nr + nc:])
        assert abs(complex(want).imag) < 1e-12              # (the Landau toy is complex-typed with zero imaginary part, Q11)
        assert np.isclose(evaluate(node, x), complex(want).real, rtol=1e-13, atol=1e-13)


def test_term_dictionary_and_reject_are_traced():
    source, names = pe.generate_source(landau_dictionary(), 2, 1, reject=wall)
    assert names == ("area", "field")
    assert "#define ME_USER_N_TERMS 2" in source and "#define ME_USER_HAS_REJECT" in source
    assert "term == 0 ? 1u : 3u" in source                  # "area": real group only; "field": both groups
    cond = pe.trace_reject(wall, 2, 1)
    assert evaluate(cond, np.array([1.5, 0.0, 0.0, 0.0])) and not evaluate(cond, np.array([-0.5, 0.0, 0.0, 0.0]))
    both = pe.trace_reject(lambda r, c: (abs(r[0]) >= 1) | (np.abs(c[0]) > 2), 2, 1)
    assert evaluate(both, np.array([0.0, 0.0, 3.0, 0.0])) and not evaluate(both, np.array([0.0, 0.0, 1.0, 0.0]))


def test_what_cannot_be_traced_fails_loudly():
    with pytest.raises(pe.TraceError, match="branches|`if`"):
        pe.trace_energy(lambda r, c: r[0] if r[0] > 0 else -r[0], 1, 0)
    with pytest.raises(pe.TraceError, match="numpy functions"):
        pe.trace_energy(lambda r, c: math.exp(r[0]), 1, 0)
    with pytest.raises(pe.TraceError, match="& \\| ~"):
        pe.trace_reject(lambda r, c: abs(r[0]) >= 1 or abs(r[1]) >= 1, 2, 0)
    with pytest.raises(pe.TraceError):
        pe.trace_energy(lambda r, c: r, 2, 0)               # not a scalar


def test_generated_plugins_compile_for_gfx950():
    """hipcc around the engine's kernels, exactly what the constructor does on first use; the libraries stay in
    metropolisengine_amd/lib/ and travel to the GPU box, where tests/test_gpu_pyenergy.py finds them up to date."""
    import os
    for energy, nr, nc, reject in ((readme_energy, 1, 0, None), (well_energy, 2, 0, None), (landau_dictionary(), 2, 1, wall)):
        spec = pe.PythonEnergy(energy, reject=reject)
        plugin = spec.build_plugin(nr, nc)
        assert os.path.exists(plugin) and spec.name.startswith("py") and spec.name in plugin
        assert spec.term_names == (("area", "field") if isinstance(energy, dict) else ("total",))
