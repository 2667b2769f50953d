"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/metropolis_engine.h
declares, reports capabilities, and fails loudly without a GPU (no compute calls here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from metropolisengine_amd import _capi, energy
from metropolisengine_amd.engine import MetropolisEngine, unpack_complex_block, unpack_real_block

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "metropolis_engine.h")).read()
    return sorted(set(re.findall(r"^\s*(?:int|const char \*)\s+(me_[a-z_]+)\s*\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libmetropolis_hip.so does not export %s" % name
    assert sorted(_capi.SYMBOLS) == declared, "ctypes binding and header disagree"
    assert lib.me_abi_version() == _capi.ABI_VERSION


def test_config_struct_layout_matches_header():
    # natural C layout of struct me_config on x86-64
    assert ctypes.sizeof(_capi.MeConfig) == 136
    assert _capi.MeConfig.n_chains.offset == 8 and _capi.MeConfig.temp.offset == 48
    assert _capi.MeConfig.energy_coeffs.offset == 80 and _capi.MeConfig.covariance_complex.offset == 120


def test_capability_query():
    lib = _capi.load()
    for nr, nc in [(1, 0), (2, 0), (16, 0), (64, 0), (4, 4), (2, 1), (2, 7), (0, 1)]:
        for dt in (_capi.ME_F32, _capi.ME_F64):
            assert lib.me_supported(dt, nr, nc, _capi.ENERGY_ISO_QUAD) == 1
            assert lib.me_supported(dt, nr, nc, _capi.ENERGY_DIAG_QUAD) == 1
    assert lib.me_supported(_capi.ME_F32, 2, 1, _capi.ENERGY_LANDAU_TOY) == 1
    assert lib.me_supported(_capi.ME_F32, 16, 0, _capi.ENERGY_LANDAU_TOY) == 0
    assert lib.me_supported(_capi.ME_F32, 64, 0, _capi.ENERGY_DENSE_QUAD) == 1
    assert lib.me_supported(_capi.ME_F32, 2, 7, _capi.ENERGY_CYLINDER) == 1
    assert lib.me_supported(_capi.ME_F32, 5, 5, _capi.ENERGY_ISO_QUAD) == 0


def test_argument_errors_match_the_reference():
    with pytest.raises(ValueError):                      # metropolis_engine.py:37-39
        MetropolisEngine(energy.IsoQuadratic())
    with pytest.raises(AssertionError):                  # :92
        MetropolisEngine(energy.IsoQuadratic(), initial_real_params=[0.0], temp=-1.0)
    with pytest.raises(TypeError):                       # neither an EnergySpec nor a callable / term dictionary
        MetropolisEngine(42.0, initial_real_params=[0.0])
    with pytest.raises(TypeError):
        MetropolisEngine(energy.IsoQuadratic(), reject_condition=lambda r, c: False, initial_real_params=[0.0])
    with pytest.raises(ValueError):
        MetropolisEngine(energy.DiagQuadratic(a=[1.0]), initial_real_params=[0.0, 0.0])


def test_create_validation_without_gpu():
    """me_create validates its arguments before touching the device; on a GPU-less host a valid config then fails
    loudly with ME_ERR_HIP instead of falling back to the CPU."""
    lib = _capi.load()
    cfg = _capi.MeConfig()
    handle = ctypes.c_void_p()
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_INVALID   # abi_version 0
    assert "abi_version" in _capi.last_error()
    init = np.zeros(5)
    coef = np.ones(1)
    cfg.abi_version, cfg.n_chains, cfg.n_real, cfg.n_complex = _capi.ABI_VERSION, 8, 5, 0
    cfg.target_acceptance, cfg.sampling_width = 0.3, 0.05
    cfg.n_energy_coeffs = 1
    cfg.energy_coeffs = coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    cfg.initial_params = init.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_UNSUPPORTED  # no (5,0) kernels
    cfg.n_real = 16
    cfg.temp = -1.0
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_INVALID
    assert "temp" in _capi.last_error()


def test_packed_layout_helpers():
    nr, nc = 3, 2
    pr = nr * (nr + 1) // 2
    packed = np.arange(1.0, pr + nc * nc + 1.0)[None, :]
    real = unpack_real_block(packed, nr)[0]
    assert np.array_equal(real, real.T) and real[2, 1] == packed[0, 4] and real[0, 0] == 1.0
    cplx = unpack_complex_block(packed, nr, nc)[0]
    assert cplx[0, 0] == packed[0, pr] and cplx[1, 0] == packed[0, pr + 1] + 1j * packed[0, pr + 2]
    assert cplx[0, 1] == np.conj(cplx[1, 0]) and cplx[1, 1] == packed[0, pr + 3]


def test_user_energy_plugin_loads_and_registers():
    """The example plugin (examples/user_energy_cylinder.h compiled around the kernels) resolves against the main
    library and registers its kernel sets; a missing path fails loudly."""
    from metropolisengine_amd import build
    lib = _capi.load()
    path = build.user_plugin_path("cylinder", 2, 7)
    assert os.path.exists(path), "run `python -m metropolisengine_amd.build` (it builds the example plugin)"
    assert lib.me_load_plugin(path.encode()) == _capi.ME_OK
    assert lib.me_load_plugin(b"/nonexistent/libme_user_x.so") == _capi.ME_ERR_INVALID
    assert "dlopen" in _capi.last_error()
    # an engine asking for an unknown plugin name is refused before any device work
    cfg = _capi.MeConfig()
    init, coef = np.zeros(16), np.ones(3)
    cfg.abi_version, cfg.n_chains, cfg.n_real, cfg.n_complex = _capi.ABI_VERSION, 8, 2, 7
    cfg.target_acceptance, cfg.sampling_width, cfg.temp = 0.3, 0.05, 0.1
    cfg.energy_kind, cfg.n_energy_coeffs = _capi.ENERGY_USER, 3
    cfg.energy_coeffs = coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    cfg.initial_params = init.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    cfg.user_energy_name = b"no_such_plugin"
    handle = ctypes.c_void_p()
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_UNSUPPORTED
    assert "no_such_plugin" in _capi.last_error()
    cfg.user_energy_name = b"cylinder"
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_HIP   # found; no GPU here


def test_reject_user_needs_a_plugin_with_a_reject_function():
    """ME_REJECT_USER on a plugin that defines no me_user_reject is refused (the wall would otherwise be dropped silently
    and chains could walk into the forbidden region; the reference evaluates reject_condition before the energy,
    metropolis_engine.py:247-249).  examples/user_energy_landau_terms.h has none; examples/user_energy_cylinder.h does."""
    from metropolisengine_amd import build
    lib = _capi.load()
    for name, nr, nc in (("landau_terms", 2, 1), ("cylinder", 2, 7)):
        assert lib.me_load_plugin(build.user_plugin_path(name, nr, nc).encode()) == _capi.ME_OK
    handle = ctypes.c_void_p()
    coef = np.ones(3)

    def create(name, nr, nc):
        cfg = _capi.MeConfig()
        init = np.zeros(nr + 2 * nc)
        cfg.abi_version, cfg.n_chains, cfg.n_real, cfg.n_complex = _capi.ABI_VERSION, 8, nr, nc
        cfg.target_acceptance, cfg.sampling_width, cfg.temp = 0.3, 0.05, 0.1
        cfg.energy_kind, cfg.n_energy_coeffs = _capi.ENERGY_USER, 3
        cfg.energy_coeffs = coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        cfg.initial_params = init.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
        cfg.user_energy_name = name
        cfg.reject_kind = _capi.REJECT_USER
        return lib.me_create(ctypes.byref(cfg), ctypes.byref(handle))

    assert create(b"landau_terms", 2, 1) == _capi.ME_ERR_UNSUPPORTED
    assert "me_user_reject" in _capi.last_error()
    assert create(b"cylinder", 2, 7) == _capi.ME_ERR_HIP          # accepted; no GPU here


def test_per_device_cache_is_keyed_by_device(tmp_path):
    """Launch properties resolved lazily (a __device__ function pointer, the raised dynamic-LDS limit, the CU count) are
    cached per DEVICE of the process, not per process (csrc/me_per_device.h)."""
    import subprocess
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "per_device_host.cpp")
    out = str(tmp_path / "libper_device.so")
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", src, "-o", out, "-pthread"], check=True)
    assert ctypes.CDLL(out).me_test_per_device() == 0


def test_runtime_dimension_kernel_set_is_the_fallback_beyond_96_dof():
    """Parameter spaces beyond the register-resident kernels (n_real + 2 n_complex > 96) resolve to the runtime-dimension
    set of the main library (csrc/me_runtime_dims.hip): separable and dense energies; identity shape or per-chain shapes, and one
    shared factor for pure real spaces."""
    lib = _capi.load()
    assert lib.me_supported(_capi.ME_F32, 200, 0, _capi.ENERGY_ISO_QUAD) == 1
    assert lib.me_supported(_capi.ME_F64, 60, 25, _capi.ENERGY_DIAG_QUAD) == 1
    assert lib.me_supported(_capi.ME_F32, 200, 0, _capi.ENERGY_DENSE_QUAD) == 1       # LDS-resident form
    assert lib.me_supported(_capi.ME_F32, 200, 0, _capi.ENERGY_LANDAU_TOY) == 0
    assert lib.me_supported(_capi.ME_F32, 5, 0, _capi.ENERGY_ISO_QUAD) == 0          # small sizes are built on demand instead
    cfg = _capi.MeConfig()
    init, coef = np.zeros(200), np.ones(1)
    cfg.abi_version, cfg.n_chains, cfg.n_real, cfg.n_complex = _capi.ABI_VERSION, 8, 200, 0
    cfg.target_acceptance, cfg.sampling_width, cfg.temp = 0.3, 0.05, 1.0
    cfg.energy_kind, cfg.n_energy_coeffs = _capi.ENERGY_ISO_QUAD, 1
    cfg.energy_coeffs = coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    cfg.initial_params = init.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    handle = ctypes.c_void_p()
    cfg.cov_mode = _capi.COV_REFERENCE
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_HIP     # per-chain shapes, pure real: accepted; no GPU here
    cfg.n_real, cfg.n_complex = 100, 50                                                  # ... with complex parameters as well
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_HIP
    cfg.cov_mode = _capi.COV_POOLED                                                      # ... but no shared factor there
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_UNSUPPORTED
    assert "identity proposal shape" in _capi.last_error()
    cfg.n_real, cfg.n_complex = 200, 0
    cfg.cov_mode = _capi.COV_FIXED
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_HIP     # accepted; no GPU here
    cfg.cov_mode = _capi.COV_POOLED
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_HIP     # one shared factor: pure real spaces
    # a dense quadratic form is staged in LDS: 1 000 parameters do not fit
    big = np.zeros(1000)
    dense = np.zeros(1000 * 1000)
    cfg.n_real, cfg.energy_kind, cfg.n_energy_coeffs, cfg.cov_mode = 1000, _capi.ENERGY_DENSE_QUAD, dense.size, _capi.COV_FIXED
    cfg.energy_coeffs = dense.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    cfg.initial_params = big.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)) == _capi.ME_ERR_UNSUPPORTED
    assert "LDS" in _capi.last_error()
