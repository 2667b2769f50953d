"""``MetropolisEngine``: the reference's class surface over N independent chains on one MI355X.

Mirrors ``metropolisengine.MetropolisEngine`` (/root/reference/metropolisengine/metropolis_engine.py:10-463):
same constructor arguments in the same order, ``step_all()`` / ``measure()``, the same attribute names.  All
arithmetic happens in libmetropolis_hip.so (HIP kernels, C ABI in include/metropolis_engine.h); this class only
marshals arguments and unpacks results.  Differences from the reference, all forced by the GPU setting:

 * ``energy_functions`` is an :class:`~metropolisengine_amd.energy.EnergySpec` (built-in energy or a hand-written HIP device
   function) OR, exactly as in the reference, a Python callable ``(real_params, complex_params) -> float`` / the dictionary of
   term callables: the callable is traced once on symbolic parameters and compiled into a plugin (``pyenergy.py``; hipcc, about a
   minute on first use, cached).  ``reject_condition`` is a :class:`~metropolisengine_amd.energy.RejectSpec` or, with a Python
   energy, a Python predicate.  (The reference silently drops a ``reject_condition`` given to the constructor -- SURVEY.md
   quirk Q6; here it is honoured.)
 * keyword-only extras: ``n_chains``, ``seed``, ``dtype``, ``device``, ``chain_offset``, ``cov_mode``, ``trace_chains``,
   ``trace_stride``, ``track_covariance``, ``reference_energy_ledgers`` (reproduce the reference's two energy ledgers,
   SURVEY.md quirk Q5: ``step_all`` of a mixed engine uses ``energy_total``, group steps ``energy[term]``).
 * with ``n_chains == 1`` attributes have the reference's shapes and ``step_all()`` returns a bool; with more
   chains they gain a leading chain axis and ``step_all()`` returns ``None`` (it stays asynchronous).
 * randomness is a seeded counter-based Philox stream per global chain id instead of numpy's global state.
 * parameter-space size: like the reference, no fixed limit on the number of parameters -- up to 96 real degrees of freedom
   (``n_real + 2 n_complex``) the register-resident kernels, up to 128 a kernel set compiled on first use for the default
   ``cov_mode="reference"``, beyond that the runtime-dimension kernels (identity shape or per-chain shapes for any space, one
   shared factor for pure real ones) as long as 64 x D values fit the LDS (float64: D <= 290); README.md "Limits".
"""
import ctypes

import numpy as np

from . import _capi
from .energy import EnergySpec, RejectSpec

_DTYPES = {"f32": _capi.ME_F32, "float32": _capi.ME_F32, "f64": _capi.ME_F64, "float64": _capi.ME_F64}
_COV_MODES = {"reference": _capi.COV_REFERENCE, "fixed": _capi.COV_FIXED, "pooled": _capi.COV_POOLED}


def _as_double_ptr(arr):
    return arr.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def unpack_real_block(packed, nr):
    """[n, P] packed lower triangles -> [n, nr, nr] symmetric matrices (real block of ME_FIELD_COV)."""
    n = packed.shape[0]
    out = np.zeros((n, nr, nr))
    il = np.tril_indices(nr)
    out[:, il[0], il[1]] = packed[:, :nr * (nr + 1) // 2]
    out[:, il[1], il[0]] = packed[:, :nr * (nr + 1) // 2]
    return out


def unpack_complex_block(packed, nr, nc, hermitian=True):
    """[n, P] -> [n, nc, nc] complex: Hermitian matrices (covariance) or lower-triangular factors."""
    n = packed.shape[0]
    pr = nr * (nr + 1) // 2
    out = np.zeros((n, nc, nc), dtype=np.complex128)
    for i in range(nc):
        for j in range(i):
            v = packed[:, pr + i * i + 2 * j] + 1j * packed[:, pr + i * i + 2 * j + 1]
            out[:, i, j] = v
            if hermitian:
                out[:, j, i] = np.conj(v)
        out[:, i, i] = packed[:, pr + i * i + 2 * i]
    return out


def unpack_real_factor(packed, nr):
    n = packed.shape[0]
    out = np.zeros((n, nr, nr))
    il = np.tril_indices(nr)
    out[:, il[0], il[1]] = packed[:, :nr * (nr + 1) // 2]
    return out


def set_cache_budget(n_bytes):
    """Process-wide tuning knob (``me_set_cache_budget``): launches whose working set exceeds ``n_bytes`` stream their
    read-once / write-once fields with the non-temporal cache policy.  Default 224 MiB."""
    _capi.check(_capi.load().me_set_cache_budget(int(n_bytes)))


class MetropolisEngine:
    def __init__(self, energy_functions, reject_condition=None, initial_real_params=None,
                 initial_complex_params=None, sampling_width=0.05, covariance_matrix_real=None,
                 covariance_matrix_complex=None, params_names=None, target_acceptance=.3, temp=0,
                 complex_sample_method="multivariate-gaussian", *, n_chains=1, seed=0, dtype="f32", device=0,
                 chain_offset=0, cov_mode="reference", trace_chains=None, trace_stride=1, track_covariance=False,
                 reference_energy_ledgers=False):
        if initial_real_params is None and initial_complex_params is None:
            print("must give list containing  at least one value for initial real or complex parameters")
            raise ValueError("no initial parameters")                                    # metropolis_engine.py:37-39
        if not isinstance(energy_functions, EnergySpec):
            # the reference's own form: a Python callable (real_params, complex_params) -> float, or its dictionary of
            # term callables (metropolis_engine.py:20, :111-116).  It is TRACED once on symbolic parameters, written out as
            # a HIP device function and compiled around the kernels (pyenergy.py) -- there is still no CPU fallback.
            if not (callable(energy_functions) or isinstance(energy_functions, dict)):
                raise TypeError("energy_functions must be an EnergySpec (IsoQuadratic, DiagQuadratic, DenseQuadratic, LandauToy, "
                                "CylinderSurrogate, UserEnergy), a callable (real_params, complex_params) -> float, or the "
                                "reference's dictionary of term callables")
            from .pyenergy import PythonEnergy, PythonReject
            traced_reject = reject_condition if (reject_condition is not None and not isinstance(reject_condition, RejectSpec)) else None
            if traced_reject is not None and not callable(traced_reject):
                raise TypeError("reject_condition must be a RejectSpec, a callable (real_params, complex_params) -> bool, or None")
            energy_functions = PythonEnergy(energy_functions, reject=traced_reject)
            if traced_reject is not None:
                reject_condition = PythonReject()
        if reject_condition is not None and not isinstance(reject_condition, RejectSpec):
            raise TypeError("reject_condition must be a metropolisengine_amd.energy.RejectSpec or None (a Python predicate is "
                            "accepted together with a Python energy: both are traced into one plugin)")
        if complex_sample_method not in ("magnitude-phase", "multivariate-gaussian"):
            print("complex_sample_method", complex_sample_method, "not recognized")       # :131-133
            print("defaulting to multivariate-gaussian")
        if temp is None or not temp >= 0:
            raise AssertionError("temp must be >= 0")                                     # :92
        if isinstance(sampling_width, (list, tuple)):
            # the reference accepts [real, complex] but then breaks in mixed engines (quirk Q7)
            if initial_real_params is not None and initial_complex_params is not None:
                raise ValueError("a [real, complex] sampling_width list is not usable with mixed parameter spaces")
            initial_widths = (float(sampling_width[0]), float(sampling_width[1]))               # :94-95
            sampling_width = sampling_width[0] if initial_real_params is not None else sampling_width[1]
        else:
            initial_widths = (float(sampling_width), float(sampling_width))                     # :97-99

        real0 = np.zeros(0) if initial_real_params is None else np.asarray(initial_real_params, dtype=np.float64).ravel()
        cplx0 = (np.zeros(0, dtype=np.complex128) if initial_complex_params is None
                 else np.asarray(initial_complex_params, dtype=np.complex128).ravel())
        self.num_real_params = int(real0.size)
        self.num_complex_params = int(cplx0.size)
        self.param_space_dims = self.num_real_params + self.num_complex_params          # :60
        nr, nc = self.num_real_params, self.num_complex_params
        self.n_chains = int(n_chains)
        self.dtype = dtype
        self.seed = int(seed)
        self.chain_offset = int(chain_offset)
        self.temp = temp
        self.target_acceptance = target_acceptance
        self._initial_widths = initial_widths     # what the width of an absent group stays at
        # "magnitude-phase" swaps step_complex_group only; step_all keeps the Gaussian sampler (:129-130, quirk Q9)
        self.complex_sample_method = ("magnitude-phase" if complex_sample_method == "magnitude-phase"
                                      else "multivariate-gaussian")
        self._energy_spec = energy_functions
        self._reject_spec = reject_condition
        if params_names:
            self.params_names = params_names                                             # :82-85
        else:
            self.params_names = ["param_" + str(i) for i in range(nr + nc)]
        self.observables_names = ["abs_param_" + str(i) for i in range(nr + nc)]        # :86-87
        self.observables_names.extend(["param_" + str(i) + "_squared" for i in range(nr)])
        self.energy_term_names = list(energy_functions.term_names)                       # :112-118
        self.df = None

        self._lib = _capi.load()
        coeffs = np.ascontiguousarray(energy_functions.coefficients(nr, nc), dtype=np.float64)
        init = np.ascontiguousarray(np.concatenate((real0, cplx0.real, cplx0.imag)), dtype=np.float64)
        cfg = _capi.MeConfig()
        cfg.abi_version = _capi.ABI_VERSION
        cfg.device_id = self.device = int(device)
        cfg.n_chains = self.n_chains
        cfg.chain_offset = self.chain_offset
        cfg.seed = self.seed
        cfg.n_real, cfg.n_complex = nr, nc
        if dtype not in _DTYPES:
            raise ValueError("dtype must be one of %s" % sorted(_DTYPES))
        cfg.dtype = _DTYPES[dtype]
        if cov_mode not in _COV_MODES:
            raise ValueError("cov_mode must be one of %s" % sorted(_COV_MODES))
        cfg.cov_mode = _COV_MODES[cov_mode]
        self.cov_mode = cov_mode
        # parameter spaces whose packed matrix is too large for registers (> 160 entries, e.g. 64 real parameters) keep
        # the per-chain matrices only where the proposals need them (cov_mode="reference": streamed kernels, pure real
        # spaces) or on request (track_covariance=True: statistics only)
        self.reference_energy_ledgers = bool(reference_energy_ledgers)
        cfg.flags = ((_capi.FLAG_TRACK_COVARIANCE if track_covariance else 0) |
                     (_capi.FLAG_REFERENCE_ENERGY_LEDGERS if reference_energy_ledgers else 0))
        cfg.temp = float(temp)
        cfg.target_acceptance = float(target_acceptance)
        cfg.sampling_width = float(sampling_width)
        cfg.energy_kind = energy_functions.kind
        if hasattr(energy_functions, "ensure_loaded"):          # UserEnergy: build + load its plugin library
            energy_functions.ensure_loaded(nr, nc)
            self._user_name = energy_functions.name.encode()
            cfg.user_energy_name = self._user_name
        cfg.n_energy_coeffs = int(coeffs.size)
        cfg.energy_coeffs = _as_double_ptr(coeffs)
        cfg.reject_kind = reject_condition.kind if reject_condition is not None else _capi.REJECT_NONE
        cfg.reject_bound = reject_condition.bound if reject_condition is not None else 0.0
        cfg.initial_params = _as_double_ptr(init)
        keep = [coeffs, init]
        if covariance_matrix_real is not None and nr:
            c_r = np.ascontiguousarray(covariance_matrix_real, dtype=np.float64)
            if c_r.shape != (nr, nr):
                raise ValueError("covariance_matrix_real must be %dx%d" % (nr, nr))
            cfg.covariance_real = _as_double_ptr(c_r)
            keep.append(c_r)
        if covariance_matrix_complex is not None and nc:
            c_c = np.asarray(covariance_matrix_complex, dtype=np.complex128)
            if c_c.shape != (nc, nc):
                raise ValueError("covariance_matrix_complex must be %dx%d" % (nc, nc))
            c_ri = np.ascontiguousarray(np.stack((c_c.real, c_c.imag), axis=-1), dtype=np.float64)
            cfg.covariance_complex = _as_double_ptr(c_ri)
            keep.append(c_ri)
        handle = ctypes.c_void_p()
        self._handle = None
        if not hasattr(energy_functions, "ensure_loaded"):
            from . import build
            d = nr + 2 * nc
            if not self._lib.me_supported(cfg.dtype, nr, nc, _capi.ENERGY_ISO_QUAD):
                # dimensions outside the prebuilt set: compile this (nr, nc) kernel set once (hipcc) and load it
                _capi.check(self._lib.me_load_plugin(build.build_dims(nr, nc).encode()))
            elif build.MAX_REGISTER_DOF < d <= build.MAX_COMPILED_DOF and cov_mode == "reference":
                # beyond 96 degrees of freedom the runtime-dimension set (no build) has per-chain shapes too, slower ones per
                # step: up to MAX_COMPILED_DOF the space's own kernel set is compiled after all for the reference's
                # semantics (streamed shapes; minutes of hipcc, cached)
                _capi.check(self._lib.me_load_plugin(build.build_dims(nr, nc).encode()))
        _capi.check(self._lib.me_create(ctypes.byref(cfg), ctypes.byref(handle)))
        self._handle = handle
        n_terms = ctypes.c_int32()
        _capi.check(self._lib.me_energy_terms(handle, ctypes.byref(n_terms)), handle)
        if n_terms.value != len(self.energy_term_names):
            raise ValueError("the energy has %d terms on the device but term_names=%r" %
                             (n_terms.value, tuple(self.energy_term_names)))
        alpha, ratio, m = ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        _capi.check(self._lib.me_constants(handle, ctypes.byref(alpha), ctypes.byref(m), ctypes.byref(ratio)), handle)
        self.alpha, self.m, self.ratio = alpha.value, m.value, ratio.value               # :101-107
        self._last_accepted = 0
        # time series (:31-35, :350-356): the reference records its one chain at every measure(); a single-chain engine
        # does the same, a many-chain engine records ``trace_chains`` chains (every ``trace_stride``-th) on request
        self.trace_chains = (1 if self.n_chains == 1 else 0) if trace_chains is None else int(trace_chains)
        self.trace_stride = int(trace_stride)
        if self.trace_chains:
            _capi.check(self._lib.me_trace_enable(handle, self.trace_chains, self.trace_stride), handle)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_handle", None) is not None:
            self._lib.me_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, status):
        _capi.check(status, self._handle)

    # ------------------------------------------------------------------ stepping (metropolis_engine.py:209-259)
    def step_all(self, n_sweeps=1):
        """One (or ``n_sweeps`` fused) propose -> energy -> accept/reject -> width-adaptation step of every chain."""
        self._check(self._lib.me_step(self._handle, int(n_sweeps)))
        if self.n_chains == 1:
            accepted, _ = self.accept_stats()
            took = accepted > self._last_accepted
            self._last_accepted = accepted
            return took if n_sweeps == 1 else None
        return None

    def step_injected(self, normals, uniforms, kind=_capi.STEP_ALL):
        """Test hook (float64 engines): step with caller-supplied draws instead of the Philox stream
        (``me_step_injected``): ``normals[sweep, chain, D]`` and ``uniforms[sweep, chain]`` for the Gaussian kinds,
        ``normals[sweep, chain, nc]`` and ``uniforms[sweep, chain, nc + 2]`` for the magnitude-phase pair."""
        normals = np.ascontiguousarray(normals, dtype=np.float64)
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64)
        nc = self.num_complex_params
        magphase = kind == _capi.STEP_COMPLEX_MAGNITUDE_PHASE
        nz = nc if magphase else self.num_real_params + 2 * nc
        want_u = normals.shape[:2] + ((nc + 2,) if magphase else ())
        if normals.ndim != 3 or normals.shape[1:] != (self.n_chains, nz) or uniforms.shape != want_u:
            raise ValueError("injected streams have the wrong shape for this step kind")
        self._check(self._lib.me_step_injected(self._handle, int(kind), normals.shape[0], _as_double_ptr(normals),
                                               _as_double_ptr(uniforms)))

    def _step_kind(self, kind, n_sweeps):
        before = self.accept_stats()[0] if self.n_chains == 1 else 0
        self._check(self._lib.me_step_kind(self._handle, int(kind), int(n_sweeps)))
        if self.n_chains == 1 and n_sweeps == 1 and kind != _capi.STEP_COMPLEX_MAGNITUDE_PHASE:
            accepted = self.accept_stats()[0]
            self._last_accepted = accepted
            return accepted > before
        return None                                   # the magnitude-phase pair returns None (:175-176)

    def step_real_group(self, n_sweeps=1):
        """metropolis_engine.py:225-239: only the real parameters move, only the real width adapts.  On pure-real
        engines this IS step_all (:56)."""
        return self._step_kind(_capi.STEP_REAL_GROUP, n_sweeps)

    def step_complex_group(self, n_sweeps=1):
        """metropolis_engine.py:209-223, or the magnitude-phase pair (:168-207) when the engine was built with
        ``complex_sample_method="magnitude-phase"``."""
        if self.complex_sample_method == "magnitude-phase":
            return self._step_kind(_capi.STEP_COMPLEX_MAGNITUDE_PHASE, n_sweeps)
        return self._step_kind(_capi.STEP_COMPLEX_GROUP, n_sweeps)

    def measure(self):
        """Update running means, covariances (once measure_step_counter > 50) and observables (:342-427)."""
        self._check(self._lib.me_measure(self._handle))

    def cycle(self, n_sweeps=1):
        """``n_sweeps`` x ``step_all()`` followed by ``measure()`` -- one iteration of the reference's driver loop
        (README.md:41-44) -- as ONE kernel launch where the engine has a fused kernel (``me_cycle``); same results as
        ``step_all(n_sweeps); measure()``."""
        self._check(self._lib.me_cycle(self._handle, int(n_sweeps)))

    def fused_cycles(self):
        """How many :meth:`cycle` calls ran as one launch (the others fell back to a step and a measure launch)."""
        count = ctypes.c_uint64()
        self._check(self._lib.me_cycle_stats(self._handle, ctypes.byref(count)))
        return count.value

    measure_real_system = measure                                                        # :57
    measure_complex_system = measure                                                     # :47

    # ------------------------------------------------------------------ setters (:136-149)
    def set_reject_condition(self, reject_fct):
        """metropolis_engine.py:142-146 (in the reference the only working way to install a constraint, quirk Q6).
        ``reject_fct`` is a :class:`~metropolisengine_amd.energy.RejectSpec`, ``None`` (no constraint) or -- on an engine whose
        energy is a Python callable -- a Python predicate ``(real_params, complex_params) -> bool``: energy and predicate are
        traced into one new plugin and the engine is re-bound to it."""
        if reject_fct is not None and not isinstance(reject_fct, RejectSpec):
            from .pyenergy import PythonEnergy, PythonReject
            if not callable(reject_fct):
                raise TypeError("reject_condition must be a RejectSpec, a callable or None")
            if not isinstance(self._energy_spec, PythonEnergy):
                raise TypeError("a Python reject_condition needs a Python energy (both are traced into one device plugin); "
                                "with a built-in energy use a RejectSpec such as AbsReal0AtLeast")
            self._bind_energy(PythonEnergy(self._energy_spec.energy, reject=reject_fct))
            reject_fct = PythonReject()
        kind = reject_fct.kind if reject_fct is not None else _capi.REJECT_NONE
        bound = reject_fct.bound if reject_fct is not None else 0.0
        self._check(self._lib.me_set_reject_condition(self._handle, int(kind), float(bound)))
        self._reject_spec = reject_fct

    def _bind_energy(self, spec):
        nr, nc = self.num_real_params, self.num_complex_params
        name = None
        if hasattr(spec, "ensure_loaded"):
            spec.ensure_loaded(nr, nc)
            name = spec.name.encode()
        coeffs = np.ascontiguousarray(spec.coefficients(nr, nc), dtype=np.float64)
        self._check(self._lib.me_set_energy(self._handle, int(spec.kind), _as_double_ptr(coeffs), int(coeffs.size), name))
        self._energy_spec = spec
        self._user_name = name
        self.energy_term_names = list(spec.term_names)

    def set_energy_function(self, energy_function):
        """metropolis_engine.py:134-138: replace the energy -- the reference's term dictionary, a single callable, or an
        :class:`~metropolisengine_amd.energy.EnergySpec` -- on the same parameter space.  The term names are collected anew
        and, unlike the reference (which leaves ``self.energy`` stale until ``initialize_energy_dict`` is called), every
        term is re-evaluated at the current state.  State, widths, running statistics and counters stay."""
        if not isinstance(energy_function, EnergySpec):
            if not (callable(energy_function) or isinstance(energy_function, dict)):
                raise TypeError("energy_function must be an EnergySpec, a callable or a dictionary of term callables")
            from .pyenergy import PythonEnergy
            keep = getattr(self._energy_spec, "reject", None) if isinstance(self._reject_spec, RejectSpec) and \
                self._reject_spec.kind == _capi.REJECT_USER else None
            energy_function = PythonEnergy(energy_function, reject=keep)
        self._bind_energy(energy_function)

    def initialize_energy_dict(self):
        """Re-evaluate every term of the energy ledger at the current state (metropolis_engine.py:152-155)."""
        self._check(self._lib.me_recompute_energy(self._handle))

    def set_initial_sampling_width(self, sampling_width):
        self.group_sampling_width = sampling_width                                       # :148-149 (unused there too)

    # ------------------------------------------------------------------ raw field access
    def _get(self, field, chain_begin=0, n_chains=None):
        n = self.n_chains - chain_begin if n_chains is None else n_chains
        comps = ctypes.c_int32()
        self._check(self._lib.me_field_components(self._handle, field, ctypes.byref(comps)))
        out = np.empty((n, comps.value), dtype=np.float64)
        self._check(self._lib.me_get(self._handle, field, chain_begin, n, _as_double_ptr(out)))
        return out

    def _set(self, field, values, chain_begin=0):
        values = np.ascontiguousarray(values, dtype=np.float64)
        self._check(self._lib.me_set(self._handle, field, chain_begin, values.shape[0], _as_double_ptr(values)))

    def _squeeze(self, arr):
        return arr[0] if self.n_chains == 1 else arr

    # ------------------------------------------------------------------ the reference's attributes
    @property
    def real_params(self):
        return self._squeeze(self._get(_capi.FIELD_PARAMS)[:, :self.num_real_params])

    @property
    def complex_params(self):
        nr, nc = self.num_real_params, self.num_complex_params
        x = self._get(_capi.FIELD_PARAMS)
        return self._squeeze(x[:, nr:nr + nc] + 1j * x[:, nr + nc:])

    @property
    def real_mean(self):
        return self._squeeze(self._get(_capi.FIELD_MEAN)[:, :self.num_real_params])

    @property
    def complex_mean(self):
        nr, nc = self.num_real_params, self.num_complex_params
        x = self._get(_capi.FIELD_MEAN)
        return self._squeeze(x[:, nr:nr + nc] + 1j * x[:, nr + nc:])

    @property
    def covariance_matrix_real(self):
        if not self.num_real_params:
            return None
        return self._squeeze(unpack_real_block(self._get(_capi.FIELD_COV), self.num_real_params))

    @property
    def covariance_matrix_complex(self):
        if not self.num_complex_params:
            return None
        return self._squeeze(unpack_complex_block(self._get(_capi.FIELD_COV), self.num_real_params,
                                                  self.num_complex_params))

    @property
    def observables(self):
        nr, nc = self.num_real_params, self.num_complex_params
        x = self._get(_capi.FIELD_PARAMS)
        z = x[:, nr:nr + nc] + 1j * x[:, nr + nc:]
        return self._squeeze(np.concatenate((np.abs(x[:, :nr]), np.abs(z), x[:, :nr] ** 2), axis=1))   # :458-463

    @property
    def observables_mean(self):
        return self._squeeze(self._get(_capi.FIELD_OBS_MEAN))

    def _width(self, row):
        w = self._get(_capi.FIELD_WIDTH)       # [n, 1], or [n, 3] = (sampling_width, real, complex) for mixed engines
        w = w[:, row if w.shape[1] == 3 else 0]
        return float(w[0]) if self.n_chains == 1 else w

    @property
    def sampling_width(self):
        return self._width(0)

    @property
    def real_group_sampling_width(self):
        # pure-complex engines never touch the real width (:449-456)
        return self._width(1) if self.num_real_params else self._initial_widths[0]

    @property
    def complex_group_sampling_width(self):
        return self._width(2) if self.num_complex_params else self._initial_widths[1]

    @property
    def energy_total(self):
        if self.reference_energy_ledgers:                  # the reference's separate attribute (quirk Q5)
            e = self._get(_capi.FIELD_ENERGY_TOTAL)[:, 0]
        else:
            e = self._get(_capi.FIELD_ENERGY).sum(axis=1)     # sum of the ledger's terms (:158-162)
        return float(e[0]) if self.n_chains == 1 else e

    @property
    def energy(self):
        """The energy ledger, ``{term name: value}`` (``self.energy``, :152-155); one entry ``"total"`` unless the
        energy is a term dictionary."""
        rows = self._get(_capi.FIELD_ENERGY)
        return {name: (float(rows[0, t]) if self.n_chains == 1 else rows[:, t])
                for t, name in enumerate(self.energy_term_names)}

    def _counters(self):
        step, meas = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.me_counters(self._handle, ctypes.byref(step), ctypes.byref(meas)))
        return step.value, meas.value

    @property
    def measure_step_counter(self):
        return self._counters()[1]

    @property
    def step_counter(self):
        # starts at 1 and counts width updates (:72, :450); every GPU step updates the width
        return self._counters()[0] + 1

    # ------------------------------------------------------------------ many-chain extras
    def sync(self):
        self._check(self._lib.me_sync(self._handle))

    def accept_stats(self):
        """(accepted, proposed) chain-steps so far (ballot-reduced on the device)."""
        acc, prop = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.me_accept_stats(self._handle, ctypes.byref(acc), ctypes.byref(prop)))
        return acc.value, prop.value

    def acceptance_rate(self):
        acc, prop = self.accept_stats()
        return acc / prop if prop else float("nan")

    def pooled_moments(self):
        """Local ensemble sums (fp64), layout of ``me_pooled_moments`` in include/metropolis_engine.h."""
        size = ctypes.c_int64()
        self._check(self._lib.me_pooled_moments_size(self._handle, ctypes.byref(size)))
        out = np.empty(size.value, dtype=np.float64)
        self._check(self._lib.me_pooled_moments(self._handle, _as_double_ptr(out), size.value))
        return out

    def pooled_moments_begin(self):
        """Enqueue the pooled-moment reduction of the CURRENT state and its copy to the host without waiting; work
        enqueued afterwards overlaps it.  Collect with :meth:`pooled_moments_end`."""
        self._check(self._lib.me_pooled_moments_begin(self._handle))

    def pooled_moments_end(self):
        size = ctypes.c_int64()
        self._check(self._lib.me_pooled_moments_size(self._handle, ctypes.byref(size)))
        out = np.empty(size.value, dtype=np.float64)
        self._check(self._lib.me_pooled_moments_end(self._handle, _as_double_ptr(out), size.value))
        return out

    # -- RCCL behind the C ABI (me_comm_*): the all-reduce of the moments runs on the engine's own streams, no PyTorch
    @staticmethod
    def comm_unique_id():
        """Rank 0: a fresh ``ncclUniqueId`` (bytes) to hand to every rank's :meth:`comm_init`."""
        buf = ctypes.create_string_buffer(_capi.COMM_ID_BYTES)
        _capi.check(_capi.load().me_comm_unique_id(buf, _capi.COMM_ID_BYTES))
        return buf.raw

    def comm_init(self, unique_id, rank, world):
        """Join the RCCL communicator named by ``unique_id`` as ``rank`` of ``world`` (collective over all ranks)."""
        uid = bytes(unique_id)
        if len(uid) != _capi.COMM_ID_BYTES:
            raise ValueError("the unique id is %d bytes" % _capi.COMM_ID_BYTES)
        self._check(self._lib.me_comm_init_rank(self._handle, uid, len(uid), int(rank), int(world)))

    def comm_destroy(self):
        self._check(self._lib.me_comm_destroy(self._handle))

    def comm_info(self):
        """(rank, world, RCCL version code); (-1, 0, 0) without a communicator."""
        rank, world, version = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self._lib.me_comm_info(self._handle, ctypes.byref(rank), ctypes.byref(world), ctypes.byref(version)))
        return rank.value, world.value, version.value

    def pooled_moments_allreduce(self):
        """Ensemble sums over ALL ranks' chains: reduction kernels, ``ncclAllReduce`` and the copy to the host enqueued on
        the engine's streams; waits for the copy only."""
        size = ctypes.c_int64()
        self._check(self._lib.me_pooled_moments_size(self._handle, ctypes.byref(size)))
        out = np.empty(size.value, dtype=np.float64)
        self._check(self._lib.me_pooled_moments_allreduce(self._handle, _as_double_ptr(out), size.value))
        return out

    def pooled_moments_allreduce_begin(self):
        """The same without waiting; collect with :meth:`pooled_moments_end`."""
        self._check(self._lib.me_pooled_moments_allreduce_begin(self._handle))

    def pooled_moments_into(self, device_ptr, n_doubles):
        """Write the local ensemble sums into caller-owned device memory (e.g. a torch CUDA tensor's data_ptr)."""
        self._check(self._lib.me_pooled_moments_device(self._handle, ctypes.c_void_p(device_ptr), int(n_doubles)))

    def shared_factor(self):
        """The packed factor last installed with :meth:`set_shared_factor`, or ``None``."""
        nr, nc = self.num_real_params, self.num_complex_params
        out = np.empty(nr * (nr + 1) // 2 + nc * nc, dtype=np.float64)
        is_set = ctypes.c_int32()
        self._check(self._lib.me_get_shared_factor(self._handle, _as_double_ptr(out), out.size, ctypes.byref(is_set)))
        return out if is_set.value else None

    def set_shared_factor(self, packed_factor):
        f = np.ascontiguousarray(packed_factor, dtype=np.float64)
        self._check(self._lib.me_set_shared_factor(self._handle, _as_double_ptr(f), f.size))

    def time_steps(self, n_launches, n_sweeps=1):
        """Device milliseconds (HIP events on the engine's stream) of ``n_launches`` step launches."""
        ms = ctypes.c_float()
        self._check(self._lib.me_time_steps(self._handle, int(n_launches), int(n_sweeps), ctypes.byref(ms)))
        return ms.value

    def proposal_factors(self):
        """Per-chain Cholesky factors the next proposals use: (real [n,nr,nr], complex [n,nc,nc])."""
        packed = self._get(_capi.FIELD_FACTOR)
        nr, nc = self.num_real_params, self.num_complex_params
        return unpack_real_factor(packed, nr), unpack_complex_block(packed, nr, nc, hermitian=False)

    # ------------------------------------------------------------------ checkpoint (the reference has only the
    # constructor warm start, metropolis_engine.py:17,24)
    _STATE_FIELDS = (("params", _capi.FIELD_PARAMS), ("energy", _capi.FIELD_ENERGY), ("width", _capi.FIELD_WIDTH),
                     ("mean", _capi.FIELD_MEAN), ("obs_mean", _capi.FIELD_OBS_MEAN), ("cov", _capi.FIELD_COV),
                     ("factor", _capi.FIELD_FACTOR), ("energy_total", _capi.FIELD_ENERGY_TOTAL))

    def state_dict(self):
        state = {}
        for name, field in self._STATE_FIELDS:
            try:
                state[name] = self._get(field)
            except NotImplementedError:
                pass
        step, meas = self._counters()
        state["step_index"], state["measure_step_counter"] = step, meas
        state["uses_per_chain_factors"] = bool(meas > 50 and self.cov_mode == "reference")
        state["accepted"], state["proposed"] = self.accept_stats()
        shared = self.shared_factor()
        if shared is not None:                       # cov_mode="pooled": the proposal shape every chain shares
            state["shared_factor"] = shared
        return state

    def load_state_dict(self, state):
        """Restore a :meth:`state_dict`.  Everything is validated against THIS engine before the first device write -- a
        checkpoint of an engine with other fields (e.g. ``reference_energy_ledgers=True``, a tracked covariance, a shared
        factor) or other shapes raises and leaves the engine untouched."""
        todo = []
        for name, field in self._STATE_FIELDS:
            if name in state and (name != "factor" or state.get("uses_per_chain_factors", False)):
                comps = ctypes.c_int32()
                self._check(self._lib.me_field_components(self._handle, field, ctypes.byref(comps)))   # NotImplementedError
                values = np.asarray(state[name], dtype=np.float64)
                if values.shape != (self.n_chains, comps.value):
                    raise ValueError("checkpoint field %r has shape %s, this engine keeps %s"
                                     % (name, values.shape, (self.n_chains, comps.value)))
                todo.append((field, values))
        if "step_index" not in state or "measure_step_counter" not in state:
            raise ValueError("checkpoint lacks the step / measure counters")
        if int(state["measure_step_counter"]) < 1:
            raise ValueError("measure_step_counter starts at 1")
        if "shared_factor" in state:
            nr, nc = self.num_real_params, self.num_complex_params
            if self.cov_mode != "pooled":
                raise ValueError("the checkpoint carries a shared proposal factor: this engine needs cov_mode='pooled'")
            if np.asarray(state["shared_factor"]).shape != (nr * (nr + 1) // 2 + nc * nc,):
                raise ValueError("checkpoint shared_factor has the wrong length")
        if "accepted" in state and "proposed" in state and int(state["accepted"]) > int(state["proposed"]):
            raise ValueError("checkpoint accept counters are inconsistent")
        for field, values in todo:
            self._set(field, values)
        self._check(self._lib.me_set_counters(self._handle, int(state["step_index"]),
                                              int(state["measure_step_counter"])))
        if "shared_factor" in state:
            self.set_shared_factor(state["shared_factor"])
        if "accepted" in state and "proposed" in state:
            self._check(self._lib.me_set_accept_stats(self._handle, int(state["accepted"]), int(state["proposed"])))

    # ------------------------------------------------------------------ time series (:31-35, :350-356, :466-479)
    def trace(self):
        """Recorded series as ``[n_measures, n_traced, D + n_terms + n_widths]``: params, energy terms, widths per
        measure()."""
        rows, cols, k = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self._lib.me_trace_shape(self._handle, ctypes.byref(rows), ctypes.byref(cols), ctypes.byref(k)))
        out = np.empty((rows.value, cols.value, k.value), dtype=np.float64)
        self._check(self._lib.me_trace_get(self._handle, _as_double_ptr(out), out.size))
        return np.ascontiguousarray(out.transpose(0, 2, 1))

    def _series(self, chain=0):
        tr = self.trace()
        if tr.shape[1] <= chain:
            raise ValueError("chain %d is not traced (trace_chains=%d)" % (chain, self.trace_chains))
        tr = tr[:, chain, :]
        nr, nc = self.num_real_params, self.num_complex_params
        d = nr + 2 * nc
        real = tr[:, :nr]
        cplx = tr[:, nr:nr + nc] + 1j * tr[:, nr + nc:d]
        obs = np.concatenate((np.abs(real), np.abs(cplx), real ** 2), axis=1)
        t = len(self.energy_term_names)
        widths = tr[:, d + t:]
        w_real = widths[:, 1 if widths.shape[1] == 3 else 0] if nr else None
        w_cplx = widths[:, 2 if widths.shape[1] == 3 else 0] if nc else None
        return real, cplx, obs, tr[:, d:d + t], w_real, w_cplx

    @property
    def real_params_time_series(self):
        return list(self._series()[0]) if self.num_real_params else None                  # :48

    @property
    def complex_params_time_series(self):
        return list(self._series()[1]) if self.num_complex_params else None               # :58

    @property
    def observables_time_series(self):
        return list(self._series()[2])

    @property
    def energy_time_series(self):
        series = self._series()[3]
        return {name: list(series[:, t]) for t, name in enumerate(self.energy_term_names)}            # :123

    @property
    def real_group_sampling_width_time_series(self):
        return list(self._series()[4]) if self.num_real_params else None

    @property
    def complex_group_sampling_width_time_series(self):
        return list(self._series()[5]) if self.num_complex_params else None

    def time_series_frame(self, chain=0):
        """The reference's DataFrame layout (metropolis_engine.py:466-478; column order of exampledata.csv:1) for one
        traced chain: observables, ``<term>_energy``, real parameters, ``real_group_sampling_width``, complex
        parameters, ``complex_group_sampling_width``."""
        import pandas
        real, cplx, obs, energy, w_real, w_cplx = self._series(chain)
        nr, nc = self.num_real_params, self.num_complex_params
        cols = {}
        for i, name in enumerate(self.observables_names):
            cols[name] = obs[:, i]
        for name in sorted(self.energy_term_names):    # one column per energy term (:468-469; the reference iterates a set)
            cols[name + "_energy"] = energy[:, self.energy_term_names.index(name)]
        if nr:
            for i in range(nr):
                cols[self.params_names[i]] = real[:, i]
            cols["real_group_sampling_width"] = w_real
        if nc:
            for i in range(nc):
                cols[self.params_names[nr + i]] = cplx[:, i]
            cols["complex_group_sampling_width"] = w_cplx
        return pandas.DataFrame.from_dict(cols)

    def save_time_series(self):
        self.df = self.time_series_frame(0)
        print(self.df)                                                                   # :479

    def equilibration_points(self, fast=True, nskip=1):
        """Equilibration start, statistical inefficiency and effective sample count of EVERY traced chain and recorded
        column in one GPU batch (``me_detect_equilibration``): ``{column: (t0[k], g[k], Neff[k])}`` over the ``k`` traced
        chains.  The many-chain form of ``get_equilibration_points`` (statistics.py:25-48); PARITY UNPINNED like it."""
        from . import statistics
        names, rows = [], []
        for chain in range(self.trace_chains):
            frame = self.time_series_frame(chain)
            if chain == 0:
                for name in frame.columns.values:
                    values = frame[name].to_numpy()
                    names += [name + "_real", name + "_imag"] if np.iscomplexobj(values) else [name]
            for name in frame.columns.values:
                values = frame[name].to_numpy()
                rows += [values.real, values.imag] if np.iscomplexobj(values) else [values.astype(np.float64)]
        t0, g, neff = statistics.detect_equilibration_batch(np.stack(rows), fast=fast, nskip=nskip, device=self.device)
        k, c = self.trace_chains, len(names)
        t0, g, neff = t0.reshape(k, c), g.reshape(k, c), neff.reshape(k, c)
        return {name: (t0[:, i], g[:, i], neff[:, i]) for i, name in enumerate(names)}

    def save_equilibrium_stats(self, external_df=None):
        """metropolis_engine.py:481-504: equilibration point per recorded column, the global cut-off (largest ``t0``
        over the columns that are not sampling widths) and the re-averaged means.  PARITY UNPINNED: the reference
        delegates to pymbar, which is absent; see metropolisengine_amd/statistics.py."""
        import pandas
        from . import statistics
        if self.df is None:
            self.save_time_series()
        if external_df is not None:
            frames = [self.df] + [df for df in external_df if isinstance(df.iloc[0, 0], (float, int))]   # :486
            all_df = pandas.concat(frames, axis=1)
        else:
            all_df = self.df
        self.eq_points = statistics.get_equilibration_points(all_df)
        self.global_eq_point = max(t for key, (t, _, _) in self.eq_points.items() if "sampling_width" not in key)
        self.equilibrated_means, self.eq_means_error = statistics.get_equilibrated_means(self.df,
                                                                                         cutoff=self.global_eq_point)
        if external_df is not None:
            profiles = [statistics.get_equilibrated_means(e_df, cutoff=self.global_eq_point)[0] for e_df in external_df]
            self.field_profile = profiles[0]                                             # :501-502
            if len(profiles) > 1:
                self.field_abs_profile = profiles[1]
        print("global t_0", self.global_eq_point)                                        # :503
        self.equilibrated_means["global_cutoff"] = self.global_eq_point
