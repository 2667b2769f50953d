"""ctypes binding of libmetropolis_hip.so (C ABI: include/metropolis_engine.h).

There is deliberately no CPU fallback: if the HIP library is missing or no GPU is present, the product fails loudly.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("METROPOLIS_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libmetropolis_hip.so")
ABI_VERSION = 1

ME_OK, ME_ERR_INVALID, ME_ERR_UNSUPPORTED, ME_ERR_HIP, ME_ERR_NUMERIC, ME_ERR_STATE = range(6)
ME_F32, ME_F64 = 0, 1
(ENERGY_ISO_QUAD, ENERGY_DIAG_QUAD, ENERGY_DENSE_QUAD, ENERGY_LANDAU_TOY, ENERGY_CYLINDER, ENERGY_USER,
 ENERGY_USER_INDIRECT, ENERGY_LANDAU_TERMS) = range(8)
FLAG_TRACK_COVARIANCE = 1
FLAG_REFERENCE_ENERGY_LEDGERS = 2
REJECT_NONE, REJECT_ABS_REAL0_GE, REJECT_USER = 0, 1, 2
STEP_ALL, STEP_REAL_GROUP, STEP_COMPLEX_GROUP, STEP_COMPLEX_MAGNITUDE_PHASE = range(4)
COV_REFERENCE, COV_FIXED, COV_POOLED = 0, 1, 2
COMM_ID_BYTES = 128
(FIELD_PARAMS, FIELD_ENERGY, FIELD_WIDTH, FIELD_MEAN, FIELD_COV, FIELD_OBS_MEAN, FIELD_FACTOR, FIELD_ENERGY_TOTAL) = range(8)

_dp = ctypes.POINTER(ctypes.c_double)


class MeConfig(ctypes.Structure):
    """struct me_config (include/metropolis_engine.h)."""
    _fields_ = [
        ("abi_version", ctypes.c_uint32), ("device_id", ctypes.c_int32),
        ("n_chains", ctypes.c_int64), ("chain_offset", ctypes.c_uint64), ("seed", ctypes.c_uint64),
        ("n_real", ctypes.c_int32), ("n_complex", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("cov_mode", ctypes.c_int32),
        ("temp", ctypes.c_double), ("target_acceptance", ctypes.c_double), ("sampling_width", ctypes.c_double),
        ("energy_kind", ctypes.c_int32), ("n_energy_coeffs", ctypes.c_int32), ("energy_coeffs", _dp),
        ("reject_kind", ctypes.c_int32), ("flags", ctypes.c_int32), ("reject_bound", ctypes.c_double),
        ("initial_params", _dp), ("covariance_real", _dp), ("covariance_complex", _dp),
        ("user_energy_name", ctypes.c_char_p),
    ]


# every symbol the header declares: name -> (restype, argtypes)
_H = ctypes.c_void_p
SYMBOLS = {
    "me_abi_version": (ctypes.c_int, []),
    "me_load_plugin": (ctypes.c_int, [ctypes.c_char_p]),
    "me_create": (ctypes.c_int, [ctypes.POINTER(MeConfig), ctypes.POINTER(_H)]),
    "me_destroy": (ctypes.c_int, [_H]),
    "me_step": (ctypes.c_int, [_H, ctypes.c_int32]),
    "me_measure": (ctypes.c_int, [_H]),
    "me_cycle": (ctypes.c_int, [_H, ctypes.c_int32]),
    "me_cycle_stats": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_uint64)]),
    "me_step_kind": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32]),
    "me_set_reject_condition": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_double]),
    "me_set_energy": (ctypes.c_int, [_H, ctypes.c_int32, _dp, ctypes.c_int32, ctypes.c_char_p]),
    "me_step_injected": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32, _dp, _dp]),
    "me_field_components": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]),
    "me_energy_terms": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32)]),
    "me_pooled_moments_begin": (ctypes.c_int, [_H]),
    "me_pooled_moments_end": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_double), ctypes.c_int64]),
    "me_detect_equilibration": (ctypes.c_int, [ctypes.c_int32, ctypes.POINTER(ctypes.c_double), ctypes.c_int64,
                                               ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                               ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_double),
                                               ctypes.POINTER(ctypes.c_double)]),
    "me_get": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, _dp]),
    "me_set": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, _dp]),
    "me_recompute_energy": (ctypes.c_int, [_H]),
    "me_constants": (ctypes.c_int, [_H, _dp, ctypes.POINTER(ctypes.c_int32), _dp]),
    "me_counters": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "me_set_counters": (ctypes.c_int, [_H, ctypes.c_uint64, ctypes.c_uint64]),
    "me_accept_stats": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "me_pooled_moments_size": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int64)]),
    "me_pooled_moments": (ctypes.c_int, [_H, _dp, ctypes.c_int64]),
    "me_pooled_moments_device": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_int64]),
    "me_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t]),
    "me_comm_init_rank": (ctypes.c_int, [_H, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32, ctypes.c_int32]),
    "me_comm_destroy": (ctypes.c_int, [_H]),
    "me_comm_info": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32),
                                    ctypes.POINTER(ctypes.c_int32)]),
    "me_pooled_moments_allreduce": (ctypes.c_int, [_H, _dp, ctypes.c_int64]),
    "me_pooled_moments_allreduce_begin": (ctypes.c_int, [_H]),
    "me_set_shared_factor": (ctypes.c_int, [_H, _dp, ctypes.c_int64]),
    "me_get_shared_factor": (ctypes.c_int, [_H, _dp, ctypes.c_int64, ctypes.POINTER(ctypes.c_int32)]),
    "me_set_accept_stats": (ctypes.c_int, [_H, ctypes.c_uint64, ctypes.c_uint64]),
    "me_set_cache_budget": (ctypes.c_int, [ctypes.c_int64]),
    "me_trace_enable": (ctypes.c_int, [_H, ctypes.c_int64, ctypes.c_int64]),
    "me_trace_shape": (ctypes.c_int, [_H, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64),
                                      ctypes.POINTER(ctypes.c_int64)]),
    "me_trace_get": (ctypes.c_int, [_H, _dp, ctypes.c_int64]),
    "me_sync": (ctypes.c_int, [_H]),
    "me_set_stream": (ctypes.c_int, [_H, ctypes.c_void_p]),
    "me_time_steps": (ctypes.c_int, [_H, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float)]),
    "me_last_error": (ctypes.c_int, [_H, ctypes.c_char_p, ctypes.c_size_t]),
    "me_supported": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
}


class MetropolisLibraryError(RuntimeError):
    """The HIP library is missing or reported a failure (status code in ``.status``)."""

    def __init__(self, message, status=None):
        super().__init__(message)
        self.status = status


_lib = None


def load():
    """Load libmetropolis_hip.so (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MetropolisLibraryError(
            "libmetropolis_hip.so not found at %s: build it with `python -m metropolisengine_amd.build` "
            "(needs hipcc). metropolisengine_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)   # plugins resolve me::register_kernel_set against it
    except OSError as exc:
        raise MetropolisLibraryError("could not load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.me_abi_version() != ABI_VERSION:
        raise MetropolisLibraryError("ABI mismatch: library %d, binding %d" % (lib.me_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def last_error(handle=None):
    buf = ctypes.create_string_buffer(1024)
    load().me_last_error(handle, buf, len(buf))
    return buf.value.decode("utf-8", "replace")


_EXC = {ME_ERR_INVALID: ValueError, ME_ERR_UNSUPPORTED: NotImplementedError}


def check(status, handle=None):
    """Map a non-zero me_status to the exception the reference would raise where it has one."""
    if status == ME_OK:
        return
    message = last_error(handle) or "me_status %d" % status
    exc = _EXC.get(status)
    if exc is not None:
        raise exc(message)
    if status == ME_ERR_NUMERIC:
        raise FloatingPointError(message)
    raise MetropolisLibraryError(message, status)
