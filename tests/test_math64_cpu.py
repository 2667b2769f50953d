"""The float64 step arithmetic of the kernels (metropolisengine_amd/csrc/me_math64.h: table-driven -2 ln u,
bounded sqrt, sin/cos of a word as a fraction of a revolution, exp of a non-positive argument) compiled for the HOST
and measured against long-double libm.  The same header is what k_step<double> compiles for gfx950; the device-only
differences are v_rsq_f64 (emulated here by a noisy estimate), v_ldexp_f64 and v_rndne_f64.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "math64_host.cpp")
LD = np.longdouble


def _build(tmp, noise):
    out = os.path.join(tmp, "libmath64_%d.so" % noise)
    cmd = ["g++", "-O2", "-march=native", "-std=c++17", "-shared", "-fPIC", SRC, "-o", out]
    if noise:
        cmd.insert(1, "-DME_MATH64_TEST_RSQ_NOISE")
    subprocess.run(cmd, check=True)
    lib = ctypes.CDLL(out)
    return lib


@pytest.fixture(scope="module", params=[0, 1], ids=["exact_rsq", "noisy_rsq"])
def lib(request, tmp_path_factory):
    return _build(str(tmp_path_factory.mktemp("math64")), request.param)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _words(n, seed):
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    edge = np.array([0, 1, 2, 3, 0xFFFFFFFF, 0xFFFFFFFE, 0x7FFFFFFF, 0x80000000, 0x80000001, 0x3FFFFFFF, 0x40000000,
                     0x40000001, 0xBFFFFFFF, 0xC0000000, 0xC0000001, 0x1FFFFFFF, 0x20000000, 0x5FFFFFFF, 0x60000000,
                     0x9FFFFFFF, 0xA0000000, 0xDFFFFFFF, 0xE0000000, 0xBFFFFFFE, 0xC0000002, 0x00010000, 0xFFFF0000],
                    dtype=np.uint32)
    # words whose (w + 0.5) mantissa sits at the table's interval edges and at the [0.75, 1.5) wrap
    k = np.arange(96, 193, dtype=np.uint64)
    for shift in (24, 16, 8):
        edge = np.concatenate([edge, ((k << shift) >> 7).astype(np.uint32), (((k << shift) >> 7) - 1).astype(np.uint32)])
    return np.concatenate([edge, w])


def _ulp_err(got, ref):
    """|got - ref| in units of the float64 ulp of ref (ref long double)."""
    ref64 = ref.astype(np.float64)
    ulp = np.spacing(np.abs(ref64)).astype(LD)
    return np.abs(got.astype(LD) - ref) / ulp


def _reference_cos_sin(w):
    """cos, sin of 2 pi (w + 0.5) / 2^32 in long double with the quarter-revolution reduction done exactly."""
    wi = w.astype(np.int64)
    q = (wi + (1 << 29)) >> 30
    rem = wi - (q << 30)
    p = (rem.astype(LD) + LD(0.5)) / LD(2 ** 32)
    two_pi = LD(2) * LD("3.14159265358979323846264338327950288")
    s, c = np.sin(two_pi * p), np.cos(two_pi * p)
    qq = q & 3
    cs = np.where(qq == 0, c, np.where(qq == 1, -s, np.where(qq == 2, -c, s)))
    sn = np.where(qq == 0, s, np.where(qq == 1, c, np.where(qq == 2, -s, -c)))
    return cs, sn


def test_radius(lib):
    w = _words(2_000_000, 1)
    y = np.empty(w.size)
    r = np.empty(w.size)
    lib.me_math64_radius(_ptr(w), ctypes.c_long(w.size), _ptr(y), _ptr(r))
    u = (w.astype(LD) + LD(0.5)) / LD(2 ** 32)
    y_ref = -2 * np.log(u)
    assert _ulp_err(y, y_ref).max() <= 2.0
    assert _ulp_err(r, np.sqrt(y_ref)).max() <= 1.5
    # u -> 1: the relative accuracy of ln survives (table row c = 1 is exact)
    top = w == 0xFFFFFFFF
    assert abs(float(y[top][0]) / float(y_ref[top][0]) - 1) < 3e-16


def test_cos_sin(lib):
    w = _words(2_000_000, 2)
    cs = np.empty(w.size)
    sn = np.empty(w.size)
    lib.me_math64_cos_sin(_ptr(w), ctypes.c_long(w.size), _ptr(cs), _ptr(sn))
    cs_ref, sn_ref = _reference_cos_sin(w)
    assert _ulp_err(cs, cs_ref).max() <= 2.0      # relative, also next to the zeros of cos / sin
    assert _ulp_err(sn, sn_ref).max() <= 2.0
    # against the oracle's formula (float64 libm on the rounded angle 2 pi u): absolute agreement
    u = (w.astype(np.float64) + 0.5) / 2.0 ** 32
    assert np.abs(cs - np.cos(2 * np.pi * u)).max() < 1.5e-15
    assert np.abs(sn - np.sin(2 * np.pi * u)).max() < 1.5e-15


def test_normal_pair_matches_oracle_formula(lib):
    wa, wb = _words(1_000_000, 3), _words(1_000_000, 4)
    g0 = np.empty(wa.size)
    g1 = np.empty(wa.size)
    lib.me_math64_normals(_ptr(wa), _ptr(wb), ctypes.c_long(wa.size), _ptr(g0), _ptr(g1))
    u1 = (wa.astype(np.float64) + 0.5) / 2.0 ** 32
    u2 = (wb.astype(np.float64) + 0.5) / 2.0 ** 32
    r = np.sqrt(-2.0 * np.log(u1))                 # oracle/philox.py: step_draws
    assert np.abs(g0 - r * np.cos(2 * np.pi * u2)).max() < 2e-14
    assert np.abs(g1 - r * np.sin(2 * np.pi * u2)).max() < 2e-14
    cs_ref, sn_ref = _reference_cos_sin(wb)
    r_ref = np.sqrt(-2 * np.log((wa.astype(LD) + LD(0.5)) / LD(2 ** 32)))
    assert _ulp_err(g0, r_ref * cs_ref).max() <= 4.0
    assert _ulp_err(g1, r_ref * sn_ref).max() <= 4.0
    a, b = g0[-1_000_000:], g1[-1_000_000:]          # the random words only (the edge words come first)
    assert abs(a.mean()) < 5e-3 and abs(a.var() - 1) < 5e-3 and abs((a * b).mean()) < 5e-3


def test_exp(lib):
    rng = np.random.default_rng(5)
    x = np.concatenate([-rng.exponential(2.0, 1_000_000), -rng.uniform(0, 750, 200_000),
                        -np.array([0.0, 1e-300, 1e-17, 1e-9, 0.5 * np.log(2), np.log(2), 700.0, 708.4, 745.0, 746.0, 1e3,
                                   np.inf]), -2.0 ** np.arange(-60, 10)])
    out = np.empty(x.size)
    lib.me_math64_exp(_ptr(x), ctypes.c_long(x.size), _ptr(out))
    ref = np.exp(x.astype(LD))
    normal = ref > LD(2.3e-308)
    assert _ulp_err(out[normal], ref[normal]).max() <= 1.5
    assert np.abs(out[~normal] - ref[~normal].astype(np.float64)).max() <= 5e-324 * 2
    assert out[x == 0][0] == 1.0 and out[np.isinf(x)][0] == 0.0
    nan = np.array([np.nan])
    lib.me_math64_exp(_ptr(nan), ctypes.c_long(1), _ptr(out))
    assert np.isnan(out[0])


def test_generated_header_is_current():
    assert subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_math64.py"), "--check"]).returncode == 0
