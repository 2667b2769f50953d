"""Post-hoc equilibration statistics -- host side, off the accelerated hot path.

Mirrors ``/root/reference/metropolisengine/statistics.py:25-64`` (``get_equilibration_points``,
``get_equilibrated_means``) and ``MetropolisEngine.save_equilibrium_stats`` (metropolis_engine.py:481-504).

PARITY UNPINNED.  The arithmetic of the reference lives in ``pymbar.timeseries.detectEquilibration`` (statistics.py:4,
:41-46): a third-party dependency that is not vendored, not version-pinned (setup.py:22 misspells
``install_requires``) and not installed offline, and the reference holds no output of it.  The two functions below
restate the published algorithm (J. D. Chodera, "A simple method for automated equilibration detection in molecular
simulations", JCTC 12:1799, 2016; statistical inefficiency after Chodera et al., JCTC 3:26, 2007, as implemented in
pymbar 3.x ``timeseries.statisticalInefficiency`` / ``detectEquilibration``) and are tested against analytic
AR(1) autocorrelation times only.
"""
import numpy as np


def statistical_inefficiency(series, mintime=3, fast=False):
    """``g = 1 + 2 tau``: sum the normalised autocorrelation function until it first turns non-positive (after
    ``mintime`` lags); ``fast`` lengthens the lag increment by one at every step."""
    a = np.asarray(series, dtype=np.float64)
    n = a.size
    da = a - a.mean()
    sigma2 = np.mean(da * da)
    if sigma2 == 0:
        raise ValueError("sample covariance is zero: cannot compute the statistical inefficiency")
    g = 1.0
    t = 1
    increment = 1
    while t < n - 1:
        c = np.sum(da[:n - t] * da[t:]) / ((n - t) * sigma2)
        if c <= 0.0 and t > mintime:
            break
        g += 2.0 * c * (1.0 - t / n) * increment
        t += increment
        if fast:
            increment += 1
    return max(g, 1.0)


def detect_equilibration(series, fast=True, nskip=1):
    """``(t0, g, Neff_max)``: the start ``t0`` of the production region that maximises the number of effectively
    uncorrelated samples ``(T - t0 + 1) / g(t0)``."""
    a = np.asarray(series, dtype=np.float64)
    big_t = a.size
    if a.std() == 0.0:
        return 0, 1.0, 1.0
    g_t = np.ones(big_t - 1)
    neff_t = np.ones(big_t - 1)
    for t in range(0, big_t - 1, nskip):
        try:
            g_t[t] = statistical_inefficiency(a[t:], fast=fast)
        except ValueError:
            g_t[t] = big_t - t + 1
        neff_t[t] = (big_t - t + 1) / g_t[t]
    t0 = int(np.argmax(neff_t))
    return t0, float(g_t[t0]), float(neff_t[t0])


def detect_equilibration_batch(series, fast=True, nskip=1, device=0):
    """``detect_equilibration`` for every row of ``series[n_series, T]`` in one call on the GPU
    (``me_detect_equilibration``: one wavefront per (series, start)); returns ``(t0, g, Neff_max)`` arrays.  An
    ensemble run records thousands of chains x columns, each an O(T^2) scan on the host."""
    import ctypes
    from . import _capi
    a = np.ascontiguousarray(series, dtype=np.float64)
    if a.ndim != 2:
        raise ValueError("series must be [n_series, length]")
    n, length = a.shape
    t0 = np.zeros(n, dtype=np.int64)
    g = np.ones(n)
    neff = np.ones(n)
    dp = ctypes.POINTER(ctypes.c_double)
    _capi.check(_capi.load().me_detect_equilibration(
        int(device), a.ctypes.data_as(dp), n, length, int(bool(fast)), int(nskip),
        t0.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), g.ctypes.data_as(dp), neff.ctypes.data_as(dp)))
    return t0, g, neff


def get_equilibration_points(df, device=None):
    """Per column ``[t0, g, Neff_max]``; constant columns are skipped and complex columns split into ``_real`` /
    ``_imag`` (statistics.py:25-48).  With ``device`` set, all columns go to the GPU in one batch."""
    names, rows = [], []
    for name in df.columns.values:
        column = df.loc[:, name]
        if column.nunique() <= 1:
            continue
        values = column.to_numpy()
        if np.iscomplexobj(values):
            names += [name + "_real", name + "_imag"]
            rows += [values.real, values.imag]
        else:
            names.append(name)
            rows.append(values.astype(np.float64))
    if device is not None and rows:
        t0, g, neff = detect_equilibration_batch(np.stack(rows), device=device)
        return {name: [int(t0[i]), float(g[i]), float(neff[i])] for i, name in enumerate(names)}
    return {name: list(detect_equilibration(row)) for name, row in zip(names, rows)}


def timeseries_from_csv(file_name, column_name=None):
    """Read a time-series file the reference writes (``df.to_csv``; layout of exampledata.csv: an index column, then
    observables, ``<term>_energy``, parameters and widths, complex values as ``(re+imj)`` strings) into
    ``{name: float array}``.  Like ``plottable_timeseries_from_csv`` (statistics.py:6-23) a non-float column becomes
    ``<name>_real`` plus, when its first entry has a non-zero imaginary part, ``<name>_imag``."""
    import pandas
    data = pandas.read_csv(file_name, index_col=0)
    names = data.columns if column_name is None else [column_name]
    out = {}
    for name in names:
        column = data[name]
        if column.dtype.kind == "f":
            out[name] = column.to_numpy(dtype=np.float64)
            continue
        values = np.array([complex(str(v).replace(" ", "")) for v in column], dtype=np.complex128)
        if values[0].imag != 0:
            out[name + "_imag"] = values.imag.copy()
        out[name + "_real"] = values.real.copy()
    return out


def get_equilibrated_means(df, cutoff=None):
    """Column means from row ``cutoff`` on (statistics.py:53-64; the reference's ``cutoff=None`` branch calls an
    undefined name, here it takes the largest ``t0``).  Returns ``(means, errors)``; like the reference, ``errors``
    stays empty."""
    if cutoff is None:
        points = get_equilibration_points(df)
        cutoff = max(t for t, _, _ in points.values()) if points else 0
    means = {name: np.average(df.loc[cutoff:, name]) for name in df.columns.values}
    return means, {}
