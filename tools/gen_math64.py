"""Generate metropolisengine_amd/csrc/me_math64_coef.h: the constants of the float64 Box-Muller / accept-rule
arithmetic in me_math64.h (a table-driven log, sin/cos of a fraction of a revolution, exp).

    python tools/gen_math64.py            # rewrites the header
    python tools/gen_math64.py --check    # exits 1 if the committed header differs

Everything is derived here in 60-digit decimal arithmetic (series for pi, ln, sin, cos, exp; polynomial coefficients
by interpolation at Chebyshev nodes, which is within a small factor of minimax) and rounded once to float64.
The accuracy of the resulting functions is measured against long-double libm in tests/test_math64_cpu.py.
"""
import os
import sys
from decimal import Decimal, getcontext

getcontext().prec = 60
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "metropolisengine_amd", "csrc", "me_math64_coef.h")
D = Decimal


def arctan_inv(n):
    """atan(1/n) by its Taylor series."""
    x = D(1) / n
    x2 = x * x
    term, total, k = x, x, 0
    while abs(term) > D(10) ** -58:
        k += 1
        term = -term * x2
        total += term / (2 * k + 1)
    return total


PI = 4 * (4 * arctan_inv(5) - arctan_inv(239))       # Machin


def ln(v):
    """ln v for v > 0: 2 atanh((v-1)/(v+1)) after scaling into [0.5, 2) by powers of two."""
    v = D(v)
    k = 0
    while v >= 2:
        v /= 2
        k += 1
    while v < D("0.5"):
        v *= 2
        k -= 1
    s = (v - 1) / (v + 1)
    s2 = s * s
    term, total, j = s, s, 0
    while abs(term) > D(10) ** -58:
        j += 1
        term *= s2
        total += term / (2 * j + 1)
    return 2 * total + (k * LN2 if k else 0)


def _ln2():
    s = D(1) / 3
    s2 = s * s
    term, total, j = s, s, 0
    while abs(term) > D(10) ** -58:
        j += 1
        term *= s2
        total += term / (2 * j + 1)
    return 2 * total


LN2 = _ln2()


def cos_dec(t):
    t = D(t)
    term, total, k = D(1), D(1), 0
    while abs(term) > D(10) ** -58:
        k += 1
        term = -term * t * t / ((2 * k - 1) * (2 * k))
        total += term
    return total


def cheb_fit(f, lo, hi, degree):
    """Monomial coefficients (in the ORIGINAL variable) of the degree-`degree` interpolant of f at the Chebyshev
    nodes of [lo, hi]."""
    n = degree + 1
    lo, hi = D(lo), D(hi)
    nodes = [cos_dec(PI * (2 * j + 1) / (2 * n)) for j in range(n)]
    vals = [f((hi + lo) / 2 + (hi - lo) / 2 * t) for t in nodes]
    # Chebyshev coefficients c_k = (2 - [k=0]) / n * sum_j f_j T_k(t_j)
    cheb = []
    for k in range(n):
        s = D(0)
        for j in range(n):
            s += vals[j] * cos_dec(PI * k * (2 * j + 1) / (2 * n))
        cheb.append(s * (1 if k == 0 else 2) / n)
    # to monomials in t (T_0 = 1, T_1 = t, T_{k+1} = 2 t T_k - T_{k-1})
    t_prev, t_cur = [D(1)], [D(0), D(1)]
    mono = [D(0)] * n
    for k in range(n):
        poly = t_prev if k == 0 else t_cur
        for i, c in enumerate(poly):
            mono[i] += cheb[k] * c
        if k >= 1:
            nxt = [D(0)] + [2 * c for c in t_cur]
            for i, c in enumerate(t_prev):
                nxt[i] -= c
            t_prev, t_cur = t_cur, nxt
    # substitute t = (v - mid) / half
    mid, half = (hi + lo) / 2, (hi - lo) / 2
    out = [D(0)] * n
    basis = [D(1)]                       # ((v - mid)/half)^i as a polynomial in v
    for i in range(n):
        for p, c in enumerate(basis):
            out[p] += mono[i] * c
        nxt = [D(0)] * (len(basis) + 1)
        for p, c in enumerate(basis):
            nxt[p + 1] += c / half
            nxt[p] -= c * mid / half
        basis = nxt
    return out


def sin_rev_over_p(z):
    """sin(2 pi p) / p as a function of z = p^2 (entire in z)."""
    a = 2 * PI
    term, total, k = a, a, 0
    while abs(term) > D(10) ** -58:
        k += 1
        term = -term * a * a * z / ((2 * k) * (2 * k + 1))
        total += term
    return total


def cos_rev(z):
    """cos(2 pi p) as a function of z = p^2."""
    a2 = 4 * PI * PI
    term, total, k = D(1), D(1), 0
    while abs(term) > D(10) ** -58:
        k += 1
        term = -term * a2 * z / ((2 * k - 1) * (2 * k))
        total += term
    return total


def log1p_tail(r):
    """(log(1 + r) - r) / r^2 = -1/2 + r/3 - r^2/4 ..."""
    r = D(r)
    total, power, k = D(0), D(1), 2
    while True:
        term = power / k * (1 if k % 2 else -1)
        total += term
        if abs(term) < D(10) ** -58:
            return total
        power *= r
        k += 1


def exp_tail(f):
    """(e^f - 1 - f) / f^2 = 1/2 + f/6 + ..."""
    f = D(f)
    total, term, k = D(0), D(1) / 2, 2
    while abs(term) > D(10) ** -58:
        total += term
        k += 1
        term = term * f / k
    return total


def c_double(v):
    return float(v).hex()


LOG_TABLE_BITS = 7                      # c = j / 128, j = 96 .. 192  (z in [0.75, 1.5))
LOG_J0, LOG_J1 = 96, 192
SIN_DEGREE = 6                          # in z = p^2, |p| <= 1/8
COS_DEGREE = 7
LOG_DEGREE = 5                          # tail polynomial in r, |r| <= 1/192
EXP_DEGREE = 10                         # tail polynomial in f, |f| <= ln2/2


def generate():
    lines = ["// GENERATED by tools/gen_math64.py -- do not edit; regenerate and commit.",
             "// Constants of me_math64.h: every value was computed in 60-digit decimal arithmetic and rounded once.",
             "#pragma once", "", "namespace me {", "namespace math64 {", ""]
    lines.append("constexpr int kLogJ0 = %d, kLogEntries = %d;   // table row j - kLogJ0 holds c = j / %d"
                 % (LOG_J0, LOG_J1 - LOG_J0 + 1, 1 << LOG_TABLE_BITS))
    lines.append("// {1/c rounded to float64, 2 ln(that rounded value)}: -2 ln z = T1 - 2 log1p(z T0 - 1), exactly")
    lines.append("constexpr double kLogTable[kLogEntries][2] = {")
    for j in range(LOG_J0, LOG_J1 + 1):
        invc = float(D(1 << LOG_TABLE_BITS) / D(j))
        two_ln = 2 * ln(D(invc)) if j != (1 << LOG_TABLE_BITS) else D(0)
        lines.append("    {%s, %s}," % (c_double(D(invc)), c_double(two_ln)))
    lines.append("};")
    rmax = D(1) / (2 * LOG_J0)           # |z/c - 1| <= 1/(2 * 128 * 0.75) plus rounding slack
    rmax = rmax * D("1.001")
    q = cheb_fit(log1p_tail, -rmax, rmax, LOG_DEGREE)
    lines.append("// -2 (log1p(r) - r) / r^2 on |r| <= %s, degree %d" % (float(rmax), LOG_DEGREE))
    lines.append("constexpr double kLogTail[%d] = {%s};" % (len(q), ", ".join(c_double(-2 * c) for c in q)))
    lines.append("constexpr double kMinusTwoLn2 = %s;" % c_double(-2 * LN2))
    zmax = D(1) / 64
    s = cheb_fit(sin_rev_over_p, D(0), zmax, SIN_DEGREE)
    c = cheb_fit(cos_rev, D(0), zmax, COS_DEGREE)
    lines.append("// sin(2 pi p) = p S(p^2), cos(2 pi p) = C(p^2) on |p| <= 1/8")
    lines.append("constexpr double kSinRev[%d] = {%s};" % (len(s), ", ".join(c_double(v) for v in s)))
    lines.append("constexpr double kCosRev[%d] = {%s};" % (len(c), ", ".join(c_double(v) for v in c)))
    fmax = LN2 / 2 * D("1.0001")
    e = cheb_fit(exp_tail, -fmax, fmax, EXP_DEGREE)
    lines.append("// (e^f - 1 - f) / f^2 on |f| <= ln2/2, degree %d" % EXP_DEGREE)
    lines.append("constexpr double kExpTail[%d] = {%s};" % (len(e), ", ".join(c_double(v) for v in e)))
    ln2_hi = float(LN2)
    # hi part with 21 trailing zero bits so that n * hi is exact for |n| < 2^11
    import struct
    bits = struct.unpack("<Q", struct.pack("<d", ln2_hi))[0] & ~((1 << 21) - 1)
    ln2_hi = struct.unpack("<d", struct.pack("<Q", bits))[0]
    lines.append("constexpr double kLn2Hi = %s, kLn2Lo = %s, kLog2e = %s;"
                 % (c_double(D(ln2_hi)), c_double(LN2 - D(ln2_hi)), c_double(1 / LN2)))
    lines += ["", "}  // namespace math64", "}  // namespace me", ""]
    return "\n".join(lines)


if __name__ == "__main__":
    text = generate()
    if "--check" in sys.argv:
        with open(OUT) as fh:
            sys.exit(0 if fh.read() == text else 1)
    with open(OUT, "w") as fh:
        fh.write(text)
    print("wrote", OUT)
