#!/usr/bin/env python3
"""Derives the constants of include/mm_math.h from first principles (80-digit Decimal arithmetic):
split pi/2 and ln 2, atan breakpoints, Taylor coefficients.  Prints C hex-float definitions."""
from decimal import Decimal, getcontext
from fractions import Fraction
import math

getcontext().prec = 90


def d_atan(x):  # Taylor/Euler series in Decimal, |x| <= 1
    x = Decimal(x)
    if x == 1:
        return 4 * d_atan(Decimal(1) / 5) - d_atan(Decimal(1) / 239)
    y = x / (1 + (1 + x * x).sqrt())  # halve the angle twice for fast convergence
    y = y / (1 + (1 + y * y).sqrt())
    s, t, k = Decimal(0), y, 0
    while abs(t) > Decimal(10) ** -85:
        s += t / (2 * k + 1) if k % 2 == 0 else -t / (2 * k + 1)
        t *= y * y
        k += 1
    return 4 * s


PI = 4 * d_atan(1)
LN2 = Decimal(2).ln()


def to_double(d):
    return float(d)  # Decimal -> nearest double


def trunc_bits(d, bits):
    """d rounded to `bits` significant bits (returned as an exact float)."""
    f = Fraction(d)
    e = math.floor(math.log2(f))
    scale = Fraction(2) ** (bits - 1 - e)
    return float(Fraction(round(f * scale)) / scale)


def hilo(d, bits=None):
    hi = to_double(d) if bits is None else trunc_bits(d, bits)
    lo = to_double(d - Decimal(hi))
    return hi, lo


def emit(name, v):
    print("#define %s %s /* %.17g */" % (name, float(v).hex(), float(v)))


pio2 = PI / 2
h, l = hilo(pio2, 33)
emit("MMM_PIO2_1", h); emit("MMM_PIO2_1T", l)
h, l = hilo(pio2)
emit("MMM_PIO2_HI", h); emit("MMM_PIO2_LO", l)
emit("MMM_TWO_OVER_PI", to_double(2 / PI))
h, l = hilo(LN2, 32)
emit("MMM_LN2_HI", h); emit("MMM_LN2_LO", l)
emit("MMM_INV_LN2", to_double(1 / LN2))
emit("MMM_SQRT_HALF", to_double(Decimal(0.5).sqrt()))
for j, c in enumerate(("0.25", "0.5", "0.75", "1")):
    h, l = hilo(d_atan(Decimal(c)))
    emit("MMM_ATAN_HI_%d" % (j + 1), h); emit("MMM_ATAN_LO_%d" % (j + 1), l)
for k in range(1, 8):  # sin: (-1)^k / (2k+1)!
    emit("MMM_S%d" % k, to_double(Decimal((-1) ** k) / Decimal(math.factorial(2 * k + 1))))
for k in range(2, 9):  # cos: (-1)^k / (2k)!
    emit("MMM_C%d" % k, to_double(Decimal((-1) ** k) / Decimal(math.factorial(2 * k))))
for k in range(1, 11):  # atan: (-1)^k / (2k+1)
    emit("MMM_A%d" % k, to_double(Decimal((-1) ** k) / Decimal(2 * k + 1)))
for k in range(2, 15):  # exp: 1/k!
    emit("MMM_E%d" % k, to_double(Decimal(1) / Decimal(math.factorial(k))))
for k in range(1, 12):  # log: 1/(2k+1)
    emit("MMM_L%d" % k, to_double(Decimal(1) / Decimal(2 * k + 1)))


def feastol_sq():
    """MM_QP_FEASTOL_SQ of include/mm_qp.h: the largest double v with RN(sqrt(v)) <= 1e-7 (math.sqrt is correctly rounded)."""
    import math
    F = 1e-7
    v = F * F
    while math.sqrt(v) <= F:
        v = math.nextafter(v, math.inf)
    while math.sqrt(v) > F:
        v = math.nextafter(v, -math.inf)
    assert math.sqrt(v) <= F < math.sqrt(math.nextafter(v, math.inf))
    return v


if __name__ == "__main__":
    print("MM_QP_FEASTOL_SQ", feastol_sq().hex())
