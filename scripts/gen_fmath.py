#!/usr/bin/env python3
"""Constants of cice5_amd/csrc/evpk_fmath.h, derived with exact rational arithmetic (no libm): the three-part split of pi/2
for the argument reduction of sin / cos, pi and pi/2 as hi + lo pairs, atan(k/16) as hi + lo pairs, Taylor coefficients.

    python scripts/gen_fmath.py          # prints the C initialisers
"""
from fractions import Fraction as F
import math
import struct

PI_DIGITS = "3.14159265358979323846264338327950288419716939937510582097494459230781640628620899862803482534211706798"
PI = F(PI_DIGITS)


def to_double(fr):
    """nearest double of a rational (round half to even) via integer arithmetic"""
    if fr == 0:
        return 0.0
    sgn = -1 if fr < 0 else 1
    fr = abs(fr)
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    # scale to 53 bits
    sh = 53 - e
    q = fr * (F(2) ** sh)
    n = q.numerator // q.denominator
    while n >= (1 << 53):
        sh -= 1; q = fr * (F(2) ** sh); n = q.numerator // q.denominator
    while n < (1 << 52):
        sh += 1; q = fr * (F(2) ** sh); n = q.numerator // q.denominator
    rem = q - n
    if rem > F(1, 2) or (rem == F(1, 2) and (n & 1)):
        n += 1
    return sgn * math.ldexp(n, -sh)


def trunc_bits(fr, bits):
    """fr truncated to `bits` significant bits (so that small integer multiples are exact in double)"""
    e = fr.numerator.bit_length() - fr.denominator.bit_length()
    sh = bits - e
    q = fr * (F(2) ** sh)
    n = q.numerator // q.denominator
    while n >= (1 << bits):
        sh -= 1; q = fr * (F(2) ** sh); n = q.numerator // q.denominator
    while n < (1 << (bits - 1)):
        sh += 1; q = fr * (F(2) ** sh); n = q.numerator // q.denominator
    return F(n) / (F(2) ** sh)


def atan_rat(x, terms=400):
    """atan of a rational 0 <= x <= 1 by its Taylor series at a point where it converges fast enough"""
    if x > F(1, 2):           # atan(x) = pi/4 - atan((1-x)/(1+x))
        return PI / 4 - atan_rat((1 - x) / (1 + x), terms)
    s, p = F(0), x
    for k in range(terms):
        s += (-1) ** k * p / (2 * k + 1)
        p = p * x * x
        if p.numerator.bit_length() - p.denominator.bit_length() < -400:
            break
        # keep the fractions from exploding
        p = F(p.numerator >> max(0, p.numerator.bit_length() - 600), p.denominator >> max(0, p.numerator.bit_length() - 600)) if p.numerator.bit_length() > 1200 else p
    return s


def hilo(fr):
    hi = to_double(fr)
    lo = to_double(fr - F(hi))
    return hi, lo


def c(x):
    return x.hex() if False else repr(x)


def main():
    pio2 = PI / 2
    p1 = trunc_bits(pio2, 33)
    p2 = trunc_bits(pio2 - p1, 33)
    p3 = to_double(pio2 - p1 - p2)
    print("/* pi/2 = P1 + P2 + P3: P1, P2 carry 33 bits each (n * P1, n * P2 exact for |n| < 2^20) */")
    print(f"#define EVPK_PIO2_1 {float(p1)!r}\n#define EVPK_PIO2_2 {float(p2)!r}\n#define EVPK_PIO2_3 {p3!r}")
    assert F(float(p1)) == p1 and F(float(p2)) == p2
    print(f"#define EVPK_2OPI {to_double(2 / PI)!r}")
    for name, v in (("PI", PI), ("PIO2", pio2)):
        hi, lo = hilo(v)
        print(f"#define EVPK_{name}_HI {hi!r}\n#define EVPK_{name}_LO {lo!r}")
    print("/* atan(k/16), k = 0..16, {hi, lo} */")
    rows = []
    for k in range(17):
        hi, lo = hilo(atan_rat(F(k, 16)))
        rows.append((hi, lo))
    his = [r[0] for r in rows]
    print("static const double evpk_atan_tab[34] = {" + ", ".join(f"{hi!r}, {lo!r}" for hi, lo in rows) + "};")
    print("/* Taylor coefficients: sin r = r + r^3 (S[0] + r^2 (S[1] + ...)), cos r = 1 - r^2/2 + r^4 (C[0] + r^2 (C[1] + ...)), atan t = t + t^3 (A[0] + ...) */")
    S = [to_double(F((-1) ** (k + 1), math.factorial(2 * k + 3))) for k in range(8)]       # r^3 .. r^17
    C = [to_double(F((-1) ** k, math.factorial(2 * k + 4))) for k in range(8)]             # r^4 .. r^18
    A = [to_double(F((-1) ** (k + 1), 2 * k + 3)) for k in range(8)]                       # t^3 .. t^17
    for n, v in (("S", S), ("C", C), ("A", A)):
        print(f"static const double evpk_{n}[8] = {{" + ", ".join(repr(x) for x in v) + "};")
    # sanity against libm (not used for the constants)
    assert abs(his[16] - math.atan(1.0)) < 1e-16 and abs(his[8] - math.atan(0.5)) < 1e-16
    assert abs(float(p1) + float(p2) + p3 - math.pi / 2) < 1e-16


if __name__ == "__main__":
    main()
