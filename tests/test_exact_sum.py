"""The order-independent floating-point PLUS of the masked product's deterministic mode (pygraphblas_amd/csrc/grb_exact.hpp), run on the HOST through
GrBX_exact_sum_host: terms -> 128-bit integers in the row's unit -> integer sum -> one rounding.  No GPU: the same two functions (fx_from_double, fx_to_fp)
are what the kernels call; the device adds the integers with atomics instead of `+`.  The check is Python's math.fsum (the exactly rounded sum) and, for the
24-bit rounding of FP32 results, exact rationals."""
import ctypes as C
import math
import random
import struct
from fractions import Fraction

import numpy as np
import pytest

import pygraphblas_amd as gb


def exact_sum(terms, bits=53):
    arr = (C.c_double * len(terms))(*terms)
    out, u = C.c_double(), C.c_int()
    gb.base.check(gb.lib.GrBX_exact_sum_host(arr, C.c_uint64(len(terms)), C.c_int(bits), C.byref(out), C.byref(u)))
    return out.value, u.value


def random_terms(rng):
    n = rng.choice([1, 2, 3, 5, 17, 64, 300])
    spread = rng.choice([0, 1, 10, 40, 60])
    base = rng.choice([-300, -100, 0, 50, 300]) if spread < 40 else rng.choice([-100, 0, 100])
    terms = [rng.choice([-1, 1]) * math.ldexp(rng.random() + 0.5, base + rng.randint(0, spread)) for _ in range(n)]
    if rng.random() < 0.3:
        terms += [-t for t in terms[: n // 2]]          # cancellation down to the last bits
    rng.shuffle(terms)
    return terms


def test_sum_of_doubles_is_the_exactly_rounded_sum():
    rng = random.Random(5)
    for _ in range(6000):
        terms = random_terms(rng)
        got, _ = exact_sum(terms)
        exp = math.fsum(terms)
        assert struct.pack("d", got + 0.0) == struct.pack("d", exp + 0.0), (len(terms), got.hex(), exp.hex())


def test_sum_rounded_to_24_bits_is_the_nearest_float_of_the_exact_sum():
    rng = random.Random(6)
    for _ in range(3000):
        terms = [float(np.float32(t)) for t in random_terms(rng) if abs(t) < 1e30 and abs(t) > 1e-30]
        if not terms:
            continue
        got, _ = exact_sum(terms, 24)
        ex = sum(Fraction(t) for t in terms)
        if ex == 0:
            assert got == 0.0
            continue
        near = np.float32(float(ex))
        cands = [np.nextafter(near, np.float32(-np.inf)), near, np.nextafter(near, np.float32(np.inf))]
        best = min(cands, key=lambda c: (abs(Fraction(float(c)) - ex), int(np.float32(c).view(np.uint32)) & 1))
        assert np.float32(got) == best and float(np.float32(got)) == got, (got, best)


def test_any_order_gives_the_same_bits_and_terms_far_below_the_bound_are_cut_at_the_unit():
    rng = random.Random(7)
    terms = [rng.uniform(-1, 1) * 10.0 ** rng.randint(-8, 8) for _ in range(500)]
    first, u = exact_sum(terms)
    for _ in range(20):
        rng.shuffle(terms)
        assert exact_sum(terms)[0].hex() == first.hex()
    # the unit: 2^(ilogb(bound) + 2 + H - 126); a term below it vanishes, one above it counts down to the unit
    big = 1.0
    got, u = exact_sum([big, math.ldexp(1.0, u - 1)])
    assert got == 1.0
    got, _ = exact_sum([big, -big, math.ldexp(1.0, u + 3)])
    assert got == math.ldexp(1.0, u + 3)
    assert u == 0 + 2 + 2 - 126


def test_non_finite_terms_are_refused():
    arr = (C.c_double * 2)(1.0, float("inf"))
    out, u = C.c_double(), C.c_int()
    assert gb.lib.GrBX_exact_sum_host(arr, C.c_uint64(2), C.c_int(53), C.byref(out), C.byref(u)) != 0


def test_subnormal_results_are_rounded_once():
    """a sum that lands in the subnormal range of its target type is rounded ONCE, at the last place that type really has there (2^-1074 for a
    double, 2^-149 for an FP32 result) — not to 53 / 24 bits first and then again by ldexp or the cast to float.  Doubles against math.fsum, the 24-bit results against
    exact rationals rounded to the nearest float32 (ties to even) by numpy."""
    rng = random.Random(9)
    tiny = 5e-324
    cases = [[3 * tiny, 2 * tiny, -tiny], [math.ldexp(1.0, -1060), math.ldexp(1.0, -1074), math.ldexp(1.0, -1073)], [math.ldexp(0.75, -1022), -math.ldexp(0.5, -1022), tiny]]
    for _ in range(300):
        cases.append([rng.choice([-1, 1]) * math.ldexp(rng.random() + 0.5, rng.randint(-1074, -1040)) for _ in range(rng.choice([2, 5, 33]))])
    for terms in cases:
        got, _ = exact_sum(terms)
        assert struct.pack("<d", got) == struct.pack("<d", math.fsum(terms)) or (got == 0.0 and math.fsum(terms) == 0.0), terms
    # FP32 results: products of floats summed exactly, the sum subnormal in float32
    for _ in range(300):
        terms = [float(np.float32(rng.choice([-1, 1]) * math.ldexp(rng.random() + 0.5, rng.randint(-149, -128)))) for _ in range(rng.choice([2, 7, 40]))]
        got, _ = exact_sum(terms, bits=24)
        exact = sum(Fraction(t) for t in terms)
        want = float(np.float32(float(exact))) if exact.denominator.bit_length() <= 1000 else None      # the double of the exact sum is itself exact here (few bits, one binade range)
        assert float(np.float32(got)) == got, terms                                                       # representable in float32
        assert got == want, (terms, got, want)
