"""The arithmetic identity the rewritten row loop of the scoring kernels rests on (tools/patches/score_body_v2.patch,
tools/micro/score_loop.hip): in fp32 the RAW product of four terms started at 2^108 has, bit for bit, the
mantissa of the product of the four mantissas (rounding commutes with powers of two while nothing leaves the normal range), and its
exponent is the exact sum of the four exponents + 108 -- for terms between 2^-58 and 2^4, in any order.  numpy float32 on the CPU;
the kernels themselves are checked on the GPU (the micro-benchmark's host fp64 check, the library's parity tests)."""
import numpy as np


def _products(x):
    n = len(x)
    raw = np.full(n, np.float32(2.0 ** 108), dtype=np.float32)
    mant = np.ones(n, dtype=np.float32)
    exps = np.zeros(n, dtype=np.int64)
    for k in range(x.shape[1]):
        raw = (raw * x[:, k]).astype(np.float32)
        mk, ek = np.frexp(x[:, k])
        mant = (mant * mk.astype(np.float32)).astype(np.float32)
        exps += ek
    return raw, mant, exps


def test_raw_product_of_four_equals_the_mantissa_product_bit_for_bit():
    rng = np.random.default_rng(20261005)
    n = 400_000
    x = np.ldexp(rng.uniform(0.5, 1.0, size=(n, 4)), rng.integers(-57, 5, size=(n, 4))).astype(np.float32)
    # the extremes, in both orders: four smallest, four largest, small then large, large then small
    lo, hi = np.float32(2.0 ** -58), np.float32(np.nextafter(np.float32(16.0), np.float32(0.0)))
    x[:4] = [[lo, lo, lo, lo], [hi, hi, hi, hi], [lo, lo, hi, hi], [hi, hi, lo, lo]]
    raw, mant, exps = _products(x)
    assert np.all(np.isfinite(raw)) and np.all(raw >= np.float32(2.0 ** -126))            # every product a normal number
    m_raw, e_raw = np.frexp(raw)
    m_ref, e_ref = np.frexp(mant)
    assert np.array_equal(m_raw.astype(np.float32), m_ref.astype(np.float32))             # the same mantissa, bit for bit
    assert np.array_equal(e_raw - 108, exps + e_ref)                                       # the exponent: exact sum + 108
    # ... and the log2 of the batch is taken of a value in [1/2, 1) instead of [1/16, 1): not worse against fp64
    ref = np.log2(x.astype(np.float64)).sum(1)
    new = np.log2(m_raw.astype(np.float32)).astype(np.float64) + (e_raw - 108)
    old = np.log2(mant).astype(np.float64) + exps
    assert np.abs(new - ref).max() <= np.abs(old - ref).max() * 1.05 + 1e-12
    assert np.abs(new - ref).max() < 5e-7


def test_a_zero_or_out_of_range_term_is_caught_by_the_guard():
    """What the kernel's guard (positive normal number above 2^-90) must send to the term-by-term path."""
    # (the threshold 2^-90 covers partial products that passed through the denormals as long as the terms behind them multiply by no
    # more than 2^36 -- three terms of at most 2^12; a term of the scoring kernels is at most a few units)
    big, huge, tiny = np.float32(2.0 ** 12), np.float32(2.0 ** 40), np.float32(2.0 ** -120)
    cases = np.array([[1.0, 0.0, 1.0, 1.0],          # an exact zero term (lambda = 0: every never co-rated pair): the product is 0
                      [tiny, tiny, 1.0, 1.0],        # a partial product below the normal range (2^-132)
                      [tiny, tiny, big, big],        # ... even when later terms bring the product back up: it stays below 2^-90
                      [huge, huge, huge, huge]],     # overflow
                     dtype=np.float32)
    with np.errstate(over="ignore", under="ignore", divide="ignore"):
        raw, _, _ = _products(cases)
    ok = np.isfinite(raw) & (raw > np.float32(2.0 ** -90))
    assert not ok.any(), raw
