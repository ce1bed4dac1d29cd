"""CPU checks of the oracle's Modular -> f32 conversions against independent definitions (no GPU): the floating-point
sample path of ConvertModularToF32Stage (int_to_float, render/stages/convert.rs:416-486) must agree with IEEE binary16 /
binary32 semantics -- the reference's own fast paths for those two formats -- and with the value a custom format
encodes, computed here in float64 from the format's definition."""
import numpy as np
import pytest


def test_binary16_samples_widen_like_ieee(oracle):
    smp = np.arange(0, 1 << 16, dtype=np.int64).astype(np.int32)          # every half-precision bit pattern
    got = oracle.modular_to_f32(smp, 16, 5)
    ref = smp.astype(np.uint16).view(np.float16).astype(np.float32)
    nan = np.isnan(ref)
    assert np.array_equal(np.isnan(got), nan)
    assert np.array_equal(got.view(np.uint32)[~nan], ref.view(np.uint32)[~nan])


def test_binary32_samples_pass_through(oracle):
    rng = np.random.default_rng(3)
    smp = rng.integers(0, 1 << 32, size=20000, dtype=np.int64).astype(np.uint32).view(np.int32)
    got = oracle.modular_to_f32(smp, 32, 8)
    assert np.array_equal(got.view(np.int32), smp)


@pytest.mark.parametrize("bits,exp_bits", [(19, 6), (12, 4), (24, 7), (10, 2)])
def test_custom_float_formats_decode_to_their_value(oracle, bits, exp_bits):
    mant = bits - exp_bits - 1
    rng = np.random.default_rng(bits)
    smp = rng.integers(0, 1 << bits, size=5000, dtype=np.int64)
    with np.errstate(invalid="ignore"):   # signalling NaN patterns among the random samples
        got = oracle.modular_to_f32(smp.astype(np.int32), bits, exp_bits).astype(np.float64)
    sign = np.where(smp >> (bits - 1), -1.0, 1.0)
    e = (smp >> mant) & ((1 << exp_bits) - 1)
    m = smp & ((1 << mant) - 1)
    bias = (1 << (exp_bits - 1)) - 1
    normal = sign * (1.0 + m / float(1 << mant)) * np.exp2(e.astype(np.float64) - bias)
    sub = sign * (m / float(1 << mant)) * np.exp2(1.0 - bias)
    want = np.where(e == 0, sub, normal)
    finite = e != (1 << exp_bits) - 1
    assert np.array_equal(got[finite], want[finite])        # every such value is exactly representable in binary32
    inf = ~finite & (m == 0)
    assert np.all(np.isinf(got[inf])) and np.array_equal(np.sign(got[inf]), sign[inf])
    assert np.all(np.isnan(got[~finite & (m != 0)]))
