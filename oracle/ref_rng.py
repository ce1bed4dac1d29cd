"""The random draw of the reference's transform tests, restated (TEST INFRASTRUCTURE, like everything under oracle/).

jxl_transforms/src/tests.rs:246-255:

    fn random_matrix(n, m) { let mut rng = ChaCha12Rng::seed_from_u64(0);
                             ... *x = rng.random_range(-1.0..1.0) }   // row-major, f64

The crates are third-party dependencies absent from /root/reference (Cargo.lock pins rand 0.9.2,
rand_chacha 0.9.0, rand_core 0.9.3); their published algorithms are restated here:

  * rand_core::SeedableRng::seed_from_u64: a PCG32 stream (multiplier 6364136223846793005, increment
    11634580027462260723; the state advances first, output = rotate_right((state >> 18 ^ state) >> 27, state >> 59))
    fills the 32-byte seed four little-endian bytes at a time.
  * rand_chacha::ChaCha12Rng: the ChaCha stream cipher with 12 rounds (D. J. Bernstein), key = the seed, 64-bit
    block counter in words 12-13 starting at 0, 64-bit stream id in words 14-15 = 0; the generator hands out the
    key stream as little-endian u32 words in order; next_u64 = two consecutive words, low word first.
  * rand::distr::uniform::UniformFloat<f64>::sample_single(low, high): value1_2 = f64 from the bits
    (next_u64 >> 12) with exponent 0, i.e. in [1, 2); res = (value1_2 - 1.0) * (high - low) + low; accepted when
    res < high (always, for the range -1.0..1.0 at this precision).

The ChaCha core is checked against the published ChaCha20 / ChaCha12 zero-key key streams in
tests/test_oracle_pin.py; the draw itself is what makes the reference's per-shape tolerances (calibrated on exactly
this draw) applicable to the oracle without slack.
"""
import struct

import numpy as np

_MASK32 = 0xFFFFFFFF
_MASK64 = 0xFFFFFFFFFFFFFFFF


def _rotl32(v, n):
    return ((v << n) & _MASK32) | (v >> (32 - n))


def chacha_block(key_words, counter, stream, rounds):
    """One 64-byte ChaCha block as 16 little-endian u32 words.  State layout: constants, 8 key words, 64-bit block
    counter (low word first), 64-bit stream id."""
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [
        counter & _MASK32, (counter >> 32) & _MASK32, stream & _MASK32, (stream >> 32) & _MASK32]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _MASK32
        x[d] = _rotl32(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _MASK32
        x[b] = _rotl32(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _MASK32
        x[d] = _rotl32(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _MASK32
        x[b] = _rotl32(x[b] ^ x[c], 7)

    assert rounds % 2 == 0
    for _ in range(rounds // 2):
        qr(0, 4, 8, 12)
        qr(1, 5, 9, 13)
        qr(2, 6, 10, 14)
        qr(3, 7, 11, 15)
        qr(0, 5, 10, 15)
        qr(1, 6, 11, 12)
        qr(2, 7, 8, 13)
        qr(3, 4, 9, 14)
    return [(x[i] + init[i]) & _MASK32 for i in range(16)]


def seed_from_u64(state, nbytes=32):
    """rand_core::SeedableRng::seed_from_u64 (PCG32 expansion of a u64 into a seed)."""
    mul, inc = 6364136223846793005, 11634580027462260723
    out = bytearray()
    while len(out) < nbytes:
        state = (state * mul + inc) & _MASK64
        xorshifted = (((state >> 18) ^ state) >> 27) & _MASK32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _MASK32
        out += struct.pack("<I", x)
    return bytes(out[:nbytes])


class ChaCha12Rng:
    """rand_chacha::ChaCha12Rng as far as the reference's tests use it (seed_from_u64, next_u64)."""

    def __init__(self, seed_bytes):
        assert len(seed_bytes) == 32
        self.key = struct.unpack("<8I", seed_bytes)
        self.counter = 0
        self.words = []

    @classmethod
    def seed_from_u64(cls, state):
        return cls(seed_from_u64(state))

    def next_u32(self):
        if not self.words:
            self.words = chacha_block(self.key, self.counter, 0, 12)
            self.counter += 1
        return self.words.pop(0)

    def next_u64(self):
        lo = self.next_u32()
        hi = self.next_u32()
        return lo | (hi << 32)

    def random_range_f64(self, low, high):
        """UniformFloat<f64>::sample_single"""
        scale = high - low
        while True:
            bits = (self.next_u64() >> 12) | (1023 << 52)
            value1_2 = struct.unpack("<d", struct.pack("<Q", bits))[0]
            res = (value1_2 - 1.0) * scale + low
            if res < high:
                return res


def random_matrix(n, m):
    """tests.rs:246-255: an n x m f64 matrix of uniform(-1, 1) values from ChaCha12Rng::seed_from_u64(0), row-major."""
    rng = ChaCha12Rng.seed_from_u64(0)
    return np.array([[rng.random_range_f64(-1.0, 1.0) for _ in range(m)] for _ in range(n)], dtype=np.float64)
