"""The product's device math (csrc/devmath.hpp), compiled for the host, against the oracle's definitions: every
primitive must be BIT-identical, including the fma-corrected UNORM decoders that replace the literal divisions."""
import numpy as np


def bits(a):
    return a.view(np.uint32)


def test_d24_decode_exhaustive(hostsim):
    x = np.arange(1 << 24, dtype=np.uint32)
    got = hostsim.eval_array(6, x.view(np.float32))
    ref = x.astype(np.float32) / np.float32(16777215.0)           # IEEE division in numpy
    assert np.array_equal(bits(got), bits(ref))


def test_d24_decode_matches_oracle_sample(hostsim, oracle):
    rng = np.random.default_rng(0)
    x = rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32)   # stencil bits set too
    assert np.array_equal(bits(hostsim.eval_array(6, x.view(np.float32))), bits(oracle.eval_array(6, x.view(np.float32))))


def test_unorm16_unorm8_half_exhaustive(hostsim, oracle):
    x = np.arange(65536, dtype=np.uint32)
    assert np.array_equal(bits(hostsim.eval_array(7, x.view(np.float32))), bits(x.astype(np.float32) / np.float32(65535.0)))
    assert np.array_equal(bits(hostsim.eval_array(7, x.view(np.float32))), bits(oracle.eval_array(7, x.view(np.float32))))
    y = np.arange(256, dtype=np.uint32)
    assert np.array_equal(bits(hostsim.eval_array(8, y.view(np.float32))), bits(y.astype(np.float32) / np.float32(255.0)))
    assert np.array_equal(bits(hostsim.eval_array(9, x.view(np.float32))), bits(oracle.eval_array(9, x.view(np.float32))))


def same_bits_or_nan(a, b):
    an, bn = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(an, bn) and np.array_equal(bits(a)[~an], bits(b)[~bn]))


def test_transcendentals_bit_identical(hostsim, oracle):
    rng = np.random.default_rng(1)
    xs = np.concatenate([rng.uniform(-400, 400, 300000), rng.uniform(-1, 1, 100000), rng.normal(0, 1e5, 50000),
                         [0.0, -0.0, np.inf, -np.inf, np.nan, 1e30, -1e30, 8388607.5, 8388608.0]]).astype(np.float32)
    for kind in (0, 1, 3):
        assert np.array_equal(bits(hostsim.eval_array(kind, xs)), bits(oracle.eval_array(kind, xs))), kind
    pos = np.concatenate([np.exp(rng.uniform(-100, 88, 300000)), [0.0, 1e-45, 1e-40, 1.0, np.inf, np.nan, -1.0]]).astype(np.float32)
    assert np.array_equal(bits(hostsim.eval_array(2, pos)), bits(oracle.eval_array(2, pos)))
    base = np.concatenate([rng.uniform(0, 1, 300000), rng.uniform(0, 50, 50000), [0.0, 1.0, np.nan, -0.5, np.inf]]).astype(np.float32)
    expo = np.full_like(base, np.float32(1.0 / 2.2))
    assert np.array_equal(bits(hostsim.eval_array(4, base, expo)), bits(oracle.eval_array(4, base, expo)))
    # the tone map's pow(x, 1/2.2): scalar and packed forms of the kernels against the oracle, every exponent, both signs, specials
    allx = np.concatenate([base, np.exp(rng.uniform(-104, 88, 300000)), -np.exp(rng.uniform(-104, 88, 20000)),
                           [-0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1.17549435e-38, -1.17549435e-38, -np.inf, 3.4028234e38]]).astype(np.float32)
    assert same_bits_or_nan(hostsim.eval_array(10, allx), oracle.eval_array(10, allx))
    assert same_bits_or_nan(hostsim.eval_array(11, allx, np.roll(allx, 7)), oracle.eval_array(10, allx))
    u = rng.uniform(-2, 3, 200000).astype(np.float32); v = rng.uniform(-2, 3, 200000).astype(np.float32)
    assert np.array_equal(bits(hostsim.eval_array(5, u, v)), bits(oracle.eval_array(5, u, v)))


def test_d24_threshold(hostsim):
    """light_core.hpp d24_threshold(ref): `ref <= d24_to_float(t)` <=> `t >= d24_threshold(ref)` -- checked against the definition
    for every D24 value (references: the decoded value, its two float neighbours) and for out-of-range / NaN references."""
    assert int(hostsim.lib.hs_check_d24_threshold()) == 0
