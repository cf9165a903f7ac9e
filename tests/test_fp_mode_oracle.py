"""fp_mode is a live switch: GCC_FMA (first product fused, as g++ -O3 -march=native builds
GET_VALUE, src/ORBextractor.cc:207-209) and STRICT (both products rounded) disagree on about 1.6e-7 of all
(angle, tap) pairs (300 of 1.8e9 on a 1e-4 degree grid).  Angles below come from that search."""
import numpy as np
import oracle


def test_fma_and_strict_differ_at_known_angles():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64)).astype(np.uint8)
    L, p = oracle.lib(), oracle.orb_oracle._p
    ndiff = 0
    for i in (26615, 52150, 68474, 115172):
        ang = float(np.float32(i) * np.float32(0.0001))
        a = np.zeros(32, np.uint8); b = np.zeros(32, np.uint8)
        L.orc_descriptor(p(img), 64, 32, 32, ang, oracle.FP_GCC_FMA, p(a))
        L.orc_descriptor(p(img), 64, 32, 32, ang, oracle.FP_STRICT, p(b))
        nb = int(np.unpackbits(a ^ b).sum())
        assert nb <= 4                      # at most the bits of the shifted taps
        ndiff += nb > 0
    assert ndiff >= 1
    # and an ordinary angle gives identical descriptors
    a = np.zeros(32, np.uint8); b = np.zeros(32, np.uint8)
    L.orc_descriptor(p(img), 64, 32, 32, 123.456, oracle.FP_GCC_FMA, p(a))
    L.orc_descriptor(p(img), 64, 32, 32, 123.456, oracle.FP_STRICT, p(b))
    assert np.array_equal(a, b)
