import numpy as np

from oracle import philox


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        got = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in got) == exp


def test_philox_vectorised_matches_scalar():
    rng = np.random.default_rng(0)
    c = rng.integers(0, 2 ** 32, size=(4, 50), dtype=np.uint64)
    v = philox.philox4x32_10(c[0], c[1], c[2], c[3], 123, 456)
    for i in (0, 7, 49):
        s = philox.philox4x32_10(int(c[0, i]), int(c[1, i]), int(c[2, i]), int(c[3, i]), 123, 456)
        assert all(int(a[i]) == int(b) for a, b in zip(v, s))


def test_dropout_mask_contract():
    m = philox.dropout_keep_mask(seed=(7 << 32) | 9, image_id=3, sample=2, layer_id=5, num_pixels=500,
                                 channels=256, rate=0.3)
    assert m.shape == (500, 256) and m.dtype == bool
    assert abs(m.mean() - 0.7) < 0.01
    # channel c: decision d = ((c>>3)&1)*4 + (c&3) of call group (c>>5)*4 + ((c>>4)&1)*2 + ((c>>2)&1)
    thr = philox.drop_threshold(0.3)
    assert thr == 19660                    # floor(float32(0.3) * 2**16)
    for c in (0, 5, 13, 77, 200, 255):
        g = (c >> 5) * 4 + ((c >> 4) & 1) * 2 + ((c >> 2) & 1)
        d = ((c >> 3) & 1) * 4 + (c & 3)
        w = philox.philox4x32_10(17, g, 2 | (5 << 16), 3, 9, 7)
        u16 = (int(w[d >> 1]) >> (16 * (d & 1))) & 0xFFFF
        assert m[17, c] == (u16 >= int(thr))
    # every channel is decided by exactly one (group, decision) pair
    cs = np.arange(256)
    pairs = set(zip(((cs >> 5) * 4 + ((cs >> 4) & 1) * 2 + ((cs >> 2) & 1)).tolist(), (((cs >> 3) & 1) * 4 + (cs & 3)).tolist()))
    assert len(pairs) == 256
    other = philox.dropout_keep_mask((7 << 32) | 9, 3, 3, 5, 500, 256, 0.3)
    assert (other != m).mean() > 0.3
    assert philox.dropout_keep_mask(1, 0, 0, 0, 10, 32, 0.0).all()


def test_categorical_uniforms_contract():
    u = philox.categorical_uniforms(seed=5, image_id=2, num_anchors=1000)
    assert u.shape == (1000, 30) and u.dtype == np.float32
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01
    w = philox.philox4x32_10(10, 6, philox.CAT_TAG, 2, 5, 0)
    assert u[10, 25] == np.float32((int(w[1]) >> 8) * 2.0 ** -24)
