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
    # contract v3: channel c looks at the 16-bit window starting at byte 4u + (0, 2, 1, 3)[c & 3], u = ((c>>4)&1)*2 + ((c>>3)&1), of
    # the 128 bits of call group (c>>5)*2 + ((c>>2)&1)
    thr = philox.drop_threshold(0.3)
    assert thr == 19660                    # floor(float32(0.3) * 2**16)
    for c in (0, 5, 13, 30, 31, 77, 200, 255):
        g = (c >> 5) * 2 + ((c >> 2) & 1)
        b = 4 * (((c >> 4) & 1) * 2 + ((c >> 3) & 1)) + (0, 2, 1, 3)[c & 3]
        w = philox.philox4x32_10(17, g, 2 | (5 << 16), 3, 9, 7)
        bits = sum(int(w[k]) << (32 * k) for k in range(4))
        bits |= (bits & 0xFF) << 128           # the last window wraps to byte 0
        u16 = (bits >> (8 * b)) & 0xFFFF
        assert m[17, c] == (u16 >= int(thr)), c
    # every channel is decided by exactly one (group, window) pair; a call's 16 windows start at 16 different bytes
    cs = np.arange(256)
    pairs = set(zip(philox.dropout_group16(cs).tolist(), philox.dropout_window_byte(cs).tolist()))
    assert len(pairs) == 256 and {p[1] for p in pairs} == set(range(16)) and {p[0] for p in pairs} == set(range(16))
    other = philox.dropout_keep_mask((7 << 32) | 9, 3, 3, 5, 500, 256, 0.3)
    assert (other != m).mean() > 0.3
    assert philox.dropout_keep_mask(1, 0, 0, 0, 10, 32, 0.0).all()


def test_dropout_windows_are_uniform_and_nearly_independent():
    """Contract v3 reads 16 decisions out of one call as overlapping 16-bit windows.  Over 2**20 decisions per rate: the keep rate
    of every window position matches 1 - thr/2**16 within 4 sigma, and the decisions of windows that share a byte (neighbouring
    bytes of the call) are as good as uncorrelated (within the 4 sigma of 4 096 samples; the bound from the construction is 2**-8)."""
    for rate in (0.3, 0.5, 0.1):
        m = philox.dropout_keep_mask(12345, 1, 0, 3, 4096, 256, rate).astype(np.float64)      # [4096, 256]
        thr = int(philox.drop_threshold(rate))
        pk = 1.0 - thr / 65536.0
        b = philox.dropout_window_byte(np.arange(256))
        g = philox.dropout_group16(np.arange(256))
        for byte in range(16):
            x = m[:, b == byte]
            n = x.size
            assert abs(x.mean() - pk) < 4.0 * np.sqrt(pk * (1 - pk) / n), (rate, byte)
        # channels of one call whose windows start at neighbouring bytes
        for grp in (0, 7, 15):
            ch = {int(b[c]): c for c in range(256) if g[c] == grp}
            for byte in range(16):
                x, y = m[:, ch[byte]], m[:, ch[(byte + 1) % 16]]
                assert abs(np.corrcoef(x, y)[0, 1]) < 0.06, (rate, grp, byte)          # 4096 samples: sigma 0.016
        x = m[:, [c for c in range(256) if g[c] == 3]]
        assert abs(np.corrcoef(x.T) - np.eye(16)).max() < 0.08


def test_categorical_uniforms_contract():
    u = philox.categorical_uniforms(seed=5, image_id=2, num_anchors=1000)
    assert u.shape == (1000, 30) and u.dtype == np.float32
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01
    w = philox.philox4x32_10(10, 6, philox.CAT_TAG, 2, 5, 0)
    assert u[10, 25] == np.float32((int(w[1]) >> 8) * 2.0 ** -24)


def test_mc_dropout_statistics_under_contract_v3_match_independent_masks():
    """Round-4 advisor: contract v3 reads sixteen OVERLAPPING 16-bit windows of one Philox call, so the keep decisions of channels that
    share a byte are only approximately independent (dependence of order 2^-8), while tf.nn.dropout draws independent uniforms
    (multitask_headers.py:104-116) -- and every parity test uses the same contract on both sides.  What the Bayesian stages consume is the
    MC SAMPLE MEAN AND COVARIANCE of a layer's outputs (inference_utils.py:220-244): a dropout layer followed by a dense mix of its 256
    channels, 2 048 MC samples, under v3 masks and under independent Bernoulli masks -- per-output means, variances and the
    covariance of output pairs must agree within the sampling noise of 2 048 draws (a window dependence at the 2^-8 level moves a
    variance by < 1e-3 of itself; a broken window scheme -- e.g. neighbours sharing a whole half-word -- moves it by tens of per cent)."""
    from oracle import philox
    rng = np.random.default_rng(12)
    S, P, C, K = 2048, 8, 256, 64
    x = np.maximum(rng.normal(0, 1, (P, C)), 0)
    w = rng.normal(0, 1, (C, K)) / np.sqrt(C)
    # adversarial mix for the window scheme: one output sums NEIGHBOURING channels with equal signs (their keep decisions share bytes)
    w[:, 0] = 1.0 / np.sqrt(C)
    w[:, 1] = np.where((np.arange(C) & 3) < 2, 1.0, -1.0) / np.sqrt(C)
    v3 = np.stack([philox.dropout_keep_mask(77, 3, s, 5, P, C, 0.3) for s in range(S)]).astype(np.float64)      # [S,P,C]
    ind = (rng.random((S, P, C)) >= 0.3).astype(np.float64)
    y3 = np.einsum("spc,ck->spk", v3 * x / 0.7, w)
    yi = np.einsum("spc,ck->spk", ind * x / 0.7, w)
    # means: standard error sigma / sqrt(S)
    sd = yi.std(0)
    assert np.all(np.abs(y3.mean(0) - yi.mean(0)) < 6 * sd * np.sqrt(2.0 / S))
    # variances: relative standard error sqrt(2 / S) = 3.1 % per output; averaged over the 8 x 64 outputs the two schemes agree to < 1 %
    ratio = y3.var(0) / yi.var(0)
    assert np.all(np.abs(ratio - 1) < 6 * np.sqrt(2 * 2.0 / S)), (ratio.min(), ratio.max())
    assert abs(ratio.mean() - 1) < 1e-2, ratio.mean()
    assert abs(ratio[:, :2].mean() - 1) < 4 * np.sqrt(2 * 2.0 / S / (2 * P))            # the adversarial outputs
    # covariance of output pairs (the off-diagonal of the epistemic covariance): correlation coefficients agree within 6 / sqrt(S)
    c3 = np.stack([np.corrcoef(y3[:, p, :8].T) for p in range(P)])
    ci = np.stack([np.corrcoef(yi[:, p, :8].T) for p in range(P)])
    assert np.abs(c3 - ci).max() < 6.0 / np.sqrt(S) * 1.5
