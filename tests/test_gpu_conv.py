"""GPU parity of the hot kernel in isolation: one convolution through the pipeline's
implicit-GEMM MFMA kernel (C ABI ``bod_stage_conv``) vs the oracle's conv on IDENTICAL inputs
(both sides see the same bf16-rounded activations and weights; accumulation is fp32 on the device,
float64 in the oracle).  Tolerance: BASELINE.json north_star, 1e-3 relative."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
REL_TOL = 1e-3


def _case(rng, b, h, w, cin, cout, k):
    x = rng.normal(0, 1, (b, h, w, cin)).astype(np.float32)
    wt = (rng.normal(0, 1, (k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    return x, wt, bias


def _oracle(x, wt, bias, stride, padding, relu=False, residual=None):
    from oracle import network
    xb = network.bf16_round(x).astype(np.float64)
    wb = network.bf16_round(wt).astype(np.float64)
    y = network.conv2d(xb, wb, bias.astype(np.float64), stride, padding)
    if residual is not None:
        y = y + network.bf16_round(residual).astype(np.float64)
    if relu:
        y = np.maximum(y, 0)
    return y


CASES = [
    # b, h, w, cin, cout, k, stride, padding          what it stands for
    (2, 16, 16, 256, 256, 3, 1, "same"),            # head tower / FPN output conv
    (1, 9, 13, 64, 64, 3, 1, "same"),               # ragged size, 64-wide cout tile
    (2, 12, 12, 64, 256, 1, 1, "valid"),            # bottleneck 1x1 expand
    (1, 15, 17, 256, 128, 1, 2, "valid"),           # ConvBlock strided 1x1 (odd input)
    (1, 16, 16, 128, 256, 3, 2, "same"),            # P6: stride 2 SAME on even input (pad 0/1)
    (1, 7, 5, 128, 256, 3, 2, "same"),              # stride 2 SAME on odd input (pad 1/1)
    (1, 8, 8, 256, 72, 1, 1, "same"),               # cls output conv (fp32 out, padded cout)
    (1, 8, 8, 256, 36, 1, 1, "same"),               # reg output conv
    (1, 8, 8, 256, 90, 1, 1, "same"),               # cov output conv (cout not a multiple of 4)
    (1, 40, 36, 512, 128, 3, 1, "same"),            # many K tiles, M not a multiple of the tile
    (1, 130, 128, 64, 256, 3, 1, "same"),           # M >= 16384: 256x256-tile 8-wave configuration
    (2, 96, 100, 128, 512, 1, 1, "valid"),          # 256-tile config, two cout tiles, ragged M
]


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding", CASES)
def test_conv_matches_oracle(b, h, w, cin, cout, k, stride, padding):
    from bayes_od_rc_amd.engine import stage_conv
    rng = np.random.default_rng(cin * 131 + cout * 7 + k + h)
    x, wt, bias = _case(rng, b, h, w, cin, cout, k)
    got = stage_conv(x, wt, bias, stride=stride, padding=padding)
    ref = _oracle(x, wt, bias, stride, padding)
    assert got.shape == ref.shape
    rms = float(np.sqrt((ref ** 2).mean()))
    assert rel_err(got, ref, floor=rms) < REL_TOL


def test_conv_residual_relu_bf16_store():
    """Bottleneck tail: conv + shortcut + ReLU, stored as bf16 (feature_extractor.py:208-213)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(5)
    x, wt, bias = _case(rng, 2, 10, 14, 128, 512, 1)
    res = rng.normal(0, 1, (2, 10, 14, 512)).astype(np.float32)
    got = stage_conv(x, wt, bias, padding="valid", relu=True, residual=res, round_output_bf16=True)
    ref = _oracle(x, wt, bias, 1, "valid", relu=True, residual=res)
    assert np.array_equal(got, network.bf16_round(got))          # really bf16-valued
    ref_b = network.bf16_round(ref.astype(np.float32))
    # identical up to rare 1-ulp bf16 rounding flips (fp32 vs float64 accumulation)
    mism = got != ref_b
    assert mism.mean() < 2e-3
    assert np.all(np.abs(got - ref_b)[mism] <= np.abs(ref_b[mism]) * 2.0 ** -7 + 1e-30)
    assert np.all(got >= 0)


def test_conv_dropout_matches_philox_contract():
    """Head-tower epilogue: ReLU -> x/(1-rate) -> Philox keep mask -> bf16
    (multitask_headers.py:102-116; DESIGN.md RNG contract)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network, philox
    rng = np.random.default_rng(9)
    b, h, w = 3, 6, 10
    x, wt, bias = _case(rng, b, h, w, 256, 256, 3)
    seed, lid, img, rate = (0xDEADBEEF << 20) + 12345, 6, 41, 0.3
    got = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=rate, seed=seed,
                     layer_id=lid, image_id=img)
    ref = _oracle(x, wt, bias, 1, "same", relu=True)
    keep = np.stack([philox.dropout_keep_mask(seed, img, s, lid, h * w, 256, rate).reshape(h, w, 256)
                     for s in range(b)])
    # dropped elements are exactly zero, kept ones are scaled by float32(1/(1-rate))
    assert np.all(got[~keep] == 0)
    assert abs(keep.mean() - (1 - rate)) < 0.01
    scaled = ref * np.float64(np.float32(1.0 / (1.0 - rate))) * keep
    ref_b = network.bf16_round(scaled.astype(np.float32))
    mism = got != ref_b
    assert mism.mean() < 2e-3
    assert np.all(np.abs(got - ref_b)[mism] <= np.abs(ref_b[mism]) * 2.0 ** -7 + 1e-30)
    # a different image id / layer id / seed gives a different mask
    other = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=rate, seed=seed,
                       layer_id=lid + 1, image_id=img)
    assert ((other == 0) != (got == 0)).mean() > 0.2


def test_stage_conv_rejects_bad_arguments():
    from bayes_od_rc_amd.engine import stage_conv
    x = np.zeros((1, 4, 4, 48), np.float32)
    w = np.zeros((3, 3, 48, 64), np.float32)
    with pytest.raises(ValueError):
        stage_conv(x, w)                     # Cin not a multiple of 64


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding", CASES[:10])
def test_fp32_conv_matches_oracle(b, h, w, cin, cout, k, stride, padding):
    """fp32 precision mode: unrounded fp32 operands through v_mfma_f32_32x32x2_f32 vs float64."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(cin + cout + k + w)
    x, wt, bias = _case(rng, b, h, w, cin, cout, k)
    got = stage_conv(x, wt, bias, stride=stride, padding=padding, precision="fp32")
    ref = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), stride, padding)
    rms = float(np.sqrt((ref ** 2).mean()))
    assert rel_err(got, ref, floor=rms) < 2e-5          # fp32 accumulation over up to K=4608 terms


def test_fp32_conv_residual_relu_dropout():
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network, philox
    rng = np.random.default_rng(8)
    b, h, w = 2, 7, 9
    x, wt, bias = _case(rng, b, h, w, 128, 256, 3)
    res = rng.normal(0, 1, (b, h, w, 256)).astype(np.float32)
    got = stage_conv(x, wt, bias, padding="same", relu=True, residual=res, precision="fp32")
    ref = np.maximum(network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same") + res, 0)
    assert rel_err(got, ref, floor=float(np.sqrt((ref ** 2).mean()))) < 2e-5
    got = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=0.3, seed=11, layer_id=2, image_id=5,
                     precision="fp32")
    keep = np.stack([philox.dropout_keep_mask(11, 5, s, 2, h * w, 256, 0.3).reshape(h, w, 256) for s in range(b)])
    ref = np.maximum(network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same"), 0)
    ref = ref * np.float64(np.float32(1.0 / 0.7)) * keep
    assert np.all(got[~keep] == 0)
    assert rel_err(got, ref, floor=float(np.sqrt((ref ** 2).mean()))) < 2e-5


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding,split,res", [
    (2, 16, 16, 2048, 256, 3, 2, "same", 8, False),      # P6: 16x16x2048 -> 8x8x256, K = 18432
    (1, 8, 8, 1024, 512, 1, 1, "valid", 4, True),        # stage-5 1x1 at batch 1, with shortcut + ReLU
    (1, 11, 9, 512, 192, 3, 1, "same", 2, False),        # ragged M, 64-wide cout tile
])
def test_split_k_matches_oracle_and_unsplit(monkeypatch, b, h, w, cin, cout, k, stride, padding, split, res):
    """Split-K (small-M layers): partial fp32 sums over channel-chunk ranges + the reduce kernel give the oracle's
    result within the conv tolerance and the un-split kernel's bf16 output up to rare 1-ulp flips."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(cin + cout + split)
    x, wt, bias = _case(rng, b, h, w, cin, cout, k)
    oh = -(-h // stride) if padding == "same" else (h - k) // stride + 1
    ow = -(-w // stride) if padding == "same" else (w - k) // stride + 1
    residual = rng.normal(0, 1, (b, oh, ow, cout)).astype(np.float32) if res else None
    kw = dict(stride=stride, padding=padding, relu=res, residual=residual, round_output_bf16=True)
    plain = stage_conv(x, wt, bias, **kw)
    monkeypatch.setenv("BOD_STAGE_KSPLIT", str(split))
    got = stage_conv(x, wt, bias, **kw)
    monkeypatch.delenv("BOD_STAGE_KSPLIT")
    ref = _oracle(x, wt, bias, stride, padding, relu=res, residual=residual)
    rms = float(np.sqrt((ref ** 2).mean()))
    assert rel_err(got, ref, floor=rms) < 6e-3                    # bf16-stored output: half an ulp of 2^-8
    mism = got != plain
    assert mism.mean() < 5e-3
    # one bf16 ulp, plus the fp32 re-association error of the K-long sum where the result nearly cancels
    assert np.all(np.abs(got - plain)[mism] <= np.abs(plain[mism]) * 2.0 ** -7 + 1e-4 * rms)
    assert np.array_equal(got, network.bf16_round(got))


# ---------------------------------------------------------------------------------------------- bf16x3 precision
@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding", CASES)
def test_bf16x3_conv_matches_oracle(b, h, w, cin, cout, k, stride, padding):
    """bf16x3 precision: both operands as (hi, lo) bf16 pairs, products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16,
    fp32 accumulate -- against float64 on the UNROUNDED fp32 operands.  Representation error 2^-17 per operand, the
    dropped lo*lo term 2^-16 of a product: 1e-4 of the output RMS over up to K = 4608 terms (fp32 MFMA mode: 2e-5)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(cin + 3 * cout + k + w)
    x, wt, bias = _case(rng, b, h, w, cin, cout, k)
    got = stage_conv(x, wt, bias, stride=stride, padding=padding, precision="bf16x3")
    ref = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), stride, padding)
    rms = float(np.sqrt((ref ** 2).mean()))
    assert got.shape == ref.shape
    assert rel_err(got, ref, floor=rms) < 1e-4


def test_bf16x3_residual_relu_dropout_and_pair_store():
    """The (hi, lo) pair store of the bf16x3 epilogue (values leave as two bf16, exact to 2^-17), with shortcut + ReLU and
    with the head-tower dropout (same Philox contract: dropped elements are exactly zero in both halves)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network, philox
    rng = np.random.default_rng(18)
    b, h, w = 2, 9, 11
    x, wt, bias = _case(rng, b, h, w, 128, 256, 3)
    res = rng.normal(0, 1, (b, h, w, 256)).astype(np.float32)
    conv = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same")
    got = stage_conv(x, wt, bias, padding="same", relu=True, residual=res, precision="bf16x3", round_output_bf16=True)
    ref = np.maximum(conv + res, 0)
    rms = float(np.sqrt((ref ** 2).mean()))
    assert rel_err(got, ref, floor=rms) < 1e-4 and np.all(got >= 0)
    got = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=0.3, seed=11, layer_id=2, image_id=5, precision="bf16x3")
    keep = np.stack([philox.dropout_keep_mask(11, 5, s, 2, h * w, 256, 0.3).reshape(h, w, 256) for s in range(b)])
    ref = np.maximum(conv, 0) * np.float64(np.float32(1.0 / 0.7)) * keep
    assert np.all(got[~keep] == 0)
    assert rel_err(got, ref, floor=float(np.sqrt((ref ** 2).mean()))) < 1e-4
    # a 64-channel layer (one cout tile of the small configuration) and a strided 1x1 with shortcut
    x, wt, bias = _case(rng, 1, 12, 10, 64, 64, 3)
    got = stage_conv(x, wt, bias, padding="same", relu=True, precision="bf16x3", round_output_bf16=True)
    ref = np.maximum(network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same"), 0)
    assert rel_err(got, ref, floor=float(np.sqrt((ref ** 2).mean()))) < 1e-4


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding,split,res", [
    (2, 16, 16, 2048, 256, 3, 2, "same", 8, False),
    (1, 8, 8, 1024, 512, 1, 1, "valid", 4, True),
])
def test_bf16x3_split_k(monkeypatch, b, h, w, cin, cout, k, stride, padding, split, res):
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(cin + cout + split + 1)
    x, wt, bias = _case(rng, b, h, w, cin, cout, k)
    oh = -(-h // stride) if padding == "same" else (h - k) // stride + 1
    ow = -(-w // stride) if padding == "same" else (w - k) // stride + 1
    residual = rng.normal(0, 1, (b, oh, ow, cout)).astype(np.float32) if res else None
    monkeypatch.setenv("BOD_STAGE_KSPLIT", str(split))
    got = stage_conv(x, wt, bias, stride=stride, padding=padding, relu=res, residual=residual, round_output_bf16=True, precision="bf16x3")
    monkeypatch.delenv("BOD_STAGE_KSPLIT")
    ref = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), stride, padding)
    if res:
        ref = np.maximum(ref + residual, 0)
    assert rel_err(got, ref, floor=float(np.sqrt((ref ** 2).mean()))) < 1e-4


# ---------------------------------------------------------------------------------------------- f16mx precision (round 5)
def _tower_case(rng, b, h, w):
    """a head-tower layer's operands: ReLU'd, dropped-out activations (65 % zeros), he-normal weights"""
    x = np.maximum(rng.normal(0, 1, (b, h, w, 256)), 0).astype(np.float32) * (rng.random((b, h, w, 256)) >= 0.3).astype(np.float32) / np.float32(0.7)
    wt = (rng.normal(0, 1, (3, 3, 256, 256)) * np.sqrt(2.0 / (9 * 256))).astype(np.float32)
    bias = rng.normal(0, 0.5, 256).astype(np.float32)
    return x, wt, bias


# per-layer bounds of the two tower formats, max |d| / (|ref| + rms): [mode 0, modes 1 / 2] (CPU model of the arithmetic, tests/tools/
# tower_numerics.py: e2m3 cross terms 1.3e-5 rms / 7e-5 max of the output RMS, e2m1 5.7e-5 / 2.9e-4)
MX_TOL = {"f16mx": (1e-4, 2e-4), "f16mx4": (4.5e-4, 5e-4)}


@pytest.mark.parametrize("precision", ["f16mx", "f16mx4"])
@pytest.mark.parametrize("mode,b,h,w", [(0, 2, 16, 16), (0, 1, 21, 37), (1, 2, 16, 16), (1, 3, 9, 13), (2, 2, 16, 16), (2, 1, 30, 7)])
def test_f16mx_tower_layer_matches_oracle(mode, b, h, w, precision):
    """f16mx precision, one head-tower layer (multitask_headers.py:98-123: 3x3, 256 -> 256) on the f16mx kernel of the row-reuse loop
    against float64 on the UNROUNDED fp32 operands: x = f16 hi + lo, hi*hi on v_mfma_f32_32x32x16_f16 (exact products), the cross
    terms hi*lo + lo*hi as ONE block-scaled e2m3 product (v_mfma_scale_f32_32x32x64_f8f6f4; conv_igemm.hip header).  Per-layer
    error 1.3e-5 rms / 7e-5 max of the output RMS on the CPU model of the arithmetic (tests/tools/tower_numerics.py): 1e-4 here,
    the bf16x3 kernel's own gate.  mode 0: hx rows in, (hi, lo) pairs out (a head's last layer: exact to 2^-17); 1: hx in, hx out
    (the decoded hx row carries f16 hi + e2m3 lo: 2^-14); 2: pairs in -- the bf16x3 loop -- hx out (the first layer).  Ragged
    sizes: tiles with invalid slots, runs of x-adjacent pixels shorter than a tile.
    precision 'f16mx4': the same layer with h4 rows -- the cross terms as block-scaled e2m1 products of twice the channels (54 K-tiles
    instead of 72; scale bytes staged as their own LDS-DMA pieces, byte ks of a scale dword chosen by the MFMA's op_sel)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(100 * mode + h + w)
    x, wt, bias = _tower_case(rng, b, h, w)
    got = stage_conv(x, wt, bias, padding="same", relu=True, precision=precision, round_output_bf16=mode)
    ref = np.maximum(network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same"), 0)
    rms = float(np.sqrt((ref ** 2).mean()))
    err = rel_err(got, ref, floor=rms)
    print("%s mode %d %dx%dx%d: max |d| / (|ref| + rms) = %.2e" % (precision, mode, b, h, w, err))
    assert got.shape == ref.shape and np.all(got >= 0)
    assert err < MX_TOL[precision][0 if mode == 0 else 1]


@pytest.mark.parametrize("precision", ["f16mx", "f16mx4"])
def test_f16mx_tower_layer_dropout_and_bf16x3_agreement(precision):
    """The head-tower dropout in the hx epilogue (values are zeroed before the split: the Philox contract's decisions, exact
    zeros) and, on the same layer, agreement with the bf16x3 kernel (both within 1e-4 of float64, hence 2e-4 of each other)."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network, philox
    rng = np.random.default_rng(77)
    b, h, w = 3, 12, 20
    x, wt, bias = _tower_case(rng, b, h, w)
    conv = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same")
    keep = np.stack([philox.dropout_keep_mask(11, 5, s, 6, h * w, 256, 0.3).reshape(h, w, 256) for s in range(b)])
    ref = np.maximum(conv, 0) * np.float64(np.float32(1.0 / 0.7)) * keep
    rms = float(np.sqrt((ref ** 2).mean()))
    for mode in (0, 1, 2):
        got = stage_conv(x, wt, bias, padding="same", relu=True, dropout_rate=0.3, seed=11, layer_id=6, image_id=5, precision=precision, round_output_bf16=mode)
        assert np.all(got[~keep] == 0), mode
        assert rel_err(got, ref, floor=rms) < MX_TOL[precision][0 if mode == 0 else 1], mode
    a = stage_conv(x, wt, bias, padding="same", relu=True, precision=precision, round_output_bf16=0)
    c = stage_conv(x, wt, bias, padding="same", relu=True, precision="bf16x3", round_output_bf16=True)
    assert rel_err(a, c, floor=float(np.sqrt((c.astype(np.float64) ** 2).mean()))) < 2 * MX_TOL[precision][0]


@pytest.mark.parametrize("precision", ["f16mx", "f16mx4"])
def test_f16mx_extreme_magnitudes(precision):
    """Blocks of tiny values (f16-subnormal range: the lo part carries them), of large ones, all-zero blocks and mixed signs in the
    weights: the block scale follows the block's maximum, nothing saturates, zeros stay zeros."""
    from bayes_od_rc_amd.engine import stage_conv
    from oracle import network
    rng = np.random.default_rng(5)
    b, h, w = 1, 16, 16
    x, wt, bias = _tower_case(rng, b, h, w)
    scale = np.ones(256, np.float32)
    scale[0:16] = 3e-6; scale[16:32] = 1e-3; scale[64:96] = 900.0; scale[128:144] = 0.0       # per-channel magnitudes of the input
    x = x * scale
    wt = wt * rng.choice([1e-3, 1.0, 30.0], size=(1, 1, 256, 1)).astype(np.float32)
    ref = network.conv2d(x.astype(np.float64), wt.astype(np.float64), bias.astype(np.float64), 1, "same")
    rms = float(np.sqrt((ref ** 2).mean()))
    for mode in (0, 1):
        got = stage_conv(x, wt, bias, padding="same", precision=precision, round_output_bf16=mode)
        assert rel_err(got, ref, floor=rms) < (2e-4 if precision == "f16mx" else 6e-4), mode


@pytest.mark.parametrize("ch", [64, 128])
def test_sliding_window_3x3_kernels_on_single_layers(ch):
    """The sliding-window 3x3 kernels of the backbone (conv_pointwise.hip: 64 -> 64 of stage 2, 128 -> 128 of stage 3 -- the latter
    splits the input channels over two waves and adds the halves) against the ORACLE conv and against the generic kernel, one
    layer at a time: strips of 64 pixels that are full, ragged (70 = 64 + 6) and several per row (130), one-row and tall planes,
    bias + ReLU, bf16 stores.  Against the oracle: bf16 rounding of the output (2^-9 relative).  Against the generic launch
    (BOD_SLIDE3X3=0): identical for 64 channels (same k order); for 128 channels the one re-associated fp32 addition may move
    a result across a rounding boundary -- at most one bf16 ulp, on a small fraction of the elements.
    (BOD_POINTWISE_MIN_M=1 sends these small shapes down the streaming kernels.)"""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shapes = [(2, 5, 64), (1, 1, 70), (2, 9, 130), (3, 33, 32), (1, 64, 64)]
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from bayes_od_rc_amd.engine import stage_conv\n"
            "out = {}\n"
            "for i, (b, h, w) in enumerate(%r):\n"
            "    rng = np.random.default_rng(100 + i)\n"
            "    x = rng.normal(0, 1, (b, h, w, %d)).astype(np.float32)\n"
            "    wt = (rng.normal(0, 1, (3, 3, %d, %d)) * np.sqrt(2.0 / (9 * %d))).astype(np.float32)\n"
            "    bias = rng.normal(0, 0.5, %d).astype(np.float32)\n"
            "    out['y%%d' %% i] = stage_conv(x, wt, bias, relu=True, round_output_bf16=True)\n"
            "np.savez(sys.argv[1], **out)\n" % (root, shapes, ch, ch, ch, ch, ch))
    outs = []
    for on in ("1", "0"):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npz")
            env = dict(os.environ, BOD_SLIDE3X3=on, BOD_POINTWISE_MIN_M="1")
            r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-3000:]
            z = np.load(path)
            outs.append({k: z[k] for k in z.files})
    for i, (b, h, w) in enumerate(shapes):
        rng = np.random.default_rng(100 + i)
        x = rng.normal(0, 1, (b, h, w, ch)).astype(np.float32)
        wt = (rng.normal(0, 1, (3, 3, ch, ch)) * np.sqrt(2.0 / (9 * ch))).astype(np.float32)
        bias = rng.normal(0, 0.5, ch).astype(np.float32)
        ref = _oracle(x, wt, bias, 1, "same", relu=True)
        slide, generic = outs[0]["y%d" % i].astype(np.float64), outs[1]["y%d" % i].astype(np.float64)
        assert slide.shape == ref.shape
        rms = float(np.sqrt((ref ** 2).mean()))
        assert rel_err(slide, ref, floor=rms) < 2.0 ** -8, (i, rel_err(slide, ref, floor=rms))
        if ch == 64:
            assert np.array_equal(slide, generic), i
        else:
            # one bf16 ulp is at most 2^-7 of the value; sums that cancel to ~0 may land on either side of the ReLU (fp32 round-off of O(1) terms)
            ulp = np.maximum(np.abs(generic), np.abs(slide)) * 2.0 ** -7 + 1e-5
            assert (np.abs(slide - generic) <= ulp).all(), (i, float(np.abs(slide - generic).max()))
            assert (slide != generic).mean() < 0.02, (i, float((slide != generic).mean()))
