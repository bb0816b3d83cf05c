"""GPU: building blocks of the training step -- the weight gradient of a Conv2D as a
pixel-reduction GEMM on the forward MFMA kernel (bod_stage_conv_wgrad) and the input gradient as the forward
kernel on flipped / swapped weights -- against torch.autograd on the CPU in float64, on identical bf16-rounded
operands.  Tolerance: 1e-3 of the gradient's RMS (fp32 accumulation over up to ~1e5 pixels)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def _torch_grads(x, w, dy, stride, padding):
    """Keras Conv2D (SAME pad before = total // 2) under autograd, float64."""
    import torch
    import torch.nn.functional as F
    xt = torch.tensor(x, dtype=torch.float64).permute(0, 3, 1, 2).requires_grad_(True)
    wt = torch.tensor(w, dtype=torch.float64).permute(3, 2, 0, 1).requires_grad_(True)
    bt = torch.zeros(w.shape[3], dtype=torch.float64, requires_grad=True)
    kh, kw = w.shape[:2]
    if padding == "same":
        def pads(n, k):
            out = -(-n // stride)
            tot = max((out - 1) * stride + k - n, 0)
            return tot // 2, tot - tot // 2
        (pt, pb), (pl, pr) = pads(x.shape[1], kh), pads(x.shape[2], kw)
        xp = F.pad(xt, (pl, pr, pt, pb))
    else:
        xp = xt
    y = F.conv2d(xp, wt, bt, stride=stride)
    g = torch.tensor(dy, dtype=torch.float64).permute(0, 3, 1, 2)
    assert tuple(y.shape) == tuple(g.shape)
    y.backward(g)
    return (xt.grad.permute(0, 2, 3, 1).numpy(), wt.grad.permute(2, 3, 1, 0).numpy(), bt.grad.numpy())


WGRAD_CASES = [
    # b, h, w, cin, cout, k, stride, padding, ksplit
    (2, 16, 16, 256, 256, 3, 1, "same", 0),        # head-tower layer
    (1, 20, 24, 64, 64, 3, 1, "same", 4),          # stage-2 3x3, fixed split
    (2, 12, 12, 64, 256, 1, 1, "valid", 0),        # bottleneck expand 1x1
    (1, 15, 17, 256, 128, 1, 2, "valid", 0),       # strided 1x1 (odd input)
    (1, 16, 16, 128, 256, 3, 2, "same", 2),        # P6-style stride 2 SAME (pad 0/1)
    (1, 9, 13, 64, 72, 1, 1, "same", 1),           # head output conv: cout not a multiple of 64, no split
    (1, 33, 31, 3, 64, 3, 2, "valid", 0),          # image-like input: cin = 3 (rows (tap, ci) = 27 + 1)
]


@pytest.mark.parametrize("b,h,w,cin,cout,k,stride,padding,ksplit", WGRAD_CASES)
def test_wgrad_matches_autograd(b, h, w, cin, cout, k, stride, padding, ksplit):
    from bayes_od_rc_amd.engine import stage_conv_wgrad
    from oracle import network
    rng = np.random.default_rng(cin + 3 * cout + k + h)
    x = rng.normal(0, 1, (b, h, w, cin)).astype(np.float32)
    oh = -(-h // stride) if padding == "same" else (h - k) // stride + 1
    ow = -(-w // stride) if padding == "same" else (w - k) // stride + 1
    dy = rng.normal(0, 1, (b, oh, ow, cout)).astype(np.float32)
    dw, db = stage_conv_wgrad(x, dy, (k, k), stride=stride, padding=padding, ksplit=ksplit)
    wz = np.zeros((k, k, cin, cout), np.float32)
    _, ref_dw, ref_db = _torch_grads(network.bf16_round(x), wz, network.bf16_round(dy), stride, padding)
    assert dw.shape == ref_dw.shape and db.shape == ref_db.shape
    assert rel_err(dw, ref_dw, floor=float(np.sqrt((ref_dw ** 2).mean()))) < 1e-3
    assert rel_err(db, ref_db, floor=float(np.sqrt((ref_db ** 2).mean()))) < 1e-3


@pytest.mark.parametrize("b,h,w,cin,cout,k", [(2, 16, 16, 256, 256, 3), (1, 9, 13, 64, 128, 3), (2, 12, 12, 256, 64, 1)])
def test_dgrad_is_the_forward_kernel_on_flipped_weights(b, h, w, cin, cout, k):
    from bayes_od_rc_amd.engine import stage_conv_dgrad
    from oracle import network
    rng = np.random.default_rng(cin + cout + k)
    x = np.zeros((b, h, w, cin), np.float32)
    wt = (rng.normal(0, 1, (k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(np.float32)
    dy = rng.normal(0, 1, (b, h, w, cout)).astype(np.float32)
    dx = stage_conv_dgrad(dy, wt)
    ref_dx, _, _ = _torch_grads(x, network.bf16_round(wt), network.bf16_round(dy), 1, "same")
    assert dx.shape == ref_dx.shape
    assert rel_err(dx, ref_dx, floor=float(np.sqrt((ref_dx ** 2).mean()))) < 1e-3
