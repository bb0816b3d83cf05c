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


def _torch_loss(cls, cls_t, box, box_t, cov, anchors, pos, neg, reg_kind, eps, w_cls, w_reg):
    """The reference's total loss (retinanet_model.py:183-323, core/losses.py:30-61) in torch float64."""
    import torch
    t = lambda x: torch.tensor(np.asarray(x), dtype=torch.float64)
    cls, box, cov = t(cls).requires_grad_(True), t(box).requires_grad_(True), t(cov).requires_grad_(True)
    cls_t, box_t, anc = t(cls_t), t(box_t), t(anchors)
    posm, negm = t(pos), t(neg)
    npos = torch.clamp(posm.sum(), min=1.0)
    c = cls.shape[-1]
    ls = torch.log_softmax(cls, dim=-1)
    q = cls_t * (1 - eps) + eps / c
    ce = -(q * ls).sum(-1)
    pt = (torch.softmax(cls, -1) * cls_t).sum(-1)
    focal = 0.5 * (1 - pt) ** 2 * ce
    total = w_cls * (focal * (posm + negm)).sum() / npos

    def decode(tg):
        return torch.stack([anc[:, 2] * tg[..., 0] / 10 + anc[:, 0], anc[:, 3] * tg[..., 1] / 10 + anc[:, 1],
                            anc[:, 2] * torch.clamp(torch.exp(tg[..., 2] / 5), 1e-4, 1e4),
                            anc[:, 3] * torch.clamp(torch.exp(tg[..., 3] / 5), 1e-4, 1e4)], -1)
    hub = torch.nn.functional.huber_loss
    if reg_kind == 1:
        l = hub(box, box_t, reduction="none", delta=1.0).mean(-1)
        total = total + w_reg * (l * posm).sum() / npos
    else:
        e = hub(decode(box), decode(box_t), reduction="none", delta=1.0)
        ld = torch.stack([cov[..., 4], cov[..., 9], cov[..., 5], cov[..., 0]], -1)
        cmp = (torch.exp(-ld) * e).sum(-1)
        if reg_kind == 3:
            off = torch.stack([cov[..., k] for k in (8, 7, 6, 3, 2, 1)], -1)
            cmp = cmp * torch.sqrt(4.0 + (off ** 2).sum(-1))
        total = total + w_reg * ((cmp + 0.5 * ld.sum(-1)) * posm).sum() / npos
    total.backward()
    z = lambda v: v.grad.numpy() if v.grad is not None else np.zeros(tuple(v.shape))      # reg_kind 1 does not touch cov
    return z(cls), z(box), z(cov)


@pytest.mark.parametrize("reg_kind", [1, 2, 3])
def test_loss_backward_matches_autograd(reg_kind):
    import ctypes as C
    from bayes_od_rc_amd import _lib
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    from conftest import ANCHOR_CFG
    rng = np.random.default_rng(reg_kind)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((64, 96, 3)).astype(np.float32)
    b, a, c = 2, anchors.shape[0], 8
    cls = rng.normal(0, 2, (b, a, c)).astype(np.float32)
    cls_t = np.eye(c, dtype=np.float32)[rng.integers(0, c, (b, a))]
    box = rng.normal(0, 1.5, (b, a, 4)).astype(np.float32)
    box_t = rng.normal(0, 1.5, (b, a, 4)).astype(np.float32)
    cov = rng.normal(0, 0.7, (b, a, 10)).astype(np.float32)
    pos = (rng.uniform(size=(b, a)) < 0.2).astype(np.uint8)
    neg = ((rng.uniform(size=(b, a)) < 0.6) & (pos == 0)).astype(np.uint8)
    lib = _lib.load()
    out4 = (C.c_double * 4)()
    dcls, dbox, dcov = np.empty_like(cls), np.empty_like(box), np.empty_like(cov)
    u8 = C.POINTER(C.c_uint8)
    st = lib.bod_loss_backward(0, b, a, c, _lib.fptr(cls), _lib.fptr(cls_t), _lib.fptr(box), _lib.fptr(box_t), _lib.fptr(cov),
                               _lib.fptr(anchors), pos.ctypes.data_as(u8), neg.ctypes.data_as(u8), 1, reg_kind, 0.001, 5.0, 1.0,
                               out4, _lib.fptr(dcls), _lib.fptr(dbox), _lib.fptr(dcov))
    _lib.check(lib, None, st)
    r_cls, r_box, r_cov = _torch_loss(cls, cls_t, box, box_t, cov, anchors, pos, neg, reg_kind, 0.001, 5.0, 1.0)
    for got, ref, name in ((dcls, r_cls, "cls"), (dbox, r_box, "box"), (dcov, r_cov, "cov")):
        rms = float(np.sqrt((ref ** 2).mean()))
        if rms == 0.0:
            assert not got.any(), name
        else:
            assert rel_err(got, ref, floor=rms) < 1e-4, name
