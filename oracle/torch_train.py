"""TEST INFRASTRUCTURE (oracle): the training step of run_training.train_single_step (:208-247) in PyTorch on the
CPU under autograd -- oracle/torch_ref.py's forward (literal BatchNorm in inference mode, Philox dropout masks
injected), RetinaNetModel.get_loss (retinanet_model.py:183-323, core/losses.py:30-61), Keras l2 on the header tower
kernels and cov_out, tf.clip_by_global_norm(5.0), keras Adam(epsilon = 1e-2)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox, torch_ref
from .network import BN_EPS, HEAD_ID, HEAD_NUM_CONVS, HEAD_PREFIX, stages_for

TRAINABLE_BN = (0, 1)        # gamma, beta (moving mean / variance are not trained)


def prepare(weights, dtype=torch.float64):
    tw, leaves = {}, {}
    for name, e in weights.items():
        if "kernel" in e:
            w = torch.tensor(np.transpose(e["kernel"], (3, 2, 0, 1)), dtype=dtype, requires_grad=True)
            b = torch.tensor(e["bias"], dtype=dtype, requires_grad=True) if e.get("bias") is not None else None
            tw[name] = (w, b)
            leaves[name + "/kernel"] = w
            if b is not None:
                leaves[name + "/bias"] = b
        else:
            g = torch.tensor(e["gamma"], dtype=dtype, requires_grad=True)
            be = torch.tensor(e["beta"], dtype=dtype, requires_grad=True)
            tw[name] = (g, be, torch.tensor(e["mean"], dtype=dtype), torch.tensor(e["var"], dtype=dtype))
            leaves[name + "/gamma"], leaves[name + "/beta"] = g, be
    return tw, leaves


def forward(tw, images, seed, first_image_id, num_classes, rate=0.3, dtype=torch.float64, emulate_bf16=False):
    """images [B,H,W,3] -> (cls [B,A,C], box [B,A,4], cov [B,A,10]) torch tensors (graph attached).  emulate_bf16: the
    device's storage roundings in the graph (see the section above); the literal float64 network otherwise."""
    x = torch.tensor(np.asarray(images), dtype=dtype)
    pyr = backbone_fpn_bf16(tw, x) if emulate_bf16 else torch_ref.backbone_fpn(tw, x)
    sizes = [p.shape[2] * p.shape[3] for p in pyr]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    ptotal = int(offs[-1])
    scale = float(np.float32(1.0 / (1.0 - rate)))
    b = x.shape[0]
    outs = {}
    for head, c_out in (("cls", num_classes), ("reg", 4), ("cov", 10)):
        per_level = []
        for li, lvl in enumerate(pyr):
            y = lvl
            _, ch, h, w = y.shape
            for layer in range(HEAD_NUM_CONVS[head]):
                lid = HEAD_ID[head] * 4 + layer
                keep = np.stack([philox.dropout_keep_mask(seed, first_image_id + i, 0, lid, ptotal, 256, rate)[offs[li]:offs[li + 1]]
                                 for i in range(b)])
                ks = scale * torch.tensor(keep.reshape(b, h, w, ch)).permute(0, 3, 1, 2)
                if emulate_bf16:
                    y = _econv(y, tw, "%s_%d" % (HEAD_PREFIX[head], layer), same=True, relu=True, keep_scale=ks)
                else:
                    y = F.relu(torch_ref.conv(y, tw, "%s_%d" % (HEAD_PREFIX[head], layer), 1, True)) * ks
            if emulate_bf16:
                z = _econv(y, tw, HEAD_PREFIX[head], same=True, store=False).permute(0, 2, 3, 1)
            else:
                z = torch_ref.conv(y, tw, HEAD_PREFIX[head], 1, True).permute(0, 2, 3, 1)
            per_level.append(z.reshape(b, h * w * (z.shape[-1] // c_out), c_out))
        outs[head] = torch.cat(per_level, dim=1)
    return outs["cls"], outs["reg"], outs["cov"]


# ----------------------------------------------------------------------------- bf16-storage emulation of the device step
# The device's training step keeps fp32 master parameters but computes like its bf16 inference mode: BatchNorm folded into
# the kernels and the folded kernels rounded to bf16 every step, every activation stored as bf16 (the head outputs stay
# fp32), fp32 accumulation; in the backward pass the gradient with respect to a layer's pre-activation (dZ = dOut * mask *
# dropout scale) is rounded to bf16 before it feeds the weight-gradient, bias-gradient and input-gradient products, while
# activation gradients (and the residual branch's share) stay fp32.  The functions below put exactly those roundings into a
# float64 autograd graph, so that what separates this from the device is fp32 summation order (and the rare bf16 flip it
# causes) -- not bf16 storage itself, which costs ~20 % per-tensor error against the literal float64 step.
def _bf16(t):
    f = t.detach().to(torch.float32).contiguous()
    u = f.view(torch.int32)
    r = (u + 0x7FFF + ((u >> 16) & 1)) & ~0xFFFF
    return r.view(torch.float32).to(t.dtype)


class _RoundValue(torch.autograd.Function):          # bf16 storage of a value; the gradient passes through
    @staticmethod
    def forward(ctx, x):
        return _bf16(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradient(torch.autograd.Function):       # identity; the gradient arriving here is what the device stores as bf16 dZ
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf16(g)


def _econv(x, tw, name, bn=None, stride=1, same=False, relu=False, residual=None, keep_scale=None, store=True):
    """One device layer: out = store_bf16(act(conv(x, bf16(fold(W, BN))) + b' + residual) [* dropout])."""
    w, b = tw[name]
    if bn is not None:
        g, be, mu, var = tw[bn]
        s = g / torch.sqrt(var + BN_EPS)
        w = w * s[:, None, None, None]
        b = ((b if b is not None else 0.0) - mu) * s + be
    if same:
        pt, pb = torch_ref._same_pads(x.shape[2], w.shape[2], stride)
        pl, pr = torch_ref._same_pads(x.shape[3], w.shape[3], stride)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
    z = _RoundGradient.apply(F.conv2d(x, _RoundValue.apply(w), b, stride=stride))
    if residual is not None:
        z = z + residual
    if relu:
        z = F.relu(z)
    if keep_scale is not None:
        z = z * keep_scale
    return _RoundValue.apply(z) if store else z


def backbone_fpn_bf16(tw, image_nhwc):
    """torch_ref.backbone_fpn with the device's bf16 storage points (see above); same reference lines."""
    x = _bf16(image_nhwc.permute(0, 3, 1, 2))
    x = _econv(x, tw, "conv1", "bn_conv1", 2, relu=True)
    x = F.max_pool2d(F.pad(x, (2, 2, 1, 1)), 3, 2)
    taps = {}
    for stage, blocks, fs in stages_for(tw):
        for blk in blocks:
            cb, bb = "res%d%s_branch" % (stage, blk), "bn%d%s_branch" % (stage, blk)
            s = fs if blk == "a" else 1
            y = _econv(x, tw, cb + "2a", bb + "2a", s, relu=True)
            y = _econv(y, tw, cb + "2b", bb + "2b", 1, True, relu=True)
            sc = _econv(x, tw, cb + "1", bb + "1", s) if blk == "a" else x
            x = _econv(y, tw, cb + "2c", bb + "2c", relu=True, residual=sc)
            if blk == "a":
                taps[stage] = x
    c5, c4, c3 = x, taps[4], taps[3]
    c5r = _econv(c5, tw, "C5_reduced")
    p5 = _econv(c5r, tw, "P5", same=True)
    p6 = _econv(c5, tw, "P6", stride=2, same=True)
    p7 = _econv(F.relu(p6), tw, "P7", stride=2, same=True)

    def up(src, hw):
        h, w = hw
        ys = torch.clamp(torch.floor((torch.arange(h) + 0.5) * (src.shape[2] / h)).long(), max=src.shape[2] - 1)
        xs = torch.clamp(torch.floor((torch.arange(w) + 0.5) * (src.shape[3] / w)).long(), max=src.shape[3] - 1)
        return src[:, :, ys][:, :, :, xs]

    m4 = _econv(c4, tw, "C4_reduced", residual=up(c5r, c4.shape[2:]))
    p4 = _econv(m4, tw, "P4", same=True)
    m3 = _econv(c3, tw, "C3_reduced", residual=up(m4, c3.shape[2:]))
    p3 = _econv(m3, tw, "P3", same=True)
    return [p3, p4, p5, p6, p7]


def total_loss(cls, box, cov, cls_t, box_t, anchors, pos, neg, reg_kind=3, eps=0.001, w_cls=5.0, w_reg=1.0):
    t = lambda v: torch.tensor(np.asarray(v), dtype=cls.dtype)
    cls_t, box_t, anc, posm, negm = t(cls_t), t(box_t), t(anchors), t(pos), t(neg)
    npos = torch.clamp(posm.sum(), min=1.0)
    c = cls.shape[-1]
    ls = torch.log_softmax(cls, dim=-1)
    ce = -((cls_t * (1 - eps) + eps / c) * ls).sum(-1)
    pt = (torch.softmax(cls, -1) * cls_t).sum(-1)
    cls_loss = w_cls * (0.5 * (1 - pt) ** 2 * ce * (posm + negm)).sum() / npos

    def decode(tg):
        return torch.stack([anc[:, 2] * tg[..., 0] / 10 + anc[:, 0], anc[:, 3] * tg[..., 1] / 10 + anc[:, 1],
                            anc[:, 2] * torch.clamp(torch.exp(tg[..., 2] / 5), 1e-4, 1e4),
                            anc[:, 3] * torch.clamp(torch.exp(tg[..., 3] / 5), 1e-4, 1e4)], -1)
    if reg_kind == 1:
        reg = (F.huber_loss(box, box_t, reduction="none", delta=1.0).mean(-1) * posm).sum() / npos
        covl = torch.zeros((), dtype=cls.dtype)
    else:
        e = F.huber_loss(decode(box), decode(box_t), reduction="none", delta=1.0)
        ld = torch.stack([cov[..., 4], cov[..., 9], cov[..., 5], cov[..., 0]], -1)
        cmp = (torch.exp(-ld) * e).sum(-1)
        if reg_kind == 3:
            off = torch.stack([cov[..., k] for k in (8, 7, 6, 3, 2, 1)], -1)
            cmp = cmp * torch.sqrt(4.0 + (off ** 2).sum(-1))
        reg = (cmp * posm).sum() / npos
        covl = (0.5 * ld.sum(-1) * posm).sum() / npos
    return cls_loss + w_reg * (reg + covl), {"cls_loss": cls_loss, "reg_loss": reg, "covariance_loss": covl}


def l2_loss(leaves, rate):
    tot = 0.0
    for name, w in leaves.items():
        layer, kind = name.rsplit("/", 1)
        tower = layer.startswith("pyramid_") and layer[-2] == "_"
        # RegHeader constructs conv_4 but never calls it (a4): Keras never builds it, so it has no variables and no loss
        if layer == "pyramid_regression_3":
            continue
        if kind == "kernel" and (tower or layer == "pyramid_cov"):
            tot = tot + rate * (w ** 2).sum()
    return tot


def train_step(weights, images, cls_t, box_t, anchors, pos, neg, seed=0, first_image_id=0, reg_kind=3, eps=0.001, w_cls=5.0,
               w_reg=1.0, l2_rate=1e-6, lr=1e-3, adam_state=None, step=1, dtype=torch.float64, emulate_bf16=False):
    """One step.  Returns (losses dict, grads {name: ndarray in the build's layout (HWIO kernels)}, new weights dict)."""
    tw, leaves = prepare(weights, dtype)
    ncls = np.asarray(cls_t).shape[-1]
    cls, box, cov = forward(tw, images, seed, first_image_id, ncls, dtype=dtype, emulate_bf16=emulate_bf16)
    loss, parts = total_loss(cls, box, cov, cls_t, box_t, anchors, pos, neg, reg_kind, eps, w_cls, w_reg)
    reg = l2_loss(leaves, l2_rate)
    total = loss + reg
    total.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    gnorm = torch.sqrt(sum((g ** 2).sum() for g in grads.values()))
    scale = 5.0 / max(float(gnorm), 5.0)
    new, state = {}, adam_state or {}
    b1, b2, aeps = 0.9, 0.999, 1e-2
    lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
    for k, v in leaves.items():
        g = grads[k] * scale
        m, vv = state.get(k, (torch.zeros_like(v), torch.zeros_like(v)))
        m = b1 * m + (1 - b1) * g
        vv = b2 * vv + (1 - b2) * g * g
        state[k] = (m, vv)
        new[k] = (v.detach() - lr_t * m / (torch.sqrt(vv) + aeps))

    def hwio(name, tsr):
        a = tsr.detach().numpy()
        return np.transpose(a, (2, 3, 1, 0)) if name.endswith("/kernel") else a
    losses = {"total_loss": float(total.detach()), "cls_loss": float(parts["cls_loss"].detach()), "reg_loss": float(parts["reg_loss"].detach()),
              "covariance_loss": float(parts["covariance_loss"].detach()), "regularization_loss": float(reg.detach()), "grad_norm": float(gnorm)}
    return losses, {k: hwio(k, g) for k, g in grads.items()}, {k: hwio(k, w) for k, w in new.items()}, state
