"""TEST INFRASTRUCTURE (oracle): the training step of run_training.train_single_step (:208-247) in PyTorch on the
CPU under autograd -- oracle/torch_ref.py's forward (literal BatchNorm in inference mode, Philox dropout masks
injected), RetinaNetModel.get_loss (retinanet_model.py:183-323, core/losses.py:30-61), Keras l2 on the header tower
kernels and cov_out, tf.clip_by_global_norm(5.0), keras Adam(epsilon = 1e-2)."""
import numpy as np
import torch
import torch.nn.functional as F

from . import philox, torch_ref
from .network import HEAD_ID, HEAD_NUM_CONVS, HEAD_PREFIX

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


def forward(tw, images, seed, first_image_id, num_classes, rate=0.3, dtype=torch.float64):
    """images [B,H,W,3] -> (cls [B,A,C], box [B,A,4], cov [B,A,10]) torch tensors (graph attached)."""
    x = torch.tensor(np.asarray(images), dtype=dtype)
    pyr = torch_ref.backbone_fpn(tw, x)
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
                y = F.relu(torch_ref.conv(y, tw, "%s_%d" % (HEAD_PREFIX[head], layer), 1, True))
                lid = HEAD_ID[head] * 4 + layer
                keep = np.stack([philox.dropout_keep_mask(seed, first_image_id + i, 0, lid, ptotal, 256, rate)[offs[li]:offs[li + 1]]
                                 for i in range(b)])
                y = y * scale * torch.tensor(keep.reshape(b, h, w, ch)).permute(0, 3, 1, 2)
            z = torch_ref.conv(y, tw, HEAD_PREFIX[head], 1, True).permute(0, 2, 3, 1)
            per_level.append(z.reshape(b, h * w * (z.shape[-1] // c_out), c_out))
        outs[head] = torch.cat(per_level, dim=1)
    return outs["cls"], outs["reg"], outs["cov"]


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
               w_reg=1.0, l2_rate=1e-6, lr=1e-3, adam_state=None, step=1, dtype=torch.float64):
    """One step.  Returns (losses dict, grads {name: ndarray in the build's layout (HWIO kernels)}, new weights dict)."""
    tw, leaves = prepare(weights, dtype)
    ncls = np.asarray(cls_t).shape[-1]
    cls, box, cov = forward(tw, images, seed, first_image_id, ncls, dtype=dtype)
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
