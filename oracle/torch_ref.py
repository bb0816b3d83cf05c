"""Second, independent CPU restatement of the forward pass in PyTorch fp32 (oneDNN convolutions
with EXPLICIT padding, NCHW) -- TEST INFRASTRUCTURE and the ``cpu_baseline`` leg of bench.py.

Purpose: (1) cross-check oracle/network.py (different code path: F.conv2d + F.pad vs per-tap
matmuls) because the TF half of the reference cannot be run here ("parity unpinned");
(2) a reference-literal CPU timing of the same pipeline on the host cores of the GPU box:
backbone + FPN once, then ALL 11 head convs on the N-times tiled pyramid
(src/retina_net/models/retinanet_model.py:78-109), dropout from the framework RNG like
``keras.layers.Dropout`` (no dedup of the first tower conv -- that is what the reference does).

Follows the same reference files as oracle/network.py.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .network import BN_EPS, HEAD_NUM_CONVS, HEAD_PREFIX, HEAD_ID, stages_for, fill_triangular_4


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def prepare(weights):
    """HWIO -> OIHW torch tensors; BN kept separate (literal)."""
    out = {}
    for name, e in weights.items():
        if "kernel" in e:
            out[name] = (_t(np.transpose(e["kernel"], (3, 2, 0, 1))),
                         _t(e["bias"]) if e.get("bias") is not None else None)
        else:
            out[name] = tuple(_t(e[k]) for k in ("gamma", "beta", "mean", "var"))
    return out


def _same_pads(n, k, s):
    o = -(-n // s)
    total = max((o - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv(x, tw, name, stride=1, same=False):
    w, b = tw[name]
    if same:
        pt, pb = _same_pads(x.shape[2], w.shape[2], stride)
        pl, pr = _same_pads(x.shape[3], w.shape[3], stride)
        if pt or pb or pl or pr:
            x = F.pad(x, (pl, pr, pt, pb))
    return F.conv2d(x, w, b, stride=stride)


def bn(x, tw, name):
    g, be, mu, var = tw[name]
    return F.batch_norm(x, mu, var, g, be, training=False, eps=BN_EPS)


def backbone_fpn(tw, image_nhwc):
    x = (image_nhwc if torch.is_tensor(image_nhwc) else _t(image_nhwc)).permute(0, 3, 1, 2)
    x = F.relu(bn(conv(x, tw, "conv1", 2), tw, "bn_conv1"))
    x = F.max_pool2d(F.pad(x, (2, 2, 1, 1)), 3, 2)
    taps = {}
    for stage, blocks, fs in stages_for(tw):
        for blk in blocks:
            cb, bb = "res%d%s_branch" % (stage, blk), "bn%d%s_branch" % (stage, blk)
            s = fs if blk == "a" else 1
            y = F.relu(bn(conv(x, tw, cb + "2a", s), tw, bb + "2a"))
            y = F.relu(bn(conv(y, tw, cb + "2b", 1, True), tw, bb + "2b"))
            y = bn(conv(y, tw, cb + "2c"), tw, bb + "2c")
            sc = bn(conv(x, tw, cb + "1", s), tw, bb + "1") if blk == "a" else x
            x = F.relu(y + sc)
            if blk == "a":
                taps[stage] = x
    c5, c4, c3 = x, taps[4], taps[3]
    c5r = conv(c5, tw, "C5_reduced")
    p5 = conv(c5r, tw, "P5", 1, True)
    p6 = conv(c5, tw, "P6", 2, True)
    p7 = conv(F.relu(p6), tw, "P7", 2, True)

    def up(src, like):
        h, w = like.shape[2], like.shape[3]
        ys = torch.clamp(torch.floor((torch.arange(h) + 0.5) * (src.shape[2] / h)).long(), max=src.shape[2] - 1)
        xs = torch.clamp(torch.floor((torch.arange(w) + 0.5) * (src.shape[3] / w)).long(), max=src.shape[3] - 1)
        return src[:, :, ys][:, :, :, xs]

    c4r = conv(c4, tw, "C4_reduced")
    m4 = up(c5r, c4r) + c4r
    p4 = conv(m4, tw, "P4", 1, True)
    c3r = conv(c3, tw, "C3_reduced")
    m3 = up(m4, c3r) + c3r
    p3 = conv(m3, tw, "P3", 1, True)
    return [p3, p4, p5, p6, p7]


def heads(tw, pyramid, n_samples, out_channels, rate=0.3, keep_masks=None, generator=None):
    """Returns dict head -> [N, A, c] numpy.  keep_masks(sample, layer_id) -> bool [P,256] injects
    the Philox masks; otherwise torch's RNG draws them (baseline timing)."""
    sizes = [p.shape[2] * p.shape[3] for p in pyramid]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    mc = n_samples > 1
    scale = float(np.float32(1.0 / (1.0 - rate)))
    res = {}
    for head, c_out in out_channels.items():
        outs = []
        for li, lvl in enumerate(pyramid):
            x = lvl.repeat(n_samples, 1, 1, 1)
            _, ch, h, w = x.shape
            for layer in range(HEAD_NUM_CONVS[head]):
                x = F.relu(conv(x, tw, "%s_%d" % (HEAD_PREFIX[head], layer), 1, True))
                if mc:
                    if keep_masks is not None:
                        lid = HEAD_ID[head] * 4 + layer
                        k = np.stack([keep_masks(n, lid)[offs[li]:offs[li + 1]] for n in range(n_samples)])
                        keep = torch.from_numpy(k.reshape(n_samples, h, w, ch)).permute(0, 3, 1, 2)
                    else:
                        keep = torch.rand(x.shape, generator=generator) >= rate
                    x = x * scale * keep
            y = conv(x, tw, HEAD_PREFIX[head], 1, True).permute(0, 2, 3, 1)
            outs.append(y.reshape(n_samples, h * w * (y.shape[-1] // c_out), c_out))
        res[head] = torch.cat(outs, dim=1).numpy()
    return res


def retinanet_forward(weights, image, n_samples, num_classes_with_bknd, keep_masks=None, rate=0.3,
                      prepared=None, threads=None):
    if threads:
        torch.set_num_threads(threads)
    tw = prepared if prepared is not None else prepare(weights)
    with torch.no_grad():
        pyr = backbone_fpn(tw, image)
        h = heads(tw, pyr, n_samples, {"cls": num_classes_with_bknd, "reg": 4, "cov": 10}, rate, keep_masks)
    return {"anchors_class_predictions": h["cls"], "anchors_box_predictions": h["reg"],
            "anchors_box_covar_predictions": fill_triangular_4(h["cov"]), "_covar_params": h["cov"],
            "_pyramid": [p.permute(0, 2, 3, 1).numpy() for p in pyr]}
