"""Oracle restatement of the RetinaNet forward.

Follows the reference files
  src/retina_net/models/feature_extractor.py   (ResNet-50, frozen BN)          :104-139,:195-213,:283-309
  src/retina_net/models/feature_decoder.py     (FPN)                            :136-171
  src/retina_net/models/multitask_headers.py   (cls / reg / cov towers)         :98-123,:209-230,:318-342
  src/retina_net/models/retinanet_model.py     (MC tiling, fill_triangular)     :67-112
and the TF/Keras op semantics of SURVEY.md App. A (SAME/VALID padding placement, BN eps 1e-3,
ZeroPadding2D((1,2)), half-pixel nearest resize, dropout keep/scale).

Two numerics modes:
  * ``literal``  -- separate conv / BN / add / ReLU in ``dtype`` (float64 = ground truth,
                    float32 = what the TF reference computes).
  * ``bf16``     -- emulates the HIP path's storage precision exactly: BN folded into the
                    conv (float64 fold -> float32), folded weights rounded to bf16 (the stem's too;
                    its fp32 pixels are NOT rounded: the device splits them hi + lo), activations rounded to bf16 (RNE) at every point where the device
                    stores them, fp32 accumulation.  See DESIGN.md "Numerics".
Weights: dict  keras_layer_name -> {"kernel": HWIO, "bias": [O]}  /  BN name ->
{"gamma","beta","mean","var"}.
"""
import numpy as np

BN_EPS = 1e-3          # keras BatchNormalization default (feature_extractor.py:30 passes none)

HEADS = ("cls", "reg", "cov")
HEAD_PREFIX = {"cls": "pyramid_classification", "reg": "pyramid_regression", "cov": "pyramid_cov"}
HEAD_NUM_CONVS = {"cls": 4, "reg": 3, "cov": 4}     # RegHeader.call never calls conv_4 (:209-230)
HEAD_ID = {"cls": 0, "reg": 1, "cov": 2}


# ----------------------------------------------------------------------------- bf16
def bf16_round(x):
    """float32 -> nearest-even bfloat16 -> float32 (NaN/inf untouched for finite inputs here)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


# ----------------------------------------------------------------------------- primitive ops
def _same_pads(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2, out


def conv2d(x, w, b, stride=1, padding="valid"):
    """TF Conv2D, NHWC x HWIO (App. A.1). Accumulates one matmul per kernel tap."""
    kh, kw, cin, cout = w.shape
    bsz, h, wd, _ = x.shape
    if padding == "same":
        pt, pb, oh = _same_pads(h, kh, stride)
        pl, pr, ow = _same_pads(wd, kw, stride)
        if pt or pb or pl or pr:
            x = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    else:
        oh = (h - kh) // stride + 1
        ow = (wd - kw) // stride + 1
    out = np.zeros((bsz * oh * ow, cout), dtype=x.dtype)
    for ky in range(kh):
        for kx in range(kw):
            patch = x[:, ky:ky + (oh - 1) * stride + 1:stride, kx:kx + (ow - 1) * stride + 1:stride, :]
            out += patch.reshape(-1, cin) @ w[ky, kx]
    out = out.reshape(bsz, oh, ow, cout)
    if b is not None:
        out = out + b
    return out


def batchnorm_eval(x, bn):
    """App. A.3."""
    t = x.dtype.type
    return (x - bn["mean"].astype(x.dtype)) / np.sqrt(bn["var"].astype(x.dtype) + t(BN_EPS)) \
        * bn["gamma"].astype(x.dtype) + bn["beta"].astype(x.dtype)


def stem_pool(x):
    """ZeroPadding2D((1,2)) + MaxPooling2D(3, s2, valid)  (feature_extractor.py:31-33; App. A.2)."""
    x = np.pad(x, ((0, 0), (1, 1), (2, 2), (0, 0)))
    _, h, w, _ = x.shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    out = None
    for ky in range(3):
        for kx in range(3):
            p = x[:, ky:ky + (oh - 1) * 2 + 1:2, kx:kx + (ow - 1) * 2 + 1:2, :]
            out = p if out is None else np.maximum(out, p)
    return out


def resize_nearest(x, oh, ow):
    """tf.image.resize(NEAREST) TF2 half-pixel rule (App. A.4)."""
    _, h, w, _ = x.shape
    ys = np.minimum(np.floor((np.arange(oh) + 0.5) * (h / oh)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor((np.arange(ow) + 0.5) * (w / ow)).astype(np.int64), w - 1)
    return x[:, ys][:, :, xs]


def fill_triangular_4(x):
    """tfp.math.fill_triangular for 10 -> 4x4 lower (retinanet_model.py:110; App. A.6)."""
    idx = [[4, -1, -1, -1], [8, 9, -1, -1], [7, 6, 5, -1], [3, 2, 1, 0]]
    out = np.zeros(x.shape[:-1] + (4, 4), dtype=x.dtype)
    for r in range(4):
        for c in range(4):
            if idx[r][c] >= 0:
                out[..., r, c] = x[..., idx[r][c]]
    return out


def fold_bn(conv, bn):
    """W' = W*s, b' = (b-mean)*s+beta, s = gamma/sqrt(var+eps)  in float64 -> float32 (App. A.3)."""
    w = conv["kernel"].astype(np.float64)
    b = conv["bias"].astype(np.float64) if conv.get("bias") is not None else np.zeros(w.shape[-1])
    if bn is not None:
        s = bn["gamma"].astype(np.float64) / np.sqrt(bn["var"].astype(np.float64) + BN_EPS)
        w = w * s
        b = (b - bn["mean"].astype(np.float64)) * s + bn["beta"].astype(np.float64)
    return w.astype(np.float32), b.astype(np.float32)


# ----------------------------------------------------------------------------- numerics policies
class _Literal:
    def __init__(self, weights, dtype):
        self.w, self.dtype = weights, dtype

    def inp(self, x):
        return x.astype(self.dtype)

    def conv(self, x, name, bn=None, stride=1, padding="valid", relu=False, residual=None,
             store=True):
        c = self.w[name]
        bias = c["bias"].astype(self.dtype) if c.get("bias") is not None else None
        y = conv2d(x, c["kernel"].astype(self.dtype), bias, stride, padding)
        if bn is not None:
            y = batchnorm_eval(y, self.w[bn])
        if residual is not None:
            y = y + residual
        if relu:
            y = np.maximum(y, 0)
        return y

    def store(self, x):
        return x


class _Bf16:
    """fp32 accumulate over bf16-valued operands; bf16 storage between layers."""

    def __init__(self, weights):
        self.w = weights
        self._cache = {}

    def inp(self, x):
        return x.astype(np.float32)

    def folded(self, name, bn, round_w=True):
        key = (name, bn)
        if key not in self._cache:
            w, b = fold_bn(self.w[name], self.w[bn] if bn else None)
            self._cache[key] = (bf16_round(w) if round_w else w, b)
        return self._cache[key]

    def conv(self, x, name, bn=None, stride=1, padding="valid", relu=False, residual=None,
             store=True):
        w, b = self.folded(name, bn)
        y = conv2d(x, w, b, stride, padding)
        if residual is not None:
            y = y + residual
        if relu:
            y = np.maximum(y, np.float32(0))
        return bf16_round(y) if store else y

    def store(self, x):
        return bf16_round(x)


def make_numerics(weights, mode="literal", dtype=np.float64):
    if mode == "literal":
        return _Literal(weights, dtype)
    if mode == "bf16":
        return _Bf16(weights)
    raise ValueError(mode)


# ----------------------------------------------------------------------------- backbone + FPN
_STAGES = ((2, "abc", 1), (3, "abcd", 2), (4, "abcdef", 2), (5, "abc", 2))


def stages_for(weights):
    """ResNet-50 (the reference) unless the weights hold the 23-block stage 4 of the build's ResNet-101 option
    (res4a .. res4w; BASELINE config 5, no reference counterpart: SURVEY F6)."""
    if "res4w_branch2a" in weights:
        return ((2, "abc", 1), (3, "abcd", 2), (4, "abcdefghijklmnopqrstuvw", 2), (5, "abc", 2))
    return _STAGES


def feature_extractor(nm, image):
    """FeatureExtractor.call (feature_extractor.py:104-139). Returns (C5, C4-tap, C3-tap)."""
    x = nm.conv(nm.inp(image), "conv1", "bn_conv1", stride=2, padding="valid", relu=True)
    x = stem_pool(x)
    taps = {}
    for stage, blocks, first_stride in stages_for(nm.w):
        for blk in blocks:
            cb, bb = "res%d%s_branch" % (stage, blk), "bn%d%s_branch" % (stage, blk)
            if blk == "a":      # ConvBlock (:283-309): stride on conv_1 (1x1, valid) and shortcut
                y = nm.conv(x, cb + "2a", bb + "2a", stride=first_stride, relu=True)
                y = nm.conv(y, cb + "2b", bb + "2b", padding="same", relu=True)
                sc = nm.conv(x, cb + "1", bb + "1", stride=first_stride)
                x = nm.conv(y, cb + "2c", bb + "2c", relu=True, residual=sc)
                taps[stage] = x     # map_3 / map_4 are taken right after block 'a' (:119-120,:126-127)
            else:               # IdentityBlock (:195-213)
                y = nm.conv(x, cb + "2a", bb + "2a", relu=True)
                y = nm.conv(y, cb + "2b", bb + "2b", padding="same", relu=True)
                x = nm.conv(y, cb + "2c", bb + "2c", relu=True, residual=x)
    return x, taps[4], taps[3]


def feature_decoder(nm, c5, c4, c3):
    """FeatureDecoder.call (feature_decoder.py:136-171). Returns [p3, p4, p5, p6, p7]."""
    c5r = nm.conv(c5, "C5_reduced")
    p5 = nm.conv(c5r, "P5", padding="same")
    p6 = nm.conv(c5, "P6", stride=2, padding="same")
    p7 = nm.conv(np.maximum(p6, 0), "P7", stride=2, padding="same")
    # lateral 1x1 + nearest-upsampled top-down map, added before storage (fused on the device)
    up4 = resize_nearest(c5r, c4.shape[1], c4.shape[2])
    m4 = nm.conv(c4, "C4_reduced", residual=up4)
    p4 = nm.conv(m4, "P4", padding="same")
    up3 = resize_nearest(m4, c3.shape[1], c3.shape[2])       # upsamples merged m4, not p4 (:162-167)
    m3 = nm.conv(c3, "C3_reduced", residual=up3)
    p3 = nm.conv(m3, "P3", padding="same")
    return [p3, p4, p5, p6, p7]


# ----------------------------------------------------------------------------- heads
def head_tower(nm, pyramid, head, n_samples, keep_masks, rate, out_channels):
    """{Cls,Reg,Cov}Header.call over all levels with MC tiling (retinanet_model.py:78-109).

    pyramid    : list of [1,h,w,256] level maps (p3..p7)
    keep_masks : None (dropout off) or callable (sample, layer_id) -> bool [P, 256] over the
                 p3..p7 concatenated pixel index
    returns    : [n_samples, sum(h*w)*A, out_channels]
    """
    prefix = HEAD_PREFIX[head]
    sizes = [p.shape[1] * p.shape[2] for p in pyramid]
    offs = np.concatenate([[0], np.cumsum(sizes)])
    scale = np.float32(1.0 / (1.0 - rate)) if rate else None
    outs = []
    for li, lvl in enumerate(pyramid):
        x = np.repeat(lvl, n_samples, axis=0)              # tf.tile over batch (:78-81)
        _, h, w, c = x.shape
        for layer in range(HEAD_NUM_CONVS[head]):
            x = nm.conv(x, "%s_%d" % (prefix, layer), padding="same", relu=True, store=False)
            if keep_masks is not None:
                lid = HEAD_ID[head] * 4 + layer
                keep = np.stack([keep_masks(n, lid)[offs[li]:offs[li + 1]] for n in range(n_samples)])
                x = x * x.dtype.type(scale) * keep.reshape(n_samples, h, w, c).astype(x.dtype)
            x = nm.store(x)
        y = nm.conv(x, prefix, padding="same", store=False)  # 1x1 output conv, fp32 out on device
        outs.append(y.reshape(n_samples, h * w * (y.shape[-1] // out_channels), out_channels))
    return np.concatenate(outs, axis=1)


def retinanet_forward(weights, image, n_samples, num_classes_with_bknd, mode="literal",
                      dtype=np.float64, keep_masks=None, dropout_rate=0.3, return_pyramid=False):
    """RetinaNetModel.call(..., 'testing') (retinanet_model.py:67-112).

    image: [1,H,W,3] normalised BGR float. MC dropout is enabled iff n_samples > 1 (:74-77);
    ``keep_masks`` must then be given (injected randomness).
    """
    nm = make_numerics(weights, mode, dtype)
    c5, c4, c3 = feature_extractor(nm, image)
    pyr = feature_decoder(nm, c5, c4, c3)
    mc = n_samples > 1
    km = keep_masks if mc else None
    cls = head_tower(nm, pyr, "cls", n_samples, km, dropout_rate if mc else 0.0, num_classes_with_bknd)
    box = head_tower(nm, pyr, "reg", n_samples, km, dropout_rate if mc else 0.0, 4)
    cov = head_tower(nm, pyr, "cov", n_samples, km, dropout_rate if mc else 0.0, 10)
    out = {"anchors_class_predictions": cls,
           "anchors_box_predictions": box,
           "anchors_box_covar_predictions": fill_triangular_4(cov),
           "_covar_params": cov}
    if return_pyramid:
        out["_pyramid"] = pyr
        out["_backbone"] = (c5, c4, c3)
    return out
