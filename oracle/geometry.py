"""Oracle restatement of the reference geometry layer.

Follows  src/retina_net/anchor_generator/fpn_anchor_generator.py  and
         src/retina_net/anchor_generator/box_utils.py  of the reference.
float32 arithmetic in the reference's operation order unless ``dtype`` says otherwise.
"""
import numpy as np


# ----------------------------------------------------------------------------- anchors (a17)
def generate_anchors(im_shape, layer_number, aspect_ratios, scales, dtype=np.float32):
    """fpn_anchor_generator.py:21-59.  Returns [9*h*w, 4] rows (v, u, h, w).

    centres (i+0.5)*stride over ``tf.range(0, dim/stride)`` (float range => ceil(dim/stride)
    entries), row-major with v outer / u inner (meshgrid + reshape, :30-34); dims per
    (ratio outer, scale inner) (:36-48); each location's anchors contiguous (tf_repeat :52-53).
    """
    f = dtype
    h_im, w_im = f(im_shape[0]), f(im_shape[1])
    stride = f(2.0) ** f(layer_number)
    u_pos = (np.arange(0, w_im / stride, dtype=f) + f(0.5)) * stride
    v_pos = (np.arange(0, h_im / stride, dtype=f) + f(0.5)) * stride
    u, v = np.meshgrid(u_pos, v_pos)
    loc = np.stack((v.reshape(-1), u.reshape(-1)), axis=1)

    side = f(2.0) ** f(layer_number + 2.0)
    dims = []
    for ar in aspect_ratios:
        ar_t = np.asarray(ar, dtype=f)
        for scale in scales:
            if ar[0] == 1 and ar[1] == 1:
                dims.append(ar_t * side * f(scale))
            else:
                sol = np.sqrt(f(side ** f(2.0)) / f(np.prod(ar))).astype(f)
                dims.append(ar_t * sol * f(scale))
    dims = np.stack(dims).astype(f)                       # [A, 2]
    n_loc, a = loc.shape[0], dims.shape[0]
    locs = np.repeat(loc, a, axis=0)                      # each location repeated A times
    dims = np.tile(dims, (n_loc, 1))
    return np.concatenate((locs, dims), axis=1).astype(f)


def generate_all_anchors(im_shape, layers, aspect_ratios, scales, dtype=np.float32):
    """Concatenation p3 -> p7 as in bdd_dataset_handler.py:160-186."""
    return np.concatenate([generate_anchors(im_shape, l, aspect_ratios, scales, dtype)
                           for l in layers], axis=0)


# ----------------------------------------------------------------------------- box utils
def vuhw_to_vuvu(vuhw):
    """box_utils.py:5-23 / :73-91."""
    v, u, h, w = vuhw[:, 0], vuhw[:, 1], vuhw[:, 2], vuhw[:, 3]
    two = vuhw.dtype.type(2.0)
    return np.stack((v - h / two, u - w / two, v + h / two, u + w / two), axis=1)


def vuvu_to_vuhw(vuvu):
    """box_utils.py:26-46 / :49-70."""
    v0, u0, v1, u1 = vuvu[:, 0], vuvu[:, 1], vuvu[:, 2], vuvu[:, 3]
    two = vuvu.dtype.type(2.0)
    return np.stack(((v1 + v0) / two, (u1 + u0) / two, v1 - v0, u1 - u0), axis=1)


def bbox_iou_vuvu(b1, b2):
    """box_utils.py:117-146, INCLUDING the area sign quirk (:140-141): areas are
    (min-max+1)*(min-max+1) = (w-1)(h-1) while the intersection uses (max-min+1)."""
    t = b1.dtype.type
    y11, x11, y12, x12 = (b1[:, i:i + 1] for i in range(4))
    y21, x21, y22, x22 = (b2[:, i:i + 1] for i in range(4))
    xi1 = np.maximum(x11, x21.T)
    yi1 = np.maximum(y11, y21.T)
    xi2 = np.minimum(x12, x22.T)
    yi2 = np.minimum(y12, y22.T)
    inter = np.maximum(xi2 - xi1 + t(1.0), t(0.0)) * np.maximum(yi2 - yi1 + t(1.0), t(0.0))
    a1 = (x11 - x12 + t(1.0)) * (y11 - y12 + t(1.0))
    a2 = (x21 - x22 + t(1.0)) * (y21 - y22 + t(1.0))
    union = (a1 + a2.T) - inter
    return inter / (union + t(0.00001))


def box_from_anchor_and_target(anchors, targets):
    """box_utils.py:149-192 (single and _bnms batched: ``anchors`` broadcasts over leading dims)."""
    t = targets.dtype.type
    a = anchors.astype(targets.dtype)
    v = a[..., 2] * targets[..., 0] / t(10.0) + a[..., 0]
    u = a[..., 3] * targets[..., 1] / t(10.0) + a[..., 1]
    h = a[..., 2] * np.clip(np.exp(targets[..., 2] / t(5.0)), t(1e-4), t(1e4))
    w = a[..., 3] * np.clip(np.exp(targets[..., 3] / t(5.0)), t(1e-4), t(1e4))
    return np.stack([v, u, h, w], axis=-1)


# ----------------------------------------------------------------------------- targets (a18)
def positive_negative_batching(ious, min_positive_iou=0.5, max_negative_iou=0.4):
    """fpn_anchor_generator.py:61-79."""
    pos = np.any(ious >= min_positive_iou, axis=1)
    neg = np.all(ious <= max_negative_iou, axis=1)
    return pos, neg, np.argmax(ious, axis=1)


def generate_anchor_targets(anchors, gt_boxes, gt_classes, max_ious, positive_mask):
    """fpn_anchor_generator.py:81-137."""
    t = anchors.dtype.type
    gt = gt_boxes[max_ious]
    tv = (gt[:, 0] - anchors[:, 0]) / anchors[:, 2] * t(10.0)
    tu = (gt[:, 1] - anchors[:, 1]) / anchors[:, 3] * t(10.0)
    th = np.log(gt[:, 2] / anchors[:, 2]) * t(5.0)
    tw = np.log(gt[:, 3] / anchors[:, 3]) * t(5.0)
    box_t = np.stack([tv, tu, th, tw], axis=1)
    cls = gt_classes[max_ious]
    c = gt_classes.shape[1]
    neg_row = np.zeros((c,), dtype=gt_classes.dtype)
    neg_row[c - 1] = 1.0
    cls_t = np.where(positive_mask[:, None], cls, neg_row[None, :])
    return box_t, cls_t
