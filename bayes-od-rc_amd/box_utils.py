"""Host-side box helpers with the reference's names and conventions
(src/retina_net/anchor_generator/box_utils.py).  NumPy in, NumPy out; the per-anchor hot-path
versions of these run inside the HIP pipeline (csrc/post_kernels.hip)."""
import numpy as np


def vuhw_to_vuvu_np(vuhw):
    """(v,u,h,w) -> (v_min,u_min,v_max,u_max); box_utils.py:73-91."""
    v, u, h, w = vuhw[:, 0], vuhw[:, 1], vuhw[:, 2], vuhw[:, 3]
    return np.stack((v - h / 2.0, u - w / 2.0, v + h / 2.0, u + w / 2.0), axis=1)


def vuvu_to_vuhw_np(vuvu):
    """box_utils.py:49-70."""
    v0, u0, v1, u1 = vuvu[:, 0], vuvu[:, 1], vuvu[:, 2], vuvu[:, 3]
    return np.stack(((v1 + v0) / 2.0, (u1 + u0) / 2.0, v1 - v0, u1 - u0), axis=1)


vuhw_to_vuvu = vuhw_to_vuvu_np
vuvu_to_vuhw = vuvu_to_vuhw_np


def bbox_iou_vuvu(bboxes1, bboxes2):
    """Pairwise IoU with the reference's +1 pixel convention and its area expression
    (box_utils.py:117-146) -- host version used by target generation only."""
    b1 = np.asarray(bboxes1, dtype=np.float32)
    b2 = np.asarray(bboxes2, dtype=np.float32)
    y11, x11, y12, x12 = np.split(b1, 4, axis=1)
    y21, x21, y22, x22 = np.split(b2, 4, axis=1)
    xi1, yi1 = np.maximum(x11, x21.T), np.maximum(y11, y21.T)
    xi2, yi2 = np.minimum(x12, x22.T), np.minimum(y12, y22.T)
    one = np.float32(1.0)
    inter = np.maximum(xi2 - xi1 + one, 0) * np.maximum(yi2 - yi1 + one, 0)
    a1 = (x11 - x12 + one) * (y11 - y12 + one)
    a2 = (x21 - x22 + one) * (y21 - y22 + one)
    return inter / ((a1 + a2.T) - inter + np.float32(0.00001))


def box_from_anchor_and_target(anchors, regressed_targets):
    """box_utils.py:149-168."""
    a = np.asarray(anchors, dtype=np.float32)
    t = np.asarray(regressed_targets, dtype=np.float32)
    v = a[..., 2] * t[..., 0] / np.float32(10.0) + a[..., 0]
    u = a[..., 3] * t[..., 1] / np.float32(10.0) + a[..., 1]
    h = a[..., 2] * np.clip(np.exp(t[..., 2] / np.float32(5.0)), 1e-4, 1e4).astype(np.float32)
    w = a[..., 3] * np.clip(np.exp(t[..., 3] / np.float32(5.0)), 1e-4, 1e4).astype(np.float32)
    return np.stack([v, u, h, w], axis=-1)


box_from_anchor_and_target_bnms = box_from_anchor_and_target
