"""TEST INFRASTRUCTURE (oracle): restatement of validation_utils.post_process_predictions
(src/retina_net/experiments/validation_utils.py:10-77) -- the deterministic validation path: one forward pass,
softmax, background filter on the arg-max class, soft-NMS on the top score, optional KITTI rescale."""
import numpy as np

from . import geometry, nms


def post_process_predictions(anchors, box_targets, class_logits, dataset_name='bdd', net_hw=None, orig_hw=None,
                             max_output_size=100, iou_threshold=0.5, soft_nms_sigma=0.5, variant='A', dtype=np.float64):
    """anchors [A,4] (v,u,h,w), box_targets [A,4], class_logits [A,C] -> (classes [K,C], corners [K,4], info)."""
    a = np.asarray(anchors, dtype=dtype)
    t = np.asarray(box_targets, dtype=dtype)
    boxes = np.stack([a[:, 2] * t[:, 0] / 10.0 + a[:, 0], a[:, 3] * t[:, 1] / 10.0 + a[:, 1],
                      a[:, 2] * np.clip(np.exp(t[:, 2] / 5.0), 1e-4, 1e4),
                      a[:, 3] * np.clip(np.exp(t[:, 3] / 5.0), 1e-4, 1e4)], axis=1)          # box_utils.py:149-168
    corners = geometry.vuhw_to_vuvu(boxes)
    z = np.asarray(class_logits, dtype=dtype)
    e = np.exp(z - z.max(axis=1, keepdims=True))
    probs = e / e.sum(axis=1, keepdims=True)
    keep = probs.argmax(axis=1) != probs.shape[1] - 1                                         # :33-43
    corners, probs = corners[keep], probs[keep]
    top = probs.max(axis=1)
    idx, _ = nms.soft_nms(corners.astype(np.float32), top.astype(np.float32), max_output_size, iou_threshold,
                          soft_nms_sigma, variant=variant)
    scaled = corners
    if dataset_name == 'kitti':                                                                # :52-57
        n = np.asarray([net_hw[0], net_hw[1]] * 2, dtype=dtype)
        s = np.asarray([orig_hw[0], orig_hw[1]] * 2, dtype=dtype)
        scaled = (corners / n) * s
    return probs[idx], scaled[idx], {"keep": keep, "top": top, "nms": idx, "boxes": boxes[keep]}
