"""Builds the reference's training/testing ``sample_dict`` from an in-memory frame and GT boxes
(what ``BddDatasetHandler.create_sample_dict`` does after decoding the image:
src/retina_net/datasets/bdd/bdd_dataset_handler.py:128-197).  Host plumbing, NumPy only."""
import numpy as np

from . import box_utils, constants
from .anchor_generator import FpnAnchorGenerator


def normalize_frame(rgb_uint8, normalization='ImageNet'):
    """uint8 RGB [H,W,3] -> float32 mean-subtracted BGR (dataset_utils.py:19-29 + BGR flip :136-139)."""
    means = np.asarray(constants.MEANS_DICT[normalization], dtype=np.float32).reshape(1, 1, 3)
    return (np.asarray(rgb_uint8).astype(np.float32) - means)[:, :, ::-1].copy()


def create_sample_dict(image_normalized, anchor_gen_config, boxes_2d_gt_vuvu=None, boxes_class_gt=None,
                       is_testing=False):
    """Returns the dict with the reference's keys (src/core/constants.py:48-55); anchors / targets are
    stacked p3 -> p7.  ``boxes_2d_gt_vuvu`` [G,4] corners, ``boxes_class_gt`` [G,C] one-hot (incl. bknd)."""
    image_normalized = np.asarray(image_normalized, dtype=np.float32)
    gen = FpnAnchorGenerator(anchor_gen_config)
    sample = {constants.IMAGE_NORMALIZED_KEY: image_normalized,
              constants.ORIGINAL_IM_SIZE_KEY: np.asarray(image_normalized.shape, dtype=np.int32)}
    anchors_l, cls_l, box_l, pos_l, neg_l = [], [], [], [], []
    gt_vuhw = box_utils.vuvu_to_vuhw_np(np.asarray(boxes_2d_gt_vuvu, dtype=np.float32)) if not is_testing else None
    for layer in anchor_gen_config['layers']:
        anchors = gen.generate_anchors(image_normalized.shape, layer)
        anchors_l.append(anchors)
        if not is_testing:
            ious = box_utils.bbox_iou_vuvu(box_utils.vuhw_to_vuvu_np(anchors), np.asarray(boxes_2d_gt_vuvu, np.float32))
            pos, neg, arg = gen.positive_negative_batching(ious, anchor_gen_config['min_positive_iou'],
                                                           anchor_gen_config['max_negative_iou'])
            box_t, cls_t = gen.generate_anchor_targets(anchors, gt_vuhw, np.asarray(boxes_class_gt, np.float32), arg, pos)
            pos_l.append(pos); neg_l.append(neg); box_l.append(box_t); cls_l.append(cls_t)
    sample[constants.ANCHORS_KEY] = np.concatenate(anchors_l, axis=0)
    if not is_testing:
        sample[constants.ANCHORS_BOX_TARGETS_KEY] = np.concatenate(box_l, axis=0)
        sample[constants.ANCHORS_CLASS_TARGETS_KEY] = np.concatenate(cls_l, axis=0)
        sample[constants.POSITIVE_ANCHORS_MASK_KEY] = np.concatenate(pos_l, axis=0)
        sample[constants.NEGATIVE_ANCHOR_MASK_KEY] = np.concatenate(neg_l, axis=0)
    return sample
