"""FPN anchor generator with the reference's interface
(src/retina_net/anchor_generator/fpn_anchor_generator.py).  Anchors are built once per image
shape on the host and uploaded with ``bod_set_anchors``; float32 in the reference's op order."""
import numpy as np


class FpnAnchorGenerator(object):
    def __init__(self, generator_config):
        self.config = generator_config
        self.aspect_ratios = generator_config['aspect_ratios']
        self.scales = generator_config['scales']
        self.anchors_per_location = int(np.size(self.aspect_ratios, axis=0) * np.size(self.scales))

    def generate_anchors(self, im_shape, layer_number):
        """[A_l, 4] rows (v, u, h, w) for pyramid level ``layer_number`` (:21-59)."""
        f = np.float32
        h_im, w_im = f(im_shape[0]), f(im_shape[1])
        stride = f(2.0) ** f(layer_number)
        u_pos = (np.arange(0, w_im / stride, dtype=f) + f(0.5)) * stride
        v_pos = (np.arange(0, h_im / stride, dtype=f) + f(0.5)) * stride
        u, v = np.meshgrid(u_pos, v_pos)
        locations = np.stack((v.reshape(-1), u.reshape(-1)), axis=1)
        side = f(2.0) ** f(layer_number + 2.0)
        dims = []
        for aspect_ratio in self.aspect_ratios:
            ar = np.asarray(aspect_ratio, dtype=f)
            for scale in self.scales:
                if aspect_ratio[0] == 1 and aspect_ratio[1] == 1:
                    dims.append(ar * side * f(scale))
                else:
                    solution = np.sqrt(f(side ** f(2.0)) / f(np.prod(aspect_ratio))).astype(f)
                    dims.append(ar * solution * f(scale))
        dims = np.stack(dims).astype(f)
        n_loc = locations.shape[0]
        grid = np.concatenate((np.repeat(locations, self.anchors_per_location, axis=0),
                               np.tile(dims, (n_loc, 1))), axis=1)
        return grid.astype(f)

    def generate_all(self, im_shape, layers=None):
        """p3 -> p7 concatenation, the order of sample_dict['anchors']
        (src/retina_net/datasets/bdd/bdd_dataset_handler.py:160-186)."""
        layers = layers if layers is not None else self.config['layers']
        return np.concatenate([self.generate_anchors(im_shape, l) for l in layers], axis=0)

    @staticmethod
    def positive_negative_batching(ious, min_positive_iou=0.5, max_negative_iou=0.4):
        """:61-79."""
        positive = np.any(ious >= min_positive_iou, axis=1)
        negative = np.all(ious <= max_negative_iou, axis=1)
        return positive, negative, np.argmax(ious, axis=1)

    @staticmethod
    def generate_anchor_targets(anchors, gt_boxes, gt_classes, max_ious, positive_anchor_mask):
        """:81-137."""
        f = np.float32
        gt = gt_boxes[max_ious]
        tv = (gt[:, 0] - anchors[:, 0]) / anchors[:, 2] * f(10.0)
        tu = (gt[:, 1] - anchors[:, 1]) / anchors[:, 3] * f(10.0)
        th = np.log(gt[:, 2] / anchors[:, 2]) * f(5.0)
        tw = np.log(gt[:, 3] / anchors[:, 3]) * f(5.0)
        box_targets = np.stack([tv, tu, th, tw], axis=1)
        cls = gt_classes[max_ious]
        c = gt_classes.shape[1]
        negative = np.zeros((c,), dtype=gt_classes.dtype)
        negative[c - 1] = 1.0
        cls_targets = np.where(np.asarray(positive_anchor_mask)[:, None], cls, negative[None, :])
        return box_targets, cls_targets
