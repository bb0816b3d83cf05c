"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's frame preprocessing, i.e. what the
dataset handlers do between image decode and ``sample_dict['image_normalized']``.

  BDD   (src/retina_net/datasets/bdd/bdd_dataset_handler.py:128-139): decode_jpeg -> float32 ->
        mean_image_subtraction (datasets/dataset_utils.py:19-29, means src/core/constants.py:12) -> RGB->BGR.
  KITTI (src/retina_net/datasets/kitti/kitti_dataset_handler.py:120-148): decode_png ->
        tf.image.resize(BILINEAR, preserve_aspect_ratio=True) to config['kitti']['resize_shape'] ->
        tf.image.resize_with_crop_or_pad -> mean subtraction -> RGB->BGR; GT boxes are divided by the original
        size and multiplied by the FINAL (padded) size (box_utils.normalize_/expand_2d_bounding_boxes, :132-135).

PARITY UNPINNED for the two TF image ops (tensorflow is not importable here and the reference holds no
fixture for them); restated from the TF2 kernels' documented behaviour:
  * resize, bilinear, antialias=False, half_pixel_centers: src = (dst + 0.5) * (in/out) - 0.5, the two taps
    clamped to [0, in-1], weight = src - floor(src) (computed in float32 like the kernel's CachedInterpolation);
  * preserve_aspect_ratio: scale = min(th/in_h, tw/in_w), new size = round(in * scale) (Python/TF round-half-even
    on float32... TF uses math_ops.round on float32 products);
  * resize_with_crop_or_pad: centred; crop offset = max(-diff // 2, 0), pad offset = max(diff // 2, 0), zeros.
tests/test_preprocess.py cross-checks the bilinear kernel against torch.nn.functional.interpolate
(align_corners=False, antialias=False), an independent implementation of the same convention."""
import numpy as np

IMAGENET_MEANS = (123.68, 116.78, 103.94)          # src/core/constants.py:12 (RGB order)


def preserve_aspect_size(in_hw, target_hw):
    in_h, in_w = int(in_hw[0]), int(in_hw[1])
    sh = np.float32(target_hw[0]) / np.float32(in_h)
    sw = np.float32(target_hw[1]) / np.float32(in_w)
    s = min(sh, sw)
    return int(np.round(np.float32(s * np.float32(in_h)))), int(np.round(np.float32(s * np.float32(in_w))))


def _interp_axis(n_in, n_out):
    scale = np.float32(n_in) / np.float32(n_out)
    src = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    lo_f = np.floor(src)
    lerp = (src - lo_f).astype(np.float32)
    lo = np.clip(lo_f.astype(np.int64), 0, n_in - 1)
    hi = np.clip(np.ceil(src).astype(np.int64), 0, n_in - 1)
    return lo, hi, lerp


def bilinear_resize(img, out_h, out_w):
    """img [H,W,C] (any real dtype) -> float32 [out_h,out_w,C], TF2 tf.image.resize(BILINEAR) semantics."""
    x = np.asarray(img, dtype=np.float32)
    ylo, yhi, yl = _interp_axis(x.shape[0], out_h)
    xlo, xhi, xl = _interp_axis(x.shape[1], out_w)
    xl = xl[None, :, None]
    top = x[ylo][:, xlo] + (x[ylo][:, xhi] - x[ylo][:, xlo]) * xl
    bot = x[yhi][:, xlo] + (x[yhi][:, xhi] - x[yhi][:, xlo]) * xl
    return (top + (bot - top) * yl[:, None, None]).astype(np.float32)


def crop_or_pad_offsets(in_hw, target_hw):
    """(crop_y, crop_x, pad_y, pad_x) of tf.image.resize_with_crop_or_pad."""
    dh, dw = target_hw[0] - in_hw[0], target_hw[1] - in_hw[1]
    return max(-dh // 2, 0), max(-dw // 2, 0), max(dh // 2, 0), max(dw // 2, 0)


def resize_with_crop_or_pad(img, th, tw):
    x = np.asarray(img)
    cy, cx, py, px = crop_or_pad_offsets(x.shape[:2], (th, tw))
    hh, ww = min(x.shape[0], th), min(x.shape[1], tw)
    out = np.zeros((th, tw) + x.shape[2:], dtype=x.dtype)
    out[py:py + hh, px:px + ww] = x[cy:cy + hh, cx:cx + ww]
    return out


def normalize_bgr(img_rgb, means=IMAGENET_MEANS):
    x = np.asarray(img_rgb, dtype=np.float32) - np.asarray(means, dtype=np.float32).reshape(1, 1, 3)
    return np.ascontiguousarray(x[:, :, ::-1])


def bdd_preprocess(rgb_u8, means=IMAGENET_MEANS):
    return normalize_bgr(rgb_u8, means)


def kitti_preprocess(rgb_u8, resize_shape, means=IMAGENET_MEANS):
    nh, nw = preserve_aspect_size(rgb_u8.shape[:2], resize_shape)
    x = bilinear_resize(rgb_u8, nh, nw)
    x = resize_with_crop_or_pad(x, resize_shape[0], resize_shape[1])
    return normalize_bgr(x, means)


def kitti_rescale_boxes(boxes_vuvu, orig_hw, final_hw):
    """GT boxes (y1,x1,y2,x2) in original pixels -> network pixels, as the handler does (:132-135):
    divide by the original (h,w), multiply by the FINAL padded (h,w) -- the pad offset is ignored there."""
    b = np.asarray(boxes_vuvu, dtype=np.float32).reshape(-1, 4)
    n = np.asarray([orig_hw[0], orig_hw[1]] * 2, dtype=np.float32)
    s = np.asarray([final_hw[0], final_hw[1]] * 2, dtype=np.float32)
    return (b / n) * s
