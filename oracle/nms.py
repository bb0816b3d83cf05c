"""Oracle restatement of ``tf.image.non_max_suppression_with_scores`` (NonMaxSuppressionV5),
called at src/retina_net/experiments/inference_utils.py:207-212 with
``max_output_size=100, iou_threshold=0.5, soft_nms_sigma=0.5`` and the default
``score_threshold=-inf``  (SURVEY.md row a14, App. A.8).

TensorFlow is a third-party dependency absent from /root/reference and **unpinned**
(requirements.txt:10; README.md:7 says "tensorflow 2.0").  This restates the published
algorithm of ``tensorflow/core/kernels/image/non_max_suppression_op.cc``
(``DoNonMaxSuppressionOp``): greedy selection from a max-priority queue; a popped candidate's
score is multiplied by ``w(iou)`` for every box selected since it was last examined, newest
first; it is selected iff its score did not change, otherwise re-queued while
``score > score_threshold``.  Two published variants of ``w`` exist:

  variant "A" (TF 2.0 - 2.2):  w = exp(-0.5/sigma * iou^2) if iou <= thr else 0
  variant "B" (TF >= 2.3)   :  w = exp(-0.5/sigma * iou^2)           (sigma > 0: never hard-suppress)

Ties in the queue are broken towards the lower box index.  All arithmetic float32 as in the op
(T = float).  **Parity unpinned**: no TF build is available to check either variant.
"""
import heapq

import numpy as np

f32 = np.float32


def _iou(boxes, i, j):
    """IOU<T> of the op: corner order normalised with min/max, no +1 convention."""
    bi, bj = boxes[i], boxes[j]
    ymin_i, xmin_i = min(bi[0], bi[2]), min(bi[1], bi[3])
    ymax_i, xmax_i = max(bi[0], bi[2]), max(bi[1], bi[3])
    ymin_j, xmin_j = min(bj[0], bj[2]), min(bj[1], bj[3])
    ymax_j, xmax_j = max(bj[0], bj[2]), max(bj[1], bj[3])
    area_i = f32(ymax_i - ymin_i) * f32(xmax_i - xmin_i)
    area_j = f32(ymax_j - ymin_j) * f32(xmax_j - xmin_j)
    if area_i <= 0 or area_j <= 0:
        return f32(0.0)
    iy0, ix0 = max(ymin_i, ymin_j), max(xmin_i, xmin_j)
    iy1, ix1 = min(ymax_i, ymax_j), min(xmax_i, xmax_j)
    inter = f32(max(f32(iy1 - iy0), f32(0.0))) * f32(max(f32(ix1 - ix0), f32(0.0)))
    return f32(inter / f32(f32(area_i + area_j) - inter))


def soft_nms(boxes, scores, max_output_size=100, iou_threshold=0.5, soft_nms_sigma=0.5,
             score_threshold=-np.inf, variant="A"):
    """boxes [M,4] (y1,x1,y2,x2) float32, scores [M] float32 -> (indices int32 [K], scores [K])."""
    boxes = np.asarray(boxes, dtype=np.float32)
    scores = np.asarray(scores, dtype=np.float32)
    m = boxes.shape[0]
    thr = f32(iou_threshold)
    sthr = f32(score_threshold)
    is_soft = soft_nms_sigma > 0.0
    scale = f32(f32(-0.5) / f32(soft_nms_sigma)) if is_soft else f32(0.0)

    def weight(sim):
        # exp evaluated in double and rounded once, so host and device agree bit-for-bit
        w = f32(np.exp(np.float64(f32(scale * f32(sim * sim)))))
        if variant == "B" and is_soft:
            return w
        return w if sim <= thr else f32(0.0)

    # max-heap on (score, -index): python heapq is a min-heap -> negate score
    heap = [(-float(scores[i]), i, 0) for i in range(m) if scores[i] > sthr]
    heapq.heapify(heap)
    cur = scores.copy()
    selected, selected_scores = [], []
    while len(selected) < max_output_size and heap:
        neg, idx, begin = heapq.heappop(heap)
        original = f32(-neg)
        s = original
        hard = False
        for j in range(len(selected) - 1, begin - 1, -1):
            sim = _iou(boxes, idx, selected[j])
            s = f32(s * weight(sim))
            if variant == "B" and (not is_soft) and sim > thr:
                hard = True
                break
            if s <= sthr:
                break
        begin = len(selected)
        if hard:
            continue
        if s == original:
            selected.append(idx)
            selected_scores.append(s)
        elif s > sthr:
            cur[idx] = s
            heapq.heappush(heap, (-float(s), idx, begin))
    return np.asarray(selected, dtype=np.int32), np.asarray(selected_scores, dtype=np.float32)
