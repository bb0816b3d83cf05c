"""``bayes_od_inference`` / ``bayes_od_clustering`` / ``map_dataset_classes`` with the reference's
names, argument meaning, return shapes and error behaviour
(src/retina_net/experiments/inference_utils.py:13-217, :285-364, :372-404), executed by the HIP
pipeline through the C ABI.  Nothing here computes on the host except array reshaping and the
(trivial, index-only) class-name mapping.
"""
import numpy as np

from . import _lib, box_utils, constants


def _bayes_testing_kwargs(bayes_od_config, nms_config, use_full_covar, dataset_name, sample_dict,
                          nms_variant):
    kw = dict(bayes_od_config=bayes_od_config, nms_config=nms_config, use_full_covar=use_full_covar,
              dataset_name=dataset_name, nms_variant=nms_variant)
    if dataset_name == 'kitti':
        orig = np.asarray(sample_dict[constants.ORIGINAL_IM_SIZE_KEY]).reshape(-1, 3)[0]
        kw['orig_size'] = (int(orig[0]), int(orig[1]))
    return kw


def bayes_od_inference(model, sample_dict, bayes_od_config, nms_config, use_full_covar=False,
                       dataset_name='bdd', seed=None, image_id=None, nms_variant='A',
                       return_iou=True, return_engine=False):
    """Same 5 return values as the reference (:217) for a batch-of-1 ``sample_dict``:

        dirichlet_posterior_count [M,C], gaussian_posterior_means [M,4,1],
        gaussian_posterior_covs [M,4,4], nms_indices [K], predicted_boxes_iou_mat [M,M]

    ``seed`` / ``image_id`` key the Philox streams that replace TF's unseeded RNG.
    ``return_iou=False`` skips materialising the M x M matrix (the device clustering does not
    need it); an empty [0,0] array is returned in its place.
    ``return_engine=True`` appends the handle that produced the results as a sixth value: passing it to
    ``bayes_od_clustering(..., engine=)`` re-uses its device buffers.  Without it the two calls are as independent as
    the reference's (run_inference.py:138-149): nothing is remembered between them.
    """
    image = np.asarray(sample_dict[constants.IMAGE_NORMALIZED_KEY], dtype=np.float32)
    if image.ndim == 3:
        image = image[None]
    if image.shape[0] != 1:
        raise ValueError("bayes_od_inference mirrors the reference's batch(1) loop; use "
                         "BayesOdPipeline for batched throughput")
    anchors = np.asarray(sample_dict[constants.ANCHORS_KEY], dtype=np.float32)
    anchors = anchors.reshape(-1, 4)
    if model.mc_dropout_samples < 2:
        raise ValueError("bayes_od needs mc_dropout_samples >= 2: the sample covariance divides by N-1 "
                         "(inference_utils.py:241-242)")
    kw = _bayes_testing_kwargs(bayes_od_config, nms_config, use_full_covar, dataset_name, sample_dict,
                               nms_variant)
    eng = model.engine_for(image.shape[1:3], batch=1, mc_samples=model.mc_dropout_samples, **kw)
    eng.set_anchors(anchors)
    seed = model.seed if seed is None else seed
    if image_id is None:
        image_id = model.image_counter
        model.image_counter += 1
    eng.forward(image, seed=seed, first_image_id=image_id)
    eng.posterior(seed=seed, first_image_id=image_id)
    eng.nms()
    post = eng.get_posterior(0)
    nms_indices = eng.get_nms(0)
    iou = eng.get_iou_matrix(0) if return_iou else np.zeros((0, 0), np.float32)
    out = (post["counts"], post["means"][:, :, None], post["covs"], nms_indices, iou)
    return out + (eng,) if return_engine else out


def bayes_od_clustering(predicted_boxes_class_counts, predicted_boxes_means, predicted_boxes_covs,
                        cluster_centers, affinity_matrix=None, affinity_threshold=0.7, engine=None):
    """Bayesian cluster-and-fuse on the device (reference :285-364).

    Returns (final_scores [K,C], final_means [K,4,1], final_covs [K,4,4] (x70), final_counts [K,C]).
    ``affinity_matrix`` [M,M] is used exactly as the reference uses it (:316, members of centre k =
    ``affinity_matrix[:, k] > affinity_threshold``): its K centre columns are uploaded (K*M floats, not M*M).
    ``affinity_matrix=None`` selects the reference pipeline's own affinity, ``bbox_iou_vuvu`` of the
    posterior means (:204-215, run_inference.py:145-149), evaluated on the fly against each centre
    on the device without ever building the M x M matrix.
    ``engine``: optional handle to run on (``bayes_od_inference(..., return_engine=True)``); by default the call runs on a
    small post-processing handle of its own, sized for M boxes -- it depends on nothing but its arguments, like the reference's.
    """
    from .engine import Engine, make_config
    counts = np.ascontiguousarray(predicted_boxes_class_counts, dtype=np.float32)
    means = np.ascontiguousarray(predicted_boxes_means, dtype=np.float32).reshape(-1, 4)
    covs = np.ascontiguousarray(predicted_boxes_covs, dtype=np.float32).reshape(-1, 4, 4)
    centres = np.ascontiguousarray(cluster_centers, dtype=np.int32).reshape(-1)
    m, c = counts.shape
    k = centres.shape[0]
    if k == 0 or m == 0:
        return (np.zeros((0, c), np.float32), np.zeros((0, 4, 1), np.float32),
                np.zeros((0, 4, 4), np.float32), np.zeros((0, c), np.float32))
    eng = engine if engine is not None else _standalone_engine(m, c, k)
    cfg = eng.cfg
    cfg.nms_iou_threshold = float(affinity_threshold)
    eng.update_config(cfg)
    eng.set_posterior(0, counts, means, covs, np.zeros(m, np.float32))
    eng._set_centres(0, centres)
    if affinity_matrix is not None:
        aff = np.asarray(affinity_matrix)
        if aff.shape != (m, m):
            raise ValueError("affinity_matrix must be [M,M] = [%d,%d], got %s" % (m, m, aff.shape))
        if centres.min() < 0 or centres.max() >= m:
            raise ValueError("cluster centre out of range [0,%d)" % m)
        eng.set_affinity(0, np.ascontiguousarray(aff[:, centres].T, dtype=np.float32))
    eng.cluster_fuse()
    scores, fmeans, fcovs, fcounts = eng.get_detections(0)
    return scores, fmeans[:, :, None], fcovs, fcounts


_standalone = {}


def _standalone_engine(m, c, k):
    """Small handle used when bayes_od_clustering is called without a model (pure post-processing):
    geometry only sizes the buffers (A >= M)."""
    from .engine import Engine, make_config
    side = 64
    while True:
        a = sum((-(-side // s)) ** 2 for s in (8, 16, 32, 64, 128)) * 9          # anchors of a side x side input
        if a >= m:
            break
        side *= 2
    key = (side, c, max(k, 100))
    if key not in _standalone:
        nms = {'max_output_size': min(max(k, 100), 512), 'iou_threshold': 0.5, 'soft_nms_sigma': 0.5}
        _standalone[key] = Engine(make_config((side, side), batch=1, mc_samples=2, num_classes=c, nms_config=nms))
    return _standalone[key]


def map_dataset_classes(input_dataset, target_dataset, output_classes):
    """Class-score remapping between label sets (reference :372-404), host-side index shuffling.
    The result has ``len(target_dict) + 1`` columns, as in the reference."""
    mapping = constants.SET_TO_SET_MAPPING_DICTS[input_dataset + '_' + target_dataset]
    if not mapping:
        return output_classes
    in_d = constants.CATEGORY_IDX_MAPPING_DICTS[input_dataset]
    tg_d = constants.CATEGORY_IDX_MAPPING_DICTS[target_dataset]
    mapped = np.zeros([output_classes.shape[0], len(tg_d) + 1])
    if len(output_classes.shape) == 1:
        output_classes = np.expand_dims(output_classes, axis=1)
    names = list(in_d.keys())
    for row, scores in zip(mapped, output_classes):
        j = int(np.argmax(scores))
        row[tg_d[mapping[names[j]]]] = scores[j]
    return mapped


class BayesOdPipeline(object):
    """Batched, fully device-resident form of the reference's per-image loop body
    (src/retina_net/experiments/run_inference.py:137-161): forward -> posterior -> soft-NMS ->
    cluster-and-fuse for ``batch`` images per call, no host round trip in between."""

    def __init__(self, model, image_hw, batch, bayes_od_config, nms_config, use_full_covar=True,
                 dataset_name='bdd', orig_size=None, nms_variant='A', anchors=None):
        self._kw = dict(bayes_od_config=bayes_od_config, nms_config=nms_config, use_full_covar=use_full_covar,
                        dataset_name=dataset_name, nms_variant=nms_variant, orig_size=orig_size)
        self.model, self._hw, self._batch = model, tuple(image_hw), batch
        self.engine = self.bind()
        if anchors is not None:
            self.engine.set_anchors(anchors)

    def bind(self, orig_size=None):
        """(Re-)apply this pipeline's testing configuration to its engine.  Engines are cached by the model per
        (network size, batch, N), so two pipelines that differ only in ``orig_size`` -- KITTI frames of 370x1224 and
        375x1242 both resize to the same network input -- share one handle: the S mu / S Sigma S^T factors
        (inference_utils.py:147-167 takes them per sample from ORIGINAL_IM_SIZE) must be set before every batch."""
        if orig_size is not None:
            self._kw['orig_size'] = tuple(orig_size)
        self.engine = self.model.engine_for(self._hw, batch=self._batch, mc_samples=self.model.mc_dropout_samples, **self._kw)
        return self.engine

    def __call__(self, images=None, seed=0, first_image_id=0):
        """images [B,H,W,3] (or None to reuse the uploaded device batch).  Returns, per image,
        (output_classes [K,C], output_boxes_vuhw [K,4], output_covs [K,4,4], output_counts [K,C])."""
        self.bind()
        self.engine.infer(images, seed=seed, first_image_id=first_image_id)
        return [self.engine.get_detections(b) for b in range(self.engine.B)]


def post_process_predictions(sample_dict, prediction_dict, dataset_name='bdd', engine=None, nms_config=None):
    """Validation post-processing with the reference's signature and return value
    (src/retina_net/experiments/validation_utils.py:10-77): ``(predicted_boxes_classes [K,C],
    predicted_boxes_corners [K,4])`` of the first image of the batch -- softmax, background filter, soft-NMS on the
    top score -- computed on the device (``bod_validation_post`` + ``bod_nms``).  ``engine``: the handle that produced
    ``prediction_dict`` (default: a post-only handle of the right geometry is created)."""
    from .engine import Engine, make_config
    anchors = _lib.as_f32(np.asarray(sample_dict[constants.ANCHORS_KEY]))
    if anchors.ndim == 3:
        anchors = anchors[0]
    cls = _lib.as_f32(np.asarray(prediction_dict[constants.ANCHORS_CLASS_PREDICTIONS_KEY]))[0:1]
    box = _lib.as_f32(np.asarray(prediction_dict[constants.ANCHORS_BOX_PREDICTIONS_KEY]))[0:1]
    image = np.asarray(sample_dict[constants.IMAGE_NORMALIZED_KEY])
    hw = image.shape[1:3] if image.ndim == 4 else image.shape[0:2]
    if engine is None:
        engine = Engine(make_config(hw, batch=1, mc_samples=1, num_classes=cls.shape[-1], nms_config=nms_config,
                                    has_covar_head=False))
        engine.set_anchors(anchors)
    if (engine.B, engine.N) != (1, 1):
        raise ValueError("post_process_predictions works on a batch-1, single-sample handle")
    engine.set_raw(cls.reshape(1, 1, -1, cls.shape[-1]), box.reshape(1, 1, -1, 4), None)
    engine.validation_post()
    engine.nms()
    post = engine.get_posterior(0)
    idx = engine.get_nms(0)
    corners = box_utils.vuhw_to_vuvu_np(post["means"]) if post["means"].size else np.zeros((0, 4), np.float32)
    if dataset_name == 'kitti':
        orig = np.asarray(sample_dict[constants.ORIGINAL_IM_SIZE_KEY]).reshape(-1)[-3:]
        n = np.asarray([hw[0], hw[1]] * 2, np.float32)
        s = np.asarray([orig[0], orig[1]] * 2, np.float32)
        corners = (corners / n) * s
    return post["score"][idx], corners[idx]
