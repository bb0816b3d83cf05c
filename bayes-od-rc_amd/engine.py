"""Object wrapper over one ``bod_handle`` (include/bayesod.h).  NumPy in / NumPy out.

One Engine = one GPU, one HIP stream, one (image size, batch, MC sample count) geometry.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import BodConfig, BodSizes, as_f32, fptr, iptr

_KINDS = {"kernel": 0, "bias": 1, "gamma": 2, "beta": 3, "mean": 4, "var": 5}
# bod_config.precision (include/bayesod.h): bf16 = throughput path; fp32 = exact-fp32 MFMA; bf16x3 = (hi, lo) bf16 pairs with
# three MFMA products, the 1e-3 end-to-end parity mode on the bf16 matrix pipe
PRECISIONS = {"bf16": 0, "fp32": 1, "bf16x3": 2, "f16mx": 3, "f16mx4": 4}


def make_config(image_hw, batch=1, mc_samples=10, num_classes=8, anchors_per_location=9, device=0,
                dropout_rate=0.3, use_full_covar=True, bayes_od_config=None, nms_config=None,
                has_covar_head=True, dataset_name='bdd', orig_size=None, nms_variant='A',
                num_categorical_draws=30, layers=(3, 4, 5, 6, 7), precision='bf16', mc_sample_base=0,
                mc_ensemble_size=0, training=False, backbone_depth=50, pipeline_overlap=False):
    """Translates the reference's yaml dictionaries (configs/retinanet_bdd_covar.yaml:61-143)
    into a ``bod_config``."""
    bo = bayes_od_config or {'ranking_method': 'score', 'dirichlet_prior': {'type': 'non_informative'},
                             'gaussian_prior': {'type': 'isotropic', 'isotropic_variance': 100000.0}}
    nms = nms_config or {'max_output_size': 100, 'iou_threshold': 0.5, 'soft_nms_sigma': 0.5}
    cfg = BodConfig()
    cfg.device = int(device)
    cfg.image_h, cfg.image_w = int(image_hw[0]), int(image_hw[1])
    cfg.batch, cfg.mc_samples = int(batch), int(mc_samples)
    cfg.num_classes = int(num_classes)
    cfg.anchors_per_location = int(anchors_per_location)
    cfg.min_level, cfg.max_level = int(min(layers)), int(max(layers))
    cfg.dropout_rate = float(dropout_rate)
    cfg.use_full_covar = int(bool(use_full_covar))
    cfg.dirichlet_non_informative = int(bo['dirichlet_prior']['type'] == 'non_informative')
    cfg.gaussian_isotropic = int(bo['gaussian_prior']['type'] == 'isotropic')
    cfg.isotropic_variance = float(bo['gaussian_prior'].get('isotropic_variance', 100000.0))
    if bo['ranking_method'] not in ('score', 'joint_entropy'):
        raise ValueError("ranking_method must be 'score' or 'joint_entropy'")
    cfg.ranking_method = int(bo['ranking_method'] == 'joint_entropy')
    cfg.nms_max_output_size = int(nms['max_output_size'])
    cfg.nms_iou_threshold = float(nms['iou_threshold'])
    cfg.nms_soft_sigma = float(nms['soft_nms_sigma'])
    cfg.nms_variant = {'A': 0, 'B': 1}[nms_variant]
    cfg.num_categorical_draws = int(num_categorical_draws)
    cfg.has_covar_head = int(bool(has_covar_head))
    if precision not in PRECISIONS:
        raise ValueError("precision must be one of %s" % (sorted(PRECISIONS),))
    cfg.precision = PRECISIONS[precision]
    cfg.mc_sample_base, cfg.mc_ensemble_size = int(mc_sample_base), int(mc_ensemble_size)
    cfg.training = int(bool(training))
    if int(backbone_depth) not in (50, 101):
        raise ValueError("backbone_depth must be 50 or 101")
    cfg.backbone_depth = int(backbone_depth)
    # infer_async overlaps the front (stem / backbone / FPN) of batch i+1 with the towers of batch i on CU-partitioned streams
    cfg.pipeline_overlap = int(bool(pipeline_overlap))
    if dataset_name == 'kitti':
        if orig_size is None:
            raise ValueError("dataset_name='kitti' needs orig_size (sample_dict['im_size'])")
        cfg.kitti_scale_h = float(orig_size[0]) / float(image_hw[0])
        cfg.kitti_scale_w = float(orig_size[1]) / float(image_hw[1])
    return cfg


class Engine(object):
    def __init__(self, cfg):
        self.lib = _lib.load()
        self.cfg = cfg
        self.h = C.c_void_p()
        st = self.lib.bod_create(C.byref(cfg), C.byref(self.h))
        _lib.check(self.lib, None, st)
        s = BodSizes()
        self._chk(self.lib.bod_query_sizes(self.h, C.byref(s)))
        self.P, self.A = s.num_pixels, s.num_anchors
        self.levels = [(s.level_h[i], s.level_w[i]) for i in range(s.num_levels)]
        self.B, self.N, self.Ccls = cfg.batch, cfg.mc_samples, cfg.num_classes
        self.K = cfg.nms_max_output_size
        self._anchors_set = False

    # ------------------------------------------------------------------ plumbing
    def _chk(self, st):
        _lib.check(self.lib, self.h, st)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.lib.bod_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self):
        s = BodSizes()
        self._chk(self.lib.bod_query_sizes(self.h, C.byref(s)))
        return int(s.device_bytes)

    def update_config(self, cfg):
        self._chk(self.lib.bod_update_config(self.h, C.byref(cfg)))
        self.cfg = cfg

    # ------------------------------------------------------------------ weights / anchors
    def load_weights(self, weights):
        """weights: {keras_layer_name: {"kernel"/"bias"/"gamma"/"beta"/"mean"/"var": ndarray}}"""
        for name, entry in weights.items():
            for field, arr in entry.items():
                if arr is None:
                    continue
                a = as_f32(arr)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                self._chk(self.lib.bod_load_weight(self.h, name.encode(), _KINDS[field], shape, a.ndim, fptr(a)))
        self._chk(self.lib.bod_finalize_weights(self.h))

    def set_anchors(self, anchors):
        a = as_f32(anchors)
        self._chk(self.lib.bod_set_anchors(self.h, fptr(a), a.shape[0]))
        self._anchors_set = True

    # ------------------------------------------------------------------ stages
    def _img(self, images):
        a = as_f32(images)
        expect = (self.B, self.cfg.image_h, self.cfg.image_w, 3)
        if a.shape != expect:
            raise ValueError("images must have shape %s, got %s" % (expect, a.shape))
        return a

    def upload_images(self, images):
        a = self._img(images)
        self._chk(self.lib.bod_upload_images(self.h, fptr(a)))

    def upload_frames_u8(self, frames_rgb_u8, means=None, aspect_resize=False):
        """Decoded uint8 RGB frames [B,h,w,3] -> the device image buffer, preprocessed on the device like the
        reference's dataset handlers (mean subtraction, BGR flip; aspect_resize=True adds KITTI's bilinear
        aspect-preserving resize + centred crop/pad).  Then call forward()/infer() with images=None."""
        from . import constants
        a = np.ascontiguousarray(frames_rgb_u8, dtype=np.uint8)
        if a.ndim != 4 or a.shape[0] != self.B or a.shape[3] != 3:
            raise ValueError("expected uint8 frames of shape (%d, h, w, 3), got %s" % (self.B, a.shape))
        m = np.ascontiguousarray(constants.MEANS_DICT['ImageNet'] if means is None else means, dtype=np.float32)
        self._chk(self.lib.bod_upload_frames_u8(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[2],
                                                fptr(m), int(bool(aspect_resize))))

    def upload_frames_u8_async(self, frames_rgb_u8, buffer, means=None, aspect_resize=False):
        """Pipelined upload (``bod_upload_frames_u8_async``): copy + preprocessing run on the handle's copy stream into
        image buffer 0/1 and return at once; pass ``image_buffer=buffer`` to forward()/infer()/infer_async().  The array must
        be C-contiguous uint8 (pinned host memory for a truly asynchronous copy) and stay alive until that batch is
        collected."""
        from . import constants
        a = frames_rgb_u8
        if a.dtype != np.uint8 or not a.flags['C_CONTIGUOUS'] or a.ndim != 4 or a.shape[0] != self.B or a.shape[3] != 3:
            raise ValueError("expected C-contiguous uint8 frames of shape (%d, h, w, 3), got %s %s" % (self.B, a.dtype, a.shape))
        m = np.ascontiguousarray(constants.MEANS_DICT['ImageNet'] if means is None else means, dtype=np.float32)
        self._chk(self.lib.bod_upload_frames_u8_async(self.h, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[2],
                                                      fptr(m), int(bool(aspect_resize)), int(buffer)))

    def _device_images(self, image_buffer):
        ptr = self.lib.bod_device_images(self.h) if image_buffer is None else self.lib.bod_device_images_buffer(self.h, int(image_buffer))
        if not ptr:
            raise ValueError("image buffer %r has not been filled" % (image_buffer,))
        return ptr

    def get_images(self):
        """The device image buffer [B,H,W,3] float32 (normalised BGR) copied to the host."""
        import torch
        from .distributed import DeviceArray
        ptr = self.lib.bod_device_images(self.h)
        t = torch.as_tensor(DeviceArray(ptr, (self.B, self.cfg.image_h, self.cfg.image_w, 3), "<f4"),
                            device=torch.device("cuda", self.cfg.device))
        return t.cpu().numpy()

    def forward(self, images=None, seed=0, first_image_id=0, image_buffer=None):
        """images=None => use the device-resident buffer filled by upload_images() (or buffer `image_buffer` of
        upload_frames_u8_async)."""
        if images is None:
            ptr = self._device_images(image_buffer)
            self._chk(self.lib.bod_forward(self.h, ptr, 1, seed, first_image_id))
        else:
            a = self._img(images)
            self._chk(self.lib.bod_forward(self.h, a.ctypes.data, 0, seed, first_image_id))

    def infer(self, images=None, seed=0, first_image_id=0, image_buffer=None):
        if images is None:
            ptr = self._device_images(image_buffer)
            self._chk(self.lib.bod_infer(self.h, ptr, 1, seed, first_image_id))
        else:
            a = self._img(images)
            self._chk(self.lib.bod_infer(self.h, a.ctypes.data, 0, seed, first_image_id))

    def infer_async(self, images=None, seed=0, first_image_id=0, image_buffer=None):
        """Enqueue a whole pass; returns the slot ticket for collect()."""
        slot = C.c_int32(-1)
        if images is None:
            ptr = self._device_images(image_buffer)
            self._chk(self.lib.bod_infer_async(self.h, ptr, 1, seed, first_image_id, C.byref(slot)))
        else:
            a = self._img(images)
            self._chk(self.lib.bod_infer_async(self.h, a.ctypes.data, 0, seed, first_image_id, C.byref(slot)))
        return slot.value

    def collect(self, slot, out=None):
        b, k, c = self.B, self.K, self.Ccls
        if out is None:
            out = {"num": np.empty(b, np.int32), "scores": np.empty((b, k, c), np.float32),
                   "means": np.empty((b, k, 4), np.float32), "covs": np.empty((b, k, 4, 4), np.float32),
                   "counts": np.empty((b, k, c), np.float32)}
        self._chk(self.lib.bod_collect(self.h, slot, iptr(out["num"]), fptr(out["scores"]), fptr(out["means"]),
                                       fptr(out["covs"]), fptr(out["counts"])))
        return out

    def synchronize(self):
        self._chk(self.lib.bod_synchronize(self.h))

    def get_raw(self):
        n = (self.B, self.N, self.A)
        cls = np.empty(n + (self.Ccls,), np.float32)
        box = np.empty(n + (4,), np.float32)
        cov = np.empty(n + (10,), np.float32) if self.cfg.has_covar_head else None
        self._chk(self.lib.bod_get_raw(self.h, fptr(cls), fptr(box), fptr(cov)))
        return cls, box, cov

    def set_raw(self, cls, box, cov=None):
        cls, box = as_f32(cls), as_f32(box)
        cov = as_f32(cov) if cov is not None else None
        n = (self.B, self.N, self.A)
        if cls.shape != n + (self.Ccls,) or box.shape != n + (4,) or (cov is not None and cov.shape != n + (10,)):
            raise ValueError("raw head outputs have the wrong shape")
        self._chk(self.lib.bod_set_raw(self.h, fptr(cls), fptr(box), fptr(cov)))

    def get_pyramid(self, level_index):
        h, w = self.levels[level_index]
        out = np.empty((self.B, h, w, 256), np.float32)
        self._chk(self.lib.bod_get_pyramid(self.h, level_index, fptr(out)))
        return out

    def posterior(self, seed=0, first_image_id=0):
        self._chk(self.lib.bod_posterior(self.h, seed, first_image_id))

    # -- training ------------------------------------------------------
    def train_step(self, images, cls_targets, box_targets, positive_mask, negative_mask, seed=0, first_image_id=0,
                   reg_kind=3, label_smoothing=0.001, w_cls=5.0, w_reg=1.0, l2_rate=1e-6, learning_rate=1e-3,
                   apply_update=True):
        """run_training.train_single_step on a handle made with make_config(training=True): returns a dict with
        total_loss, cls_loss, reg_loss, covariance_loss, regularization_loss and the global gradient norm."""
        b, a = self.B, self.A
        ct = as_f32(cls_targets).reshape(b, a, self.Ccls)
        bt = as_f32(box_targets).reshape(b, a, 4)
        pm = np.ascontiguousarray(np.asarray(positive_mask).reshape(b, a), dtype=np.uint8)
        nm = np.ascontiguousarray(np.asarray(negative_mask).reshape(b, a), dtype=np.uint8)
        out = (C.c_double * 6)()
        u8 = C.POINTER(C.c_uint8)
        if images is None:
            ptr, on_dev = self.lib.bod_device_images(self.h), 1
        else:
            img = self._img(images)
            ptr, on_dev = img.ctypes.data, 0
        self._chk(self.lib.bod_train_step(self.h, ptr, on_dev, fptr(ct), fptr(bt), pm.ctypes.data_as(u8), nm.ctypes.data_as(u8),
                                          seed, first_image_id, int(reg_kind), float(label_smoothing), float(w_cls), float(w_reg),
                                          float(l2_rate), float(learning_rate), int(bool(apply_update)), out))
        keys = ("total_loss", "cls_loss", "reg_loss", "covariance_loss", "regularization_loss", "grad_norm")
        return dict(zip(keys, [out[i] for i in range(6)]))

    def train_gradients_view(self):
        """torch tensor aliasing the contiguous fp32 gradient arena (for the data-parallel all-reduce)."""
        import torch
        from .distributed import DeviceArray
        ptr, n = C.c_void_p(), C.c_int64()
        self._chk(self.lib.bod_train_gradients(self.h, C.byref(ptr), C.byref(n)))
        return torch.as_tensor(DeviceArray(ptr.value, (n.value,), "<f4"), device=torch.device("cuda", self.cfg.device))

    def train_apply(self, learning_rate):
        """Clip + Adam on the gradient arena's current contents; returns the global gradient norm."""
        g = C.c_double()
        self._chk(self.lib.bod_train_apply(self.h, float(learning_rate), C.byref(g)))
        return g.value

    def train_get(self, layer, kind, shape, what="value"):
        """A trainable tensor (kind: 'kernel' | 'bias' | 'gamma' | 'beta') or its gradient / Adam moments."""
        kinds = {"kernel": 0, "bias": 1, "gamma": 2, "beta": 3}
        whats = {"value": 0, "grad": 1, "adam_m": 2, "adam_v": 3}
        out = np.empty(shape, np.float32)
        self._chk(self.lib.bod_train_get(self.h, layer.encode(), kinds[kind], whats[what], fptr(out), out.size))
        return out

    def train_set_moment(self, layer, kind, what, value):
        """Restore an Adam moment (what: 'adam_m' | 'adam_v') of a trainable tensor (checkpoint resume)."""
        kinds = {"kernel": 0, "bias": 1, "gamma": 2, "beta": 3}
        whats = {"adam_m": 2, "adam_v": 3}
        a = as_f32(value)
        self._chk(self.lib.bod_train_set(self.h, layer.encode(), kinds[kind], whats[what], fptr(a), a.size))

    def train_step_count(self, set_to=None):
        """Number of optimizer updates applied (enters Adam's bias correction); ``set_to`` restores it."""
        got = C.c_int64(0)
        self._chk(self.lib.bod_train_step_count(self.h, C.byref(got), -1 if set_to is None else int(set_to)))
        return int(got.value)

    def validation_post(self):
        """validation_utils.post_process_predictions up to the NMS input (softmax, background filter, ranking)."""
        self._chk(self.lib.bod_validation_post(self.h))

    def num_kept(self):
        out = np.zeros(self.B, np.int32)
        self._chk(self.lib.bod_get_num_kept(self.h, iptr(out)))
        return out

    def get_posterior(self, image_index=0):
        m = int(self.num_kept()[image_index])
        counts = np.empty((m, self.Ccls), np.float32)
        score = np.empty((m, self.Ccls), np.float32)
        means = np.empty((m, 4), np.float32)
        covs = np.empty((m, 4, 4), np.float32)
        ranking = np.empty((m,), np.float32)
        aidx = np.empty((m,), np.int32)
        self._chk(self.lib.bod_get_posterior(self.h, image_index, fptr(counts), fptr(score), fptr(means),
                                             fptr(covs), fptr(ranking), iptr(aidx)))
        return {"counts": counts, "score": score, "means": means, "covs": covs, "ranking": ranking,
                "anchor_index": aidx}

    def set_posterior(self, image_index, counts, means, covs, ranking):
        counts, means, covs, ranking = as_f32(counts), as_f32(means).reshape(-1, 4), as_f32(covs), as_f32(ranking)
        m = means.shape[0]
        self._chk(self.lib.bod_set_posterior(self.h, image_index, m, fptr(counts), fptr(means), fptr(covs),
                                             fptr(ranking)))

    def nms(self):
        self._chk(self.lib.bod_nms(self.h))

    def get_nms(self, image_index=0):
        idx = np.zeros(self.K, np.int32)
        n = C.c_int32(0)
        self._chk(self.lib.bod_get_nms(self.h, image_index, iptr(idx), C.byref(n)))
        return idx[:n.value].copy()

    def _set_centres(self, image_index, centres):
        c = np.ascontiguousarray(centres, dtype=np.int32)
        self._chk(self.lib.bod_set_nms(self.h, image_index, iptr(c), c.shape[0]))

    def set_affinity(self, image_index, centre_columns):
        """centre_columns [K,M]: affinity_matrix[:, centre_k] for every cluster centre (one-shot, next cluster_fuse)."""
        c = as_f32(centre_columns)
        self._chk(self.lib.bod_set_affinity(self.h, image_index, fptr(c), c.shape[0], c.shape[1]))

    def get_iou_matrix(self, image_index=0):
        m = int(self.num_kept()[image_index])
        out = np.empty((m, m), np.float32)
        if m:
            self._chk(self.lib.bod_get_iou_matrix(self.h, image_index, fptr(out)))
        return out

    def cluster_fuse(self):
        self._chk(self.lib.bod_cluster_fuse(self.h))

    def get_detections(self, image_index=0):
        k = self.K
        scores = np.empty((k, self.Ccls), np.float32)
        means = np.empty((k, 4), np.float32)
        covs = np.empty((k, 4, 4), np.float32)
        counts = np.empty((k, self.Ccls), np.float32)
        n = C.c_int32(0)
        self._chk(self.lib.bod_get_detections(self.h, image_index, C.byref(n), fptr(scores), fptr(means),
                                              fptr(covs), fptr(counts)))
        n = n.value
        return scores[:n].copy(), means[:n].copy(), covs[:n].copy(), counts[:n].copy()

    def get_detections_batch(self, out=None):
        """One D2H per array for the whole batch. Returns dict of padded arrays + 'num' [B]."""
        b, k, c = self.B, self.K, self.Ccls
        if out is None:
            out = {"num": np.empty(b, np.int32), "scores": np.empty((b, k, c), np.float32),
                   "means": np.empty((b, k, 4), np.float32), "covs": np.empty((b, k, 4, 4), np.float32),
                   "counts": np.empty((b, k, c), np.float32)}
        self._chk(self.lib.bod_get_detections_batch(self.h, iptr(out["num"]), fptr(out["scores"]),
                                                    fptr(out["means"]), fptr(out["covs"]), fptr(out["counts"])))
        return out

    def wait_slot(self, slot):
        """Block until the batch in ``slot`` is complete without copying anything."""
        self._chk(self.lib.bod_collect(self.h, slot, None, None, None, None, None))

    def device_raw_pointers(self, mark_ready=False):
        """Device addresses (cls, box, cov-or-None) of the raw head outputs [B,N,A,.] fp32."""
        ptrs = (C.c_void_p * 3)()
        self._chk(self.lib.bod_device_raw(self.h, ptrs, int(mark_ready)))
        return [ptrs[i] for i in range(3)]

    def device_detection_pointers(self, slot=0):
        ptrs = (C.c_void_p * 5)()
        self._chk(self.lib.bod_device_detections(self.h, slot, ptrs))
        b, k, c = self.B, self.K, self.Ccls
        shapes = [(b,), (b, k, c), (b, k, 4), (b, k, 16), (b, k, c)]
        names = ["num", "scores", "means", "covs", "counts"]
        return {n: (int(p), s) for n, p, s in zip(names, ptrs, shapes)}

    def bench_head_conv(self, layer=1, variant=0, iters=10):
        ms, fl = C.c_double(0), C.c_double(0)
        self._chk(self.lib.bod_bench_head_conv(self.h, layer, variant, iters, C.byref(ms), C.byref(fl)))
        return ms.value, fl.value

    # ------------------------------------------------------------------ measurement
    def gather_detections(self, slot=-1, comm=None, world=1, rank=0, root=0, want_host=True):
        """The path's one multi-GPU exchange through the C ABI (``bod_gather_detections``): packs this batch's detection
        records on the device and gathers every rank's block on ``root`` with ONE RCCL gather.  ``comm``: an ``ncclComm_t``
        as an integer / ``c_void_p`` (None: single process).  ``slot``: ticket of ``infer_async`` (-1 after ``infer``).
        Returns [world, B, K, 1+4+16+2C] float32 on the root (None elsewhere); unpack with distributed.unpack_records.
        ``want_host=False`` (root only): nothing is copied or waited for; returns ``(device pointer, shape)`` of the gathered
        block, complete after ``collect(slot)`` (a ticket) or ``synchronize()`` (slot -1) -- see include/bayesod.h."""
        w = int(self.lib.bod_record_width(self.h))
        out = None
        dev = C.c_void_p(0)
        if rank == root and want_host:
            out = np.empty((world, self.B, self.K, w), np.float32)
        self._chk(self.lib.bod_gather_detections(self.h, int(slot), C.c_void_p(int(comm) if comm else 0), int(world), int(rank), int(root),
                                                 out.ctypes.data if out is not None else None,
                                                 C.byref(dev) if rank == root else None))
        if rank == root and not want_host:
            return int(dev.value or 0), (world, self.B, self.K, w)
        return out

    def plan_info(self):
        """{'aggregating', 'fused_head_outputs', 'row_reuse', 'fan_out_row_reuse', 'ops', 'plane_row_reuse_layers', 'tower_mx'} of the
        forward plan (bod_plan_info)."""
        info = (C.c_int32 * 8)()
        self._chk(self.lib.bod_plan_info(self.h, info))
        return {"aggregating": bool(info[0]), "fused_head_outputs": bool(info[1]), "row_reuse": bool(info[2]),
                "fan_out_row_reuse": bool(info[3]), "ops": int(info[4]), "plane_row_reuse_layers": int(info[5]), "tower_mx": bool(info[6]), "tower_mx_format": int(info[6])}

    @property
    def aggregating(self):
        return self.plan_info()["aggregating"]

    def profile_begin(self, which=None):
        """which: None keeps the current selection; 0 = every head 3x3 launch, 1 = the row-reuse tower kernel's launches
        only (one kernel symbol), 2 = the others (the fan-out launch of the first tower layer)."""
        if which is not None:
            self._chk(self.lib.bod_profile_select(self.h, int(which)))
        self._chk(self.lib.bod_profile_begin(self.h))

    def profile_end(self):
        hm, pm, fl = C.c_double(0), C.c_double(0), C.c_double(0)
        hl, pl = C.c_int64(0), C.c_int64(0)
        self._chk(self.lib.bod_profile_end(self.h, C.byref(hm), C.byref(hl), C.byref(fl), C.byref(pm), C.byref(pl)))
        return {"head_conv_ms": hm.value, "head_conv_launches": hl.value, "head_conv_flops": fl.value,
                "posterior_ms": pm.value, "posterior_launches": pl.value}


def stage_conv(x, w, bias=None, stride=1, padding="same", relu=False, residual=None, dropout_rate=0.0,
               seed=0, layer_id=0, image_id=0, round_output_bf16=False, device=0, precision='bf16'):
    """One convolution through the pipeline's MFMA kernel (``bod_stage_conv``), for parity tests.
    x [B,H,W,Cin], w HWIO; returns [B,OH,OW,Cout] float32.  precision='f16mx' (a head-tower layer: 3x3, SAME, 256 -> 256):
    round_output_bf16 = 0 / 1 / 2 selects hx -> pairs / hx -> hx / pairs -> hx (include/bayesod.h)."""
    lib = _lib.load()
    x, w = as_f32(x), as_f32(w)
    b, h, wd, cin = x.shape
    kh, kw, cin2, cout = w.shape
    if cin2 != cin:
        raise ValueError("kernel Cin %d != input Cin %d" % (cin2, cin))
    if padding == "same":
        oh, ow = -(-h // stride), -(-wd // stride)
    else:
        oh, ow = (h - kh) // stride + 1, (wd - kw) // stride + 1
    out = np.empty((b, oh, ow, cout), np.float32)
    bias = as_f32(bias) if bias is not None else None
    residual = as_f32(residual) if residual is not None else None
    st = lib.bod_stage_conv(device, fptr(x), b, h, wd, cin, fptr(w), fptr(bias), kh, kw, cout, stride,
                            int(padding == "same"), int(relu), fptr(residual), float(dropout_rate), seed,
                            layer_id, image_id, int(round_output_bf16), PRECISIONS[precision], fptr(out))
    _lib.check(lib, None, st)
    return out


def stage_conv_wgrad(x, dy, kernel_hw, stride=1, padding="same", ksplit=0, device=0):
    """Weight / bias gradient of one Conv2D through the MFMA kernel (``bod_stage_conv_wgrad``): returns
    (dw [KH,KW,Cin,Cout], db [Cout]) for layer input x [B,H,W,Cin] and output gradient dy [B,OH,OW,Cout]."""
    lib = _lib.load()
    x, dy = as_f32(x), as_f32(dy)
    b, h, wd, cin = x.shape
    kh, kw = int(kernel_hw[0]), int(kernel_hw[1])
    cout = dy.shape[-1]
    dw = np.empty((kh, kw, cin, cout), np.float32)
    db = np.empty((cout,), np.float32)
    st = lib.bod_stage_conv_wgrad(device, fptr(x), b, h, wd, cin, fptr(dy), kh, kw, cout, stride,
                                  int(padding == "same"), int(ksplit), fptr(dw), fptr(db))
    _lib.check(lib, None, st)
    return dw, db


def stage_conv_dgrad(dy, w, padding="same", device=0):
    """Input gradient of a stride-1 Conv2D: the forward kernel on dy with the spatially flipped, cin/cout-swapped
    weights (SAME: symmetric padding for odd kernels; VALID forward = full correlation, not covered here)."""
    w = as_f32(w)
    if padding != "same" or w.shape[0] % 2 == 0 or w.shape[1] % 2 == 0:
        raise ValueError("stage_conv_dgrad covers stride-1 SAME convolutions with odd kernels")
    wt = np.ascontiguousarray(np.transpose(w[::-1, ::-1], (0, 1, 3, 2)))
    return stage_conv(dy, wt, None, stride=1, padding="same", device=device)
