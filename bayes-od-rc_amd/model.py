"""``RetinaNetModel`` with the reference's constructor / call surface
(src/retina_net/models/retinanet_model.py:17-149) over the HIP engine.

    model = RetinaNetModel(config['model_config'])
    model.load_weights(weights_dict)                      # stands in for ckpt.restore (run_inference.py:120)
    prediction_dict = model(image_normalized, train_val_test='testing')

The engine (device buffers, row tables, packed weights) is built on the first call for the
image shape seen, like ``@tf.function`` tracing per input signature (inference_utils.py:13).
'testing' = N MC-dropout samples, 'validation' = one deterministic sample, 'training' = one sample with dropout on
(retinanet_model.py:113-147) through a training handle; the training STEP itself is run_training.Trainer.
"""
import numpy as np

from . import constants
from .engine import Engine, make_config


def fill_triangular_4(x):
    """tfp.math.fill_triangular for 10 -> 4x4 lower (retinanet_model.py:110)."""
    idx = ((4, -1, -1, -1), (8, 9, -1, -1), (7, 6, 5, -1), (3, 2, 1, 0))
    out = np.zeros(x.shape[:-1] + (4, 4), dtype=x.dtype)
    for r in range(4):
        for c in range(4):
            if idx[r][c] >= 0:
                out[..., r, c] = x[..., idx[r][c]]
    return out


class RetinaNetModel(object):
    def __init__(self, model_config, device=0, batch=1, seed=0, precision='bf16'):
        self.model_config = model_config
        names = model_config['output_names']
        self.compute_cls = 'classification' in names
        self.compute_reg = 'regression' in names
        self.compute_covar = 'regression_covar' in names
        if not (self.compute_cls and self.compute_reg):
            raise ValueError("the BayesOD path needs both 'classification' and 'regression' outputs")
        header = model_config['header']
        if 'num_classes' not in header or 'anchors_per_location' not in header:
            raise ValueError("model_config['header'] lacks num_classes / anchors_per_location: "
                             "run config_utils.setup(config, args) first")
        self.num_classes = int(header['num_classes'])                  # excl. background
        self.anchors_per_location = int(header['anchors_per_location'])
        self.dropout_rate = float(header['dropout_rate'])
        self.mc_dropout_samples = int(model_config['mc_dropout_samples'])
        self.device = device
        self.batch = batch
        self.seed = seed
        self.precision = precision        # 'bf16' (throughput), 'f16mx' / 'bf16x3' (the 1e-3 end-to-end parity modes: f16 + MX-fp6 towers / (hi, lo) bf16 pairs), 'f16mx4' (f16mx with fp4 cross terms: faster, covariances at 2.5e-3) or 'fp32' (exact fp32)
        self.backbone_depth = 101 if '101' in str(model_config.get('feature_extractor', {}).get('name', '')) else 50
        self.image_counter = 0
        self.prediction_dict = None
        self._weights = None
        self._engines = {}
        self._testing_overrides = {}

    # -- weights ---------------------------------------------------------------------------
    def load_weights(self, weights):
        """weights: dict or path to an .npz with keys '<layer>/<field>' (field in kernel, bias,
        gamma, beta, mean, var) -- Keras layer names, HWIO kernels."""
        if isinstance(weights, str):
            z = np.load(weights)
            d = {}
            for k in z.files:
                if k.startswith('__optimizer__/'):          # run_training checkpoints carry the Adam state beside the weights
                    continue
                layer, field = k.rsplit('/', 1)
                d.setdefault(layer, {})[field] = z[k]
            weights = d
        self._weights = weights
        for e in self._engines.values():
            e.close()
        self._engines = {}

    @staticmethod
    def save_weights_npz(weights, path):
        flat = {"%s/%s" % (l, f): a for l, e in weights.items() for f, a in e.items() if a is not None}
        np.savez(path, **flat)

    # -- engine ----------------------------------------------------------------------------
    def engine_for(self, image_hw, batch=None, mc_samples=None, training=False, **testing):
        if self._weights is None:
            raise ValueError("no weights loaded: call model.load_weights(...) (the reference raises "
                             "ValueError for a missing checkpoint, run_inference.py:56-58)")
        batch = batch or self.batch
        n = mc_samples or self.mc_dropout_samples
        precision = 'bf16' if training else self.precision          # training handles are bf16 (include/bayesod.h)
        key = (int(image_hw[0]), int(image_hw[1]), batch, n, precision, self.backbone_depth, bool(training))
        cfg = make_config(image_hw, batch=batch, mc_samples=n, num_classes=self.num_classes + 1,
                          anchors_per_location=self.anchors_per_location, device=self.device,
                          dropout_rate=self.dropout_rate, has_covar_head=self.compute_covar,
                          precision=precision, backbone_depth=self.backbone_depth, training=training, **testing)
        eng = self._engines.get(key)
        if eng is None:
            eng = Engine(cfg)
            eng.load_weights(self._weights)
            self._engines[key] = eng
        else:
            eng.update_config(cfg)
        return eng

    # -- call ------------------------------------------------------------------------------
    def __call__(self, input_tensor, train_val_test='testing', seed=None, image_id=None):
        """input_tensor: [B,H,W,3] float32 normalised BGR.  Returns the prediction dict with the
        reference's keys; tensors are [B*N, A, .] with the MC samples of an image contiguous,
        which for B == 1 is exactly the reference's [N, A, .] (retinanet_model.py:78-112)."""
        x = np.asarray(input_tensor, dtype=np.float32)
        if x.ndim != 4 or x.shape[-1] != 3:
            raise ValueError("expected an NHWC image batch with 3 channels, got shape %s" % (x.shape,))
        if train_val_test not in ('training', 'validation', 'testing'):
            raise ValueError("train_val_test must be 'training', 'validation' or 'testing', got %r" % (train_val_test,))
        # 'training': no MC tiling, dropout ON (retinanet_model.py:113-129) -- a training handle's forward;
        # 'validation': no tiling, dropout off (:130-147); 'testing': N tiled samples, dropout on iff N > 1 (:73-112)
        training = train_val_test == 'training'
        n = self.mc_dropout_samples if train_val_test == 'testing' else 1
        eng = self.engine_for(x.shape[1:3], batch=x.shape[0], mc_samples=n, training=training)
        seed = self.seed if seed is None else seed
        if image_id is None:
            image_id = self.image_counter
            self.image_counter += x.shape[0]
        eng.forward(x, seed=seed, first_image_id=image_id)
        cls, box, cov = eng.get_raw()
        b, nn, a = cls.shape[:3]
        self.prediction_dict = {
            constants.ANCHORS_CLASS_PREDICTIONS_KEY: cls.reshape(b * nn, a, -1),
            constants.ANCHORS_BOX_PREDICTIONS_KEY: box.reshape(b * nn, a, 4),
        }
        if self.compute_covar:
            self.prediction_dict[constants.ANCHORS_COVAR_PREDICTIONS_KEY] = \
                fill_triangular_4(cov.reshape(b * nn, a, 10))
        return self.prediction_dict

    call = __call__

    _REG_KIND = {'regression': 1, 'regression_var': 2, 'regression_covar': 3}

    def get_loss(self, sample_dict, prediction_dict):
        """Loss FORWARD with the reference's signature and return value ``(total_loss, loss_dict)``
        (retinanet_model.py:151-328; keys src/core/constants.py:66-71), evaluated on the device.
        The backward pass / optimizer belong to the training step."""
        import ctypes as C
        from . import _lib
        losses = self.model_config['losses']
        names, weights = list(losses['loss_names']), list(losses['loss_weights'])
        for n in names:
            if n != 'classification' and n not in self._REG_KIND:
                raise ValueError('Invalid Loss! Not implemented yet.', n)
        reg = [n for n in names if n in self._REG_KIND]
        if len(reg) > 1:
            raise ValueError("only one regression loss can be active")
        cls = _lib.as_f32(prediction_dict[constants.ANCHORS_CLASS_PREDICTIONS_KEY])
        box = _lib.as_f32(prediction_dict[constants.ANCHORS_BOX_PREDICTIONS_KEY])
        b, a, c = cls.shape
        cov = None
        kind = self._REG_KIND[reg[0]] if reg else 0
        if kind >= 2:
            m = _lib.as_f32(prediction_dict[constants.ANCHORS_COVAR_PREDICTIONS_KEY])
            if m.shape[-2:] == (4, 4):           # undo fill_triangular (retinanet_model.py:110)
                idx = {4: (0, 0), 8: (1, 0), 9: (1, 1), 7: (2, 0), 6: (2, 1), 5: (2, 2), 3: (3, 0), 2: (3, 1), 1: (3, 2), 0: (3, 3)}
                cov = np.stack([m[..., idx[k][0], idx[k][1]] for k in range(10)], axis=-1)
            else:
                cov = m
            cov = np.ascontiguousarray(cov, dtype=np.float32)
        anchors = _lib.as_f32(np.asarray(sample_dict[constants.ANCHORS_KEY]).reshape(-1, 4))
        pos = np.ascontiguousarray(np.asarray(sample_dict[constants.POSITIVE_ANCHORS_MASK_KEY]).reshape(b, a), dtype=np.uint8)
        neg = np.ascontiguousarray(np.asarray(sample_dict[constants.NEGATIVE_ANCHOR_MASK_KEY]).reshape(b, a), dtype=np.uint8)
        cls_t = _lib.as_f32(sample_dict[constants.ANCHORS_CLASS_TARGETS_KEY]).reshape(b, a, c)
        box_t = _lib.as_f32(sample_dict[constants.ANCHORS_BOX_TARGETS_KEY]).reshape(b, a, 4)
        out = (C.c_double * 4)()
        lib = _lib.load()
        u8 = C.POINTER(C.c_uint8)
        st = lib.bod_loss_forward(self.device, b, a, c, _lib.fptr(cls), _lib.fptr(cls_t), _lib.fptr(box), _lib.fptr(box_t),
                                  _lib.fptr(cov), _lib.fptr(anchors), pos.ctypes.data_as(u8), neg.ctypes.data_as(u8),
                                  int('classification' in names), kind,
                                  float(losses.get('label_smoothing_epsilon', 0.001)), out)
        _lib.check(lib, None, st)
        s_cls, s_cmp, s_reg, n_pos = out[0], out[1], out[2], out[3]
        denom = max(n_pos, 1.0)
        total, loss_dict = 0.0, {}
        for n in names:
            w = float(weights[names.index(n)])
            if n == 'classification':
                loss_dict['cls_loss'] = s_cls / denom * w
                total += loss_dict['cls_loss']
            elif n == 'regression':
                loss_dict['reg_loss'] = s_cmp / denom * w
                total += loss_dict['reg_loss']
            else:
                loss_dict['reg_loss'] = s_cmp / denom
                loss_dict['covariance_loss'] = s_reg / denom
                total += w * (s_cmp + s_reg) / denom
        return total, loss_dict
