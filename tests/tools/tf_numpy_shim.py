"""TEST INFRASTRUCTURE: a NumPy stand-in for the handful of TensorFlow / TensorFlow-Probability calls that
``bayes_od_inference`` makes (src/retina_net/experiments/inference_utils.py:13-277 and the box helpers it calls in
src/retina_net/anchor_generator/box_utils.py), so that the reference's OWN SOURCE can be executed in the build container --
which has no TensorFlow -- and its results compared with oracle/bayes_od.py (tests/golden/make_transcription_golden.py,
tests/test_reference_transcription.py).

What this pins and what it does not.  It pins the TRANSCRIPTION: every formula, axis, transpose, mixing weight and branch of
the reference function is executed as written, so a mis-restated line in the oracle shows up.  It does NOT pin TensorFlow's
op semantics -- each stand-in below follows the op's documented behaviour as the builder read it -- and the two stochastic /
library pieces are injected: ``Categorical.sample`` draws from the oracle's uniforms and
``tf.image.non_max_suppression_with_scores`` calls oracle/nms.py.  The oracle therefore stays "parity unpinned" for a1-a15.

Never imported by the product; never shipped to the GPU box as anything but this file (the reference source it runs stays
under /root/reference).
"""
import sys
import types

import numpy as np


def _arr(x):
    return x if isinstance(x, np.ndarray) else np.asarray(x)


class _Categorical:
    """tfp.distributions.Categorical(probs=p): .sample(n) -> [n, *batch] class indices.  The draws come from the uniforms
    installed with ``set_uniforms`` by the oracle's rule: class = first c with cumsum(p)[c] > u * cumsum(p)[C-1]."""
    uniforms = None

    def __init__(self, probs=None, logits=None):
        assert probs is not None and logits is None
        self.probs = _arr(probs)

    def sample(self, n):
        u = _Categorical.uniforms
        assert u is not None and u.shape == (self.probs.shape[0], n), "install uniforms [A, n] first"
        a, c = self.probs.shape
        cdf = np.zeros_like(self.probs)
        acc = np.zeros(a, dtype=self.probs.dtype)
        for j in range(c):
            acc = acc + self.probs[:, j]
            cdf[:, j] = acc
        t = u.astype(self.probs.dtype) * cdf[:, -1:]
        cls = np.minimum((cdf[:, None, :] <= t[:, :, None]).sum(axis=2), c - 1)
        return cls.T.astype(np.int64)                      # [n, A]


def set_uniforms(u):
    _Categorical.uniforms = None if u is None else np.asarray(u)


def _one_hot(indices, depth, on_value=1.0, off_value=0.0, axis=None, dtype=None):
    idx = _arr(indices)
    depth = int(depth)
    out = np.where(idx[..., None] == np.arange(depth), on_value, off_value)
    assert axis in (None, -1)
    return out.astype(dtype or np.result_type(type(on_value), np.float32))


def _boolean_mask(tensor, mask, axis=0):
    t, m = _arr(tensor), _arr(mask).astype(bool)
    axis = int(axis or 0)
    assert m.ndim == 1
    return np.compress(m, t, axis=axis)


def _set_diag(x, diagonal):
    out = np.array(_arr(x), copy=True)
    d = _arr(diagonal)
    n = min(out.shape[-2], out.shape[-1])
    i = np.arange(n)
    out[..., i, i] = d
    return out


def _diag_part(x):
    return np.diagonal(_arr(x), axis1=-2, axis2=-1).copy()


def _tensor_diag(d):
    d = _arr(d)
    assert d.ndim == 1
    return np.diag(d)


def _matmul(a, b, transpose_a=False, transpose_b=False):
    a, b = _arr(a), _arr(b)
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return np.matmul(a, b)


def _softmax(x, axis=-1):
    x = _arr(x)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _tile(x, multiples):
    return np.tile(_arr(x), [int(m) for m in _arr(multiples).reshape(-1)])


def _cast(x, dtype):
    return _arr(x).astype(dtype)


def _nms_with_scores(boxes, scores, max_output_size, iou_threshold=0.5, score_threshold=float("-inf"), soft_nms_sigma=0.0):
    from oracle import nms as oracle_nms
    idx, sc = oracle_nms.soft_nms(_arr(boxes), _arr(scores), max_output_size=int(max_output_size), iou_threshold=float(iou_threshold),
                                  soft_nms_sigma=float(soft_nms_sigma))[:2]
    return np.asarray(idx), np.asarray(sc)


class _NameScope:
    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class _Reduction:
    NONE, SUM, AUTO, SUM_OVER_BATCH_SIZE = "none", "sum", "auto", "sum_over_batch_size"


class _Loss:
    """keras.losses.Loss with reduction NONE: __call__ returns call()'s per-sample values unreduced."""

    def __init__(self, reduction=_Reduction.AUTO, name=None):
        assert reduction == _Reduction.NONE, "only Reduction.NONE is stood in for"
        self.reduction, self.name = reduction, name

    def __call__(self, y_true, y_pred, sample_weight=None):
        assert sample_weight is None
        return self.call(_arr(y_true), _arr(y_pred))


class _CategoricalCrossentropy(_Loss):
    """from_logits=True, label_smoothing=e: y = y_true * (1 - e) + e / C;  loss = -sum_c y * log_softmax(logits)."""

    def __init__(self, from_logits=False, label_smoothing=0, reduction=_Reduction.AUTO, name="categorical_crossentropy"):
        super().__init__(reduction=reduction, name=name)
        assert from_logits
        self.label_smoothing = label_smoothing

    def call(self, y_true, y_pred):
        c = y_true.shape[-1]
        y = y_true * (1.0 - self.label_smoothing) + self.label_smoothing / c
        z = y_pred - y_pred.max(axis=-1, keepdims=True)
        ls = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
        return -(y * ls).sum(axis=-1)


class _Huber(_Loss):
    """keras.losses.Huber as the reference USES it (retinanet_model.py:215-226: the result is reduced over axis 2 afterwards, so it
    must still have that axis): the element-wise form of TensorFlow 2.0 / 2.1, the version the reference's README names --
    0.5 e^2 for |e| <= delta, delta |e| - 0.5 delta^2 beyond.  (Later releases average over the last axis inside the loss.)"""

    def __init__(self, delta=1.0, reduction=_Reduction.AUTO, name="huber_loss"):
        super().__init__(reduction=reduction, name=name)
        self.delta = delta

    def call(self, y_true, y_pred):
        e = y_pred - y_true
        a = np.abs(e)
        return np.where(a <= self.delta, 0.5 * e * e, self.delta * a - 0.5 * self.delta * self.delta)


class _Anything:
    """keras.regularizers.X / keras.initializers.X ...: constructor arguments of the layers that the forward pass never reads."""

    def __getattr__(self, name):
        return type(name, (object,), {"__init__": lambda self, *a, **k: None})


# ---- Keras layers for the model files (feature_extractor.py, feature_decoder.py, multitask_headers.py, retinanet_model.py):
# the reference's __init__ / call graphs are executed as written; what a LAYER computes comes from oracle/network.py's primitives
# (the builder's reading of the Keras / TF op semantics) and the weights from ``set_weights``.  The comparison
# therefore checks the WIRING -- which tensor feeds which layer, strides, names, the RegHeader's uncalled conv, MC tiling,
# reshape / concat order -- not the primitives.
_WEIGHTS = {}
_DROPOUT_HOOK = [None]


def set_weights(w):
    _WEIGHTS.clear()
    _WEIGHTS.update(w or {})


def set_dropout_hook(fn):
    """fn(dropout_layer, call_index, x) -> keep mask broadcastable to x (bool / 0-1)."""
    _DROPOUT_HOOK[0] = fn


class _Layer:
    def __init__(self, *a, name=None, **k):
        self.name = name

    def __call__(self, *a, **k):
        return self.call(*a, **k)


class _Model(_Layer):
    pass


class _Conv2D(_Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", use_bias=True, activation=None, name=None, **k):
        super().__init__(name=name)
        assert activation in (None, "linear")
        self.filters, self.kernel_size, self.use_bias = filters, tuple(np.atleast_1d(kernel_size)), use_bias
        self.stride = int(np.atleast_1d(strides)[0])
        assert all(int(v) == self.stride for v in np.atleast_1d(strides))
        self.padding = padding

    def call(self, x):
        from oracle import network
        w = _WEIGHTS[self.name]
        k = np.asarray(w["kernel"], dtype=x.dtype)
        assert k.shape[0] == self.kernel_size[0] and k.shape[-1] == self.filters, (self.name, k.shape, self.kernel_size, self.filters)
        b = np.asarray(w["bias"], dtype=x.dtype) if (self.use_bias and w.get("bias") is not None) else None
        return network.conv2d(x, k, b, self.stride, self.padding)


class _BatchNormalization(_Layer):
    def call(self, x, training=False):
        from oracle import network
        assert training is False or training == 0
        return network.batchnorm_eval(x, _WEIGHTS[self.name])


class _ZeroPadding2D(_Layer):
    def __init__(self, padding=(1, 1), name=None, **k):
        super().__init__(name=name)
        self.padding = padding

    def call(self, x):
        ph, pw = self.padding                      # Keras: a tuple of 2 ints = symmetric (height, width) padding
        return np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))


class _MaxPooling2D(_Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", name=None, **k):
        super().__init__(name=name)
        self.pool, self.strides = tuple(pool_size), tuple(strides or pool_size)
        assert padding == "valid"

    def call(self, x):
        (kh, kw), (sh, sw) = self.pool, self.strides
        _, h, w, _ = x.shape
        oh, ow = (h - kh) // sh + 1, (w - kw) // sw + 1
        out = None
        for ky in range(kh):
            for kx in range(kw):
                p = x[:, ky:ky + (oh - 1) * sh + 1:sh, kx:kx + (ow - 1) * sw + 1:sw, :]
                out = p if out is None else np.maximum(out, p)
        return out


class _ReLU(_Layer):
    def call(self, x):
        return np.maximum(x, 0)


class _Dropout(_Layer):
    def __init__(self, rate=0.5, name=None, **k):
        super().__init__(name=name)
        self.rate, self.calls = rate, 0

    def call(self, x, training=False):
        i = self.calls
        self.calls += 1
        if not training:
            return x
        keep = _DROPOUT_HOOK[0](self, i, x)
        return x * x.dtype.type(np.float32(1.0 / (1.0 - self.rate))) * np.asarray(keep, dtype=x.dtype)


class _Layers(_Anything):
    Conv2D, BatchNormalization, ZeroPadding2D, MaxPooling2D, ReLU, Dropout = _Conv2D, _BatchNormalization, _ZeroPadding2D, _MaxPooling2D, _ReLU, _Dropout

    @staticmethod
    def add(tensors, name=None):
        out = tensors[0]
        for t in tensors[1:]:
            out = out + t
        return out


def _resize(images, size, method=None, name=None):
    from oracle import network
    assert method == "nearest"
    return network.resize_nearest(_arr(images), int(size[0]), int(size[1]))


def _fill_triangular(x):
    from oracle import network
    return network.fill_triangular_4(_arr(x))


def install():
    """Puts the stand-in modules into sys.modules (``tensorflow``, ``tensorflow_probability``)."""
    tf = types.ModuleType("tensorflow")
    tf.function = lambda f=None, **kw: f if f is not None else (lambda g: g)
    tf.float32, tf.float64, tf.int32, tf.int64 = np.float32, np.float64, np.int32, np.int64
    tf.shape = lambda x: np.asarray(_arr(x).shape, dtype=np.int32)
    tf.size = lambda x: np.int32(_arr(x).size)
    tf.cast = _cast
    tf.exp = lambda x: np.exp(_arr(x))
    tf.equal = lambda a, b: np.equal(a, b)
    tf.not_equal = lambda a, b: np.not_equal(a, b)
    tf.maximum = lambda a, b: np.maximum(a, b)
    tf.minimum = lambda a, b: np.minimum(a, b)
    tf.argmax = lambda x, axis=None: np.argmax(_arr(x), axis=axis).astype(np.int64)
    tf.reduce_mean = lambda x, axis=None, keepdims=False: np.mean(_arr(x), axis=axis, keepdims=keepdims)
    tf.reduce_sum = lambda x, axis=None, keepdims=False: np.sum(_arr(x), axis=axis, keepdims=keepdims)
    tf.reduce_max = lambda x, axis=None, keepdims=False: np.max(_arr(x), axis=axis, keepdims=keepdims)
    tf.reduce_min = lambda x, axis=None, keepdims=False: np.min(_arr(x), axis=axis, keepdims=keepdims)
    tf.one_hot = _one_hot
    tf.boolean_mask = _boolean_mask
    tf.zeros_like = lambda x: np.zeros_like(_arr(x))
    tf.ones_like = lambda x: np.ones_like(_arr(x))
    tf.matmul = _matmul
    tf.tile = _tile
    tf.expand_dims = lambda x, axis: np.expand_dims(_arr(x), axis)
    tf.squeeze = lambda x, axis=None: np.squeeze(_arr(x), axis=axis)
    tf.transpose = lambda x, perm=None: np.transpose(_arr(x), perm)
    tf.stack = lambda xs, axis=0: np.stack([_arr(x) for x in xs], axis=axis)
    tf.split = lambda x, n, axis=0: np.split(_arr(x), n, axis=axis)
    tf.clip_by_value = lambda x, lo, hi: np.clip(_arr(x), lo, hi)
    tf.nn = types.SimpleNamespace(softmax=_softmax)
    tf.math = types.SimpleNamespace(log=lambda x: np.log(_arr(x)))
    tf.linalg = types.SimpleNamespace(inv=lambda x: np.linalg.inv(_arr(x)), det=lambda x: np.linalg.det(_arr(x)), diag_part=_diag_part,
                                      set_diag=_set_diag, tensor_diag=_tensor_diag)
    tf.image = types.SimpleNamespace(non_max_suppression_with_scores=_nms_with_scores)
    tf.name_scope = _NameScope
    tf.constant = lambda v, dtype=None: np.asarray(v, dtype=dtype or (np.float32 if isinstance(v, float) else None))
    tf.divide = lambda x, y, name=None: _arr(x) / y
    tf.pow = lambda x, y: np.power(_arr(x), y)
    tf.identity = lambda x: x
    tf.linalg.norm = lambda x, ord="fro", axis=None: np.sqrt((_arr(x) ** 2).sum(axis=axis))
    tf.gather = lambda params, indices, axis=0: np.take(_arr(params), np.asarray(indices, dtype=np.int64), axis=axis)
    tf.concat = lambda xs, axis=0: np.concatenate([_arr(x) for x in xs], axis=axis)
    tf.reshape = lambda x, shape: np.reshape(_arr(x), [int(v) for v in shape])
    tf.image.resize = _resize
    tf.image.ResizeMethod = types.SimpleNamespace(NEAREST_NEIGHBOR="nearest")
    tf.keras = types.SimpleNamespace(
        Model=_Model, layers=_Layers(), regularizers=_Anything(), initializers=_Anything(),
        backend=types.SimpleNamespace(learning_phase=lambda: 0),
        losses=types.SimpleNamespace(Loss=_Loss, Reduction=_Reduction, CategoricalCrossentropy=_CategoricalCrossentropy, Huber=_Huber))
    tfp = types.ModuleType("tensorflow_probability")
    tfp.distributions = types.SimpleNamespace(Categorical=_Categorical)
    tfp.math = types.SimpleNamespace(fill_triangular=_fill_triangular)
    sys.modules["tensorflow"] = tf
    sys.modules["tensorflow_probability"] = tfp
    for name, typ in (("int", int), ("float", float), ("bool", bool)):      # numpy aliases a few reference helpers still use
        if not hasattr(np, name):
            setattr(np, name, typ)
    return tf, tfp
