"""Training driver with the reference's CLI and loop structure (src/retina_net/experiments/run_training.py:22-299):

    python -m bayes_od_rc_amd.run_training --gpu_device 0 --yaml_path <cfg.yaml> --data_split train \
        [--weights init.npz] [--dataset | --synthetic N] [--image_size H W] [--steps K]

The step itself -- forward in training mode, losses, backward, global-norm clipping, Adam -- runs on the GPU in
``bod_train_step``; this file keeps what the reference keeps on the host: the piecewise-constant learning-rate
schedule (:48-61), the batching of the dataset handler's sample dictionaries, the summary print and the checkpoint
cadence.  Checkpoints are ``.npz`` files in the schema ``RetinaNetModel.load_weights`` reads (the TF-checkpoint
format is the converter's business, convert_checkpoint.py)."""
import argparse
import os
import time

# the training step is a chain of ~1 200 small launches on three streams: 8 hardware queues instead of the runtime's 4 keep its
# weight-gradient streams from sharing a queue with the main one (11.2 -> 10.85 ms, docs/DESIGN_HISTORY.md A.1).  Set by this entry
# point, before HIP initialises; an explicit GPU_MAX_HW_QUEUES wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

from . import config_utils, constants, synthetic
from .anchor_generator import FpnAnchorGenerator
from .engine import Engine, make_config
from .sample_builder import create_sample_dict

_REG_KIND = {'regression': 1, 'regression_var': 2, 'regression_covar': 3}


def piecewise_learning_rate(training_config, epoch_size):
    """tf.keras PiecewiseConstantDecay of run_training.py:48-61 -> function(step) (boundaries are inclusive on the left
    value: step <= boundary keeps the earlier rate)."""
    bounds = [b * epoch_size for b in training_config['decay_boundaries']]
    factors = [training_config['decay_factor'] ** i for i in range(len(bounds) + 1)]
    values = [float(np.round(training_config['initial_learning_rate'] * f, 8)) for f in factors]

    def lr(step):
        for b, v in zip(bounds, values):
            if step <= b:
                return v
        return values[-1]
    return lr


class Trainer(object):
    """Holds the training handle; ``train_single_step(sample_dicts)`` mirrors run_training.train_single_step (:208-247)
    and returns ``(total_loss, loss_dict)`` with the reference's keys."""

    def __init__(self, config, image_hw, weights, device=0, seed=0):
        model_config = config['model_config']
        header = model_config['header']
        losses = model_config['losses']
        self.loss_names, self.loss_weights = list(losses['loss_names']), list(losses['loss_weights'])
        reg = [n for n in self.loss_names if n in _REG_KIND]
        if 'classification' not in self.loss_names or len(reg) != 1:
            raise ValueError('Invalid Loss! Not implemented yet.', self.loss_names)
        self.reg_kind = _REG_KIND[reg[0]]
        self.w_cls = float(self.loss_weights[self.loss_names.index('classification')])
        self.w_reg = float(self.loss_weights[self.loss_names.index(reg[0])])
        self.label_smoothing = float(losses.get('label_smoothing_epsilon', 0.001))
        self.l2_rate = float(header.get('l2_norm_rate', 1e-6))
        self.batch = int(config['training_config']['minibatch_size'])
        self.seed = seed
        self.step = 0
        self.engine = Engine(make_config(image_hw, batch=self.batch, mc_samples=1, num_classes=int(header['num_classes']) + 1,
                                         anchors_per_location=int(header['anchors_per_location']), device=device,
                                         dropout_rate=float(header['dropout_rate']),
                                         has_covar_head='regression_covar' in model_config['output_names'], training=True,
                                         backbone_depth=101 if '101' in str(model_config.get('feature_extractor', {}).get('name', '')) else 50))
        self.engine.load_weights(weights)
        self._init = weights

    def train_single_step(self, sample_dicts, learning_rate):
        imgs = np.stack([s[constants.IMAGE_NORMALIZED_KEY] for s in sample_dicts]).astype(np.float32)
        if not self.engine._anchors_set:
            self.engine.set_anchors(np.asarray(sample_dicts[0][constants.ANCHORS_KEY], np.float32))
        out = self.engine.train_step(
            imgs, np.stack([s[constants.ANCHORS_CLASS_TARGETS_KEY] for s in sample_dicts]),
            np.stack([s[constants.ANCHORS_BOX_TARGETS_KEY] for s in sample_dicts]),
            np.stack([s[constants.POSITIVE_ANCHORS_MASK_KEY] for s in sample_dicts]),
            np.stack([s[constants.NEGATIVE_ANCHOR_MASK_KEY] for s in sample_dicts]),
            seed=self.seed, first_image_id=self.step * self.batch, reg_kind=self.reg_kind, label_smoothing=self.label_smoothing,
            w_cls=self.w_cls, w_reg=self.w_reg, l2_rate=self.l2_rate, learning_rate=learning_rate)
        self.step += 1
        loss_dict = {'cls_loss': out['cls_loss'], 'reg_loss': out['reg_loss'], 'regularization_loss': out['regularization_loss']}
        if self.reg_kind >= 2:
            loss_dict['covariance_loss'] = out['covariance_loss']
        return out['total_loss'], loss_dict

    def optimizer_state(self):
        """{'<layer>/<field>/adam_m' | '.../adam_v': array} + 'optimizer/step', 'trainer/step' -- what tf.train.Checkpoint(step,
        optimizer, net) stores beside the weights (run_training.py:68-82)."""
        out = {'optimizer/step': np.asarray(self.engine.train_step_count(), np.int64), 'trainer/step': np.asarray(self.step, np.int64)}
        for layer, fields in self._init.items():
            for f, a in fields.items():
                if a is None or f not in ('kernel', 'bias', 'gamma', 'beta'):
                    continue
                try:
                    for what in ('adam_m', 'adam_v'):
                        out['%s/%s/%s' % (layer, f, what)] = self.engine.train_get(layer, f, np.asarray(a).shape, what=what)
                except ValueError:
                    pass                                 # not a variable of the model (RegHeader's never-called conv_4)
        return out

    def restore_optimizer_state(self, state):
        """Inverse of optimizer_state(): Adam moments, the update counter (bias correction) and the trainer's own step (dropout
        image ids, learning-rate schedule)."""
        for key, a in state.items():
            parts = key.split('/')
            if len(parts) == 3 and parts[2] in ('adam_m', 'adam_v'):
                self.engine.train_set_moment(parts[0], parts[1], parts[2], a)
        self.engine.train_step_count(set_to=int(state['optimizer/step']))
        self.step = int(state['trainer/step'])

    def weights(self):
        """Current weights in the ``load_weights`` schema: trained tensors from the device, everything the step does not
        train (batch-norm moving statistics; RegHeader's never-called conv_4) as loaded."""
        out = {}
        for layer, fields in self._init.items():
            out[layer] = {}
            for f, a in fields.items():
                if a is None:
                    continue
                val = np.asarray(a, np.float32)
                if f in ('kernel', 'bias', 'gamma', 'beta'):
                    try:
                        val = self.engine.train_get(layer, f, val.shape)
                    except ValueError:
                        pass                             # not a variable of the model
                out[layer][f] = val
        return out


def synthetic_samples(n, image_hw, anchor_gen_config, num_classes, seed=0):
    """Random frames with a few random ground-truth boxes, through the same target generation as the dataset handlers."""
    rng = np.random.default_rng(seed)
    frames = synthetic.make_frames(n, image_hw[0], image_hw[1], seed=seed)
    out = []
    for i in range(n):
        g = int(rng.integers(2, 6))
        y1, x1 = rng.uniform(0, image_hw[0] * 0.6, g), rng.uniform(0, image_hw[1] * 0.6, g)
        hh, ww = rng.uniform(24, image_hw[0] * 0.4, g), rng.uniform(24, image_hw[1] * 0.4, g)
        boxes = np.stack([y1, x1, y1 + hh, x1 + ww], 1).astype(np.float32)
        onehot = np.eye(num_classes + 1, dtype=np.float32)[rng.integers(0, num_classes, g)]
        out.append(create_sample_dict(frames[i], anchor_gen_config, boxes, onehot, is_testing=False))
    return out


def train(config, args):
    training_config = config['training_config']
    dataset_config = config['dataset_config']
    num_classes = int(config['model_config']['header']['num_classes'])
    if args.dataset:
        from . import datasets
        handler = datasets.build_dataset(dataset_config, 'train')
        samples = list(handler.create_dataset())
        if dataset_config['dataset'] == 'kitti':
            raise ValueError("KITTI training needs the resized frames on the host; use --synthetic or BDD")
    else:
        samples = synthetic_samples(args.synthetic, args.image_size, dataset_config['anchor_generator'], num_classes, seed=args.seed)
    hw = samples[0][constants.IMAGE_NORMALIZED_KEY].shape[:2]
    mb = int(training_config['minibatch_size'])
    epoch_size = max(len(samples) // mb, 1)
    lr = piecewise_learning_rate(training_config, epoch_size)
    weights = args.weights if args.weights else synthetic.make_weights(num_classes + 1, int(config['model_config']['header']['anchors_per_location']))
    if isinstance(weights, str):
        z = np.load(weights)
        d = {}
        for k in z.files:
            if k.startswith(OPT_PREFIX):
                continue
            layer, field = k.rsplit('/', 1)
            d.setdefault(layer, {})[field] = z[k]
        weights = d
    ckpt_dir = os.path.join(config_utils.data_dir(), 'outputs', config['checkpoint_name'], 'checkpoints')
    os.makedirs(ckpt_dir, exist_ok=True)
    # ckpt.restore(manager.latest_checkpoint) (run_training.py:75-82): resume weights, Adam moments and the step counter
    opt_state, restored = None, None
    superseded = []
    if getattr(args, 'no_resume', False):
        # a fresh run: nothing is restored, and the previous run's checkpoints must not stay the "latest" of this one -- they are
        # MOVED ASIDE (never deleted), and only once the Trainer below has been built, so a bad config or weight file costs nothing
        superseded = sorted_checkpoints(ckpt_dir)
    else:
        for cand in reversed(sorted_checkpoints(ckpt_dir)):           # newest first; a damaged file falls back to the one before
            try:
                with np.load(cand) as z:
                    w, o = {}, {}
                    for k in z.files:
                        if k.startswith(OPT_PREFIX):
                            o[k[len(OPT_PREFIX):]] = z[k]
                        else:
                            layer, field = k.rsplit('/', 1)
                            w.setdefault(layer, {})[field] = z[k]
            except Exception as e:                                    # truncated / corrupt archive (zipfile.BadZipFile, EOFError, ...)
                print('Skipping unreadable checkpoint {}: {}'.format(cand, e))
                continue
            weights, opt_state, restored = w, o, cand
            break
    if restored is not None:
        print('Restored from {}'.format(restored))
    else:
        print('Initializing from scratch.')
    trainer = Trainer(config, hw, weights, device=int(args.gpu_device), seed=args.seed)
    if opt_state:
        trainer.restore_optimizer_state(opt_state)
    if superseded:
        aside = os.path.join(ckpt_dir, 'superseded-%s-%d' % (time.strftime('%Y%m%d-%H%M%S'), os.getpid()))
        os.makedirs(aside)
        for old in superseded:
            os.replace(old, os.path.join(aside, os.path.basename(old)))
        print('--no_resume: moved {} checkpoint(s) of the previous run to {}'.format(len(superseded), aside))
    total_steps = args.steps or epoch_size * int(training_config['max_epochs'])
    ckpt_every = max(int(epoch_size * training_config['checkpoint_interval']), 1)
    keep = int(training_config.get('max_checkpoints_to_keep', 10000))
    last = time.time()
    history = []
    for step in range(trainer.step, total_steps):
        lo = (step * mb) % max(len(samples) - mb + 1, 1)
        total_loss, loss_dict = trainer.train_single_step(samples[lo:lo + mb], lr(step))
        history.append(total_loss)
        if step % int(training_config['summary_interval']) == 0:
            print('Step {}, Total Loss {:0.3f}, Time Elapsed {:0.3f} s'.format(step, total_loss, time.time() - last))
            last = time.time()
        if (step + 1) % ckpt_every == 0 or step + 1 == total_steps:
            save_checkpoint(trainer, os.path.join(ckpt_dir, 'ckpt-%d.npz' % (step + 1)))
            for old in sorted_checkpoints(ckpt_dir)[:-keep]:          # CheckpointManager(max_to_keep) (run_training.py:72-74)
                os.remove(old)
    return history, ckpt_dir


OPT_PREFIX = '__optimizer__/'


def _checkpoint_step(name):
    """'ckpt-<int>.npz' -> int, anything else (ckpt-foo.npz, a temp file of an interrupted save) -> None"""
    if not (name.startswith('ckpt-') and name.endswith('.npz')):
        return None
    try:
        return int(name[5:-4])
    except ValueError:
        return None


def sorted_checkpoints(ckpt_dir):
    if not os.path.isdir(ckpt_dir):
        return []
    files = [f for f in os.listdir(ckpt_dir) if _checkpoint_step(f) is not None]
    return [os.path.join(ckpt_dir, f) for f in sorted(files, key=_checkpoint_step)]


def latest_checkpoint(ckpt_dir):
    c = sorted_checkpoints(ckpt_dir)
    return c[-1] if c else None


def save_checkpoint(trainer, path):
    """Weights in the schema RetinaNetModel.load_weights reads (keys '<layer>/<field>') + the optimizer state under
    '__optimizer__/...' (ignored by load_weights' consumers: the inference handle only looks weights up by layer name).
    Written to a temp file in the same directory and renamed into place, so a kill during the save never leaves a
    truncated 'latest' checkpoint."""
    flat = {"%s/%s" % (l, f): a for l, e in trainer.weights().items() for f, a in e.items() if a is not None}
    flat.update({OPT_PREFIX + k: v for k, v in trainer.optimizer_state().items()})
    tmp = os.path.join(os.path.dirname(path), '.tmp-%d-%s' % (os.getpid(), os.path.basename(path)))
    try:
        with open(tmp, 'wb') as fp:
            np.savez(fp, **flat)
            fp.flush()
            os.fsync(fp.fileno())
        os.replace(tmp, path)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)


def main(argv=None):
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpu_device', type=str, default='0')
    ap.add_argument('--yaml_path', type=str, default=os.path.join(here, 'configs', 'retinanet_bdd_covar.yaml'))
    ap.add_argument('--data_split', type=str, default='train')
    ap.add_argument('--weights', type=str, default=None)
    ap.add_argument('--dataset', action='store_true')
    ap.add_argument('--synthetic', type=int, default=6)
    ap.add_argument('--image_size', type=int, nargs=2, default=[256, 256])
    ap.add_argument('--steps', type=int, default=0)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no_resume', action='store_true', help='start from scratch: existing checkpoints of this run are not restored; they are moved to '
                    'checkpoints/superseded-<time>/ (never deleted) once the trainer has been built')
    args = ap.parse_args(argv)
    config = config_utils.setup(config_utils.load_yaml(args.yaml_path), args)
    return train(config, args)


if __name__ == '__main__':
    main()
