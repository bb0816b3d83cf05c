"""Validation of the checkpoints a training run writes -- the counterpart of
src/retina_net/experiments/run_validation.py:27-228 (loop body :108-204, ``val_single_step`` :230-260):

    python -m bayes_od_rc_amd.run_validation --gpu_device 0 --yaml_path <yaml> [--data_split val]
                                             [--dataset | --synthetic N --image_size H W] [--poll SECONDS]

For every checkpoint of ``<data_dir>/outputs/<checkpoint_name>/checkpoints`` that ``evaluated_ckpts.txt`` does not list
yet: one plain forward per frame (``train_val_test='validation'``: no MC dropout), the losses of the frame, the
validation post-processing (softmax, background filter, soft-NMS on the top score: ``post_process_predictions``,
validation_utils.py:10-77, on the device) and the predictions in the dataset's format under
``predictions/validation/<ckpt_id>/data`` (BDD: one ``predictions.json``; KITTI: one text file per frame), followed by
the AP report when the frames carry BDD-format labels.  Differences from the reference, on purpose: checkpoints are
the ``.npz`` files ``run_training`` writes (the TF checkpoint format is ``convert_checkpoint``'s business), losses are
returned / printed instead of going to TensorBoard, and the process stops after one pass unless ``--poll`` asks for the
reference's wait-for-new-checkpoints loop.
"""
import argparse
import json
import os
import time

import numpy as np

from . import config_utils, constants
from .inference_utils import post_process_predictions
from .model import RetinaNetModel
from .writers import predictions_to_bdd_format, predictions_to_kitti_format, strip_checkpoint_id


def get_evaluated_ckpts(predictions_dir):
    """Checkpoint ids already validated (validation_utils.py:80-93): ``evaluated_ckpts.txt``, one integer per line."""
    path = os.path.join(predictions_dir, 'evaluated_ckpts.txt')
    if not os.path.exists(path):
        return np.zeros((0,), np.int32)
    return np.atleast_1d(np.loadtxt(path, dtype=np.int64)).astype(np.int32)


def write_evaluated_ckpts(predictions_dir, ckpt_ids):
    """Append (validation_utils.py:110-114)."""
    with open(os.path.join(predictions_dir, 'evaluated_ckpts.txt'), 'ba') as fp:
        np.savetxt(fp, np.atleast_1d(ckpt_ids), fmt='%d')


def val_single_step(model, sample_dict):
    """run_validation.py:230-260: (total_loss, loss_dict, prediction_dict) of one frame, plain forward."""
    image = np.asarray(sample_dict[constants.IMAGE_NORMALIZED_KEY], np.float32)
    image = image[None] if image.ndim == 3 else image
    prediction_dict = model(image, train_val_test='validation')
    total_loss, loss_dict = model.get_loss(sample_dict, prediction_dict)
    return total_loss, loss_dict, prediction_dict


def validate_checkpoint(config, checkpoint_path, samples, sample_ids, predictions_dir, categories=None):
    """One checkpoint over the validation frames.  Returns {'ckpt_id', 'mean_total_loss', 'mean_losses', 'num_frames',
    'num_detections', 'predictions'} ('predictions' = the BDD records, or None for KITTI)."""
    dataset = config['dataset_config']['dataset']
    ckpt_id = strip_checkpoint_id(checkpoint_path[:-4] if checkpoint_path.endswith('.npz') else checkpoint_path)
    out_dir = os.path.join(predictions_dir, 'validation', str(ckpt_id), 'data')
    os.makedirs(out_dir, exist_ok=True)
    model = RetinaNetModel(config['model_config'])
    model.load_weights(checkpoint_path)
    records, totals, sums, ndet = [], [], {}, 0
    for sample, sid in zip(samples, sample_ids):
        total_loss, loss_dict, prediction_dict = val_single_step(model, sample)
        totals.append(float(total_loss))
        for k, v in loss_dict.items():
            sums[k] = sums.get(k, 0.0) + float(v)
        batched = dict(sample)
        batched[constants.IMAGE_NORMALIZED_KEY] = np.asarray(sample[constants.IMAGE_NORMALIZED_KEY])[None]
        classes, boxes = post_process_predictions(batched, prediction_dict, dataset_name=dataset)
        ndet += len(boxes)
        if dataset == 'kitti':
            rows = predictions_to_kitti_format(boxes, classes)
            path = os.path.join(out_dir, sid + '.txt')
            np.savetxt(path, rows if rows.size else [], newline='\r\n', fmt='%s')
        else:
            records.extend(predictions_to_bdd_format(boxes, classes, sid, category_list=categories))
    if dataset != 'kitti':
        with open(os.path.join(out_dir, 'predictions.json'), 'w') as fp:
            json.dump(records, fp, indent=4, separators=(',', ': '))
    n = max(len(totals), 1)
    return {'ckpt_id': int(ckpt_id), 'mean_total_loss': float(np.mean(totals)) if totals else 0.0,
            'mean_losses': {k: v / n for k, v in sums.items()}, 'num_frames': len(totals), 'num_detections': int(ndet),
            'predictions': None if dataset == 'kitti' else records}


def list_checkpoints(checkpoint_dir):
    """(id, path) of every ckpt-<id>.npz, by id."""
    found = []
    for f in os.listdir(checkpoint_dir):
        if f.endswith('.npz'):
            found.append((int(strip_checkpoint_id(f[:-4])), os.path.join(checkpoint_dir, f)))
    return sorted(found)


def validate(config, samples, sample_ids, categories=None, gt_records=None, poll_seconds=None, max_polls=None):
    """The loop of run_validation.py:86-228: every checkpoint not yet listed in evaluated_ckpts.txt, in id order; with
    ``poll_seconds`` keep waiting for new ones (``max_polls`` bounds the waiting, for tests)."""
    root = os.path.join(config_utils.data_dir(), 'outputs', config['checkpoint_name'])
    checkpoint_dir = os.path.join(root, 'checkpoints')
    predictions_dir = os.path.join(root, 'predictions')
    os.makedirs(predictions_dir, exist_ok=True)
    if not os.path.exists(checkpoint_dir):
        raise ValueError('{} must have at least one checkpoint entry.'.format(checkpoint_dir))
    results, last_id, polls = [], -1, 0
    while True:
        done = set(int(v) for v in get_evaluated_ckpts(predictions_dir))
        for ckpt_id, path in list_checkpoints(checkpoint_dir):
            if ckpt_id in done or ckpt_id <= last_id:
                continue
            print('\nRunning checkpoint ' + str(ckpt_id) + '\n')
            r = validate_checkpoint(config, path, samples, sample_ids, predictions_dir, categories)
            if gt_records is not None and r['predictions'] is not None:
                from .offline_eval import ap_report
                r['ap'] = ap_report(gt_records, r['predictions']) if r['predictions'] else None
            print('checkpoint {}: mean total loss {:0.3f} over {} frames, {} detections'.format(
                ckpt_id, r['mean_total_loss'], r['num_frames'], r['num_detections']))
            write_evaluated_ckpts(predictions_dir, np.array([ckpt_id]))
            results.append(r)
            last_id = ckpt_id
        polls += 1
        if not poll_seconds or (max_polls is not None and polls >= max_polls):
            return results
        print('\nNo new checkpoints found in %s. Will try again in %d seconds.' % (checkpoint_dir, poll_seconds))
        time.sleep(poll_seconds)


def main(argv=None):
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpu_device', type=str, default='0')
    ap.add_argument('--yaml_path', type=str, default=os.path.join(here, 'configs', 'retinanet_bdd_covar.yaml'))
    ap.add_argument('--data_split', type=str, default='val')
    ap.add_argument('--dataset', action='store_true', help='read the yaml\'s dataset (default: synthetic frames)')
    ap.add_argument('--synthetic', type=int, default=4)
    ap.add_argument('--image_size', type=int, nargs=2, default=[256, 256])
    ap.add_argument('--poll', type=int, default=0, help='seconds between scans for new checkpoints (0: one pass)')
    ap.add_argument('--seed', type=int, default=1)
    args = ap.parse_args(argv)
    config = config_utils.setup(config_utils.load_yaml(args.yaml_path), args)
    dataset_config = config['dataset_config']
    categories = None
    if args.dataset:
        from . import datasets
        handler = datasets.build_dataset(dataset_config, args.data_split)
        samples, sample_ids = list(handler.create_dataset()), list(handler.sample_ids)
        categories = handler.training_data_config['categories'] if dataset_config['dataset'] == 'bdd' else None
    else:
        from .run_training import synthetic_samples
        num_classes = int(config['model_config']['header']['num_classes'])
        samples = synthetic_samples(args.synthetic, args.image_size, dataset_config['anchor_generator'], num_classes, seed=args.seed)
        sample_ids = ['synthetic_%04d.jpg' % i for i in range(len(samples))]
        categories = ['car', 'truck', 'bus', 'person', 'rider', 'bike', 'motor'][:num_classes]
    return validate(config, samples, sample_ids, categories=categories, poll_seconds=args.poll or None)


if __name__ == '__main__':
    main()
