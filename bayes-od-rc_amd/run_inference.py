"""Inference driver with the reference's CLI (src/retina_net/experiments/run_inference.py:254-299):

    python -m bayes_od_rc_amd.run_inference --gpu_device 0 --yaml_path <cfg.yaml> --data_split test \
        [--weights weights.npz] [--frames frames.npy | --synthetic N] [--image_size H W]

Datasets and TF checkpoints are outside the hot path: frames come
from an .npy of normalised BGR images or are synthetic, weights from an .npz with Keras names.
"""
import argparse
import os
import sys
import time

import numpy as np

from . import box_utils, config_utils, inference_utils, synthetic, writers
from .anchor_generator import FpnAnchorGenerator
from .model import RetinaNetModel


def test_model(config, args):
    test_config = config['testing_config']
    if test_config['uncertainty_method'] != 'bayes_od':
        raise ValueError("only uncertainty_method 'bayes_od' is supported (as in the reference release)")
    dataset_config = config['dataset_config']
    training_dataset = dataset_config['dataset']
    test_dataset = test_config['test_dataset']
    nms_config = test_config['nms_config']
    model = RetinaNetModel(config['model_config'], device=int(args.gpu_device), seed=args.seed)
    if args.weights:
        if not os.path.exists(args.weights):
            raise ValueError('%s must exist (no checkpoint entry)' % args.weights)
        model.load_weights(args.weights)
    else:
        model.load_weights(synthetic.make_weights(config['model_config']['header']['num_classes'] + 1,
                                                  config['model_config']['header']['anchors_per_location']))
    if args.frames:
        frames = np.load(args.frames).astype(np.float32)
    else:
        frames = synthetic.make_frames(args.synthetic, args.image_size[0], args.image_size[1])
    hw = frames.shape[1:3]
    gen = FpnAnchorGenerator(dataset_config['anchor_generator'])
    anchors = gen.generate_all((hw[0], hw[1], 3))
    batch = max(1, min(args.batch, len(frames)))
    orig = tuple(args.orig_size) if args.orig_size else (hw[0], hw[1])
    pipe = inference_utils.BayesOdPipeline(model, hw, batch, test_config['bayes_od_config'], nms_config,
                                           use_full_covar=test_config['use_full_covar'],
                                           dataset_name=test_dataset, orig_size=orig, anchors=anchors)
    predictions_dir = os.path.join(config_utils.data_dir(), 'outputs', config['checkpoint_name'], 'predictions')
    writer = writers.PredictionWriter(predictions_dir, test_dataset, test_config['ckpt_idx'],
                                      test_config['uncertainty_method'],
                                      test_config['bayes_od_config']['fusion_method'])
    categories = dataset_config[training_dataset]['training_data_config']['categories']
    start = time.time()
    n_done = 0
    for lo in range(0, len(frames) - batch + 1, batch):
        dets = pipe(frames[lo:lo + batch], seed=args.seed, first_image_id=lo)
        for b, (classes, boxes_vuhw, covs, counts) in enumerate(dets):
            boxes = box_utils.vuhw_to_vuvu_np(boxes_vuhw) if boxes_vuhw.size else boxes_vuhw
            mapped = classes
            if training_dataset != test_dataset and boxes.size > 0:
                mapped = inference_utils.map_dataset_classes(training_dataset, test_dataset, classes)
            writer.write('%06d' % (lo + b), boxes, mapped, boxes_vuhw, covs, classes, counts, categories)
            n_done += 1
        sys.stdout.write('\r{}'.format(n_done) + ' /' + str(len(frames)))
    writer.close()
    elapsed = time.time() - start
    print("\nMean frame rate: " + str(n_done / max(elapsed, 1e-9)))
    return writer.root


def main(argv=None):
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpu_device', type=str, default='0')
    ap.add_argument('--yaml_path', type=str, default=os.path.join(here, 'configs', 'retinanet_bdd_covar.yaml'))
    ap.add_argument('--data_split', type=str, default='test')
    ap.add_argument('--weights', type=str, default=None)
    ap.add_argument('--frames', type=str, default=None)
    ap.add_argument('--synthetic', type=int, default=8)
    ap.add_argument('--image_size', type=int, nargs=2, default=[512, 512])
    ap.add_argument('--orig_size', type=int, nargs=2, default=None)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args(argv)
    config = config_utils.load_yaml(args.yaml_path)
    config = config_utils.setup(config, args)
    return test_model(config, args)


if __name__ == '__main__':
    main()
