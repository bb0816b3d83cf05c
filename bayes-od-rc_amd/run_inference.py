"""Inference driver with the reference's CLI (src/retina_net/experiments/run_inference.py:254-299):

    python -m bayes_od_rc_amd.run_inference --gpu_device 0 --yaml_path <cfg.yaml> --data_split test \
        [--weights weights.npz] [--dataset | --frames frames.npy | --synthetic N] [--image_size H W]

``--dataset`` reads the yaml's BDD / KITTI tree through ``datasets.build_dataset`` (the reference's default,
run_inference.py:60-72): frames are decoded on the host, uploaded as uint8 and normalised / resized on the GPU
(``bod_upload_frames_u8``); frames of one batch must share a size.  Otherwise frames come from an .npy of
normalised BGR images or are synthetic.  Weights: an .npz with Keras layer names (the TF-checkpoint converter is
SURVEY.md section 8 f3).
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
    if args.dataset:
        return _test_model_on_dataset(config, args, model)
    if args.frames:
        frames = np.load(args.frames).astype(np.float32)
    else:
        frames = synthetic.make_frames(args.synthetic, args.image_size[0], args.image_size[1])
    hw = frames.shape[1:3]
    gen = FpnAnchorGenerator(dataset_config['anchor_generator'])
    anchors = gen.generate_all((hw[0], hw[1], 3))
    batch = max(1, min(args.batch, len(frames)))
    orig = tuple(args.orig_size) if args.orig_size else (hw[0], hw[1])
    pipes = {}

    def pipe_for(b):                      # the tail (len(frames) % batch frames) runs through a smaller-batch handle
        if b not in pipes:
            pipes[b] = inference_utils.BayesOdPipeline(model, hw, b, test_config['bayes_od_config'], nms_config,
                                                       use_full_covar=test_config['use_full_covar'],
                                                       dataset_name=test_dataset, orig_size=orig, anchors=anchors)
        return pipes[b]
    predictions_dir = os.path.join(config_utils.data_dir(), 'outputs', config['checkpoint_name'], 'predictions')
    writer = writers.PredictionWriter(predictions_dir, test_dataset, test_config['ckpt_idx'],
                                      test_config['uncertainty_method'],
                                      test_config['bayes_od_config']['fusion_method'])
    categories = dataset_config[training_dataset]['training_data_config']['categories']
    start = time.time()
    n_done = 0
    for lo in range(0, len(frames), batch):              # every frame, like the reference's loop (run_inference.py:137)
        chunk = frames[lo:lo + batch]
        dets = pipe_for(len(chunk))(chunk, seed=args.seed, first_image_id=lo)
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


def _test_model_on_dataset(config, args, model):
    """The reference's loop over the dataset handler (run_inference.py:60-72,137-171): batch(1) there, batches
    of equally sized frames here; preprocessing on the device."""
    from . import constants, datasets
    test_config = config['testing_config']
    dataset_config = config['dataset_config']
    training_dataset, test_dataset = dataset_config['dataset'], test_config['test_dataset']
    handler = datasets.build_dataset(dict(dataset_config, dataset=test_dataset), 'test')
    kitti = test_dataset == 'kitti'
    predictions_dir = os.path.join(config_utils.data_dir(), 'outputs', config['checkpoint_name'], 'predictions')
    writer = writers.PredictionWriter(predictions_dir, test_dataset, test_config['ckpt_idx'],
                                      test_config['uncertainty_method'], test_config['bayes_od_config']['fusion_method'])
    categories = dataset_config[training_dataset]['training_data_config']['categories']
    gen = FpnAnchorGenerator(dataset_config['anchor_generator'])
    pipes = {}
    pending = []

    def flush():
        if not pending:
            return
        frames = np.stack([p[1] for p in pending])
        src_hw = frames.shape[1:3]
        hw = tuple(handler.resize_shape) if kitti else src_hw
        key = (src_hw, len(pending))
        if key not in pipes:
            pipes[key] = inference_utils.BayesOdPipeline(
                model, hw, len(pending), test_config['bayes_od_config'], test_config['nms_config'],
                use_full_covar=test_config['use_full_covar'], dataset_name=test_dataset, orig_size=src_hw,
                anchors=gen.generate_all((hw[0], hw[1], 3)))
        pipe = pipes[key]
        pipe.bind(orig_size=src_hw)      # the handle may be shared with a pipe of another source size: re-apply the KITTI scale
        pipe.engine.upload_frames_u8(frames, constants.MEANS_DICT[handler.im_normalization], aspect_resize=kitti)
        dets = pipe(None, seed=args.seed, first_image_id=pending[0][2])
        for (name, _, _), (classes, boxes_vuhw, covs, counts) in zip(pending, dets):
            boxes = box_utils.vuhw_to_vuvu_np(boxes_vuhw) if boxes_vuhw.size else boxes_vuhw
            mapped = classes
            if training_dataset != test_dataset and boxes.size > 0:
                mapped = inference_utils.map_dataset_classes(training_dataset, test_dataset, classes)
            writer.write(name, boxes, mapped, boxes_vuhw, covs, classes, counts, categories)
        del pending[:]

    start, n_done = time.time(), 0
    for idx, sample in enumerate(handler.create_dataset()):
        rgb = sample[datasets.IMAGE_UINT8_KEY]
        name = os.path.splitext(os.path.basename(handler.im_paths[idx]))[0]
        if pending and (pending[0][1].shape != rgb.shape or len(pending) == args.batch):
            flush()
        pending.append((name, rgb, idx))
        n_done += 1
    flush()
    writer.close()
    print("\nMean frame rate: " + str(n_done / max(time.time() - start, 1e-9)))
    return writer.root


def main(argv=None):
    here = os.path.dirname(os.path.abspath(__file__))
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpu_device', type=str, default='0')
    ap.add_argument('--yaml_path', type=str, default=os.path.join(here, 'configs', 'retinanet_bdd_covar.yaml'))
    ap.add_argument('--data_split', type=str, default='test')
    ap.add_argument('--weights', type=str, default=None)
    ap.add_argument('--dataset', action='store_true', help='read the yaml\'s test dataset from disk')
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
