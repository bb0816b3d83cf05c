"""Dataset handlers with the reference's constructor arguments, directory layout and label parsing
(src/retina_net/datasets/bdd/bdd_dataset_handler.py, src/retina_net/datasets/kitti/kitti_dataset_handler.py,
src/core/abstract_classes/dataset_handler.py), SURVEY.md section 8 f2.  ``create_dataset()`` returns a
Python generator of ``sample_dict``s instead of a ``tf.data.Dataset``; image files are decoded with PIL.

Frames are kept as decoded uint8 RGB under the extra key ``'image_uint8'``: the normalisation (mean
subtraction, BGR flip) and KITTI's bilinear aspect-preserving resize + crop/pad run on the GPU when the frames
are uploaded (``Engine.upload_frames_u8`` -> ``bod_upload_frames_u8``).  ``'image_normalized'`` is filled on
the host only where that is the plain mean subtraction of the BDD handler; the KITTI handler leaves it to
``normalized_on_device``."""
import csv
import glob
import json
import os
import random

import numpy as np

from . import box_utils, constants
from .sample_builder import create_sample_dict, normalize_frame

KITTI_DIFF_DICTS = {            # src/core/constants.py:5-9
    'easy': {'min_height': 40, 'max_occlusion': 0, 'max_truncation': 0.15},
    'moderate': {'min_height': 25, 'max_occlusion': 1, 'max_truncation': 0.30},
    'hard': {'min_height': 25, 'max_occlusion': 2, 'max_truncation': 0.50},
    'all': {'min_height': 0, 'max_occlusion': 3, 'max_truncation': 1.0}}
IMAGE_UINT8_KEY = 'image_uint8'


def check_data_dirs(folders):
    """datasets/dataset_utils.py:7-16: FileNotFoundError for a missing folder."""
    for folder in folders:
        if not os.path.exists(folder):
            raise FileNotFoundError('Folder does not exist: {}'.format(folder))


def kitti_labels_to_boxes_2d(labels):
    """datasets/dataset_utils.py:35-60: label rows -> [y1, x1, y2, x2]."""
    labels = np.asarray(labels)
    if labels.ndim < 2:
        labels = np.array([labels.tolist()])
    return np.array([[float(l[5]), float(l[4]), float(l[7]), float(l[6])] for l in labels])


def _decode(path):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


class DatasetHandler(object):
    def __init__(self, dataset_config):
        self.data_split = dataset_config['data_split']
        self.im_normalization = dataset_config['im_normalization']


class BddDatasetHandler(DatasetHandler):
    """bdd_dataset_handler.py:16-259."""

    def __init__(self, config, train_val_test):
        super().__init__(config)
        self.anchor_gen_config = config['anchor_generator']
        self.training_data_config = config['bdd']['training_data_config']
        paths_config = config['bdd']['paths_config']
        self.dataset_dir = os.path.expanduser(paths_config['dataset_dir'])
        if train_val_test == 'train':
            self.data_split_dir, self.label_file_name = 'train', 'train.json'
            self.frac_training_data = self.training_data_config['frac_training_data']
        else:
            self.data_split_dir, self.label_file_name = 'val', 'val.json'
            self.frac_training_data = 1.0
        self.im_dir = os.path.join(self.dataset_dir, 'images', paths_config['100k_or_10k'], self.data_split_dir)
        self.gt_label_dir = os.path.join(self.dataset_dir, 'labels')
        check_data_dirs([self.im_dir, self.gt_label_dir])
        self._load_sample_ids()
        self.epoch_size = len(self.sample_ids)
        with open(os.path.join(self.gt_label_dir, self.label_file_name), 'r') as fp:
            self.labels = json.load(fp)
        self.is_testing = (train_val_test == 'test')

    def _load_sample_ids(self):
        sample_ids = sorted(os.listdir(self.im_dir))          # os.listdir order is arbitrary; sorted for reproducibility
        if self.frac_training_data != 1.0 and self.data_split_dir == 'train':
            k = int(len(sample_ids) * self.frac_training_data)
            inds = np.random.choice(len(sample_ids), k, replace=False)
            sample_ids = [sample_ids[i] for i in inds]
        elif self.data_split_dir == 'train':
            random.shuffle(sample_ids)
        self.sample_ids = sample_ids
        self.im_paths = [self.im_dir + '/' + s for s in self.sample_ids]

    def set_sample_id(self, sample_index):
        self.im_paths = [self.im_paths[sample_index]]
        self.sample_ids = [self.sample_ids[sample_index]]

    def _read_labels(self, sample_id):
        """:204-259 -> (one-hot classes [G, C+1], boxes [G,4] (y1,x1,y2,x2), no_gt)."""
        categories = self.training_data_config['categories']
        frame = [l for l in self.labels if l['name'] == sample_id and l['category'] in categories]
        boxes = np.array([[l['bbox'][1], l['bbox'][0], l['bbox'][3], l['bbox'][2]] for l in frame])
        if boxes.size == 0:
            return (np.zeros((1, len(categories) + 1), np.float32), np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), True)
        onehot = np.zeros((len(frame), len(categories) + 1), np.float32)
        for i, l in enumerate(frame):
            onehot[i, categories.index(l['category'].lower())] = 1
        return onehot, boxes.astype(np.float32), False

    def create_sample_dict(self, im_path, sample_id):
        rgb = _decode(im_path)
        norm = normalize_frame(rgb, self.im_normalization)
        cls_gt, box_gt, _ = self._read_labels(sample_id)
        sample = create_sample_dict(norm, self.anchor_gen_config, box_gt, cls_gt, is_testing=self.is_testing)
        sample[constants.ORIGINAL_IM_SIZE_KEY] = np.asarray(rgb.shape, dtype=np.int32)
        sample[IMAGE_UINT8_KEY] = rgb
        return sample

    def create_dataset(self):
        return (self.create_sample_dict(p, s) for p, s in zip(self.im_paths, self.sample_ids))


class KittiDatasetHandler(DatasetHandler):
    """kitti_dataset_handler.py:17-300."""

    def __init__(self, config, train_val_test):
        super().__init__(config)
        self.resize_shape = [int(v) for v in config['kitti']['resize_shape']]
        self.training_data_config = config['kitti']['training_data_config']
        self.anchor_gen_config = config['anchor_generator']
        paths_config = config['kitti']['paths_config']
        self.dataset_dir = os.path.expanduser(paths_config['dataset_dir'])
        self.data_split_dir = paths_config['data_split_dir']
        self.im_dir = os.path.join(self.dataset_dir, self.data_split_dir, 'image_2')
        self.gt_label_dir = os.path.join(self.dataset_dir, self.data_split_dir, 'label_2')
        check_data_dirs([self.im_dir, self.gt_label_dir])
        splits = [os.path.splitext(os.path.basename(p))[0] for p in glob.glob(self.dataset_dir + '/*.txt')]
        if self.data_split not in splits:
            raise ValueError('Invalid dataset_split: {}. Possible splits include: {}'.format(self.data_split, splits))
        self.sample_ids = list(self._load_sample_ids())
        if train_val_test == 'train':
            random.shuffle(self.sample_ids)
        self.epoch_size = len(self.sample_ids)
        self._create_sample_paths(self.sample_ids)
        self.is_testing = (train_val_test == 'test')

    def _load_sample_ids(self):
        out = []
        with open(os.path.join(self.dataset_dir, self.data_split + '.txt'), 'r') as f:
            for row in csv.reader(f, delimiter=' '):
                out.extend(s for s in row if s)
        return np.array(out)

    def _create_sample_paths(self, sample_ids):
        self.im_paths = [self.im_dir + '/' + s + '.png' for s in sample_ids]
        self.label_paths = [self.gt_label_dir + '/' + s + '.txt' for s in sample_ids]

    def set_paths(self, sample):
        self.im_paths = [self.im_dir + '/' + sample + '.png']
        self.label_paths = [self.gt_label_dir + '/' + sample + '.txt']

    def _read_labels(self, label_path):
        """:234-299: difficulty / category filter, 4-wide one-hot (car, pedestrian, cyclist, bknd)."""
        diff = KITTI_DIFF_DICTS[self.training_data_config['difficulty'].lower()]
        categories = self.training_data_config['categories']
        labels = np.loadtxt(label_path, delimiter=' ', dtype=str, usecols=np.arange(0, 15), ndmin=2)
        heights = labels[:, 7].astype(np.float32) - labels[:, 5].astype(np.float32)
        keep = (np.asarray([c.lower() in categories for c in labels[:, 0]], dtype=bool)
                & (heights >= diff['min_height'])
                & (labels[:, 1].astype(np.float64) <= diff['max_truncation'])
                & (labels[:, 2].astype(np.float64) <= diff['max_occlusion']))
        labels = labels[keep]
        if labels.shape[0] == 0:
            return np.array([[0, 0, 0, 1]], np.float32), np.array([[0.0, 0.0, 1.0, 1.0]], np.float32), True
        onehot = {'car': [1, 0, 0, 0], 'pedestrian': [0, 1, 0, 0], 'cyclist': [0, 0, 1, 0]}
        cls = [onehot[c.lower()] for c in labels[:, 0] if c.lower() in onehot]
        return np.array(cls, np.float32), kitti_labels_to_boxes_2d(labels).astype(np.float32), False

    def create_sample_dict(self, im_path, label_path):
        """Anchors / targets are generated for the RESIZED frame (resize_shape); GT boxes are divided by the
        original size and multiplied by the final size as the reference does (:132-135)."""
        rgb = _decode(im_path)
        cls_gt, box_gt, _ = self._read_labels(label_path)
        oh, ow = rgb.shape[:2]
        nh, nw = self.resize_shape
        box_gt = (box_gt / np.array([oh, ow, oh, ow], np.float32)) * np.array([nh, nw, nh, nw], np.float32)
        placeholder = np.zeros((nh, nw, 3), np.float32)       # shape carrier: the pixels are produced on the device
        sample = create_sample_dict(placeholder, self.anchor_gen_config, box_gt, cls_gt, is_testing=self.is_testing)
        sample[constants.IMAGE_NORMALIZED_KEY] = None
        sample[constants.ORIGINAL_IM_SIZE_KEY] = np.asarray(rgb.shape, dtype=np.int32)
        sample[IMAGE_UINT8_KEY] = rgb
        return sample

    def create_dataset(self):
        return (self.create_sample_dict(p, l) for p, l in zip(self.im_paths, self.label_paths))


def normalized_on_device(engine, frames_u8, im_normalization='ImageNet', aspect_resize=False):
    """uint8 RGB frames [B,h,w,3] -> the reference's 'image_normalized' tensors [B,H,W,3], computed by the device
    preprocessing kernel and left resident in the engine's image buffer (forward(None) / infer(None) use them)."""
    engine.upload_frames_u8(frames_u8, constants.MEANS_DICT[im_normalization], aspect_resize=aspect_resize)
    return engine.get_images()


def build_dataset(dataset_config, train_val_test):
    """src/retina_net/builders/dataset_handler_builder.py:5-25."""
    if dataset_config['dataset'] == 'kitti':
        return KittiDatasetHandler(dataset_config, train_val_test)
    if dataset_config['dataset'] == 'bdd':
        return BddDatasetHandler(dataset_config, train_val_test)
    raise ValueError('Invalid dataset type {}'.format(dataset_config['dataset']))
