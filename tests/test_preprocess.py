"""CPU: the preprocessing oracle (oracle/preprocess.py) against known answers and an independent bilinear
implementation, and the dataset handlers (bayes_od_rc_amd/datasets.py) on tiny synthetic BDD / KITTI trees."""
import json
import os

import numpy as np
import pytest

from oracle import preprocess as pp


def test_kitti_geometry_known_answers():
    # KITTI 375x1242 -> resize_shape [384, 1248] (BASELINE config 4): limited by the width, 377x1248, 3 rows of top pad
    assert pp.preserve_aspect_size((375, 1242), (384, 1248)) == (377, 1248)
    assert pp.crop_or_pad_offsets((377, 1248), (384, 1248)) == (0, 0, 3, 0)
    # yaml default [512, 1696]: limited by the height
    assert pp.preserve_aspect_size((375, 1242), (512, 1696)) == (512, 1696)
    assert pp.preserve_aspect_size((370, 1224), (512, 1696)) == (512, 1694)
    assert pp.crop_or_pad_offsets((512, 1694), (512, 1696)) == (0, 0, 0, 1)
    # centred crop when the resized frame is larger (floor division of a negative difference)
    assert pp.crop_or_pad_offsets((10, 21), (7, 16)) == (1, 2, 0, 0)


@pytest.mark.parametrize("src,dst", [((37, 53), (40, 57)), ((375, 1242), (377, 1248)), ((64, 64), (23, 31))])
def test_bilinear_matches_independent_implementation(src, dst):
    import torch
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=src + (3,), dtype=np.uint8)
    got = pp.bilinear_resize(img, dst[0], dst[1])
    ref = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], size=dst,
                                          mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() < 2e-2           # same convention; the source coordinate is rounded differently (fp32 vs fp64 scale)
    # constant images stay constant, identity size is the identity
    assert np.array_equal(pp.bilinear_resize(np.full((5, 7, 3), 9, np.uint8), 11, 13), np.full((11, 13, 3), 9, np.float32))
    assert np.array_equal(pp.bilinear_resize(img, src[0], src[1]), img.astype(np.float32))


def test_kitti_preprocess_layout():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(375, 1242, 3), dtype=np.uint8)
    out = pp.kitti_preprocess(img, (384, 1248))
    assert out.shape == (384, 1248, 3) and out.dtype == np.float32
    bgr_pad = -np.asarray(pp.IMAGENET_MEANS, np.float32)[::-1]
    assert np.array_equal(out[:3], np.broadcast_to(bgr_pad, (3, 1248, 3)))       # zero pad, then mean subtraction
    assert np.array_equal(out[380:], np.broadcast_to(bgr_pad, (4, 1248, 3)))
    assert not np.array_equal(out[3], np.broadcast_to(bgr_pad, (1248, 3)))
    b = pp.bdd_preprocess(img)
    assert np.array_equal(b[..., 0], img[..., 2].astype(np.float32) - np.float32(103.94))
    assert np.allclose(pp.kitti_rescale_boxes([[10, 20, 100, 200]], (375, 1242), (384, 1248)),
                       [[10 / 375 * 384, 20 / 1242 * 1248, 100 / 375 * 384, 200 / 1242 * 1248]], rtol=1e-6)


ANCHOR_GEN = {'layers': [3, 4, 5, 6, 7], 'aspect_ratios': [[1, 1], [1, 2], [2, 1]], 'scales': [1.0, 1.26, 1.59],
              'min_positive_iou': 0.5, 'max_negative_iou': 0.4}


def _png(path, arr):
    from PIL import Image
    Image.fromarray(arr).save(path)


def test_bdd_handler(tmp_path):
    from bayes_od_rc_amd import constants, datasets
    root = tmp_path / "bdd100k"
    (root / "images" / "100k" / "val").mkdir(parents=True)
    (root / "labels").mkdir()
    rng = np.random.default_rng(2)
    frames = {}
    for name in ("b.png", "a.png"):
        frames[name] = rng.integers(0, 256, size=(128, 192, 3), dtype=np.uint8)
        _png(str(root / "images" / "100k" / "val" / name), frames[name])
    labels = [{"name": "a.png", "category": "car", "bbox": [10.0, 20.0, 90.0, 100.0]},
              {"name": "a.png", "category": "traffic sign", "bbox": [1, 2, 3, 4]},
              {"name": "a.png", "category": "person", "bbox": [100.0, 30.0, 120.0, 90.0]}]
    (root / "labels" / "val.json").write_text(json.dumps(labels))
    cfg = {'dataset': 'bdd', 'data_split': 'val', 'im_normalization': 'ImageNet', 'anchor_generator': ANCHOR_GEN,
           'bdd': {'paths_config': {'dataset_dir': str(root), '100k_or_10k': '100k'},
                   'training_data_config': {'categories': ['car', 'truck', 'bus', 'person', 'rider', 'bike', 'motor'],
                                            'frac_training_data': 1.0}}}
    h = datasets.build_dataset(cfg, 'val')
    assert h.epoch_size == 2 and h.sample_ids == ["a.png", "b.png"] and not h.is_testing
    samples = list(h.create_dataset())
    s = samples[0]
    assert np.array_equal(s[datasets.IMAGE_UINT8_KEY], frames["a.png"])
    assert np.array_equal(s[constants.IMAGE_NORMALIZED_KEY], pp.bdd_preprocess(frames["a.png"]))
    assert list(s[constants.ORIGINAL_IM_SIZE_KEY]) == [128, 192, 3]
    a = s[constants.ANCHORS_KEY].shape[0]
    assert a == 9 * (16 * 24 + 8 * 12 + 4 * 6 + 2 * 3 + 1 * 2)
    assert s[constants.ANCHORS_CLASS_TARGETS_KEY].shape == (a, 8) and s[constants.POSITIVE_ANCHORS_MASK_KEY].any()
    cls, box, no_gt = h._read_labels("a.png")
    assert not no_gt and box.tolist() == [[20.0, 10.0, 100.0, 90.0], [30.0, 100.0, 90.0, 120.0]]       # (y1,x1,y2,x2)
    assert cls.argmax(1).tolist() == [0, 3]
    cls, box, no_gt = h._read_labels("b.png")
    assert no_gt and box.tolist() == [[0.0, 0.0, 1.0, 1.0]] and cls.shape == (1, 8) and cls.sum() == 0
    t = datasets.build_dataset(cfg, 'test')
    assert t.is_testing and constants.ANCHORS_BOX_TARGETS_KEY not in next(iter(t.create_dataset()))
    with pytest.raises(FileNotFoundError):
        datasets.build_dataset(dict(cfg, bdd=dict(cfg['bdd'], paths_config={'dataset_dir': str(tmp_path / "nope"), '100k_or_10k': '100k'})), 'val')
    with pytest.raises(ValueError):
        datasets.build_dataset(dict(cfg, dataset='coco'), 'val')


def test_kitti_handler(tmp_path):
    from bayes_od_rc_amd import constants, datasets
    root = tmp_path / "object"
    (root / "training" / "image_2").mkdir(parents=True)
    (root / "training" / "label_2").mkdir()
    (root / "val.txt").write_text("000007\n000003\n")
    rng = np.random.default_rng(3)
    for sid in ("000003", "000007"):
        _png(str(root / "training" / "image_2" / (sid + ".png")), rng.integers(0, 256, size=(94, 310, 3), dtype=np.uint8))
    (root / "training" / "label_2" / "000007.txt").write_text(
        "Car 0.00 0 -1.57 100.00 20.00 200.00 80.00 1.5 1.6 3.9 1.0 1.5 10.0 -1.5\n"
        "Pedestrian 0.00 3 0.2 10.00 10.00 30.00 70.00 1.8 0.6 0.8 1.0 1.5 10.0 0.1\n"          # occlusion 3 > hard
        "Cyclist 0.60 0 0.2 40.00 10.00 60.00 70.00 1.8 0.6 0.8 1.0 1.5 10.0 0.1\n"             # truncation 0.6 > hard
        "Van 0.00 0 0.2 40.00 10.00 60.00 70.00 1.8 0.6 0.8 1.0 1.5 10.0 0.1\n"
        "Pedestrian 0.10 1 0.2 250.00 30.00 270.00 90.00 1.8 0.6 0.8 1.0 1.5 10.0 0.1\n")
    (root / "training" / "label_2" / "000003.txt").write_text("DontCare -1 -1 -10 5.0 5.0 9.0 9.0 -1 -1 -1 -1000 -1000 -1000 -10\n")
    cfg = {'dataset': 'kitti', 'data_split': 'val', 'im_normalization': 'ImageNet', 'anchor_generator': ANCHOR_GEN,
           'kitti': {'resize_shape': [128, 416], 'paths_config': {'dataset_dir': str(root), 'data_split_dir': 'training'},
                     'training_data_config': {'categories': ['car', 'pedestrian', 'cyclist'], 'difficulty': 'hard'}}}
    h = datasets.build_dataset(cfg, 'val')
    assert h.sample_ids == ["000007", "000003"]
    cls, box, no_gt = h._read_labels(h.label_paths[0])
    assert not no_gt and cls.tolist() == [[1, 0, 0, 0], [0, 1, 0, 0]]
    assert box.tolist() == [[20.0, 100.0, 80.0, 200.0], [30.0, 250.0, 90.0, 270.0]]
    cls, box, no_gt = h._read_labels(h.label_paths[1])
    assert no_gt and cls.tolist() == [[0, 0, 0, 1]] and box.tolist() == [[0.0, 0.0, 1.0, 1.0]]
    s = next(iter(h.create_dataset()))
    assert s[datasets.IMAGE_UINT8_KEY].shape == (94, 310, 3) and s[constants.IMAGE_NORMALIZED_KEY] is None
    assert s[constants.ANCHORS_KEY].shape[0] == 9 * (16 * 52 + 8 * 26 + 4 * 13 + 2 * 7 + 1 * 4)
    with pytest.raises(ValueError):
        datasets.build_dataset(dict(cfg, data_split='trainval'), 'val')
