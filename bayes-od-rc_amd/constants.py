"""Dictionary keys and dataset tables shared with callers.

Same key strings / values as the reference's src/core/constants.py:12-63 (data, not logic), so
sample and prediction dictionaries are interchangeable.
"""
MEANS_DICT = {'ImageNet': [123.68, 116.78, 103.94], 'Kitti': [92.84, 97.80, 93.58]}

CATEGORY_IDX_MAPPING_DICTS = {
    'kitti': {'car': 0, 'pedestrian': 1, 'cyclist': 2, 'bknd': 3},
    'bdd': {'car': 0, 'truck': 1, 'bus': 2, 'person': 3, 'rider': 4, 'bike': 5, 'motor': 6, 'bknd': 7},
}
SET_TO_SET_MAPPING_DICTS = {
    'bdd_kitti': {'car': 'car', 'truck': 'car', 'bus': 'car', 'person': 'pedestrian',
                  'rider': 'cyclist', 'bike': 'cyclist', 'motor': 'cyclist', 'bknd': 'bknd'},
    'coco_rvc': {}, 'coco_pascal': {},
}

IMAGE_NORMALIZED_KEY = 'image_normalized'
ORIGINAL_IM_SIZE_KEY = 'im_size'
ANCHORS_KEY = 'anchors'
ANCHORS_BOX_TARGETS_KEY = 'anchors_box_targets'
ANCHORS_CLASS_TARGETS_KEY = 'anchors_class_targets'
POSITIVE_ANCHORS_MASK_KEY = 'positive_anchors_mask'
NEGATIVE_ANCHOR_MASK_KEY = 'negative_anchors_mask'

ANCHORS_BOX_PREDICTIONS_KEY = 'anchors_box_predictions'
ANCHORS_COVAR_PREDICTIONS_KEY = 'anchors_box_covar_predictions'
ANCHORS_CLASS_PREDICTIONS_KEY = 'anchors_class_predictions'
