"""GPU: bod_upload_frames_u8 (device-side dataset-handler preprocessing) against oracle/preprocess.py --
bit-exact: both sides perform the same fp32 operations in the same order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(hw, batch):
    from bayes_od_rc_amd.engine import Engine, make_config
    return Engine(make_config(hw, batch=batch, mc_samples=2))


def test_bdd_normalisation_bit_exact():
    from oracle import preprocess as pp
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(3, 128, 192, 3), dtype=np.uint8)
    eng = _engine((128, 192), 3)
    eng.upload_frames_u8(frames)
    got = eng.get_images()
    for b in range(3):
        assert np.array_equal(got[b], pp.bdd_preprocess(frames[b]))
    with pytest.raises(ValueError):
        eng.upload_frames_u8(frames[:, :100])                      # size mismatch without aspect_resize


@pytest.mark.parametrize("src_hw,net_hw", [((375, 1242), (384, 1248)),      # KITTI -> BASELINE config 4: pad rows
                                           ((94, 310), (128, 416)),         # up-scaling, pad
                                           ((200, 150), (96, 128)),         # down-scaling, pad columns
                                           ((370, 1224), (128, 416))])
def test_kitti_resize_crop_pad_bit_exact(src_hw, net_hw):
    from oracle import preprocess as pp
    from bayes_od_rc_amd import constants
    rng = np.random.default_rng(1)
    frames = rng.integers(0, 256, size=(2,) + src_hw + (3,), dtype=np.uint8)
    eng = _engine(net_hw, 2)
    eng.upload_frames_u8(frames, constants.MEANS_DICT['Kitti'], aspect_resize=True)
    got = eng.get_images()
    for b in range(2):
        ref = pp.kitti_preprocess(frames[b], net_hw, constants.MEANS_DICT['Kitti'])
        assert got[b].shape == ref.shape
        assert np.array_equal(got[b], ref), float(np.abs(got[b] - ref).max())


def test_uint8_upload_feeds_the_pipeline():
    """forward() on device-preprocessed uint8 frames == forward() on the host-normalised float frames."""
    from oracle import preprocess as pp
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    rng = np.random.default_rng(2)
    frames = rng.integers(0, 256, size=(2, 160, 160, 3), dtype=np.uint8)
    eng = Engine(make_config((160, 160), batch=2, mc_samples=3))
    eng.load_weights(synthetic.make_weights())
    eng.upload_frames_u8(frames)
    eng.forward(None, seed=4, first_image_id=9)
    a = eng.get_raw()
    eng.forward(np.stack([pp.bdd_preprocess(f) for f in frames]), seed=4, first_image_id=9)
    b = eng.get_raw()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
