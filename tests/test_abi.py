"""CPU checks of the C-ABI boundary: the library builds, loads, and exports exactly the symbols
include/bayesod.h declares; without a GPU every compute entry point fails loudly (no fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "bayes-od-rc_amd", "lib", "libbayesod_hip.so")):
        g.build()
    from bayes_od_rc_amd import _lib
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "bayesod.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bod_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_and_library_agree(lib):
    from bayes_od_rc_amd import _lib
    declared = _header_symbols()
    assert len(declared) >= 30
    assert sorted(_lib.SIGNATURES) == declared, set(declared) ^ set(_lib.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.bod_version().decode().startswith("bayesod-hip")


def test_cdef_header_for_cffi_is_current_and_plain_c():
    """INTEGRATION.md's cffi route: include/bayesod_cdef.h must be the header minus comments / preprocessor lines (what
    ffi.cdef accepts), declare exactly the header's symbols, and compile as plain C given <stdint.h>."""
    import subprocess
    from bayes_od_rc_amd import build
    path = os.path.join(ROOT, "include", "bayesod_cdef.h")
    text = open(path).read()
    assert text == build.cdef_text(), "include/bayesod_cdef.h is stale: run bayes_od_rc_amd.build.write_cdef()"
    body = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "#" not in body and "extern" not in body and body.count("{") == body.count("}")
    assert sorted(set(re.findall(r"\b(bod_[a-z0-9_]+)\s*\(", body))) == _header_symbols()
    subprocess.check_call(["gcc", "-fsyntax-only", "-include", "stdint.h", "-x", "c", path])


def test_struct_layout_matches_header():
    from bayes_od_rc_amd._lib import BodConfig, BodSizes
    assert ctypes.sizeof(BodConfig) == 23 * 4 + 8 * 4
    assert ctypes.sizeof(BodSizes) == 2 * 4 + 16 * 4 + 2 * 4 + 8


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayes_od_rc_amd.engine import Engine, make_config, stage_conv
    import numpy as np
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        Engine(make_config((128, 128)))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        stage_conv(np.zeros((1, 4, 4, 64), np.float32), np.zeros((1, 1, 64, 64), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "bayes-od-rc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "/root/reference" not in src, f


def test_config_translation_errors():
    from bayes_od_rc_amd.engine import make_config
    with pytest.raises(ValueError):
        make_config((128, 128), bayes_od_config={"ranking_method": "bogus", "dirichlet_prior": {"type": "None"},
                                                 "gaussian_prior": {"type": "None"}})
    with pytest.raises(ValueError):
        make_config((128, 128), dataset_name="kitti")
    cfg = make_config((384, 1248), dataset_name="kitti", orig_size=(375, 1242))
    assert abs(cfg.kitti_scale_h - 375 / 384) < 1e-6 and abs(cfg.kitti_scale_w - 1242 / 1248) < 1e-6
