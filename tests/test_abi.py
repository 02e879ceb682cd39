"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads here (no GPU) and exports
every symbol include/lgd_hip.h declares; kernel calls without a GPU fail loudly (no fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from lgd_amd import hip
    return hip.load()


def _declared():
    src = open(os.path.join(ROOT, "include", "lgd_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lgd_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared()
    assert len(names) >= 10
    raw = ctypes.CDLL(os.path.join(ROOT, "lgd_amd", "_lib", "liblgd_hip.so"))
    for n in names:
        assert hasattr(raw, n), "missing export: " + n


def test_binding_table_matches_header(lib):
    from lgd_amd import hip
    assert sorted(hip.SIGNATURES) == _declared()


def test_version_and_arch(lib):
    from lgd_amd import hip
    assert lib.lgd_abi_version() == hip.ABI_VERSION
    assert lib.lgd_arch() == b"gfx950"


def test_size_helpers_no_gpu_needed(lib):
    L, B, T, mx = 5, 2, 22, 11
    assert lib.lgd_geom_ints(L, B, T, mx) == L * B * mx * 4 + L * B + L * B * (2 * mx + 2)
    from lgd_amd import hip
    hw = hip.int_array([64, 64, 32, 32])
    # chunks of 4096 elements: 1 chunk per plane at both levels; 5 doubles per chunk + block terms
    assert lib.lgd_distill_ws_doubles(hw, 2, 2, 256) == 2 * 2 * 256 * 5 + (2 * 2 * 256 + 255) // 256


def test_invalid_arguments_are_rejected(lib):
    from lgd_amd import hip
    hw = hip.int_array([8, 8])
    assert lib.lgd_box_prep(None, None, 1, 1, 1, 64, 64, hw, 1, None, None) == -1
    assert lib.lgd_box_sum(None, hw, 1, 1, 256, 1, 1, None, None, None, None, 1, 0, None) == -1
    assert lib.lgd_box_sum(None, hw, 17, 1, 256, 1, 1, None, None, None, None, 1, 0, None) == -1  # L > LGD_MAX_LEVELS


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu(lib):
    from lgd_amd import hip, ops
    with pytest.raises(hip.LgdHipError):
        ops.BoxGeometry(torch.zeros(1, 4), [1], (64, 64), [(8, 8)])
    with pytest.raises(hip.LgdHipError):
        ops.distill_in_mse([torch.zeros(1, 4, 8, 8)], [torch.zeros(1, 4, 8, 8)], 1.0)
