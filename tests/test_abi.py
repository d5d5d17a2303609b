"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol
include/mcmcgpu.h declares, and refuses to compute without a GPU (no CPU path).
No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest

import oracle_lib
from slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mcmcgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcmcgpu_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(capi.EXPORTS)


def test_library_exports_every_declared_symbol():
    lib = capi.load_library()
    for name in declared_symbols():
        assert hasattr(lib, name), f"libmcmcgpu.so does not export {name}"


def test_struct_layouts_match_the_header():
    assert capi.STEP_RECORD.itemsize == 56
    assert capi.CHAIN_SUMMARY.itemsize == 16
    assert ctypes.sizeof(capi.Config) == 56


def test_host_seed_helpers_match_oracle():
    """The only entry points that need no device: the generator used for chain seeds."""
    assert np.array_equal(capi.seed_sequence(5489, 3), [3499211612, 581869302, 3890346734])
    assert np.array_equal(capi.seed_sequence(2024, 100), oracle_lib.seed_sequence(2024, 100))
    assert np.array_equal(capi.uniform_sequence(7, 100), oracle_lib.uniform_sequence(7, 100))


def _gpu_visible():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_visible(), reason="this check is about machines without a GPU")
def test_no_cpu_fallback():
    with pytest.raises(RuntimeError, match="mcmcgpu_create failed"):
        capi.McmcGpu(1, 10, 2, 20)


def test_bad_arguments_are_rejected_before_touching_a_device():
    lib = capi.load_library()
    h = ctypes.c_void_p()
    assert lib.mcmcgpu_create(None, ctypes.byref(h)) == -1
    cfg = capi.Config(0, 0, 10, 2, 20, 0, 1, 0, None)  # zero chains
    assert lib.mcmcgpu_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"max_chains" in lib.mcmcgpu_last_error(None)
    cfg = capi.Config(0, 1, 10, 2, 20, 100, 1, 0, None)  # 100 threads: not a power of two
    assert lib.mcmcgpu_create(ctypes.byref(cfg), ctypes.byref(h)) == -1
    assert b"threads_per_chain" in lib.mcmcgpu_last_error(None)
