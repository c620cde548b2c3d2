"""
CPU tests of the drop-in boundary: libgance_hip.so builds, loads, exports every symbol that
include/gance_hip.h declares, and refuses to run without a GPU (no silent CPU fallback).
"""

import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from gance_amd import hip_lib
from gance_amd.stylegan2 import spec as sg2_spec

REPO_ROOT = Path(__file__).resolve().parent.parent
HEADER = REPO_ROOT / "include" / "gance_hip.h"


@pytest.fixture(scope="module")
def library() -> ctypes.CDLL:
    """The in-tree shared library, built on demand (hipcc cross-compiles without a GPU)."""
    if not hip_lib.LIBRARY_PATH.exists():
        import __graft_entry__  # pylint: disable=import-outside-toplevel

        __graft_entry__.build()
    return hip_lib.load_library()


def declared_functions() -> list:
    """Names of every function prototype in the public header."""
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(gance_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary() -> None:
    names = declared_functions()
    for required in ("gance_engine_create", "gance_engine_destroy", "gance_synthesize_w", "gance_synthesize_z"):
        assert required in names


def test_library_exports_every_declared_symbol(library: ctypes.CDLL) -> None:
    for name in declared_functions():
        assert hasattr(library, name), f"{name} declared in gance_hip.h but not exported"
        assert name in hip_lib.SIGNATURES, f"{name} has no ctypes prototype in hip_lib.SIGNATURES"
    assert library.gance_abi_version() == 6


def test_blob_size_agrees_between_python_and_c(library: ctypes.CDLL) -> None:
    for resolution in (8, 32, 256, 1024):
        spec = sg2_spec.make_spec(resolution)
        assert library.gance_weight_blob_floats(resolution) == sg2_spec.blob_size(spec)
    assert library.gance_weight_blob_floats(1000) == 0
    assert library.gance_weight_blob_floats(2048) == 0


def test_bad_arguments_are_rejected_before_touching_a_device(library: ctypes.CDLL) -> None:
    handle = ctypes.c_void_p()
    config = hip_lib.EngineConfig(32, 1, 0, 0)
    blob = np.zeros(10, dtype=np.float32)
    status = library.gance_engine_create(
        ctypes.byref(config), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 10, ctypes.byref(handle)
    )
    assert status == 2  # GANCE_ERR_BAD_WEIGHTS
    assert b"expected" in library.gance_last_error()
    config = hip_lib.EngineConfig(33, 1, 0, 0)
    status = library.gance_engine_create(
        ctypes.byref(config), blob.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), 10, ctypes.byref(handle)
    )
    assert status == 1  # GANCE_ERR_INVALID_ARGUMENT
    library.gance_engine_destroy(None)  # NULL is a no-op


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_means_loud_failure_not_cpu_fallback(library: ctypes.CDLL) -> None:
    variables = sg2_spec.make_random_variables(8, seed=0)
    with pytest.raises(hip_lib.GanceHipError) as error:
        hip_lib.Engine(variables, 8)
    assert error.value.status == 5  # GANCE_ERR_NO_DEVICE
