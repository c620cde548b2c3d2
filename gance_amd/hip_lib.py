"""
ctypes binding of libgance_hip.so (C ABI in include/gance_hip.h).

This is the ONLY route from Python to the synthesis / audio kernels. If the shared library has not
been built (`python -c "import __graft_entry__ as g; g.build()"` or `make -C gance_amd/csrc`) or no
MI355X is visible, calls raise: there is no CPU fallback in the product path.
"""

import ctypes
import os
from pathlib import Path
from typing import Dict, List, NamedTuple, Optional, Tuple

import numpy as np
import torch  # noqa: F401  pylint: disable=unused-import
# torch is imported BEFORE the dlopen below on purpose: the PyTorch-ROCm wheel carries its own HIP /
# HSA runtime, and a process must end up with exactly one. Loading libgance_hip.so first would pull
# in /opt/rocm's copy and the two runtimes then fight over the device (observed: hipGetDeviceCount
# fails). With torch loaded first the library binds to the runtime torch already mapped.

from gance_amd.stylegan2 import spec as sg2_spec

LIBRARY_NAME = "libgance_hip.so"
# GANCE_HIP_LIBRARY: another build of the same ABI (A/B timing of kernel variants); default in-tree
LIBRARY_PATH = Path(os.environ.get("GANCE_HIP_LIBRARY") or Path(__file__).resolve().parent / LIBRARY_NAME)

GANCE_OK = 0
GANCE_FLAG_PROFILE_STEPS = 1
GANCE_FLAG_DIRECT_CONV = 2
GANCE_FLAG_FORCE_WINOGRAD = 4
GANCE_FLAG_SPLIT_UPFIR = 8
GANCE_FLAG_FORCE_FUSED_UPFIR = 16
GANCE_FLAG_PRIVATE_WORKSPACE = 32
GANCE_FLAG_WINOGRAD43 = 64

STATUS_NAMES = {
    1: "GANCE_ERR_INVALID_ARGUMENT",
    2: "GANCE_ERR_BAD_WEIGHTS",
    3: "GANCE_ERR_HIP",
    4: "GANCE_ERR_OUT_OF_MEMORY",
    5: "GANCE_ERR_NO_DEVICE",
}


class GanceHipError(RuntimeError):
    """A libgance_hip call returned a non-zero status."""

    def __init__(self, status: int, message: str) -> None:
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class EngineConfig(ctypes.Structure):
    """`gance_engine_config` of include/gance_hip.h."""

    _fields_ = [
        ("resolution", ctypes.c_int32),
        ("max_batch", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("flags", ctypes.c_int32),
    ]


class BlendConfig(ctypes.Structure):
    """`gance_blend_config` of include/gance_hip.h."""

    _fields_ = [
        ("num_frames", ctypes.c_int32),
        ("vector_length", ctypes.c_int32),
        ("num_projection_frames", ctypes.c_int32),
        ("latent_depth", ctypes.c_int32),
        ("blend_depth", ctypes.c_int32),
        ("fft_roll_enabled", ctypes.c_int32),
        ("num_networks", ctypes.c_int32),
        ("has_amplitude_range", ctypes.c_int32),
        ("alpha", ctypes.c_double),
        ("amplitude_lo", ctypes.c_double),
        ("amplitude_hi", ctypes.c_double),
        ("index_savgol_window_length", ctypes.c_int32),
        ("index_savgol_polyorder", ctypes.c_int32),
    ]


_F32P = ctypes.POINTER(ctypes.c_float)
_U8P = ctypes.POINTER(ctypes.c_uint8)

# name -> (restype, argtypes); every symbol include/gance_hip.h declares.
SIGNATURES = {
    "gance_last_error": (ctypes.c_char_p, []),
    "gance_abi_version": (ctypes.c_int, []),
    "gance_engine_create": (
        ctypes.c_int,
        [ctypes.POINTER(EngineConfig), _F32P, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)],
    ),
    "gance_engine_destroy": (None, [ctypes.c_void_p]),
    "gance_engine_set_profiling": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_char_p]),
    "gance_engine_vector_length": (ctypes.c_int32, [ctypes.c_void_p]),
    "gance_engine_num_layers": (ctypes.c_int32, [ctypes.c_void_p]),
    "gance_engine_resolution": (ctypes.c_int32, [ctypes.c_void_p]),
    "gance_engine_max_batch": (ctypes.c_int32, [ctypes.c_void_p]),
    "gance_weight_blob_floats": (ctypes.c_uint64, [ctypes.c_int32]),
    "gance_synthesize_w": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "gance_synthesize_z": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "gance_synthesize_w_host": (ctypes.c_int, [ctypes.c_void_p, _F32P, ctypes.c_int32, _U8P, _F32P]),
    "gance_synthesize_z_host": (
        ctypes.c_int,
        [ctypes.c_void_p, _F32P, ctypes.c_int32, ctypes.c_float, _U8P, _F32P],
    ),
    "gance_engine_randomize_noise": (
        ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
    ),
    "gance_engine_restore_noise": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "gance_engine_debug_read_noise": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, _F32P, ctypes.c_uint64]),
    "gance_engine_step_count": (ctypes.c_int32, [ctypes.c_void_p]),
    "gance_engine_step_info": (
        ctypes.c_int,
        [
            ctypes.c_void_p,
            ctypes.c_int32,
            ctypes.c_char_p,
            ctypes.POINTER(ctypes.c_float),
            ctypes.POINTER(ctypes.c_double),
            ctypes.POINTER(ctypes.c_double),
        ],
    ),
    "gance_engine_debug_stop_after": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32]),
    "gance_engine_debug_read_activation": (
        ctypes.c_int,
        [
            ctypes.c_void_p,
            ctypes.c_int32,
            _F32P,
            ctypes.c_uint64,
            ctypes.POINTER(ctypes.c_int32),
            ctypes.POINTER(ctypes.c_int32),
        ],
    ),
    "gance_blend_create": (
        ctypes.c_int,
        [ctypes.POINTER(BlendConfig), ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)],
    ),
    "gance_blend_destroy": (None, [ctypes.c_void_p]),
    "gance_blend_run": (
        ctypes.c_int,
        [
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_uint64,
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_void_p,
            ctypes.c_int32,
            ctypes.c_void_p,
        ],
    ),
    "gance_blend_read_stage": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_uint64]),
    "gance_vec_savgol_f64": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "gance_vec_fourier_resample_f64": (
        ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    ),
    "gance_vec_spectrogram_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gance_vec_spectrogram2_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gance_vec_minmax_scale_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p]),
    "gance_vec_remap_f64": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p],
    ),
    "gance_vec_rms_rolling_average": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p],
    ),
    "gance_vec_quantize_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]),
    "gance_debug_fourier_resample_matrix": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]),
    "gance_resize_bicubic_u8": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p],
    ),
    "gance_phash_crops_u8": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), ctypes.c_int32,
         ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p],
    ),
    "gance_overlay_boxes_u8": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
         ctypes.POINTER(ctypes.c_int32), ctypes.c_int32, ctypes.c_void_p],
    ),
    "gance_resample_audio_f32": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p],
    ),
    "gance_resample_audio_f64": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p],
    ),
    "gance_debug_resample_filter": (ctypes.c_int, [ctypes.POINTER(ctypes.c_double), ctypes.c_uint64]),
    "gance_vec_rms_rolling_max": (
        ctypes.c_int,
        [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p],
    ),
    "gance_gaussian_noise": (
        ctypes.c_int,
        [
            ctypes.c_void_p,
            ctypes.c_int32,
            ctypes.c_int32,
            ctypes.c_double,
            ctypes.c_double,
            ctypes.POINTER(ctypes.c_double),
            ctypes.c_void_p,
            ctypes.c_void_p,
        ],
    ),
}

_LIB: Optional[ctypes.CDLL] = None


def load_library() -> ctypes.CDLL:
    """
    dlopen the in-tree library and declare every prototype.
    :raises RuntimeError: if the library has not been built. Never falls back to another backend.
    """
    global _LIB  # pylint: disable=global-statement
    if _LIB is not None:
        return _LIB
    if not LIBRARY_PATH.exists():
        raise RuntimeError(
            f"{LIBRARY_PATH} is missing: build the HIP extension first "
            "(`make -C gance_amd/csrc` or `__graft_entry__.build()`). "
            "gance_amd has no CPU fallback."
        )
    lib = ctypes.CDLL(str(LIBRARY_PATH))
    for name, (restype, argtypes) in SIGNATURES.items():
        function = getattr(lib, name)  # AttributeError if the .so does not export it
        function.restype = restype
        function.argtypes = argtypes
    _LIB = lib
    return lib


def _check(lib: ctypes.CDLL, status: int) -> None:
    if status != GANCE_OK:
        raise GanceHipError(status, lib.gance_last_error().decode("utf-8", "replace"))


def _f32(array: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(array, dtype=np.float32)


class StepInfo(NamedTuple):
    """One kernel launch of the last synthesize call (GANCE_FLAG_PROFILE_STEPS)."""

    name: str
    ms: float
    flops: float
    bytes: float


class Engine:
    """
    One StyleGAN2 generator resident in HBM. Thin owner of a `gance_engine*`.
    Replaces the TF session + unpickled `Network` the reference keeps per worker process
    (gance/network_interface/network_functions.py:93-111).
    """

    def __init__(
        self,
        variables: Dict[str, np.ndarray],
        resolution: int,
        max_batch: int = 1,
        device: int = 0,
        profile: bool = False,
        conv_form: str = "auto",
        up_form: str = "auto",
        private_workspace: bool = False,
    ) -> None:
        """
        :param private_workspace: give this engine its own activation scratch instead of the one every engine of
        the same (device, resolution, max_batch) shares (only needed to overlap calls of different engines on
        different streams; costs ~0.8 GB per frame of max_batch at 1024^2).
        :param up_form: Conv0_up layers with an input >= 64 wide: "auto" = one fused kernel (transposed conv +
        FIR + noise + bias + leaky ReLU) when the launch fills the chip, else two passes; "split" = always two
        passes; "fused" = the fused kernel whatever the batch.
        :param conv_form: "auto" = every Conv1 from 32x32 up in Winograd F(4x4,3x3) form when its launch has a tile per
        CU, else F(2x2,3x3) when that launch has a block per CU, else the direct form; "direct" = never Winograd;
        "winograd" = the F(2x2,3x3) kernels on every layer that supports them, whatever the batch; "winograd43" = the
        F(4x4,3x3) kernel on every Conv1 from 32x32 up (F(2x2,3x3) on what is left), whatever the batch.
        """
        self._lib = load_library()
        self._handle = ctypes.c_void_p()
        spec = sg2_spec.make_spec(resolution)
        blob = sg2_spec.pack_variables(variables, spec)
        form_flags = {
            "auto": 0, "direct": GANCE_FLAG_DIRECT_CONV, "winograd": GANCE_FLAG_FORCE_WINOGRAD,
            "winograd43": GANCE_FLAG_FORCE_WINOGRAD | GANCE_FLAG_WINOGRAD43,
        }[conv_form]
        form_flags |= {"auto": 0, "split": GANCE_FLAG_SPLIT_UPFIR, "fused": GANCE_FLAG_FORCE_FUSED_UPFIR}[up_form]
        if private_workspace:
            form_flags |= GANCE_FLAG_PRIVATE_WORKSPACE
        config = EngineConfig(resolution, max_batch, device, (GANCE_FLAG_PROFILE_STEPS if profile else 0) | form_flags)
        _check(
            self._lib,
            self._lib.gance_engine_create(
                ctypes.byref(config),
                blob.ctypes.data_as(_F32P),
                ctypes.c_uint64(blob.size),
                ctypes.byref(self._handle),
            ),
        )
        self.resolution = resolution
        self.max_batch = max_batch
        self.device = device
        self.noise_randomized = False
        self.vector_length = int(self._lib.gance_engine_vector_length(self._handle))
        self.num_layers = int(self._lib.gance_engine_num_layers(self._handle))

    def close(self) -> None:
        """Free the engine's HBM. Idempotent."""
        if self._handle:
            if getattr(self, "_op_handle", None) is not None:
                from gance_amd import torch_ops  # pylint: disable=import-outside-toplevel

                torch_ops.unregister(self._op_handle)
                self._op_handle = None
            self._lib.gance_engine_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    @property
    def op_handle(self) -> int:
        """Integer handle of this engine for the `torch.ops.gance.*` custom ops (gance_amd/torch_ops.py)."""
        self._require_open()
        if getattr(self, "_op_handle", None) is None:
            from gance_amd import torch_ops  # pylint: disable=import-outside-toplevel

            self._op_handle = torch_ops.register_engine(self)
        return self._op_handle

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    def _require_open(self) -> None:
        if not self._handle:
            raise ValueError("Engine has been closed")

    # ---- host-buffer calls (numpy in, numpy out) ----

    def synthesize_w(self, dlatents: np.ndarray, want_float: bool = False):
        """dlatents [B, W, 512] -> uint8 [B, R, R, 3] (and float32 [B, 3, R, R] if want_float)."""
        self._require_open()
        dl = _f32(dlatents)
        if dl.ndim != 3 or dl.shape[1] != self.num_layers or dl.shape[2] != self.vector_length:
            raise ValueError(f"dlatents must be [B, {self.num_layers}, {self.vector_length}], got {dl.shape}")
        batch = dl.shape[0]
        out = np.empty((batch, self.resolution, self.resolution, 3), dtype=np.uint8)
        fout = np.empty((batch, 3, self.resolution, self.resolution), dtype=np.float32) if want_float else None
        _check(
            self._lib,
            self._lib.gance_synthesize_w_host(
                self._handle,
                dl.ctypes.data_as(_F32P),
                batch,
                out.ctypes.data_as(_U8P),
                fout.ctypes.data_as(_F32P) if fout is not None else None,
            ),
        )
        return (out, fout) if want_float else out

    def synthesize_z(self, z: np.ndarray, truncation_psi: float = 1.2, want_float: bool = False):
        """z [B, 512] -> uint8 [B, R, R, 3] (and float32 [B, 3, R, R] if want_float)."""
        self._require_open()
        zz = _f32(z)
        if zz.ndim != 2 or zz.shape[1] != self.vector_length:
            raise ValueError(f"z must be [B, {self.vector_length}], got {zz.shape}")
        batch = zz.shape[0]
        out = np.empty((batch, self.resolution, self.resolution, 3), dtype=np.uint8)
        fout = np.empty((batch, 3, self.resolution, self.resolution), dtype=np.float32) if want_float else None
        _check(
            self._lib,
            self._lib.gance_synthesize_z_host(
                self._handle,
                zz.ctypes.data_as(_F32P),
                batch,
                ctypes.c_float(truncation_psi),
                out.ctypes.data_as(_U8P),
                fout.ctypes.data_as(_F32P) if fout is not None else None,
            ),
        )
        return (out, fout) if want_float else out

    # ---- device-pointer calls (torch tensors or raw pointers; asynchronous on `stream`) ----

    def synthesize_w_device(self, d_dlatents: int, batch: int, d_out_u8: int, d_out_f32: int = 0, stream: int = 0) -> None:
        """Raw device pointers (ints). Asynchronous on `stream`."""
        self._require_open()
        _check(
            self._lib,
            self._lib.gance_synthesize_w(self._handle, d_dlatents, batch, d_out_u8 or None, d_out_f32 or None, stream or None),
        )

    def synthesize_z_device(
        self, d_z: int, batch: int, truncation_psi: float, d_out_u8: int, d_out_f32: int = 0, stream: int = 0
    ) -> None:
        """Raw device pointers (ints). Asynchronous on `stream`."""
        self._require_open()
        _check(
            self._lib,
            self._lib.gance_synthesize_z(
                self._handle, d_z, batch, ctypes.c_float(truncation_psi), d_out_u8 or None, d_out_f32 or None, stream or None
            ),
        )

    # ---- randomize_noise (the reference's vector path draws fresh noise per call) ----

    def randomize_noise(
        self, seed: Optional[int] = None, count: Optional[int] = None, first_sample: int = 0, d_sample_ids: int = 0, stream: int = 0
    ) -> None:
        """
        Draw fresh standard-normal noise planes, one per layer and PER SAMPLE (upstream: tf.random_normal([N, 1, H, W])),
        for `count` samples (default: max_batch), asynchronously on `stream`; the following calls read them until
        `restore_noise`. Sample b reads the plane of id `first_sample + b` (or of the int64 id at `d_sample_ids[b]`, a raw
        device pointer); a plane is a function of (seed, layer, id) only. `seed` None: a new seed from the OS, as
        `tf.random_normal` would give. Layers whose noise strength is zero (all of them at random init) are skipped.
        """
        self._require_open()
        if seed is None:
            seed = int.from_bytes(os.urandom(8), "little")
        _check(
            self._lib,
            self._lib.gance_engine_randomize_noise(
                self._handle, ctypes.c_uint64(seed & (2**64 - 1)), int(count or 0), ctypes.c_uint64(int(first_sample)), d_sample_ids or None,
                stream or None,
            ),
        )
        self.noise_randomized = True

    def restore_noise(self, stream: int = 0) -> None:
        """Go back to the stored noise buffers (no-op if they are in use)."""
        self._require_open()
        if getattr(self, "noise_randomized", False):
            _check(self._lib, self._lib.gance_engine_restore_noise(self._handle, stream or None))
            self.noise_randomized = False

    def debug_noise(self, conv_layer: int, sample: int = 0) -> np.ndarray:
        """The noise plane sample `sample` of conv layer `conv_layer` currently reads, [res, res] float32 (synchronises)."""
        self._require_open()
        side = 2 ** sg2_spec.make_spec(self.resolution).convs[conv_layer].res_log2
        out = np.empty((side, side), dtype=np.float32)
        _check(
            self._lib,
            self._lib.gance_engine_debug_read_noise(self._handle, conv_layer, sample, out.ctypes.data_as(_F32P), ctypes.c_uint64(out.size)),
        )
        return out

    # ---- profiling / debugging ----

    def set_profiling(self, enabled: bool, only_step: Optional[str] = None) -> None:
        """Per-launch HIP events on / off; `only_step` restricts them to launches whose name contains it."""
        self._require_open()
        _check(
            self._lib,
            self._lib.gance_engine_set_profiling(
                self._handle, GANCE_FLAG_PROFILE_STEPS if enabled else 0, only_step.encode() if only_step else None
            ),
        )

    def steps(self) -> List[StepInfo]:
        """Per-launch timings of the last call (engine created with profile=True)."""
        self._require_open()
        count = int(self._lib.gance_engine_step_count(self._handle))
        result = []
        name = ctypes.create_string_buffer(64)
        ms = ctypes.c_float()
        flops = ctypes.c_double()
        nbytes = ctypes.c_double()
        for index in range(count):
            _check(
                self._lib,
                self._lib.gance_engine_step_info(
                    self._handle, index, name, ctypes.byref(ms), ctypes.byref(flops), ctypes.byref(nbytes)
                ),
            )
            result.append(StepInfo(name.value.decode(), float(ms.value), float(flops.value), float(nbytes.value)))
        return result

    def debug_activation_after(self, dlatents: np.ndarray, num_conv_layers: int) -> np.ndarray:
        """Run only the first `num_conv_layers` conv layers and return x [B, C, res, res]."""
        self._require_open()
        dl = _f32(dlatents)
        batch = dl.shape[0]
        _check(self._lib, self._lib.gance_engine_debug_stop_after(self._handle, num_conv_layers))
        try:
            _check(
                self._lib,
                self._lib.gance_synthesize_w_host(self._handle, dl.ctypes.data_as(_F32P), batch, None, None),
            )
            capacity = batch * 512 * self.resolution * self.resolution
            channels = ctypes.c_int32()
            side = ctypes.c_int32()
            spec = sg2_spec.make_spec(self.resolution)
            conv = spec.convs[num_conv_layers - 1]
            count = batch * conv.cout * (2 ** conv.res_log2) ** 2
            out = np.empty(count, dtype=np.float32)
            _check(
                self._lib,
                self._lib.gance_engine_debug_read_activation(
                    self._handle, batch, out.ctypes.data_as(_F32P), ctypes.c_uint64(min(capacity, count)),
                    ctypes.byref(channels), ctypes.byref(side),
                ),
            )
            return out.reshape(batch, channels.value, side.value, side.value)
        finally:
            self._lib.gance_engine_debug_stop_after(self._handle, 0)


# stage ids of `enum gance_blend_stage`: name -> (id, dtype, per-frame width or None for [N])
BLEND_STAGES = {
    "db": (0, np.float64, "bins"),
    "scaled": (1, np.float64, "L"),
    "smoothed_time": (2, np.float64, "L"),
    "smoothed": (3, np.float64, "L"),
    "rolled": (4, np.float64, "L"),
    "final": (5, np.float64, "L"),
    "blend_row": (6, np.float64, "L"),
    "raw_rms": (7, np.float32, None),
    "roll_values": (8, np.int32, None),
    "roll_cumulative": (9, np.int32, None),
    "network_indices": (10, np.int32, None),
    "rolling_average": (11, np.float64, None),
    "rolling_smoothed": (12, np.float64, None),
    "index_smoothed": (13, np.float64, None),
}


class Blend:
    """
    Audio -> blended latents on the GPU. Thin owner of a `gance_blend*` (operator tables and
    workspace for one (N, F, depth, alpha, ...) configuration).
    """

    def __init__(
        self,
        num_frames: int,
        num_projection_frames: int,
        alpha: float,
        fft_roll_enabled: bool,
        fft_amplitude_range,
        blend_depth: int,
        num_networks: int,
        vector_length: int = 512,
        latent_depth: int = 18,
        device: int = 0,
        index_savgol: Tuple[int, int] = (3, 2),
    ) -> None:
        """
        :param index_savgol: (window_length, polyorder) of the savgol_filter in front of the
        network-index quantisation: (3, 2) for projection-file-blend, (7, 3) for noise-blend.
        """
        self._lib = load_library()
        self._handle = ctypes.c_void_p()
        has_range = fft_amplitude_range is not None
        lo, hi = (fft_amplitude_range if has_range else (0.0, 0.0))
        self.config = BlendConfig(
            num_frames, vector_length, num_projection_frames, latent_depth, blend_depth,
            int(bool(fft_roll_enabled)), num_networks, int(has_range), float(alpha), float(lo), float(hi),
            int(index_savgol[0]), int(index_savgol[1]),
        )
        status = self._lib.gance_blend_create(ctypes.byref(self.config), device, ctypes.byref(self._handle))
        if status == 1 and (
            b"Cannot duplicate" in self._lib.gance_last_error() or b"num_frames must be >= 7" in self._lib.gance_last_error()
        ):
            # the reference raises ValueError in both cases (vector_sources_common.py:318-331;
            # scipy.signal.savgol_filter: window_length must not exceed the number of frames)
            raise ValueError(self._lib.gance_last_error().decode())
        _check(self._lib, status)
        self.device = device

    def close(self) -> None:
        """Free the tables and workspace. Idempotent."""
        if self._handle:
            self._lib.gance_blend_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self) -> None:
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass

    def run_device(
        self, d_audio: int, num_samples: int, d_latent_row0: int, d_dlatents: int = 0, d_network_indices: int = 0,
        debug_stages: bool = False, stream: int = 0,
    ) -> None:
        """Raw device pointers (ints). Asynchronous on `stream`."""
        if not self._handle:
            raise ValueError("Blend has been closed")
        _check(
            self._lib,
            self._lib.gance_blend_run(
                self._handle, d_audio, ctypes.c_uint64(num_samples), d_latent_row0, d_dlatents or None,
                d_network_indices or None, int(debug_stages), stream or None,
            ),
        )

    def check_finite(self) -> None:
        """
        The reference fails on audio with a silent 510-sample window: log10(0) = -inf reaches
        sklearn's minmax_scale, which raises ValueError (apply_spectrogram.py:43,81). Same here:
        inspect the global extrema of the last run (one 24-byte read-back).
        :raises ValueError: if the spectrogram holds a non-finite value.
        """
        extrema = np.empty(3, dtype=np.float64)
        _check(self._lib, self._lib.gance_blend_read_stage(self._handle, 14, extrema.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(24)))
        if not np.all(np.isfinite(extrema)) or extrema[0] == 0.0:
            raise ValueError("Input contains infinity or a value too large for dtype('float64').")

    def read_stage(self, name: str) -> np.ndarray:
        """Copy one stage of the last run to the host (synchronises)."""
        stage_id, dtype, width = BLEND_STAGES[name]
        frames = self.config.num_frames
        if width == "L":
            shape = (frames, self.config.vector_length)
        elif width == "bins":
            shape = (frames, (self.config.vector_length - 2) // 2)
        else:
            shape = (frames,)
        out = np.empty(shape, dtype=dtype)
        _check(
            self._lib,
            self._lib.gance_blend_read_stage(self._handle, stage_id, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(out.nbytes)),
        )
        return out


def resize_bicubic_u8_device(d_in: int, batch: int, src_side: int, d_out: int, dst_side: int, stream: int = 0) -> None:
    """Bicubic (a = -0.75) resize of uint8 NHWC frames in HBM; raw device pointers; asynchronous."""
    lib = load_library()
    _check(lib, lib.gance_resize_bicubic_u8(d_in, batch, src_side, d_out, dst_side, stream or None))


def gaussian_noise_device(  # pylint: disable=too-many-arguments
    d_randn: int,
    num_vectors: int,
    vector_length: int,
    sigma_across: float,
    sigma_within: float,
    feature_range: Optional[Tuple[float, float]],
    d_out: int,
    stream: int = 0,
) -> None:
    """
    Gaussian-filtered ("wrap"), RMS-normalised and optionally min-max scaled noise field from
    float32 standard-normal draws already in HBM (raw device pointers, [N][L]); returns once
    `stream` has drained.
    :raises ValueError: for a feature range sklearn's minmax_scale rejects.
    """
    lib = load_library()
    bounds = None
    if feature_range is not None:
        if not feature_range[0] < feature_range[1]:
            raise ValueError(f"Minimum of desired feature range must be smaller than maximum. Got {feature_range}.")
        bounds = (ctypes.c_double * 2)(float(feature_range[0]), float(feature_range[1]))
    _check(
        lib,
        lib.gance_gaussian_noise(
            d_randn, num_vectors, vector_length, float(sigma_across), float(sigma_within), bounds, d_out, stream or None
        ),
    )


def phash_crops_device(d_frames: int, num_frames: int, side: int, crops: np.ndarray, stream: int = 0) -> np.ndarray:
    """
    Perceptual hashes of crops of uint8 NHWC frames in HBM. `crops` is (n, 5) int32: frame index,
    x, y, width, height. Returns (n,) uint64 (bit 63 = DCT coefficient (0, 0)).
    """
    lib = load_library()
    crops = np.ascontiguousarray(crops, dtype=np.int32).reshape(-1, 5)
    hashes = np.zeros((crops.shape[0],), dtype=np.uint64)
    _check(
        lib,
        lib.gance_phash_crops_u8(
            d_frames, num_frames, side, crops.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), crops.shape[0],
            hashes.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), stream or None,
        ),
    )
    return hashes


def overlay_boxes_device(  # pylint: disable=too-many-arguments
    d_foreground: int, d_background: int, d_out: int, num_frames: int, side: int, boxes: np.ndarray, stream: int = 0
) -> None:
    """Foreground regions around `boxes` ((n, 5) int32: frame, x, y, width, height) written over the background."""
    lib = load_library()
    boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 5)
    _check(
        lib,
        lib.gance_overlay_boxes_u8(
            d_foreground, d_background, d_out, num_frames, side, boxes.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
            boxes.shape[0], stream or None,
        ),
    )


def resample_audio_device(
    d_in: int, num_in: int, sr_orig: float, sr_new: float, d_out: int, num_out: int, stream: int = 0, double_precision: bool = False
) -> None:
    """
    resampy's kaiser_best resampling of a mono float32 (or float64) signal in HBM (raw device pointers);
    num_out = int(num_in * sr_new / sr_orig). Returns once `stream` has drained.
    :raises ValueError: for the arguments resampy rejects (non-positive rates, an output shorter than one sample).
    """
    lib = load_library()
    entry = lib.gance_resample_audio_f64 if double_precision else lib.gance_resample_audio_f32
    _value_error_on_invalid_argument(lib, entry(d_in, num_in, float(sr_orig), float(sr_new), d_out, num_out, stream or None))


def resample_filter_table() -> np.ndarray:
    """The 32 769-entry kaiser_best half window the resampler interpolates (host only, no GPU needed)."""
    lib = load_library()
    table = np.empty(32769, dtype=np.float64)
    _check(lib, lib.gance_debug_resample_filter(table.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.c_uint64(table.size)))
    return table


# ---- stand-alone stages of the audio -> latent chain (gance_vec_*): numpy in, numpy out ------------
# The arrays involved are a few MB; every call uploads, runs one stage on the GPU and downloads. The
# fused pipeline (`Blend`) is what the hot path uses; these exist so that a caller of the reference's
# stand-alone functions finds each of them (gance_amd/apply_spectrogram.py, gance_amd/vector_sources/*).


def _value_error_on_invalid_argument(lib: ctypes.CDLL, status: int) -> None:
    """The argument checks restate scipy / sklearn ValueErrors; everything else stays a GanceHipError."""
    if status == 1:
        raise ValueError(lib.gance_last_error().decode("utf-8", "replace"))
    _check(lib, status)


def _cuda(device: int) -> "torch.device":
    if not torch.cuda.is_available():
        raise RuntimeError("gance_amd needs an MI355X: there is no CPU fallback for the audio stages")
    return torch.device("cuda", device)


def vec_savgol(data: np.ndarray, axis: int, window_length: int, polyorder: int, device: int = 0) -> np.ndarray:
    """scipy.signal.savgol_filter (mode "interp") along `axis` of a 2-D float64 array, on the GPU."""
    lib = load_library()
    host = np.ascontiguousarray(data, dtype=np.float64)
    d_in = torch.from_numpy(host).to(_cuda(device))
    d_out = torch.empty_like(d_in)
    stream = torch.cuda.current_stream(d_in.device).cuda_stream
    _value_error_on_invalid_argument(
        lib,
        lib.gance_vec_savgol_f64(d_in.data_ptr(), host.shape[0], host.shape[1], axis, int(window_length), int(polyorder), d_out.data_ptr(), stream or None),
    )
    return d_out.cpu().numpy()


def vec_fourier_resample(data: np.ndarray, out_length: int, device: int = 0) -> np.ndarray:
    """scipy.signal.resample of every row of a 2-D float64 array to `out_length` points, on the GPU."""
    lib = load_library()
    host = np.ascontiguousarray(data, dtype=np.float64)
    d_in = torch.from_numpy(host).to(_cuda(device))
    d_out = torch.empty((host.shape[0], int(out_length)), dtype=torch.float64, device=d_in.device)
    stream = torch.cuda.current_stream(d_in.device).cuda_stream
    _value_error_on_invalid_argument(
        lib, lib.gance_vec_fourier_resample_f64(d_in.data_ptr(), host.shape[0], host.shape[1], int(out_length), d_out.data_ptr(), stream or None)
    )
    return d_out.cpu().numpy()


def vec_spectrogram(audio: np.ndarray, num_frequency_bins: int, device: int = 0, truncate: bool = True) -> np.ndarray:
    """compute_spectrogram: float32 mono samples -> float64 [(bins - 2) // 2][frames] dB magnitudes ([bins - 2][frames] with `truncate=False`)."""
    lib = load_library()
    host = np.ascontiguousarray(audio, dtype=np.float32)
    window = num_frequency_bins - 2
    if host.shape[0] < window:
        raise ValueError("fewer samples than one window")
    frames = (host.shape[0] - window) // num_frequency_bins + 1
    d_audio = torch.from_numpy(host).to(_cuda(device))
    d_out = torch.empty((window // 2 if truncate else window, frames), dtype=torch.float64, device=d_audio.device)
    stream = torch.cuda.current_stream(d_audio.device).cuda_stream
    _value_error_on_invalid_argument(
        lib,
        lib.gance_vec_spectrogram2_f64(
            d_audio.data_ptr(), ctypes.c_uint64(host.shape[0]), int(num_frequency_bins), 1 if truncate else 0, d_out.data_ptr(), stream or None
        ),
    )
    return d_out.cpu().numpy()


def vec_minmax_scale(data: np.ndarray, feature_range: Tuple[float, float], device: int = 0) -> np.ndarray:
    """sklearn.preprocessing.minmax_scale of a whole array (any shape), on the GPU."""
    lib = load_library()
    host = np.ascontiguousarray(data, dtype=np.float64)
    d_data = torch.from_numpy(host).to(_cuda(device))
    stream = torch.cuda.current_stream(d_data.device).cuda_stream
    _value_error_on_invalid_argument(
        lib, lib.gance_vec_minmax_scale_f64(d_data.data_ptr(), ctypes.c_uint64(host.size), float(feature_range[0]), float(feature_range[1]), stream or None)
    )
    return d_data.cpu().numpy()


def vec_remap(data: np.ndarray, input_range: Tuple[float, float], output_range: Tuple[float, float], device: int = 0) -> np.ndarray:
    """scipy.interpolate.interp1d(input_range, output_range) applied to every value, on the GPU."""
    lib = load_library()
    host = np.ascontiguousarray(data, dtype=np.float64)
    d_in = torch.from_numpy(host).to(_cuda(device))
    d_out = torch.empty_like(d_in)
    stream = torch.cuda.current_stream(d_in.device).cuda_stream
    _value_error_on_invalid_argument(
        lib,
        lib.gance_vec_remap_f64(
            d_in.data_ptr(), ctypes.c_uint64(host.size), float(input_range[0]), float(input_range[1]), float(output_range[0]),
            float(output_range[1]), d_out.data_ptr(), stream or None,
        ),
    )
    return d_out.cpu().numpy()


def vec_rms_rolling_average(
    audio: np.ndarray, vector_length: int, rolling_window: int, savgol_window_length: int, savgol_polyorder: int, device: int = 0
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(raw RMS float32, rolling mean float64, Savitzky-Golay smoothed float64), one value per hop of 512 samples."""
    lib = load_library()
    host = np.ascontiguousarray(audio, dtype=np.float32)
    if host.shape[0] < vector_length:
        raise ValueError("fewer samples than one frame")
    count = 1 + (host.shape[0] - vector_length) // 512
    d_audio = torch.from_numpy(host).to(_cuda(device))
    d_rms = torch.empty((count,), dtype=torch.float32, device=d_audio.device)
    d_rolling = torch.empty((count,), dtype=torch.float64, device=d_audio.device)
    d_smoothed = torch.empty((count,), dtype=torch.float64, device=d_audio.device)
    stream = torch.cuda.current_stream(d_audio.device).cuda_stream
    _value_error_on_invalid_argument(
        lib,
        lib.gance_vec_rms_rolling_average(
            d_audio.data_ptr(), ctypes.c_uint64(host.shape[0]), int(vector_length), int(rolling_window), int(savgol_window_length),
            int(savgol_polyorder), d_rms.data_ptr(), d_rolling.data_ptr(), d_smoothed.data_ptr(), count, stream or None,
        ),
    )
    return d_rms.cpu().numpy(), d_rolling.cpu().numpy(), d_smoothed.cpu().numpy()


def vec_rms_rolling_max(audio: np.ndarray, vector_length: int, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(raw RMS float32, its rolling maximum over len // 80 values), one value per hop of 512 samples."""
    lib = load_library()
    host = np.ascontiguousarray(audio, dtype=np.float32)
    if host.shape[0] < vector_length:
        raise ValueError("fewer samples than one frame")
    count = 1 + (host.shape[0] - vector_length) // 512
    d_audio = torch.from_numpy(host).to(_cuda(device))
    d_rms = torch.empty((count,), dtype=torch.float32, device=d_audio.device)
    d_out = torch.empty((count,), dtype=torch.float32, device=d_audio.device)
    stream = torch.cuda.current_stream(d_audio.device).cuda_stream
    _value_error_on_invalid_argument(
        lib,
        lib.gance_vec_rms_rolling_max(
            d_audio.data_ptr(), ctypes.c_uint64(host.shape[0]), int(vector_length), d_rms.data_ptr(), d_out.data_ptr(), count, stream or None
        ),
    )
    return d_rms.cpu().numpy(), d_out.cpu().numpy()


def vec_quantize(data: np.ndarray, num_indices: int, device: int = 0) -> np.ndarray:
    """Remap [min, max] of a series onto [0, num_indices - 1] and round half to even -> int64."""
    lib = load_library()
    host = np.ascontiguousarray(data, dtype=np.float64)
    d_in = torch.from_numpy(host).to(_cuda(device))
    d_out = torch.empty((host.size,), dtype=torch.int64, device=d_in.device)
    stream = torch.cuda.current_stream(d_in.device).cuda_stream
    _value_error_on_invalid_argument(lib, lib.gance_vec_quantize_f64(d_in.data_ptr(), host.size, int(num_indices), d_out.data_ptr(), stream or None))
    return d_out.cpu().numpy()
