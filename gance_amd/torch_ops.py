"""
The C ABI of libgance_hip.so surfaced as PyTorch-ROCm custom ops (namespace `gance`).

    torch.ops.gance.synthesize_w(dlatents, engine)         [B, W, 512] f32  -> [B, R, R, 3] u8
    torch.ops.gance.synthesize_z(z, engine, psi)           [B, 512]    f32  -> [B, R, R, 3] u8
    torch.ops.gance.synthesize_w_image(dlatents, engine)   ... -> ([B, R, R, 3] u8, [B, 3, R, R] f32)
    torch.ops.gance.resize_bicubic(frames, side)           [B, S, S, 3] u8  -> [B, side, side, 3] u8
    torch.ops.gance.synthesize_w_out / synthesize_z_out / resize_bicubic_out   the same, into a caller-owned `out`
    torch.ops.gance.blend(audio, latent_row0, blend)       [samples] f32, [F, L] f32 -> ([N, depth, L] f32, [N] i32)

Tensors are CUDA (HIP) tensors; every op launches on torch's CURRENT stream of the tensor's device and returns
without synchronising, so the ops compose with torch code and with `torch.distributed` collectives in stream
order. `engine` / `blend` are integer handles of objects created through `register_engine` /
`register_blend` (the ops cannot take Python objects); status codes of the C ABI become Python exceptions
(`hip_lib.GanceHipError`, or ValueError for argument errors the reference reports as ValueError).

The ops are thin: they validate shapes, allocate the outputs with torch and pass raw pointers + the stream to the
ctypes binding (gance_amd/hip_lib.py). There is no CPU implementation: on a CPU tensor an op raises.
"""

import weakref
from typing import Tuple

import torch

from gance_amd import hip_lib

# weak: a handle must not keep an engine's HBM (weights + its share of the workspace) alive after its owner let go
_ENGINES: "weakref.WeakValueDictionary[int, hip_lib.Engine]" = weakref.WeakValueDictionary()
_BLENDS: "weakref.WeakValueDictionary[int, hip_lib.Blend]" = weakref.WeakValueDictionary()
_NEXT_HANDLE = [1]


def register_engine(engine: hip_lib.Engine) -> int:
    """Make an Engine addressable from the ops; returns its integer handle."""
    handle = _NEXT_HANDLE[0]
    _NEXT_HANDLE[0] += 1
    _ENGINES[handle] = engine
    return handle


def register_blend(blend: hip_lib.Blend) -> int:
    """Make a Blend addressable from the ops; returns its integer handle."""
    handle = _NEXT_HANDLE[0]
    _NEXT_HANDLE[0] += 1
    _BLENDS[handle] = blend
    return handle


def unregister(handle: int) -> None:
    """Forget a handle (the object itself is closed by its owner)."""
    _ENGINES.pop(handle, None)
    _BLENDS.pop(handle, None)


def _engine(handle: int) -> hip_lib.Engine:
    try:
        return _ENGINES[handle]
    except KeyError:
        raise ValueError(f"unknown engine handle {handle} (never registered, closed or garbage-collected)") from None


def _require_cuda(tensor: torch.Tensor, dtype: torch.dtype, name: str) -> None:
    if not tensor.is_cuda:
        raise RuntimeError(f"gance ops run on an MI355X only: `{name}` is on {tensor.device} (there is no CPU fallback)")
    if tensor.dtype != dtype:
        raise TypeError(f"`{name}` must be {dtype}, got {tensor.dtype}")


def _stream(tensor: torch.Tensor) -> int:
    return torch.cuda.current_stream(tensor.device).cuda_stream


@torch.library.custom_op("gance::synthesize_w", mutates_args=(), device_types="cuda")
def synthesize_w(dlatents: torch.Tensor, engine: int) -> torch.Tensor:
    """create_image_matrix for a batch (network_functions.py:160-169): synthesis with the stored noise."""
    eng = _engine(engine)
    _require_cuda(dlatents, torch.float32, "dlatents")
    if dlatents.dim() != 3 or dlatents.shape[1] != eng.num_layers or dlatents.shape[2] != eng.vector_length:
        raise ValueError(f"dlatents must be [B, {eng.num_layers}, {eng.vector_length}], got {tuple(dlatents.shape)}")
    dlatents = dlatents.contiguous()
    frames = torch.empty((dlatents.shape[0], eng.resolution, eng.resolution, 3), dtype=torch.uint8, device=dlatents.device)
    eng.synthesize_w_device(dlatents.data_ptr(), dlatents.shape[0], frames.data_ptr(), 0, _stream(dlatents))
    return frames


@synthesize_w.register_fake
def _(dlatents: torch.Tensor, engine: int) -> torch.Tensor:
    side = _engine(engine).resolution
    return dlatents.new_empty((dlatents.shape[0], side, side, 3), dtype=torch.uint8)


@torch.library.custom_op("gance::synthesize_w_image", mutates_args=(), device_types="cuda")
def synthesize_w_image(dlatents: torch.Tensor, engine: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """As synthesize_w, also returning the float image [B, 3, R, R] before the uint8 conversion."""
    eng = _engine(engine)
    _require_cuda(dlatents, torch.float32, "dlatents")
    if dlatents.dim() != 3 or dlatents.shape[1] != eng.num_layers or dlatents.shape[2] != eng.vector_length:
        raise ValueError(f"dlatents must be [B, {eng.num_layers}, {eng.vector_length}], got {tuple(dlatents.shape)}")
    dlatents = dlatents.contiguous()
    batch, side = dlatents.shape[0], eng.resolution
    frames = torch.empty((batch, side, side, 3), dtype=torch.uint8, device=dlatents.device)
    image = torch.empty((batch, 3, side, side), dtype=torch.float32, device=dlatents.device)
    eng.synthesize_w_device(dlatents.data_ptr(), batch, frames.data_ptr(), image.data_ptr(), _stream(dlatents))
    return frames, image


@synthesize_w_image.register_fake
def _(dlatents: torch.Tensor, engine: int) -> Tuple[torch.Tensor, torch.Tensor]:
    side = _engine(engine).resolution
    batch = dlatents.shape[0]
    return dlatents.new_empty((batch, side, side, 3), dtype=torch.uint8), dlatents.new_empty((batch, 3, side, side))


@torch.library.custom_op("gance::synthesize_z", mutates_args=(), device_types="cuda")
def synthesize_z(z: torch.Tensor, engine: int, truncation_psi: float) -> torch.Tensor:
    """create_image_vector for a batch (network_functions.py:144-158): mapping, truncation, synthesis."""
    eng = _engine(engine)
    _require_cuda(z, torch.float32, "z")
    if z.dim() != 2 or z.shape[1] != eng.vector_length:
        raise ValueError(f"z must be [B, {eng.vector_length}], got {tuple(z.shape)}")
    z = z.contiguous()
    frames = torch.empty((z.shape[0], eng.resolution, eng.resolution, 3), dtype=torch.uint8, device=z.device)
    eng.synthesize_z_device(z.data_ptr(), z.shape[0], float(truncation_psi), frames.data_ptr(), 0, _stream(z))
    return frames


@synthesize_z.register_fake
def _(z: torch.Tensor, engine: int, truncation_psi: float) -> torch.Tensor:
    side = _engine(engine).resolution
    return z.new_empty((z.shape[0], side, side, 3), dtype=torch.uint8)


@torch.library.custom_op("gance::synthesize_w_out", mutates_args=("out",), device_types="cuda")
def synthesize_w_out(dlatents: torch.Tensor, engine: int, out: torch.Tensor) -> None:
    """synthesize_w writing into caller-owned frames `out` [B, R, R, 3] u8 (a contiguous slice of a larger buffer)."""
    eng = _engine(engine)
    _require_cuda(dlatents, torch.float32, "dlatents")
    _require_cuda(out, torch.uint8, "out")
    batch = dlatents.shape[0]
    if dlatents.dim() != 3 or dlatents.shape[1] != eng.num_layers or dlatents.shape[2] != eng.vector_length:
        raise ValueError(f"dlatents must be [B, {eng.num_layers}, {eng.vector_length}], got {tuple(dlatents.shape)}")
    if tuple(out.shape) != (batch, eng.resolution, eng.resolution, 3) or not out.is_contiguous() or out.device != dlatents.device:
        raise ValueError(f"out must be a contiguous [{batch}, {eng.resolution}, {eng.resolution}, 3] uint8 tensor on {dlatents.device}")
    dlatents = dlatents.contiguous()
    eng.synthesize_w_device(dlatents.data_ptr(), batch, out.data_ptr(), 0, _stream(dlatents))


@torch.library.custom_op("gance::synthesize_z_out", mutates_args=("out",), device_types="cuda")
def synthesize_z_out(z: torch.Tensor, engine: int, truncation_psi: float, out: torch.Tensor) -> None:
    """synthesize_z writing into caller-owned frames `out` [B, R, R, 3] u8."""
    eng = _engine(engine)
    _require_cuda(z, torch.float32, "z")
    _require_cuda(out, torch.uint8, "out")
    batch = z.shape[0]
    if z.dim() != 2 or z.shape[1] != eng.vector_length:
        raise ValueError(f"z must be [B, {eng.vector_length}], got {tuple(z.shape)}")
    if tuple(out.shape) != (batch, eng.resolution, eng.resolution, 3) or not out.is_contiguous() or out.device != z.device:
        raise ValueError(f"out must be a contiguous [{batch}, {eng.resolution}, {eng.resolution}, 3] uint8 tensor on {z.device}")
    z = z.contiguous()
    eng.synthesize_z_device(z.data_ptr(), batch, float(truncation_psi), out.data_ptr(), 0, _stream(z))


@torch.library.custom_op("gance::resize_bicubic_out", mutates_args=("out",), device_types="cuda")
def resize_bicubic_out(frames: torch.Tensor, out: torch.Tensor) -> None:
    """resize_bicubic writing into caller-owned `out` [B, side, side, 3] u8."""
    _require_cuda(frames, torch.uint8, "frames")
    _require_cuda(out, torch.uint8, "out")
    if frames.dim() != 4 or frames.shape[1] != frames.shape[2] or frames.shape[3] != 3 or out.dim() != 4 or out.shape[0] != frames.shape[0]:
        raise ValueError("frames must be [B, S, S, 3] and out [B, side, side, 3]")
    if out.shape[1] != out.shape[2] or out.shape[3] != 3 or not out.is_contiguous() or out.device != frames.device:
        raise ValueError("out must be a contiguous [B, side, side, 3] uint8 tensor on the frames' device")
    frames = frames.contiguous()
    if frames.shape[0]:
        hip_lib.resize_bicubic_u8_device(frames.data_ptr(), frames.shape[0], frames.shape[1], out.data_ptr(), out.shape[1], _stream(frames))


@torch.library.custom_op("gance::resize_bicubic", mutates_args=(), device_types="cuda")
def resize_bicubic(frames: torch.Tensor, side: int) -> torch.Tensor:
    """cv2.resize(..., INTER_CUBIC) of video_common.py:416-418 on uint8 NHWC frames in HBM."""
    _require_cuda(frames, torch.uint8, "frames")
    if frames.dim() != 4 or frames.shape[1] != frames.shape[2] or frames.shape[3] != 3:
        raise ValueError(f"frames must be [B, S, S, 3], got {tuple(frames.shape)}")
    frames = frames.contiguous()
    out = torch.empty((frames.shape[0], side, side, 3), dtype=torch.uint8, device=frames.device)
    if frames.shape[0]:
        hip_lib.resize_bicubic_u8_device(frames.data_ptr(), frames.shape[0], frames.shape[1], out.data_ptr(), side, _stream(frames))
    return out


@resize_bicubic.register_fake
def _(frames: torch.Tensor, side: int) -> torch.Tensor:
    return frames.new_empty((frames.shape[0], side, side, 3))


@torch.library.custom_op("gance::blend", mutates_args=(), device_types="cuda")
def blend(audio: torch.Tensor, latent_row0: torch.Tensor, blend_handle: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """alpha_blend_projection_file (visualization_inputs.py:169-270): audio + projected latents -> per-frame latent matrices and network indices."""
    try:
        owner = _BLENDS[blend_handle]
    except KeyError:
        raise ValueError(f"unknown blend handle {blend_handle}") from None
    _require_cuda(audio, torch.float32, "audio")
    _require_cuda(latent_row0, torch.float32, "latent_row0")
    config = owner.config
    audio, latent_row0 = audio.contiguous(), latent_row0.contiguous()
    if latent_row0.numel() != config.num_projection_frames * config.vector_length:
        raise ValueError("latent_row0 must hold num_projection_frames * vector_length values")
    dlatents = torch.empty((config.num_frames, config.latent_depth, config.vector_length), dtype=torch.float32, device=audio.device)
    indices = torch.empty((config.num_frames,), dtype=torch.int32, device=audio.device)
    owner.run_device(audio.data_ptr(), audio.numel(), latent_row0.data_ptr(), dlatents.data_ptr(), indices.data_ptr(), stream=_stream(audio))
    return dlatents, indices


@blend.register_fake
def _(audio: torch.Tensor, latent_row0: torch.Tensor, blend_handle: int) -> Tuple[torch.Tensor, torch.Tensor]:
    config = _BLENDS[blend_handle].config
    return (
        audio.new_empty((config.num_frames, config.latent_depth, config.vector_length)),
        audio.new_empty((config.num_frames,), dtype=torch.int32),
    )
