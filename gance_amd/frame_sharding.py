"""
Frame sharding across the GPUs of one node: one process per GPU, `torch.distributed` (backend
"nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no multi-GPU path at all (SURVEY.md §2 "NCCL / MPI / Gloo: none"); frames are
independent once `combined` and `network_indices` exist
(gance/data_into_network_visualization/network_visualization.py:233-251,625-628), so the only
exchanges are a scatter of per-frame latent chunks from rank 0 and a gather of the finished uint8
frames, in frame order, back to rank 0. No all-reduce anywhere.

Partitioning: contiguous blocks, rank g owns frames [g*ceil(N/G), min(N, (g+1)*ceil(N/G))).
"""

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Half-open frame range owned by `rank` (contiguous blocks, last ranks may be short/empty)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    per_rank = -(-num_frames // world_size)
    start = min(num_frames, rank * per_rank)
    return start, min(num_frames, start + per_rank)


def scatter_latents(all_latents: Optional[torch.Tensor], num_frames: int, device: torch.device) -> torch.Tensor:
    """
    Rank 0 holds `all_latents` [N, ...] (other ranks pass None); every rank returns its own block
    [n_local, ...] on `device`. Blocks are padded to a common length for the collective and trimmed
    after it.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size == 1:
        assert all_latents is not None
        return all_latents.to(device)
    per_rank = -(-num_frames // world_size)
    meta = [None]
    if rank == 0:
        assert all_latents is not None and all_latents.shape[0] == num_frames
        meta = [(tuple(all_latents.shape[1:]), all_latents.dtype)]
    dist.broadcast_object_list(meta, src=0)
    tail_shape, dtype = meta[0]
    recv = torch.empty((per_rank, *tail_shape), dtype=dtype, device=device)
    chunks: Optional[List[torch.Tensor]] = None
    if rank == 0:
        padded = torch.zeros((per_rank * world_size, *tail_shape), dtype=dtype, device=device)
        padded[:num_frames] = all_latents.to(device)
        chunks = [chunk.contiguous() for chunk in padded.chunk(world_size, dim=0)]
    dist.scatter(recv, scatter_list=chunks, src=0)
    start, end = shard_bounds(num_frames, world_size, rank)
    return recv[: end - start]


def gather_frames(
    local_frames: torch.Tensor, num_frames: int, async_op: bool = False, out: Optional[torch.Tensor] = None
):
    """
    Gather every rank's frames [n_local, H, W, 3] uint8 into rank 0, in frame order.
    Returns (frames [N, H, W, 3] on rank 0 / None elsewhere, work handle or None).
    With async_op the caller must `work.wait()` before reading `frames` or reusing `local_frames`.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size == 1:
        return local_frames[:num_frames], None
    per_rank = -(-num_frames // world_size)
    if local_frames.shape[0] != per_rank:
        padded = torch.zeros((per_rank, *local_frames.shape[1:]), dtype=local_frames.dtype, device=local_frames.device)
        padded[: local_frames.shape[0]] = local_frames
        local_frames = padded
    gather_list = None
    if rank == 0:
        if out is None:
            out = torch.empty(
                (world_size * per_rank, *local_frames.shape[1:]), dtype=local_frames.dtype, device=local_frames.device
            )
        gather_list = list(out.chunk(world_size, dim=0))
    work = dist.gather(local_frames.contiguous(), gather_list=gather_list, dst=0, async_op=async_op)
    frames = out[:num_frames] if rank == 0 else None
    return frames, (work if async_op else None)
