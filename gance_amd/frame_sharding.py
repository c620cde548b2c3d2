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


# ---- ordered frame stream -------------------------------------------------------------------------
# The reference pulls frames lazily through an iterator chain into the video writer
# (gance/projection_file_blend.py:343, gance/image_sources/video_common.py:346-349): no more than a few
# frames exist at once. The sharded equivalent: the frame sequence is cut into chunks of world_size * C
# frames; rank g synthesises frames [g*C, (g+1)*C) of every chunk, so ONE gather per chunk lands that chunk
# in frame order on rank 0, while every rank already synthesises the next chunk. Rank 0 drains each
# gathered chunk to a pinned-host ring on a copy stream. HBM holds two chunks, the host three.


def stream_piece(num_frames: int, world_size: int, frames_per_rank: int, chunk: int, rank: int) -> Tuple[int, int]:
    """Half-open global frame range rank `rank` synthesises in chunk `chunk` (may be empty at the tail)."""
    base = chunk * world_size * frames_per_rank + rank * frames_per_rank
    start = min(num_frames, base)
    return start, min(num_frames, start + frames_per_rank)


def stream_chunks(num_frames: int, world_size: int, frames_per_rank: int) -> int:
    """Number of chunks of world_size * frames_per_rank frames that cover the sequence."""
    per_chunk = world_size * frames_per_rank
    return -(-num_frames // per_chunk) if num_frames > 0 else 0


def stream_order(num_frames: int, world_size: int, frames_per_rank: int, rank: int) -> List[int]:
    """Global indices of the frames `rank` synthesises, in the order it synthesises them."""
    order: List[int] = []
    for chunk in range(stream_chunks(num_frames, world_size, frames_per_rank)):
        start, end = stream_piece(num_frames, world_size, frames_per_rank, chunk, rank)
        order.extend(range(start, end))
    return order


def scatter_for_stream(all_inputs: Optional[torch.Tensor], num_frames: int, frames_per_rank: int, device: torch.device) -> torch.Tensor:
    """
    Rank 0 holds the per-frame network inputs [N, ...] (None elsewhere); every rank receives the inputs of the
    frames it will synthesise, in its `stream_order`: one scatter of per-rank, per-chunk latent pieces.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size == 1:
        assert all_inputs is not None
        return all_inputs.to(device)
    meta = [None]
    if rank == 0:
        assert all_inputs is not None and all_inputs.shape[0] == num_frames
        meta = [(tuple(all_inputs.shape[1:]), all_inputs.dtype)]
    dist.broadcast_object_list(meta, src=0)
    tail_shape, dtype = meta[0]
    longest = max(len(stream_order(num_frames, world_size, frames_per_rank, r)) for r in range(world_size))
    recv = torch.empty((longest, *tail_shape), dtype=dtype, device=device)
    pieces: Optional[List[torch.Tensor]] = None
    if rank == 0:
        on_device = all_inputs.to(device)
        pieces = []
        for r in range(world_size):
            order = stream_order(num_frames, world_size, frames_per_rank, r)
            piece = torch.zeros((longest, *tail_shape), dtype=dtype, device=device)
            if order:
                piece[: len(order)] = on_device.index_select(0, torch.tensor(order, dtype=torch.long, device=device))
            pieces.append(piece)
    dist.scatter(recv, scatter_list=pieces, src=0)
    return recv[: len(stream_order(num_frames, world_size, frames_per_rank, rank))]


def ordered_frame_stream(synthesize_piece, num_frames: int, frames_per_rank: int, frame_shape: Tuple[int, int, int], device: torch.device):
    """
    Generator (collective: every rank must exhaust it). `synthesize_piece(offset, count)` returns this rank's
    next `count` frames as a uint8 tensor [count, *frame_shape] on `device`, where `offset` counts the frames
    the rank has produced so far (an index into its `scatter_for_stream` inputs).
    On rank 0 it yields (first_frame_index, frames) per chunk, in frame order, `frames` a uint8 numpy view of
    a pinned host ring slot [n, *frame_shape] that stays valid until the generator is advanced twice more;
    other ranks yield nothing. Chunk k's gather and host drain overlap chunk k+1's synthesis.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    chunks = stream_chunks(num_frames, world_size, frames_per_rank)
    per_chunk = world_size * frames_per_rank
    on_gpu = device.type == "cuda"
    ring_slots = 3
    gathered = [torch.empty((per_chunk, *frame_shape), dtype=torch.uint8, device=device) for _ in range(2)] if rank == 0 else None
    local = [torch.zeros((frames_per_rank, *frame_shape), dtype=torch.uint8, device=device) for _ in range(2)]
    host_ring = (
        [torch.empty((per_chunk, *frame_shape), dtype=torch.uint8, pin_memory=on_gpu) for _ in range(ring_slots)] if rank == 0 else None
    )
    copy_stream = torch.cuda.Stream(device) if on_gpu else None
    copied = [None] * ring_slots  # events: ring slot filled
    works = [None, None]
    produced = 0

    def drain(chunk: int) -> None:
        """Gather of `chunk` done -> copy it to its host ring slot (rank 0)."""
        slot = chunk & 1
        if works[slot] is not None:
            works[slot].wait()
            works[slot] = None
        if rank != 0:
            return
        ring = chunk % ring_slots
        if on_gpu:
            copy_stream.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(copy_stream):
                host_ring[ring].copy_(gathered[slot], non_blocking=True)
                event = torch.cuda.Event()
                event.record(copy_stream)
            copied[ring] = event
        else:
            host_ring[ring].copy_(gathered[slot])

    def emit(chunk: int):
        ring = chunk % ring_slots
        if copied[ring] is not None:
            copied[ring].synchronize()
            copied[ring] = None
        first = chunk * per_chunk
        count = min(num_frames, first + per_chunk) - first
        return first, host_ring[ring][:count].numpy()

    for chunk in range(chunks):
        slot = chunk & 1
        start, end = stream_piece(num_frames, world_size, frames_per_rank, chunk, rank)
        count = end - start
        if on_gpu and copy_stream is not None and rank == 0 and chunk >= 2:
            # the drain of chunk-2 read gathered[slot]: the gather below may only overwrite it afterwards
            torch.cuda.current_stream(device).wait_stream(copy_stream)
        if count:
            frames = synthesize_piece(produced, count)
            local[slot][:count].copy_(frames)
            produced += count
        if world_size > 1:
            works[slot] = dist.gather(local[slot], gather_list=list(gathered[slot].chunk(world_size, dim=0)) if rank == 0 else None, dst=0, async_op=True)
        elif rank == 0:
            gathered[slot][:frames_per_rank].copy_(local[slot])
        if chunk >= 1:
            drain(chunk - 1)
        if rank == 0 and chunk >= 2:
            yield emit(chunk - 2)
    if chunks >= 1:
        drain(chunks - 1)
    if rank == 0:
        for chunk in range(max(0, chunks - 2), chunks):
            yield emit(chunk)
