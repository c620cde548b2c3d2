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

import datetime
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

# Host-side control exchanges (one status word per chunk) must not outlive a dead rank by gloo's default 30 minutes:
# the same knob bench.py passes to init_process_group
CONTROL_GROUP_TIMEOUT = datetime.timedelta(seconds=int(os.environ.get("GANCE_PROCESS_GROUP_TIMEOUT_S", "180")))
DRAIN_MODES = ("rank0", "per-rank")


def collectives_forced() -> bool:
    """
    GANCE_FORCE_COLLECTIVES=1: with a process group of ONE rank the scatter / gather / status exchange still go through
    `torch.distributed` instead of the world-size-1 short cuts. This is how a one-GPU box runs the RCCL code path
    (backend "nccl", world size 1: `dist.scatter`, the asynchronous `dist.gather`, the gloo control group beside it, the
    reader stream's `work.wait()`) -- tests/test_full_size_stream_gpu.py; nothing else sets it.
    """
    return os.environ.get("GANCE_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()


def single_process() -> bool:
    """No collective is needed: not initialised, or one rank and the collectives are not forced."""
    return not dist.is_initialized() or (dist.get_world_size() == 1 and not collectives_forced())


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
    if single_process():
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
    if single_process():
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
# gathered chunk to a pinned-host ring on a side stream. HBM holds two chunks, the host three. With a
# per-chunk stage between gather and drain (the eye-tracking overlay: `ordered_device_chunks` + `HostRing`) the
# same pieces are used with that stage in the middle.


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
    if single_process():
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


class StreamRankError(RuntimeError):
    """Another rank of the frame stream failed (or this one did and is telling the others); the job cannot continue."""


_CONTROL_GROUP = [None]  # the host-side (gloo) group the ranks exchange their per-chunk status word on


def control_group():
    """
    A gloo group over all ranks for host-side control messages (one status word per chunk), created on first use.
    Collective. With a gloo default group that one is used. Status words must not ride on RCCL: reading them back
    would put a stream synchronisation into every chunk, and a rank that died leaves an RCCL collective hanging until
    the watchdog aborts the process, whereas a gloo collective raises after the group's timeout.
    """
    if single_process():
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    if _CONTROL_GROUP[0] is None:
        _CONTROL_GROUP[0] = dist.new_group(backend="gloo", timeout=CONTROL_GROUP_TIMEOUT)
    return _CONTROL_GROUP[0]


def exchange_status(failed: bool, where: str, consumer: bool = False) -> None:
    """
    Every rank contributes one word; if any rank reports a failure every rank leaves with StreamRankError (the
    failing rank re-raises its own exception instead: the caller does that). The reference relays a worker's
    start-up error to its parent the same way (network_functions.py:270-278). A rank that has died makes the
    exchange time out (the control group's timeout) and raise on the survivors. `consumer`: the failure is the
    rank's consumer side (it left the stream between two chunks), not the step `where` names.
    """
    group = control_group()
    if group is None:
        return
    status = torch.tensor([(dist.get_rank() + 1) | ((1 << 16) if consumer else 0) if failed else 0], dtype=torch.int32)
    dist.all_reduce(status, op=dist.ReduceOp.MAX, group=group)
    word = int(status.item())
    if word != 0 and not failed:
        what = "consuming the stream" if word >> 16 else where
        raise StreamRankError(f"rank {(word & 0xFFFF) - 1} failed {what}; rank {dist.get_rank()} stops")


def ordered_device_chunks(  # pylint: disable=too-many-arguments,too-many-locals,too-many-branches,too-many-statements
    synthesize_piece, num_frames: int, frames_per_rank: int, frame_shape: Tuple[int, int, int], device: torch.device, drain: str = "rank0"
):
    """
    Generator (collective: every rank must exhaust it). `synthesize_piece(offset, count)` returns this rank's
    next `count` frames as a uint8 tensor [count, *frame_shape] on `device`, where `offset` counts the frames
    the rank has produced so far (an index into its `scatter_for_stream` inputs). A callable with a true attribute
    `writes_into` is called as `synthesize_piece(offset, count, out)` instead and writes the frames into `out`, a
    view of the stream's own buffer: no copy of the chunk (a device-to-device copy of 64 frames of 1024^2 is a blit
    kernel of about a millisecond in the stream of the synthesis). With one rank the chunk is handed out from that
    buffer; only several ranks need gather buffers.

    `drain="rank0"` (default): ONE gather per chunk lands it in frame order on rank 0, which yields
    (first_frame_index, frames, reader_stream) per chunk; other ranks yield nothing. This is the form for consumers that
    need the ordered stream in one place in HBM (the overlay stage, an encoder on rank 0's host).
    `drain="per-rank"`: no gather at all -- EVERY rank yields its own piece of every chunk,
    (first_frame_index of the piece, frames, reader_stream), and drains it over its OWN PCIe link (the caller's HostRing):
    the host-bound legs (2160^2 frames: 14 MB each, ~57 GB/s of pinned D2H per link = ~4 000 frames/s through rank 0
    alone) then scale with the node. The ranks stay in step through the per-chunk status word, which is also what
    sequences the chunks: piece (chunk k, rank g) holds frames [k W C + g C, k W C + (g + 1) C).

    `frames` is a view of one of two buffers in HBM, [n, *frame_shape] uint8, valid ONLY UNTIL THE GENERATOR IS ADVANCED
    AGAIN: whoever reads it must enqueue the read on `reader_stream` (a side stream that already waits for the chunk's
    gather / synthesis; None on the CPU, where the read must finish) before advancing -- the generator makes whatever
    overwrites the buffer next wait for that stream. (Chunk k is handed out during iteration k + 1, and iteration k + 2
    writes the same buffer.) Chunk k is handed out after chunk k+1's synthesis and gather have been issued, so the
    consumer's work on it (overlay, host drain) overlaps them: the reader stream waits for chunk k's OWN gather (an event
    recorded behind it), not for what was issued since -- a consumer that blocks the host on the reader stream (the
    overlay gate copies frames to the host for the landmark detector) would otherwise sit out chunk k+1's synthesis
    first and leave the GPU idle while it works.

    Failure handling: every rank exchanges one status word per chunk on the host-side control group; an exception in
    `synthesize_piece` on any rank ends the generator on EVERY rank (the failing rank re-raises its exception, the
    others raise StreamRankError) instead of leaving them blocked in the next gather. A consumer that fails or drops
    the generator between two chunks (GeneratorExit / an exception thrown in at the yield) reports that in its next
    status word, so the other ranks leave with StreamRankError in that chunk instead of waiting out the timeout.
    """
    if drain not in DRAIN_MODES:
        raise ValueError(f"drain must be one of {DRAIN_MODES}, got {drain!r}")
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    collective = not single_process()  # (several ranks -- or one, with the collectives forced: collectives_forced)
    per_rank_drain = drain == "per-rank" and collective
    chunks = stream_chunks(num_frames, world_size, frames_per_rank)
    per_chunk = world_size * frames_per_rank
    on_gpu = device.type == "cuda"
    local = [torch.zeros((frames_per_rank, *frame_shape), dtype=torch.uint8, device=device) for _ in range(2)]
    hands_out = rank == 0 or per_rank_drain
    gathered = None
    if hands_out:
        gathered = local if not collective or per_rank_drain else [torch.empty((per_chunk, *frame_shape), dtype=torch.uint8, device=device) for _ in range(2)]
    writes_into = bool(getattr(synthesize_piece, "writes_into", False))
    reader_stream = torch.cuda.Stream(device) if on_gpu and hands_out else None
    issued = [None, None]  # per buffer: event behind the last synthesis / gather issued into it
    works = [None, None]
    produced = 0

    def hand_out(chunk: int):
        """Gather of `chunk` done (in stream order on the reader stream) -> its view of the buffer (None: an empty tail piece)."""
        slot = chunk & 1
        if per_rank_drain:
            first, end = stream_piece(num_frames, world_size, frames_per_rank, chunk, rank)
            count = end - first
        else:
            first = chunk * per_chunk
            count = min(num_frames, first + per_chunk) - first
        if count <= 0:
            return None
        if reader_stream is None:
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
            return first, gathered[slot][:count], None
        with torch.cuda.stream(reader_stream):  # (an asynchronous collective's wait() makes the CURRENT stream wait for it)
            if works[slot] is not None:
                works[slot].wait()
                works[slot] = None
            reader_stream.wait_event(issued[slot])
        return first, gathered[slot][:count], reader_stream

    exchanges = 0      # status words this rank has contributed
    settled = False    # the stream ended in a way every rank already knows about (normal end, or a relayed failure)
    try:
        for chunk in range(chunks):
            slot = chunk & 1
            start, end = stream_piece(num_frames, world_size, frames_per_rank, chunk, rank)
            count = end - start
            if reader_stream is not None and chunk >= 2:
                # the consumer of chunk-2 read gathered[slot] on the reader stream: the synthesis / gather below may only overwrite it afterwards
                torch.cuda.current_stream(device).wait_stream(reader_stream)
            if works[slot] is not None:  # (ranks > 0 hand nothing out: retire the gather that last read local[slot] here)
                works[slot].wait()
                works[slot] = None
            failure = None
            if count:
                try:
                    if writes_into:
                        synthesize_piece(produced, count, local[slot][:count])
                    else:
                        local[slot][:count].copy_(synthesize_piece(produced, count))
                except Exception as error:  # pylint: disable=broad-except
                    failure = error
                produced += count
            exchanges += 1
            try:
                exchange_status(failure is not None, f"synthesising chunk {chunk}")
            except StreamRankError:
                settled = True
                if failure is None:
                    raise
            except Exception:  # (the exchange itself failed: a dead rank, a timeout -- nothing more can be relayed)
                settled = True
                raise
            if failure is not None:
                settled = True
                raise failure
            if collective and not per_rank_drain:
                works[slot] = dist.gather(local[slot], gather_list=list(gathered[slot].chunk(world_size, dim=0)) if rank == 0 else None, dst=0, async_op=True)
            if reader_stream is not None:
                issued[slot] = torch.cuda.Event()
                issued[slot].record(torch.cuda.current_stream(device))
            if hands_out and chunk >= 1:
                piece = hand_out(chunk - 1)
                if piece is not None:
                    yield piece
        settled = True
        if hands_out and chunks >= 1:
            piece = hand_out(chunks - 1)
            if piece is not None:
                yield piece
    finally:
        if not settled and exchanges < chunks:
            # the consumer failed or dropped the generator between two chunks: the other ranks are on their way into the
            # next status exchange -- tell them there, instead of letting them wait for a word that never comes
            try:
                exchange_status(True, "consuming the stream", consumer=True)
            except Exception:  # pylint: disable=broad-except
                pass
        for work in works:
            if work is not None:
                work.wait()


class HostRing:
    """
    Pinned host buffers that device chunks are drained to on the reader stream, one chunk behind: `push` starts the
    copy of a chunk and returns the PREVIOUS chunk as a numpy view once its copy has landed (None the first time),
    `flush` returns the last one. A returned view stays valid until `push` has been called `slots - 1` more times.
    """

    def __init__(self, chunk_frames: int, frame_shape: Tuple[int, int, int], device: torch.device, slots: int = 3) -> None:
        self._on_gpu = device.type == "cuda"
        self._ring = [torch.empty((chunk_frames, *frame_shape), dtype=torch.uint8, pin_memory=self._on_gpu) for _ in range(slots)]
        self._next = 0
        self._pending = None  # (first, count, slot, event)

    def _finish(self):
        if self._pending is None:
            return None
        first, count, slot, event = self._pending
        self._pending = None
        if event is not None:
            event.synchronize()
        return first, self._ring[slot][:count].numpy()

    def push(self, first: int, frames: torch.Tensor, reader_stream):
        """Start draining `frames` (device) into the next slot; returns the previous chunk (first, numpy view) or None."""
        done = self._finish()
        slot = self._next
        self._next = (self._next + 1) % len(self._ring)
        count = int(frames.shape[0])
        event = None
        if self._on_gpu and reader_stream is not None:
            with torch.cuda.stream(reader_stream):
                self._ring[slot][:count].copy_(frames, non_blocking=True)
                event = torch.cuda.Event()
                event.record(reader_stream)
        else:
            self._ring[slot][:count].copy_(frames)
        self._pending = (first, count, slot, event)
        return done

    def flush(self):
        """The last chunk pushed (first, numpy view), or None."""
        return self._finish()


def ordered_frame_stream(
    synthesize_piece, num_frames: int, frames_per_rank: int, frame_shape: Tuple[int, int, int], device: torch.device, drain: str = "rank0"
):
    """
    `ordered_device_chunks` drained to the host (collective: every rank must exhaust it). On rank 0 it yields
    (first_frame_index, frames) per chunk, in frame order, `frames` a uint8 numpy view of a pinned host ring slot
    [n, *frame_shape] that stays valid until the generator is advanced twice more; other ranks yield nothing.
    Chunk k's gather and host drain overlap chunk k+1's synthesis. HBM holds two chunks, the host three.
    With `drain="per-rank"` every rank yields its OWN pieces (first_frame_index of the piece, frames), drained over its
    own PCIe link to its own pinned ring; no gather runs.
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    ring = None
    for first, frames, reader_stream in ordered_device_chunks(synthesize_piece, num_frames, frames_per_rank, frame_shape, device, drain=drain):
        if ring is None:
            ring = HostRing(frames_per_rank if drain == "per-rank" else world_size * frames_per_rank, frame_shape, device)
        done = ring.push(first, frames, reader_stream)
        if done is not None:
            yield done
    if ring is not None:
        last = ring.flush()
        if last is not None:
            yield last
