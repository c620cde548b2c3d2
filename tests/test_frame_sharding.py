"""
CPU tests of the multi-GPU path with the gloo backend, world_size 2 (and 3 for ragged shards):
scatter of per-frame latent chunks from rank 0, gather of the ordered frame stream to rank 0.
"""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gance_amd import frame_sharding


def test_shard_bounds_cover_all_frames_once() -> None:
    for num_frames in (0, 1, 7, 8, 1800):
        for world_size in (1, 2, 3, 8):
            seen = []
            for rank in range(world_size):
                start, end = frame_sharding.shard_bounds(num_frames, world_size, rank)
                assert 0 <= start <= end <= num_frames
                seen.extend(range(start, end))
            assert seen == list(range(num_frames))
    with pytest.raises(ValueError):
        frame_sharding.shard_bounds(10, 2, 2)


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _worker(rank: int, world_size: int, port: int, num_frames: int) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        device = torch.device("cpu")
        all_latents = None
        if rank == 0:
            all_latents = torch.arange(num_frames * 2 * 4, dtype=torch.float32).reshape(num_frames, 2, 4)
        local = frame_sharding.scatter_latents(all_latents, num_frames, device)
        start, end = frame_sharding.shard_bounds(num_frames, world_size, rank)
        expected = torch.arange(num_frames * 8, dtype=torch.float32).reshape(num_frames, 2, 4)[start:end]
        assert torch.equal(local, expected), (rank, local, expected)

        # "synthesize": frame f is a 2x2x3 image filled with f (mod 256)
        frames = torch.stack(
            [torch.full((2, 2, 3), f % 256, dtype=torch.uint8) for f in range(start, end)]
        ) if end > start else torch.empty((0, 2, 2, 3), dtype=torch.uint8)
        gathered, work = frame_sharding.gather_frames(frames, num_frames, async_op=True)
        work.wait()
        if rank == 0:
            assert gathered.shape == (num_frames, 2, 2, 3)
            assert gathered[:, 0, 0, 0].tolist() == [f % 256 for f in range(num_frames)]
        else:
            assert gathered is None
        gathered_sync, work_sync = frame_sharding.gather_frames(frames, num_frames)
        assert work_sync is None
        if rank == 0:
            assert torch.equal(gathered_sync, gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,num_frames", [(2, 8), (2, 7), (3, 10)])
def test_scatter_then_gather_keeps_frame_order(world_size: int, num_frames: int) -> None:
    mp.spawn(_worker, args=(world_size, _free_port(), num_frames), nprocs=world_size, join=True)


def test_single_process_is_a_passthrough() -> None:
    latents = torch.randn(5, 18, 512)
    assert torch.equal(frame_sharding.scatter_latents(latents, 5, torch.device("cpu")), latents)
    frames = torch.zeros((5, 4, 4, 3), dtype=torch.uint8)
    out, work = frame_sharding.gather_frames(frames, 5)
    assert work is None and out.shape == (5, 4, 4, 3)


def test_stream_orders_partition_the_frames() -> None:
    for num_frames in (0, 1, 7, 23, 64, 1800):
        for world_size in (1, 2, 3, 8):
            for per_rank in (1, 4, 16):
                seen = sorted(sum((frame_sharding.stream_order(num_frames, world_size, per_rank, r) for r in range(world_size)), []))
                assert seen == list(range(num_frames))
                chunks = frame_sharding.stream_chunks(num_frames, world_size, per_rank)
                for chunk in range(chunks):  # a chunk's pieces, rank by rank, are consecutive frames
                    pieces = [frame_sharding.stream_piece(num_frames, world_size, per_rank, chunk, r) for r in range(world_size)]
                    flat = [f for start, end in pieces for f in range(start, end)]
                    first = chunk * world_size * per_rank
                    assert flat == list(range(first, min(num_frames, first + world_size * per_rank)))


def _stream_worker(rank: int, world_size: int, port: int, num_frames: int, per_rank: int) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        device = torch.device("cpu")
        all_inputs = torch.arange(num_frames, dtype=torch.float32).reshape(num_frames, 1) if rank == 0 else None
        mine = frame_sharding.scatter_for_stream(all_inputs, num_frames, per_rank, device)
        assert mine[:, 0].tolist() == [float(f) for f in frame_sharding.stream_order(num_frames, world_size, per_rank, rank)]

        def synthesize_piece(offset: int, count: int) -> torch.Tensor:
            # "frame f" = a 2x2x3 image filled with f mod 251, made from the scattered input (not from the index)
            values = mine[offset : offset + count, 0].to(torch.int64) % 251
            return values.to(torch.uint8).reshape(count, 1, 1, 1).expand(count, 2, 2, 3).contiguous()

        class WritesInto:  # pylint: disable=too-few-public-methods
            """The product's form: the frames go straight into the stream's chunk buffer."""

            writes_into = True

            def __call__(self, offset: int, count: int, out: torch.Tensor) -> None:
                assert tuple(out.shape) == (count, 2, 2, 3) and out.dtype == torch.uint8
                out.copy_(synthesize_piece(offset, count))

        for producer in (synthesize_piece, WritesInto()):
            got = []
            for first, frames in frame_sharding.ordered_frame_stream(producer, num_frames, per_rank, (2, 2, 3), device):
                assert rank == 0
                assert first == len(got)
                got.extend(int(frame[0, 0, 0]) for frame in frames)
                assert all(bool((frame == frame[0, 0, 0]).all()) for frame in frames)
            if rank == 0:
                assert got == [f % 251 for f in range(num_frames)]
            else:
                assert not got
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,num_frames,per_rank", [(2, 37, 4), (3, 50, 3), (2, 8, 4), (2, 3, 4)])
def test_ordered_frame_stream_gloo(world_size: int, num_frames: int, per_rank: int) -> None:
    """The product's multi-rank path: scatter of per-chunk latent pieces, chunked synthesis, ordered gather, host drain."""
    mp.spawn(_stream_worker, args=(world_size, _free_port(), num_frames, per_rank), nprocs=world_size, join=True)


def _failing_stream_worker(rank: int, world_size: int, port: int, failing_rank: int, results) -> None:
    import datetime  # pylint: disable=import-outside-toplevel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=60))
    try:
        device = torch.device("cpu")
        calls = []

        def synthesize_piece(offset: int, count: int) -> torch.Tensor:
            calls.append(offset)
            if rank == failing_rank and len(calls) == 2:
                raise ValueError("engine call refused")
            return torch.zeros((count, 2, 2, 3), dtype=torch.uint8)

        outcome = "finished"
        try:
            for _ in frame_sharding.ordered_frame_stream(synthesize_piece, 40, 4, (2, 2, 3), device):
                pass
        except frame_sharding.StreamRankError as error:
            outcome = f"relayed: {error}"
        except ValueError as error:
            outcome = f"own: {error}"
        results[rank] = (outcome, len(calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [0, 1])
def test_a_failing_rank_ends_the_stream_on_every_rank(failing_rank: int) -> None:
    """
    An exception inside one rank's synthesis must not leave the others blocked in the next gather: the per-chunk
    status exchange makes every rank leave the generator in the same chunk -- the failing rank with its own
    exception, the others with StreamRankError naming it.
    """
    world_size = 2
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_failing_stream_worker, args=(world_size, _free_port(), failing_rank, results), nprocs=world_size, join=True)
    assert results[failing_rank] == ("own: engine call refused", 2)
    other = 1 - failing_rank
    assert results[other][0].startswith(f"relayed: rank {failing_rank} failed synthesising chunk 1") and results[other][1] == 2


@pytest.mark.parametrize("num_frames,per_call", [(37, 4), (8, 4), (3, 4)])
def test_ordered_frame_stream_without_a_process_group(num_frames: int, per_call: int) -> None:
    """One rank: the chunk is handed out of the buffer the producer wrote into; every frame arrives once, in order."""

    class Producer:  # pylint: disable=too-few-public-methods
        writes_into = True

        def __call__(self, offset: int, count: int, out: torch.Tensor) -> None:
            out.copy_((torch.arange(offset, offset + count) % 251).to(torch.uint8).reshape(count, 1, 1, 1).expand(count, 2, 2, 3))

    got = []
    for first, frames in frame_sharding.ordered_frame_stream(Producer(), num_frames, per_call, (2, 2, 3), torch.device("cpu")):
        assert first == len(got)
        got.extend(int(frame[0, 0, 0]) for frame in frames.copy())
    assert got == [f % 251 for f in range(num_frames)]


def _per_rank_stream_worker(rank: int, world_size: int, port: int, num_frames: int, per_rank: int, results) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        device = torch.device("cpu")
        all_inputs = torch.arange(num_frames, dtype=torch.float32).reshape(num_frames, 1) if rank == 0 else None
        mine = frame_sharding.scatter_for_stream(all_inputs, num_frames, per_rank, device)

        class Producer:  # pylint: disable=too-few-public-methods
            writes_into = True

            def __call__(self, offset: int, count: int, out: torch.Tensor) -> None:
                values = (mine[offset : offset + count, 0].to(torch.int64) % 251).to(torch.uint8)
                out.copy_(values.reshape(count, 1, 1, 1).expand(count, 2, 2, 3))

        got = {}
        firsts = []
        for first, frames in frame_sharding.ordered_frame_stream(Producer(), num_frames, per_rank, (2, 2, 3), device, drain="per-rank"):
            firsts.append(first)
            for i, frame in enumerate(frames):
                assert bool((frame == frame[0, 0, 0]).all())
                got[first + i] = int(frame[0, 0, 0])
        assert firsts == sorted(firsts)
        # every rank received exactly the frames of its own stream order, each with its own value
        assert sorted(got) == frame_sharding.stream_order(num_frames, world_size, per_rank, rank)
        assert all(value == index % 251 for index, value in got.items())
        results[rank] = sorted(got)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,num_frames,per_rank", [(2, 37, 4), (3, 50, 3), (2, 3, 4)])
def test_per_rank_drain_hands_every_rank_its_own_pieces(world_size: int, num_frames: int, per_rank: int) -> None:
    """`drain="per-rank"`: no gather; every rank drains the pieces it synthesised; together they are every frame once."""
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_per_rank_stream_worker, args=(world_size, _free_port(), num_frames, per_rank, results), nprocs=world_size, join=True)
    assert sorted(sum((list(results[rank]) for rank in range(world_size)), [])) == list(range(num_frames))


def _failing_consumer_worker(rank: int, world_size: int, port: int, results) -> None:
    import datetime  # pylint: disable=import-outside-toplevel
    import time  # pylint: disable=import-outside-toplevel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size, timeout=datetime.timedelta(seconds=60))
    try:
        def synthesize_piece(offset: int, count: int) -> torch.Tensor:
            return torch.zeros((count, 2, 2, 3), dtype=torch.uint8)

        outcome = "finished"
        start = time.perf_counter()
        try:
            for index, _ in enumerate(frame_sharding.ordered_frame_stream(synthesize_piece, 40, 4, (2, 2, 3), torch.device("cpu"))):
                if index == 1:
                    raise KeyError("the consumer of rank 0 failed")  # (the overlay stage, the writer, ...)
        except frame_sharding.StreamRankError as error:
            outcome = f"relayed: {error}"
        except KeyError:
            outcome = "own"
        results[rank] = (outcome, time.perf_counter() - start)
    finally:
        dist.destroy_process_group()


def test_a_failing_consumer_on_rank_0_ends_the_stream_on_the_other_ranks_promptly() -> None:
    """Rank 0's consumer raises between two chunks: the other rank leaves in its next status exchange, not after the group's timeout."""
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_failing_consumer_worker, args=(2, _free_port(), results), nprocs=2, join=True)
    assert results[0][0] == "own", dict(results)
    assert results[1][0].startswith("relayed: rank 0 failed consuming the stream") and results[1][1] < 30.0, dict(results)


def _forced_collectives_worker(rank: int, world_size: int, port: int, results) -> None:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GANCE_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        assert frame_sharding.collectives_forced() and not frame_sharding.single_process()
        assert frame_sharding.control_group() is not None
        device = torch.device("cpu")
        num_frames, per_rank = 11, 4
        mine = frame_sharding.scatter_for_stream(torch.arange(num_frames, dtype=torch.float32).reshape(num_frames, 1), num_frames, per_rank, device)
        assert mine[:, 0].tolist() == [float(f) for f in range(num_frames)]
        calls = {"gather": 0}
        real_gather = dist.gather

        def counting_gather(*args, **kwargs):
            calls["gather"] += 1
            return real_gather(*args, **kwargs)

        dist.gather = counting_gather
        try:
            class Producer:  # pylint: disable=too-few-public-methods
                writes_into = True

                def __call__(self, offset: int, count: int, out: torch.Tensor) -> None:
                    out.copy_(mine[offset : offset + count, 0].to(torch.uint8).reshape(count, 1, 1, 1).expand(count, 2, 2, 3))

            got = []
            for first, frames in frame_sharding.ordered_frame_stream(Producer(), num_frames, per_rank, (2, 2, 3), device):
                assert first == len(got)
                got.extend(int(frame[0, 0, 0]) for frame in frames.copy())
        finally:
            dist.gather = real_gather
        assert got == list(range(num_frames))
        latents = frame_sharding.scatter_latents(torch.arange(6, dtype=torch.float32).reshape(6, 1), 6, device)
        gathered, _ = frame_sharding.gather_frames(torch.zeros((6, 2, 2, 3), dtype=torch.uint8), 6)
        results[rank] = (calls["gather"], latents[:, 0].tolist(), tuple(gathered.shape))
    finally:
        dist.destroy_process_group()
        os.environ.pop("GANCE_FORCE_COLLECTIVES", None)


def test_one_rank_with_the_collectives_forced_runs_them() -> None:
    """
    GANCE_FORCE_COLLECTIVES=1 with a process group of one rank: scatter, one gather per chunk and the status exchange go
    through torch.distributed (how the GPU suite runs the RCCL path on a one-GPU box); same frames as the short cut.
    """
    manager = mp.Manager()
    results = manager.dict()
    mp.spawn(_forced_collectives_worker, args=(1, _free_port(), results), nprocs=1, join=True)
    assert results[0] == (3, [0.0, 1.0, 2.0, 3.0, 4.0, 5.0], (6, 2, 2, 3))
