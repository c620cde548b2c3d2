"""
Stand-alone rehearsal of the PRODUCT's multi-rank path on a one-GPU box: `projection_file_blend_frame_chunks`
with WORLD_SIZE ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device), compared bit for bit with
the same job on one rank. Launch from the shell, never from a process that has touched the GPU:

    python tools/rehearse_stream_ranks.py --prepare /tmp/rehearsal              # inputs + the 1-rank result
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        tools/rehearse_stream_ranks.py --check /tmp/rehearsal                    # 2 ranks, compare
    ... --check /tmp/rehearsal --drain per-rank    # no gather: every rank writes its own pieces into one memory-mapped
                                                   # .npy through projection_file_blend_api (not with the overlay)

The numbers of such a run mean nothing (the ranks share one GPU); it only shows that the scatter of per-chunk
latent pieces, the chunked synthesis with resident networks, the ordered gather and the host drain give the
frames of the single-rank run.
"""

import argparse
import datetime
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

L, NUM_PROJECTION, FPS_IN, FPS_OUT, NETWORKS = 512, 45, 15.0, 30.0, 3
# GANCE_REHEARSAL="side,out_side,frames_per_call,overlay": e.g. "1024,2160,8,0" (configs[3]'s shape) or "128,200,8,1"
# (three networks + the streaming overlay gate: configs[4]'s shape); default 128 -> 200, 8 frames per call, no overlay
SIDE, OUT_SIDE, PER_CALL, OVERLAY = (int(v) for v in os.environ.get("GANCE_REHEARSAL", "128,200,8,0").split(","))


class BrightPicturesHaveAFace:  # pylint: disable=too-few-public-methods
    """Stand-in landmark detector (the real one is external): a picture with a bright first pixel has one face."""

    @staticmethod
    def face_landmarks(face_image):
        if int(face_image[0, 0].sum()) < 300:
            return []
        scale = face_image.shape[0] / 128.0
        return [{
            "left_eye": ((int(30 * scale), int(40 * scale)), (int(50 * scale), int(52 * scale))),
            "right_eye": ((int(70 * scale), int(41 * scale)), (int(95 * scale), int(55 * scale))),
        }]


def job_per_rank(directory: Path):
    """The API function with drain="per-rank": every rank writes the pieces it synthesised into one memory-mapped file."""
    import torch  # pylint: disable=import-outside-toplevel
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel

    from gance_amd import projection_file_blend  # pylint: disable=import-outside-toplevel

    torch.cuda.set_device(0)
    output = directory / "per_rank_drain.npy"
    if dist.get_rank() == 0 and output.exists():
        output.unlink()
    dist.barrier()
    projection_file_blend.projection_file_blend_api(
        wav=[str(directory / "audio.wav")], output_path=str(output), network_paths=[directory / f"net_{i}.pkl" for i in range(NETWORKS)],
        frames_to_visualize=None, output_fps=FPS_OUT, output_side_length=OUT_SIDE, debug_path=None, debug_window=None, debug_side_length=None,
        alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5), projection_file_path=str(directory / "projection.npz"), blend_depth=12,
        complexity_change_rolling_sum_window=None, complexity_change_threshold=None, phash_distance=None, bbox_distance=None, track_length=None,
        drain="per-rank",
    )
    dist.barrier()  # (every rank has flushed its pieces)
    if dist.get_rank() != 0:
        return None, []
    return np.load(output), [0]


def job(directory: Path):
    import torch  # pylint: disable=import-outside-toplevel

    from gance_amd import projection_file_blend  # pylint: disable=import-outside-toplevel

    torch.cuda.set_device(0)
    chunks = projection_file_blend.projection_file_blend_frame_chunks(
        wav=[str(directory / "audio.wav")], network_paths=[directory / f"net_{i}.pkl" for i in range(NETWORKS)],
        frames_to_visualize=None, output_fps=FPS_OUT, output_side_length=OUT_SIDE, alpha=0.25, fft_roll_enabled=True,
        fft_amplitude_range=(-5, 5), projection_file_path=str(directory / "projection.npz"), blend_depth=12,
        frames_per_call=PER_CALL,
        overlay=projection_file_blend.OverlayParameters(phash_distance=64, bbox_distance=5.0, track_length=4, face_finder=BrightPicturesHaveAFace())
        if OVERLAY
        else None,
    )
    collected, firsts = None, []
    for first, total, frames in chunks:
        if collected is None:
            collected = np.empty((total, *frames.shape[1:]), dtype=np.uint8)
        collected[first : first + len(frames)] = frames
        firsts.append(first)
    return collected, firsts


def main() -> int:
    parser = argparse.ArgumentParser()
    parser.add_argument("--prepare", type=Path)
    parser.add_argument("--check", type=Path)
    parser.add_argument("--drain", choices=["rank0", "per-rank"], default="rank0")
    args = parser.parse_args()
    if args.prepare is not None:
        from scipy.io import wavfile  # pylint: disable=import-outside-toplevel

        from gance_amd import network_file, synthetic  # pylint: disable=import-outside-toplevel
        from gance_amd.projection import projection_file_reader as pfr  # pylint: disable=import-outside-toplevel

        directory = args.prepare
        directory.mkdir(parents=True, exist_ok=True)
        num_frames = int(NUM_PROJECTION * FPS_OUT / FPS_IN)
        wavfile.write(str(directory / "audio.wav"), int(L * FPS_OUT), synthetic.synthetic_audio(num_frames, L, seed=61, frames_per_second=FPS_OUT))
        latents = synthetic.synthetic_final_latents(NUM_PROJECTION, L, seed=62)
        targets = (np.kron(np.random.RandomState(63).rand(NUM_PROJECTION, 8, 8, 3), np.ones((1, 16, 16, 1))) * 255).astype(np.uint8)
        pfr.write_projection_npz(
            directory / "projection.npz", latents.reshape(18, NUM_PROJECTION, L).transpose(1, 0, 2), projection_fps=FPS_IN, target_images=targets
        )
        for seed in range(NETWORKS):
            network_file.write_random_network(directory / f"net_{seed}.pkl", SIDE, seed=seed)
        frames, firsts = job(directory)
        np.save(directory / "single_rank.npy", frames)
        print(f"single rank: {frames.shape[0]} frames {frames.shape[1:]} in {len(firsts)} chunks, checksum {int(frames.astype(np.uint64).sum())}")
        return 0

    import torch.distributed as dist  # pylint: disable=import-outside-toplevel

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
    rank, world_size = dist.get_rank(), dist.get_world_size()
    start = time.perf_counter()
    frames, firsts = job_per_rank(args.check) if args.drain == "per-rank" else job(args.check)
    elapsed = time.perf_counter() - start
    status = 0
    if rank == 0:
        want = np.load(args.check / "single_rank.npy")
        # (the ranks' engine calls group the frames differently -- other batch sizes pick other split-K factors / kernel
        # forms on the small layers --, so frames may differ by fp32 rounding: at most 1 LSB on a few pixels; with an
        # overlay the written regions are copies of the target pictures and must sit on the same frames)
        same = frames is not None and frames.shape == want.shape
        differing, worst = 0.0, 0
        if same:
            diff = np.abs(frames.astype(np.int16) - want.astype(np.int16))
            differing, worst = float((diff > 0).mean()), int(diff.max())
            same = worst <= (2 if OUT_SIDE != SIDE else 1) and differing < 1e-3  # (the bicubic resize can double a 1-LSB difference)
        ordered = firsts == sorted(firsts) and firsts[0] == 0
        print(
            f"{world_size} ranks on one GPU over gloo, drain {args.drain}, {SIDE}^2 -> {OUT_SIDE}^2, {NETWORKS} networks, {PER_CALL} frames per call, overlay {bool(OVERLAY)}: "
            f"{0 if frames is None else frames.shape[0]} frames in {len(firsts)} ordered chunks "
            f"({elapsed:.2f} s incl. network loading), same frames as the single-rank run: {same} (max |diff| {worst} LSB on "
            f"{100.0 * differing:.4f} % of the values), chunk order ok: {ordered}"
        )
        status = 0 if same and ordered else 1
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
