"""
Stand-alone rehearsal of the PRODUCT's multi-rank path on a one-GPU box: `projection_file_blend_frame_chunks`
with WORLD_SIZE ranks sharing cuda:0 over gloo (RCCL refuses two ranks on one device), compared bit for bit with
the same job on one rank. Launch from the shell, never from a process that has touched the GPU:

    python tools/rehearse_stream_ranks.py --prepare /tmp/rehearsal              # inputs + the 1-rank result
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
        tools/rehearse_stream_ranks.py --check /tmp/rehearsal                    # 2 ranks, compare

The numbers of such a run mean nothing (the ranks share one GPU); it only shows that the scatter of per-chunk
latent pieces, the chunked synthesis with resident networks, the ordered gather and the host drain give the
frames of the single-rank run.
"""

import argparse
import os
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

L, NUM_PROJECTION, FPS_IN, FPS_OUT, SIDE, OUT_SIDE, NETWORKS, PER_CALL = 512, 45, 15.0, 30.0, 128, 200, 3, 8


def job(directory: Path):
    import torch  # pylint: disable=import-outside-toplevel

    from gance_amd import projection_file_blend  # pylint: disable=import-outside-toplevel

    torch.cuda.set_device(0)
    chunks = projection_file_blend.projection_file_blend_frame_chunks(
        wav=[str(directory / "audio.wav")], network_paths=[directory / f"net_{i}.pkl" for i in range(NETWORKS)],
        frames_to_visualize=None, output_fps=FPS_OUT, output_side_length=OUT_SIDE, alpha=0.25, fft_roll_enabled=True,
        fft_amplitude_range=(-5, 5), projection_file_path=str(directory / "projection.npz"), blend_depth=12,
        frames_per_call=PER_CALL,
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
        pfr.write_projection_npz(directory / "projection.npz", latents.reshape(18, NUM_PROJECTION, L).transpose(1, 0, 2), projection_fps=FPS_IN)
        for seed in range(NETWORKS):
            network_file.write_random_network(directory / f"net_{seed}.pkl", SIDE, seed=seed)
        frames, firsts = job(directory)
        np.save(directory / "single_rank.npy", frames)
        print(f"single rank: {frames.shape[0]} frames {frames.shape[1:]} in {len(firsts)} chunks, checksum {int(frames.astype(np.uint64).sum())}")
        return 0

    import torch.distributed as dist  # pylint: disable=import-outside-toplevel

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend="gloo")
    rank, world_size = dist.get_rank(), dist.get_world_size()
    start = time.perf_counter()
    frames, firsts = job(args.check)
    elapsed = time.perf_counter() - start
    status = 0
    if rank == 0:
        want = np.load(args.check / "single_rank.npy")
        same = frames is not None and frames.shape == want.shape and np.array_equal(frames, want)
        ordered = firsts == sorted(firsts) and firsts[0] == 0
        print(
            f"{world_size} ranks on one GPU over gloo: {0 if frames is None else frames.shape[0]} frames in {len(firsts)} ordered chunks "
            f"({elapsed:.2f} s incl. network loading), identical to the single-rank run: {same}, chunk order ok: {ordered}"
        )
        status = 0 if same and ordered else 1
    else:
        assert frames is None
    dist.barrier()
    dist.destroy_process_group()
    return status


if __name__ == "__main__":
    sys.exit(main())
