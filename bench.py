"""
Benchmark of the hot path on MI355X: synthesized frames/sec at 1024x1024 config-f.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[1]: FFHQ config-f 1024x1024, random-init weights (seed 0),
batched random-z synthesis. One step = one batch of `--batch` z vectors (already resident in HBM)
-> mapping -> truncation (psi 1.2) -> synthesis -> uint8 NHWC frames in HBM, through the C ABI of
libgance_hip.so. With N > 1 every rank runs its own batches (frames shard embarrassingly, weak
scaling) and the finished frames are gathered into rank 0 over RCCL, in frame order, overlapped
with the next step.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (modulated conv on the fp32 matrix cores) timed with HIP
                  events on the launch stream during the timed region
  cpu_baseline -- the CPU oracle (oracle/stylegan2_ref.py, torch fp32, all host cores) on a
                  bounded sample of the same workload; rank 0, N = 1 only. Baseline, not target.
"""

import argparse
import json
import re
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO_ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO_ROOT))

from gance_amd import frame_sharding, hip_lib  # noqa: E402
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402

import datetime  # noqa: E402

# a rank that dies must not leave the others in a collective for ever (default: 10 minutes for RCCL, 30 for gloo)
PROCESS_GROUP_TIMEOUT = datetime.timedelta(seconds=int(os.environ.get("GANCE_PROCESS_GROUP_TIMEOUT_S", "180")))
FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
BF16_MFMA_PEAK_TFLOPS = 2516.6  # the same guide: 512 MACs per clock and SIMD x 1024 SIMDs x 2.4 GHz ("~2.5 PF dense"; v_mfma_f32_16x16x32_bf16: 8192 MACs in 16 cycles)
SPLIT_DTYPE = "f32 (bf16x3 split operands, 6 terms, fp32 accumulate)"
HBM_PEAK_GBS = 8000.0
ALGORITHMIC_GFLOP_PER_FRAME_1024 = 148.5


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, cores)


SPLIT_SUFFIXES = ("/s3", "/s3r")  # launches whose products run on the bf16 matrix cores from split operands (engine.hip step names)


def kernel_of_step(step_name: str) -> str:
    """The HIP kernel behind a conv launch of the engine's step table (names: engine.hip)."""
    if step_name.startswith("convTF"):  # ("p": input pre-scaled by the Winograd launch before it; "/16": the 16-channel geometry)
        pre = "_pre" if step_name.startswith("convTFp") else ""
        if step_name.endswith("/s3r"):  # the split-operand form in two wave roles (upfir_split_roles.hip): matrix waves + vector waves
            return f"upfirr_fused{pre}_kernel"
        if step_name.endswith("/s3"):  # the split-operand form (upfir_split.hip): bf16 x 3 parts, six product terms, fp32 accumulation
            return f"upfirs_fused{pre}_kernel"
        if not step_name.endswith(("/16", "/16x")):
            return f"upfir_fused{pre}_kernel"
        side = int(re.search(r"_(\d+)x\1_", step_name).group(1)) // 2  # the layer's INPUT width picks the strip geometry
        geometry = "" if side % 64 == 0 else ("_w32" if side == 32 else "_w16")
        if step_name.endswith("/16x"):
            return f"upfir16x_fused{geometry}_pre_kernel"  # the pair form (F(2,2) along x)
        return f"upfir16_fused{geometry}{pre}_kernel"  # (the launches of a network with noise: ..._noise_kernel)
    if step_name.startswith(("convTG", "convVG")):  # (convVG: Winograd F(4x4,3x3) in GEMM form at 8x8 / 16x16: the same GEMM kernel + its transforms)
        return "tile_gemm_kernel"  # the scatter form of the two smallest up layers (gemm_forms.hip; + its pack and gather kernels)
    if step_name.startswith("convV"):
        narrow = "_32x32_" in step_name  # the 32 x 32 pixel geometry
        return ("winograd43_w32" if narrow else "winograd43") + ("_rgb_kernel" if "+rgb" in step_name else "_kernel")
    if step_name.startswith("convW"):
        narrow = step_name.endswith("->32")
        if "+rgb" in step_name:
            return "winograd64_c32_rgb_kernel" if narrow else "winograd64_rgb_kernel"
        return "winograd64_c32_kernel" if narrow else "winograd64_kernel"
    return "modconv_mfma_kernel"


def executed_fraction(step_name: str) -> float:
    """
    Matrix-core flops a launch EXECUTES per algorithmic (direct-form) flop: Winograd F(2x2,3x3) does 16 of 36, F(4x4,3x3) 36 of 144,
    the pair form of the fused up kernel (F(2,2) along x, launch names ending in "/16x") 15 of 18.
    """
    if step_name.startswith("convV"):
        return 0.25
    if step_name.endswith("/16x"):
        return 15.0 / 18.0
    if step_name.endswith(SPLIT_SUFFIXES):
        return 6.0  # six bf16 x bf16 part products per fp32 product, on the bf16 matrix cores (matrix_peak: 2516.6 TFLOP/s)
    return 4.0 / 9.0 if step_name.startswith("convW") else 1.0


def matrix_peak(step_name: str) -> float:
    """Dense peak of the matrix pipe a launch's products execute on: bf16 MFMA for the split-operand form, fp32 MFMA otherwise."""
    return BF16_MFMA_PEAK_TFLOPS if step_name.endswith(SPLIT_SUFFIXES) else FP32_MFMA_PEAK_TFLOPS


def matrix_pipe(step_name: str) -> str:
    """Name of that pipe, for the roofline record."""
    return "bf16 MFMA (fp32 products from three bf16 parts per operand, six part products each, fp32 accumulation)" if step_name.endswith(SPLIT_SUFFIXES) else "fp32 MFMA"


def kernel_sources_digest() -> str:
    """sha256 over the HIP sources: a traffic record taken on other kernels is reported as stale."""
    import hashlib  # pylint: disable=import-outside-toplevel

    digest = hashlib.sha256()
    for path in sorted((REPO_ROOT / "gance_amd" / "csrc").glob("*.hip")) + [REPO_ROOT / "gance_amd" / "csrc" / "kernels.h"]:
        digest.update(path.read_bytes())
    return digest.hexdigest()[:16]


def measured_traffic(step_name: str, resolution: int, batch: int):
    """
    HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 cannot run
    inside this process): profiles/traffic_latest.json, written by tools/make_traffic_record.py from the
    round's --pmc CSVs; only if it was taken on this workload and holds this launch (the two largest launches
    trade places from run to run: both are recorded). Returns (bytes or None, note).
    """
    path = REPO_ROOT / "profiles" / "traffic_latest.json"
    try:
        record = json.loads(path.read_text())
    except (OSError, ValueError):
        return None, "no profiles/traffic_latest.json"
    workload = record.get("workload", {})
    if workload.get("resolution") != resolution or workload.get("frames_per_step_per_gpu") != batch:
        return None, "profiles/traffic_latest.json was taken on another workload"
    stale = record.get("kernel_sources_digest") not in (None, kernel_sources_digest())
    lookup = step_name.replace("convTFp", "convTF", 1)  # (the record keys both forms of the fused up kernel as convTF)
    for prefix, entry in record.get("launches", {}).items():
        if lookup.startswith(prefix):
            note = "from %s" % record.get("source", "profiles/traffic_latest.json")
            return entry.get("hbm_bytes_per_launch"), note + (" -- STALE: the kernels changed since that PMC pass" if stale else "")
    return None, "profiles/traffic_latest.json does not hold this launch"


def cpu_baseline(resolution: int, variables, budget_seconds: float = 15.0) -> dict:
    """Time the CPU oracle (checker infrastructure, used here only as the reported baseline)."""
    from oracle import stylegan2_ref  # pylint: disable=import-outside-toplevel

    cores = usable_cores()
    torch.set_num_threads(cores)
    spec = sg2_spec.make_spec(resolution)
    rng = np.random.RandomState(1)
    dlatents = rng.randn(1, spec.num_layers, 512).astype(np.float32)
    stylegan2_ref.synthesize_w(dlatents, variables, resolution, dtype=torch.float32)  # warm-up
    frames = 0
    start = time.perf_counter()
    while True:
        image = stylegan2_ref.synthesize_w(dlatents, variables, resolution, dtype=torch.float32)
        stylegan2_ref.convert_images_to_uint8(image)
        frames += 1
        elapsed = time.perf_counter() - start
        if elapsed > budget_seconds or frames >= 10:
            break
    return {
        "value": round(frames / elapsed, 4),
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{frames} frames of the same {resolution}x{resolution} config-f network, batch 1, torch fp32 CPU oracle, dlatent (synthesis-only) input",
    }


class SyntheticFaceFinder:  # pylint: disable=too-few-public-methods
    """
    Stand-in for the landmark detector of the overlay gate (face_recognition / dlib is external and absent: SURVEY.md
    §8 f-4 has the benchmark feed synthetic boxes): every picture whose top-left pixel is bright enough "has a face"
    with the same two eye boxes. Costs what a table look-up costs; the gate's GPU work (phash of both eye regions,
    the overlay write) and the run-length filter are the real ones.
    """

    def __init__(self, side: int) -> None:
        scale = side / 128.0
        self._landmarks = [{
            "left_eye": ((int(30 * scale), int(40 * scale)), (int(50 * scale), int(52 * scale))),
            "right_eye": ((int(70 * scale), int(41 * scale)), (int(95 * scale), int(55 * scale))),
        }]

    def face_landmarks(self, face_image):
        """One face, unless the picture's first pixel is dark."""
        return self._landmarks if int(face_image[0, 0].sum()) >= 96 else []


def product_stream_measurement(  # pylint: disable=too-many-arguments,too-many-locals,too-many-statements
    label: str, resolution: int, batch: int, num_networks: int, output_side, device, with_cpu_baseline: bool, overlay: bool = False,
    drain: str = "rank0",
) -> dict:
    """
    BASELINE.json configs[2] / [3] / [4] THROUGH THE PRODUCT, on one GPU or -- under torch.distributed.run, one rank per GPU:
    collective, every rank calls this -- frame-sharded over the node: a 30 s synthetic WAV and a projection file of
    900 projected latents ON DISK -> `projection_file_blend_frame_chunks` (read + stretch the WAV, audio -> latents on
    the GPU, chunked synthesis with resident networks, optional bicubic resize / overlay gate in HBM, ordered chunks
    drained to the pinned host ring) -> every chunk consumed on the host. Timed: the call -> the last uint8 frame on
    the host; the networks are resident before the call (the reference loads them before
    its frame loop too, projection_file_blend.py:122). A first, short run warms the process (operator tables,
    kernel attributes, pinned ring) and names the last conv launch; the second, full run is the one reported.
    With N ranks a chunk is N x `batch` frames; rank 0 alone writes the input files (the others read them: one node, one
    file system), prints and returns the record (the other ranks return {}); `drain` as in
    `projection_file_blend_frame_chunks` ("per-rank": every rank consumes its own pieces on its own host link; the time
    reported is then the slowest rank's).
    """
    import tempfile  # pylint: disable=import-outside-toplevel

    from scipy.io import wavfile  # pylint: disable=import-outside-toplevel

    from gance_amd import network_file, projection_file_blend, synthetic  # pylint: disable=import-outside-toplevel
    from gance_amd.network_interface.network_functions import MultiNetwork  # pylint: disable=import-outside-toplevel
    from gance_amd.projection import projection_file_reader as pfr  # pylint: disable=import-outside-toplevel

    num_frames, vector_length, fps_in, fps_out = 1800, 512, 30.0, 60.0
    side = output_side or resolution
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    holder = tempfile.TemporaryDirectory(prefix="gance_bench_") if rank == 0 else None  # pylint: disable=consider-using-with
    shared = [holder.name if holder is not None else None]
    if world_size > 1:
        dist.broadcast_object_list(shared, src=0)
    try:
        directory = Path(shared[0])
        wav_path = directory / "audio.wav"
        projection_path = directory / "projection.npz"
        network_paths = [directory / f"net_{seed}.pkl" for seed in range(num_networks)]
        audio = latents = None
        if rank == 0:
            audio, latents = synthetic.benchmark_blend_inputs(num_frames)
            wavfile.write(str(wav_path), int(vector_length * fps_out), audio)
            targets = None
            if overlay:  # target frames of the projection: blocky pictures at 128^2, the gate scales them to the output side
                rng = np.random.RandomState(73)
                targets = (np.kron(rng.rand(num_frames // 2, 8, 8, 3), np.ones((1, 16, 16, 1))) * 255).astype(np.uint8)
            pfr.write_projection_npz(
                projection_path, latents.reshape(18, num_frames // 2, vector_length).transpose(1, 0, 2), projection_fps=fps_in, target_images=targets
            )
            for seed, path in enumerate(network_paths):
                network_file.write_random_network(path, resolution, seed=seed)
        if world_size > 1:
            dist.barrier()  # (the files exist)
        networks = MultiNetwork(network_paths=network_paths, load=True, max_batch=batch, device=device.index)
        engines = [networks._network_at(index).engine for index in range(num_networks)]  # pylint: disable=protected-access
        overlay_parameters = (
            projection_file_blend.OverlayParameters(phash_distance=64, bbox_distance=5.0, track_length=5, face_finder=SyntheticFaceFinder(side))
            if overlay
            else None
        )
        last_conv_pattern = re.compile(r"^conv[A-Z]*%d[+_].*_%dx%d_" % (2 * int(np.log2(resolution)) - 4, resolution, resolution))

        def run(frames_to_visualize, timings):
            if world_size > 1:
                torch.cuda.synchronize(device)
                dist.barrier()
            start = time.perf_counter()
            checksum, chunks, last_index, received = 0, 0, -1, 0
            for first, _total, frames in projection_file_blend.projection_file_blend_frame_chunks(
                wav=[str(wav_path)], network_paths=network_paths, frames_to_visualize=frames_to_visualize, output_fps=fps_out,
                output_side_length=side, alpha=0.25, fft_roll_enabled=True, fft_amplitude_range=(-5, 5),
                projection_file_path=str(projection_path), blend_depth=12, frames_per_call=batch, overlay=overlay_parameters,
                networks=networks, timings=timings, drain=drain,
            ):
                assert first > last_index  # (rank 0 drain: consecutive chunks; per-rank drain: this rank's pieces, ascending)
                last_index = first + len(frames) - 1
                received += len(frames)
                checksum += int(frames[-1, -1, -1, 0]) + int(frames[0, 0, 0, 0])  # the chunk is on the host: touch both ends
                chunks += 1
            elapsed = time.perf_counter() - start
            if world_size > 1:  # the job is done when its slowest rank is; frames received over all ranks
                totals = torch.tensor([elapsed, float(received)], dtype=torch.float64, device=device)
                slowest = totals.clone()
                dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
                dist.all_reduce(totals, op=dist.ReduceOp.SUM)
                elapsed, received = float(slowest[0].item()), int(totals[1].item())
            return elapsed, received, chunks, checksum

        try:
            for engine in engines:
                engine.set_profiling(True)
            run(4 * batch * world_size, {})  # warm-up, every launch bracketed: it names the last conv launch
            last_conv = next((s.name for s in engines[0].steps() if last_conv_pattern.match(s.name)), "conv")
            for engine in engines:
                engine.set_profiling(True, only_step=last_conv)
            timings: dict = {}
            elapsed, produced, chunks, _ = run(None, timings)
            assert produced == num_frames, (produced, num_frames)
            timed = [s for engine in engines for s in engine.steps() if s.name.startswith(last_conv)]
        finally:
            networks.unload()
    finally:
        if world_size > 1:
            dist.barrier()  # (nobody still reads the files)
        if holder is not None:
            holder.cleanup()
    if rank != 0:
        return {}
    split = timings.get("audio_to_latents_split_ms", {})
    audio_ms = float(timings.get("audio_to_latents_ms", 0.0))
    timed_flops, timed_ms = sum(s.flops for s in timed), sum(s.ms for s in timed)
    # audio -> latents: algorithmic bytes = the samples read once + the per-frame latent rows written once
    # (two distinct rows per frame) + the [N][L] float64 spectrogram written and read once per stage (5 stages)
    audio_bytes = num_frames * 512 * 4 + num_frames * 18 * 512 * 4 + 5 * 2 * num_frames * 512 * 8
    kernels_ms = float(split.get("kernels_ms", 0.0)) or audio_ms
    result = {
        "metric": label,
        "value": round(num_frames / elapsed, 3), "unit": "frames/s", "n_gpus": world_size,
        "timed": "WAV + projection file on disk -> last uint8 frame on the host, through projection_file_blend_frame_chunks (networks resident)",
        "drain": drain + (" (RCCL gather of every chunk to rank 0, which drains it to its pinned ring)" if drain == "rank0" else " (no gather: every rank drains its own pieces over its own PCIe link)"),
        "frames": num_frames, "chunks_on_rank_0": chunks, "frames_per_call": batch, "frames_per_chunk": batch * world_size, "seconds": round(elapsed, 4),
        "read_projection_file_ms": round(float(timings.get("read_projection_file_ms", 0.0)), 3),
        "read_and_stretch_wav_ms": round(float(timings.get("read_and_stretch_wav_ms", 0.0)), 3),
        "audio_to_latents_ms": round(audio_ms, 3),
        "audio_to_latents_split_ms": {key: round(float(value), 3) for key, value in split.items()},
        "synthesis_to_host_ms": round(float(timings.get("synthesis_to_host_ms", 0.0)), 3),
        "d2h_gb_per_s": round(float(timings.get("d2h_gb_per_s") or 0.0), 3),
        "networks_resident": num_networks, "output_side_length": side,
        "dtype": "f64 (audio) / f32 (synthesis)", "data": "synthetic",
        "roofline": {
            "bound": "mfma",
            "kernel": "%s (%s)" % (kernel_of_step(last_conv), last_conv),
            "launches_averaged": len(timed),
            # flops-weighted over the launches (the tail chunk and ragged multi-network windows issue smaller launches of the same
            # layer: sum of flops over sum of durations, not one launch's flops over the mean duration). `achieved` / `frac` = the flops
            # the matrix cores EXECUTE (a Winograd F(4x4,3x3) launch: 1/4 of the layer's direct-form flops), always <= 1 of the peak;
            # the layer priced as a direct convolution stands beside it as `*_direct_form` (it may exceed the peak).
            "achieved": round(executed_fraction(last_conv) * timed_flops / (timed_ms * 1e-3) / 1e12, 3) if timed else None,
            "peak": matrix_peak(last_conv), "unit": "TFLOP/s", "pipe": matrix_pipe(last_conv),
            "frac": round(executed_fraction(last_conv) * timed_flops / (timed_ms * 1e-3) / 1e12 / matrix_peak(last_conv), 4) if timed else None,
            "achieved_direct_form": round(timed_flops / (timed_ms * 1e-3) / 1e12, 3) if timed else None,
            "frac_direct_form": round(timed_flops / (timed_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if timed else None,
            "audio_stage": {
                "bound": "hbm (launch-latency in practice: six kernels over < 60 MB; rocprofv3 per-kernel times: profiles/r03_audio_kernel_stats.csv)",
                "algorithmic_bytes": audio_bytes,
                "achieved": round(audio_bytes / (kernels_ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s over the six kernels (stream drained)",
                "frac": round(audio_bytes / (kernels_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
            },
        },
    }
    if overlay:
        result["overlay"] = {
            "gate": "synthetic landmark detector (the real one is external: dlib), real phash + box distance + track_length 5 + overlay write",
            "chunks_held_max": timings.get("overlay_chunks_held_max"), "overlays_written": timings.get("overlays_written"),
        }
    if with_cpu_baseline:
        from oracle import audio_ref  # pylint: disable=import-outside-toplevel

        t0 = time.perf_counter()
        audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, audio, 512, list(range(num_networks)))
        cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(cpu_s * 1e3, 1), "unit": "ms for audio -> latents of the same 1800 frames (lower is better)", "cores": 1, "kind": "port",
            "sample": "oracle/audio_ref.alpha_blend_projection_file (numpy / scipy restatement of the reference's chain, its Python loops vectorised), the whole 30 s workload once, single thread; the GPU stage beside it: %.1f ms" % audio_ms,
        }
    return result


EXTRAS_WATCHDOG_S = int(os.environ.get("GANCE_BENCH_EXTRAS_WATCHDOG_S", "240"))


def rehearsal_skips_extras() -> bool:
    """GANCE_BENCH_REHEARSAL_EXTRAS=0: a one-GPU rehearsal of the N-rank line without the N-rank extras."""
    return os.environ.get("GANCE_BENCH_REHEARSAL_EXTRAS", "1") == "0"


def guarded(measure, *args, **kwargs) -> dict:
    """An extra measurement must never cost the contract line: its failure is recorded in its slot."""
    try:
        return measure(*args, **kwargs)
    except Exception as error:  # pylint: disable=broad-except
        return {"error": f"{type(error).__name__}: {error}"}


def one_frame_latency(resolution: int, device) -> dict:
    """The reference's call pattern: one frame per call through the host-buffer entries (numpy in, numpy out, PCIe included)."""
    engine = hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=0), resolution, max_batch=1, device=device.index)
    rng = np.random.RandomState(0)
    out = {}
    for name, make, call in (
        ("create_image_matrix", lambda: rng.randn(1, engine.num_layers, 512).astype(np.float32), engine.synthesize_w),
        ("create_image_vector", lambda: rng.randn(1, 512).astype(np.float32), engine.synthesize_z),
    ):
        for _ in range(4):
            call(make())
        times = []
        for _ in range(40):
            data = make()
            start = time.perf_counter()
            call(data)
            times.append(time.perf_counter() - start)
        out[name] = {"median_ms": round(1e3 * float(np.median(times)), 4), "min_ms": round(1e3 * min(times), 4), "frames_per_s": round(1.0 / float(np.median(times)), 2)}
    engine.close()
    out["note"] = "one %dx%d frame per call, 74 KB of latents in, 3 MiB of pixels out over PCIe, synchronous; launch sequence replayed from a hipGraph" % (resolution, resolution)
    return out


def batch_sweep(resolution: int, variables, device, batches, steps: int = 20, warmup: int = 3) -> dict:
    """
    SURVEY.md section 8(d) config 2 / BASELINE.md section 5: synthesis only, z vectors resident in HBM, frames stay in HBM, at
    every batch size of `batches`: HIP-event timed (events on the launch stream), `warmup` + `steps` engine calls each. One
    engine with the largest batch as capacity (as the product has); per batch the launch name of every Conv0_up / Conv1
    layer (the form the engine chose: engine.hip conv_form_of / up_runs_fused) and, at one frame per call, the three
    slowest launches.
    """
    batches = sorted(set(int(b) for b in batches))
    engine = hip_lib.Engine(variables, resolution, max_batch=max(batches), device=device.index, profile=False)
    stream = torch.cuda.current_stream(device)
    rows = {}
    try:
        for batch in batches:
            z = torch.from_numpy(np.random.RandomState(batch).randn(batch, 512).astype(np.float32)).to(device)
            frames = torch.empty((batch, resolution, resolution, 3), dtype=torch.uint8, device=device)
            engine.set_profiling(False)
            for _ in range(warmup):
                engine.synthesize_z_device(z.data_ptr(), batch, 1.2, frames.data_ptr(), 0, stream.cuda_stream)
            begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            begin.record(stream)
            for _ in range(steps):
                engine.synthesize_z_device(z.data_ptr(), batch, 1.2, frames.data_ptr(), 0, stream.cuda_stream)
            end.record(stream)
            end.synchronize()
            ms_per_call = begin.elapsed_time(end) / steps
            engine.set_profiling(True)  # one more call with every launch bracketed: the forms, and the slowest launches
            engine.synthesize_z_device(z.data_ptr(), batch, 1.2, frames.data_ptr(), 0, stream.cuda_stream)
            torch.cuda.synchronize(device)
            launches = engine.steps()
            forms = {}
            for info in launches:
                if info.name.startswith("conv") and info.flops > 0:
                    kind, _, rest = info.name.partition("_")
                    forms[rest.split("_")[0] + ("_up" if kind.startswith("convT") else "")] = kind + ("/" + info.name.rsplit("/", 1)[1] if "/" in info.name else "")  # (/16, /16x, /s3, /s3r)
            row = {
                "frames_per_s": round(batch / (ms_per_call * 1e-3), 2), "ms_per_frame": round(ms_per_call / batch, 4), "ms_per_call": round(ms_per_call, 4),
                "direct_form_frac_of_fp32_mfma_peak": round(
                    batch / (ms_per_call * 1e-3) * ALGORITHMIC_GFLOP_PER_FRAME_1024 * (resolution / 1024) ** 2 / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4
                ),
                "launches": len(launches), "forms": forms,
            }
            if batch == 1:
                slowest = sorted(launches, key=lambda info: -info.ms)[:3]
                row["gpu_ms_sum_of_launches"] = round(sum(info.ms for info in launches), 4)
                row["slowest_launches"] = [{"name": info.name, "us": round(info.ms * 1e3, 1)} for info in slowest]
            rows[str(batch)] = row
    finally:
        engine.close()
    return {
        "workload": "BASELINE.json configs[1] at other batch sizes (SURVEY.md section 8(d) config 2): random-z synthesis, z and frames resident in HBM",
        "timing": f"HIP events on the launch stream around {steps} engine calls after {warmup} warm-up calls, per batch size",
        "forms_legend": "conv<N> direct form, convW F(2x2,3x3), convV F(4x4,3x3), convVG F(4x4,3x3) as 36 dense GEMMs (8x8, 16x16), +rgb ToRGB channel sum in the epilogue, +torgb fused ToRGB + uint8; "
        "convT two-pass up layer (+ fir pass), convTG its scatter form (one dense GEMM + gather, + fir pass), convTF / convTFp one fused up kernel (p: input pre-scaled by its style)",
        "by_batch": rows,
    }


METRIC_CONFIG_2 = "projection-file-blend frames/sec at 1024x1024 through the product stream (BASELINE.json configs[2]: 30 s WAV -> FFT + fft-roll -> alpha-blended latents -> synthesis @60 fps), files on disk -> frames on the host"
METRIC_CONFIG_3 = "projection-file-blend frames/sec at --output-side-length 2160 on ONE GPU through the product stream (BASELINE.json configs[3] without the 8-GPU sharding), files on disk -> 2160x2160 frames on the host"
METRIC_CONFIG_4 = "projection-file-blend frames/sec at 1024x1024 with three resident networks switched by the RMS index, through the product stream (BASELINE.json configs[4] on one GPU, no overlay)"
METRIC_CONFIG_4_OVERLAY = "projection-file-blend frames/sec at 1024x1024 with three resident networks AND the streaming phash / bbox overlay gate (BASELINE.json configs[4] on one GPU; synthetic landmark detector)"


def sharded(label: str, world_size: int) -> str:
    """The metric label of a product-stream measurement taken on `world_size` GPUs."""
    if world_size == 1:
        return label
    return label.replace("on ONE GPU ", "").replace(" on one GPU", "").replace("without the 8-GPU sharding", "frame-sharded") + f" -- frame-sharded over {world_size} GPUs"


def blend_workload(args, device) -> int:
    """
    `--workload blend`: only the configs[2] line (with --networks / --output-side / --overlay: configs[4] / configs[3]), on
    --gpus N GPUs (collective: launched by torch.distributed.run like the contract line; rank 0 prints).
    """
    world_size = dist.get_world_size() if dist.is_initialized() else 1
    label = METRIC_CONFIG_4_OVERLAY if args.overlay else (METRIC_CONFIG_4 if args.networks > 1 else (METRIC_CONFIG_3 if args.output_side else METRIC_CONFIG_2))
    record = product_stream_measurement(
        sharded(label, world_size), args.resolution, args.batch, args.networks, args.output_side, device, not args.no_cpu_baseline and world_size == 1,
        args.overlay, args.drain,
    )
    if record:
        print(json.dumps(record), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main() -> int:
    parser = argparse.ArgumentParser()
    parser.add_argument("--gpus", type=int, default=1)
    parser.add_argument("--steps", type=int, default=20)
    parser.add_argument("--warmup", type=int, default=3)
    parser.add_argument("--batch", type=int, default=64, help="frames per step per GPU (the engine's maximum; 32: -2 %, 16: -8 %)")
    parser.add_argument("--resolution", type=int, default=1024)
    parser.add_argument("--no-cpu-baseline", action="store_true")
    parser.add_argument(
        "--all-terms", action="store_true",
        help="random weights with every noise strength and bias non-zero (StyleGAN2's own init has them at zero and the engine skips "
        "absent noise): what a trained network costs; not the contract line",
    )
    parser.add_argument("--no-extras", action="store_true", help="skip the extra measurements (configs[2], 3 networks, 2160 output, one-frame latency) a 1-GPU run adds to the line")
    parser.add_argument("--print-steps", action="store_true", help="per-launch table on stderr")
    parser.add_argument(
        "--workload", choices=["synthesis", "blend"], default="synthesis",
        help="synthesis = BASELINE configs[1] (the contract line); blend = configs[2]: 30 s synthetic WAV -> "
        "FFT + fft-roll -> alpha-blended latents -> 1024 synthesis, single GPU, extra line for DESIGN.md",
    )
    parser.add_argument("--networks", type=int, default=1, help="blend workload: resident networks the RMS index switches between")
    parser.add_argument("--output-side", type=int, default=None, help="blend workload: --output-side-length (bicubic resize in HBM), e.g. 2160")
    parser.add_argument("--overlay", action="store_true", help="blend workload: the streaming phash / bbox overlay gate (synthetic landmark detector)")
    parser.add_argument(
        "--drain", choices=list(frame_sharding.DRAIN_MODES), default="rank0",
        help="blend workload with --gpus N > 1: rank0 = RCCL gather of every chunk to rank 0, drained over its PCIe link; per-rank = every rank "
        "drains the pieces it synthesised over its own link (no gather; not with --overlay)",
    )
    parser.add_argument(
        "--rccl-channels", type=int, default=0,
        help="N > 1 GPUs: cap RCCL at this many channels (NCCL_MAX_NCHANNELS = NCCL_MIN_NCHANNELS = N, set before the process group exists): "
        "fewer copy kernels holding CUs while a gather overlaps the synthesis, for longer; 0 = RCCL's default (the first A/B on a node)",
    )
    parser.add_argument("--batch-sweep", default="1,4,8,16,32,64", help="batch sizes of extras.batch_sweep (SURVEY.md section 8(d) config 2); empty: skip")
    args = parser.parse_args()

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback in the product path")
    # rehearsal knob for a one-GPU box: GANCE_BENCH_REHEARSAL=1 puts every rank on cuda:0 over gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearsal = os.environ.get("GANCE_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rccl_channels > 0:
            os.environ["NCCL_MAX_NCHANNELS"] = os.environ["NCCL_MIN_NCHANNELS"] = str(args.rccl_channels)
        if rehearsal:
            dist.init_process_group(backend="gloo", timeout=PROCESS_GROUP_TIMEOUT)
        else:
            dist.init_process_group(backend="nccl", device_id=device, timeout=PROCESS_GROUP_TIMEOUT)

    resolution, batch = args.resolution, args.batch
    if args.workload == "blend":
        return blend_workload(args, device)
    variables = sg2_spec.make_random_variables(resolution, seed=0, perturb=args.all_terms)
    engine = hip_lib.Engine(variables, resolution, max_batch=batch, device=local_rank, profile=True)

    # inputs resident in HBM before the timed region: rank 0 draws every rank's z and scatters them
    all_z = None
    if rank == 0:
        all_z = torch.from_numpy(np.random.RandomState(1).randn(world_size * batch, 512).astype(np.float32))
    z = frame_sharding.scatter_latents(all_z, world_size * batch, device).contiguous()
    frames = [torch.empty((batch, resolution, resolution, 3), dtype=torch.uint8, device=device) for _ in range(2)]
    gathered = None
    if world_size > 1 and rank == 0:
        gathered = [torch.empty((world_size * batch, resolution, resolution, 3), dtype=torch.uint8, device=device) for _ in range(2)]
    stream = torch.cuda.current_stream(device)
    pending = [None, None]

    def step(index: int) -> None:
        slot = index & 1
        if pending[slot] is not None:
            pending[slot].wait()
            pending[slot] = None
        engine.synthesize_z_device(z.data_ptr(), batch, 1.2, frames[slot].data_ptr(), 0, stream.cuda_stream)
        if world_size > 1:
            _, work = frame_sharding.gather_frames(
                frames[slot], world_size * batch, async_op=True, out=gathered[slot] if rank == 0 else None
            )
            pending[slot] = work

    def drain() -> None:
        for slot in (0, 1):
            if pending[slot] is not None:
                pending[slot].wait()
                pending[slot] = None

    def fence() -> None:
        drain()
        torch.cuda.synchronize(device)
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Warm-up steps carry HIP events around every launch (they name the dominant kernel); in the
    # timed region only that kernel's launches are bracketed (2 events per step: ~100 events per
    # step cost 1.4 % of the step), and its average over the K timed launches is the roofline figure.
    for index in range(args.warmup):
        step(index)
    fence()
    warm_steps = [s for s in engine.steps() if s.flops > 0 and s.name.startswith("conv")]
    # without warm-up steps: the last stride-1 conv, the largest M x N of the network
    last_conv = "conv%d_%dx%d_" % (2 * int(np.log2(resolution)) - 4, resolution, resolution)
    dominant_name = max(warm_steps, key=lambda s: s.ms).name if warm_steps else last_conv
    engine.set_profiling(True, only_step=dominant_name)
    fence()
    start = time.perf_counter()
    for index in range(args.steps):
        step(index)
    fence()
    elapsed = time.perf_counter() - start
    if world_size > 1:
        elapsed_t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(elapsed_t, op=dist.ReduceOp.MAX)
        elapsed = float(elapsed_t.item())
    timed = [s for s in engine.steps() if s.name.startswith(dominant_name)]
    dominant_name = timed[0].name
    dominant = hip_lib.StepInfo(dominant_name, sum(s.ms for s in timed) / len(timed), timed[0].flops, timed[0].bytes)

    # one more, untimed, step with every launch bracketed: the per-launch table and the all-conv figure
    engine.set_profiling(True)
    step(args.steps)
    fence()
    steps_info = engine.steps()
    conv_steps = [s for s in steps_info if s.flops > 0 and s.name.startswith("conv")]
    conv_ms = sum(s.ms for s in conv_steps)
    conv_flops = sum(s.flops for s in conv_steps)
    # Winograd F(2x2,3x3) launches ("convW..."): flops above are the ALGORITHMIC (direct-form) ones; the
    # matrix cores execute 4/9 of them
    executed_flops = sum(s.flops * executed_fraction(s.name) for s in conv_steps if not s.name.endswith(SPLIT_SUFFIXES))  # on the fp32 matrix pipe
    split_steps = [s for s in conv_steps if s.name.endswith(SPLIT_SUFFIXES)]
    # matrix-pipe time the step's products need at the pipes' peaks (each launch on its own pipe) -- over the wall time: the utilisation
    pipe_seconds = sum(s.flops * executed_fraction(s.name) / (matrix_peak(s.name) * 1e12) for s in conv_steps)
    winograd_launches = sum(1 for s in conv_steps if s.name.startswith(("convW", "convV")))
    winograd43_launches = sum(1 for s in conv_steps if s.name.startswith("convV"))
    total_ms = sum(s.ms for s in steps_info)
    if args.print_steps and rank == 0:
        for info in steps_info:
            tflops = info.flops / (info.ms * 1e-3) / 1e12 if info.ms > 0 else 0.0
            gbs = info.bytes / (info.ms * 1e-3) / 1e9 if info.ms > 0 else 0.0
            print(f"  {info.name:34s} {info.ms * 1e3:9.1f} us {tflops:8.2f} TFLOP/s {gbs:9.1f} GB/s", file=sys.stderr)
        print(f"  sum of launches {total_ms:.3f} ms", file=sys.stderr)

    if rank == 0:
        traffic_bytes, traffic_note = measured_traffic(dominant.name, resolution, batch)
        frames_total = world_size * batch * args.steps
        fps = frames_total / elapsed
        result = {
            "metric": "synthesized frames/sec at 1024x1024 config-f",
            "value": round(fps, 3),
            "unit": "frames/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": SPLIT_DTYPE if split_steps else "f32",
            "data": "synthetic (random-init weights seed 0, RandomState(1) z vectors)" + (" --all-terms: noise strengths and biases non-zero" if args.all_terms else "") + (" REHEARSAL: ranks share one GPU over gloo, not a measurement" if rehearsal else ""),
            "config": {
                "workload": "BASELINE.json configs[1]: FFHQ config-f %dx%d random-init, batched random-z synthesis (mapping + truncation psi=1.2 + synthesis + uint8 NHWC), frames resident in HBM" % (resolution, resolution),
                "frames_per_step_per_gpu": batch,
                "parallelism": f"frame-sharded x{world_size}" + (" + RCCL gather to rank 0" if world_size > 1 else "")
                + (f" (RCCL capped at {args.rccl_channels} channels)" if world_size > 1 and args.rccl_channels > 0 else ""),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "%s (%s)" % (kernel_of_step(dominant.name), dominant.name),
                "launches_averaged": len(timed),
                # `achieved` / `frac`: the flops the matrix cores EXECUTE in the dominant launch over its duration (HIP events on the launch
                # stream, inside the timed region), always <= 1 of the peak. The dominant launch is an up layer in the pair form, which
                # executes 15/18 of the layer's direct-form flops (a Winograd launch 1/4 or 4/9); SURVEY.md section 8(d)'s ALGORITHMIC
                # figure -- the layer priced as a direct convolution -- stands beside it as `*_direct_form` and may exceed the peak.
                "achieved": round(executed_fraction(dominant.name) * dominant.flops / (dominant.ms * 1e-3) / 1e12, 3),
                "peak": matrix_peak(dominant.name),
                "unit": "TFLOP/s",
                "pipe": matrix_pipe(dominant.name),
                "frac": round(executed_fraction(dominant.name) * dominant.flops / (dominant.ms * 1e-3) / 1e12 / matrix_peak(dominant.name), 4),
                "achieved_direct_form": round(dominant.flops / (dominant.ms * 1e-3) / 1e12, 3),
                "frac_direct_form": round(dominant.flops / (dominant.ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "executed_share_of_direct_form": round(executed_fraction(dominant.name), 4),
                "ms_per_launch": round(dominant.ms, 4),
                "traffic": traffic_bytes,
                "traffic_note": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes) %s; algorithmic bytes per launch = %d"
                % (traffic_note, int(dominant.bytes)),
                # the same launch against the HBM roof (it is not the bound): measured traffic, and the algorithmic bytes, over the duration
                "hbm": {
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "achieved": round(traffic_bytes / (dominant.ms * 1e-3) / 1e9, 1) if traffic_bytes else None,
                    "frac": round(traffic_bytes / (dominant.ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic_bytes else None,
                    "algorithmic_achieved": round(dominant.bytes / (dominant.ms * 1e-3) / 1e9, 1),
                    "algorithmic_frac": round(dominant.bytes / (dominant.ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                },
                "all_conv_launches": {
                    # executed matrix-core flops of all conv launches over their summed durations (Winograd launches
                    # count 4/9 of their direct-form flops: that is what runs)
                    "frac": round(pipe_seconds / (conv_ms * 1e-3), 4),
                    "frac_note": "time the launches' executed products need at the dense peak of the matrix pipe each runs on (fp32 MFMA 157.3 TFLOP/s; "
                    "the %d split-operand launches: bf16 MFMA 2516.6 TFLOP/s, six part products per fp32 product) over their measured time" % len(split_steps),
                    "fp32_pipe_achieved": round(executed_flops / (sum(s.ms for s in conv_steps if not s.name.endswith(SPLIT_SUFFIXES)) * 1e-3) / 1e12, 3),
                    "bf16_pipe_achieved": round(sum(6.0 * s.flops for s in split_steps) / (sum(s.ms for s in split_steps) * 1e-3) / 1e12, 3) if split_steps else None,
                    "algorithmic_direct_form": round(conv_flops / (conv_ms * 1e-3) / 1e12, 3),
                    "share_of_step_time": round(conv_ms / total_ms, 4),
                    "winograd_launches": winograd_launches,
                    "note": "%d of the %d conv launches run in Winograd form (%d of them F(4x4,3x3): 1/4 of their direct-form flops executed; the others F(2x2,3x3): 4/9); "
                    "algorithmic_direct_form prices every launch as a direct convolution and may exceed the peak" % (winograd_launches, len(conv_steps), winograd43_launches),
                },
                # the whole step (mapping, styles, ToRGB, uint8 included) against the matrix peak: executed flops of a
                # step over its wall time; and the same in direct-form flops, the work a direct implementation would do
                "whole_path_frac": round(pipe_seconds / (elapsed / args.steps), 4),
                "whole_path_direct_form_frac": round(
                    (fps / world_size) * ALGORITHMIC_GFLOP_PER_FRAME_1024 * (resolution / 1024) ** 2 / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4
                ),
            },
        }
        if world_size == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(resolution, variables)
    engine.close()
    extras = None
    if not args.no_extras and world_size == 1:
        # measured beside the contract line, same process, same GPU (each is also reachable alone: --workload blend)
        extras = {
            "config_2_blend_stream": guarded(product_stream_measurement, METRIC_CONFIG_2, resolution, batch, 1, None, device, not args.no_cpu_baseline),
            "config_4_three_networks_stream": guarded(product_stream_measurement, METRIC_CONFIG_4, resolution, batch, 3, None, device, False),
            "config_4_three_networks_overlay_stream": guarded(product_stream_measurement, METRIC_CONFIG_4_OVERLAY, resolution, batch, 3, None, device, False, True),
            "config_3_output_side_2160_one_gpu_stream": guarded(product_stream_measurement, METRIC_CONFIG_3, resolution, batch, 1, 2160, device, False),
            "one_frame_latency": guarded(one_frame_latency, resolution, device),
        }
        if args.batch_sweep and not args.all_terms:
            extras["batch_sweep"] = guarded(batch_sweep, resolution, variables, device, [int(b) for b in args.batch_sweep.split(",") if int(b) <= batch])
    elif not args.no_extras and not rehearsal_skips_extras():
        # N GPUs: BASELINE.json configs[3] (2160^2 output, frame-sharded, both drains) and configs[4] (three networks + overlay gate) through
        # the product stream, collectively, beside the weak-scaling line. A watchdog guarantees the contract line: should a
        # collective of the extras hang (a rank lost), rank 0 prints the line without them and every rank leaves.
        import threading  # pylint: disable=import-outside-toplevel

        printed = threading.Lock()  # whoever holds it prints the contract line: the watchdog or the normal end, never both

        def give_up() -> None:
            if not printed.acquire(blocking=False):  # pylint: disable=consider-using-with
                return  # (the extras finished while the timer fired: the normal end prints)
            if rank == 0:
                result["extras"] = {"error": "the multi-GPU extras did not finish within %d s; the contract line above them is complete" % EXTRAS_WATCHDOG_S}
                print(json.dumps(result), flush=True)
            # non-zero: a lost rank or a stuck collective is a failure of the job, whatever was printed (the driver takes the last JSON line)
            os._exit(3)  # pylint: disable=protected-access

        watchdog = threading.Timer(EXTRAS_WATCHDOG_S, give_up)
        watchdog.daemon = True
        watchdog.start()
        extras = {
            "config_3_output_side_2160_sharded_rank0_drain": guarded(product_stream_measurement, sharded(METRIC_CONFIG_3, world_size), resolution, batch, 1, 2160, device, False, False, "rank0"),
            "config_3_output_side_2160_sharded_per_rank_drain": guarded(product_stream_measurement, sharded(METRIC_CONFIG_3, world_size), resolution, batch, 1, 2160, device, False, False, "per-rank"),
            "config_4_three_networks_overlay_sharded": guarded(product_stream_measurement, sharded(METRIC_CONFIG_4_OVERLAY, world_size), resolution, batch, 3, None, device, False, True, "rank0"),
        }
        watchdog.cancel()
        if not printed.acquire(blocking=False):  # pylint: disable=consider-using-with
            threading.Event().wait()  # the watchdog fired first: it prints and ends the process
    if rank == 0:
        if extras is not None:
            result["extras"] = extras
        print(json.dumps(result), flush=True)

    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
