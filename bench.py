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

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
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


def kernel_of_step(step_name: str) -> str:
    """The HIP kernel behind a conv launch of the engine's step table (names: engine.hip)."""
    if step_name.startswith("convTF"):
        return "upfir_fused_kernel"
    if step_name.startswith("convW"):
        narrow = step_name.endswith("->32")
        if "+rgb" in step_name:
            return "winograd64_c32_rgb_kernel" if narrow else "winograd64_rgb_kernel"
        return "winograd64_c32_kernel" if narrow else "winograd64_kernel"
    return "modconv_mfma_kernel"


def executed_fraction(step_name: str) -> float:
    """Matrix-core flops a launch EXECUTES per algorithmic (direct-form) flop: Winograd F(2x2,3x3) does 16 of 36."""
    return 4.0 / 9.0 if step_name.startswith("convW") else 1.0


def measured_traffic(step_name: str, resolution: int, batch: int):
    """
    HBM bytes per launch of the dominant kernel from the committed PMC pass (rocprofv3 cannot run
    inside this process): profiles/traffic_latest.json, only if it was taken on this workload and
    holds this launch (the two largest launches trade places from run to run: both are recorded).
    """
    path = REPO_ROOT / "profiles" / "traffic_latest.json"
    try:
        record = json.loads(path.read_text())
    except (OSError, ValueError):
        return None
    workload = record.get("workload", {})
    if workload.get("resolution") != resolution or workload.get("frames_per_step_per_gpu") != batch:
        return None
    for prefix, entry in record.get("launches", {}).items():
        if step_name.startswith(prefix):
            return entry.get("hbm_bytes_per_launch")
    return None


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


def blend_measurement(resolution: int, batch: int, num_networks: int, output_side, device, with_cpu_baseline: bool) -> dict:
    """
    BASELINE.json configs[2] on one GPU: 30 s synthetic WAV (30 720 Hz) + 900 projected latents
    -> spectrogram, fft-roll, alpha blend (alpha 0.25, amplitude +-5, depth 12) -> 1800 frames at
    1024^2, frames left in HBM. Timed: host audio array -> last uint8 frame in HBM. With `num_networks` > 1 the
    RMS-driven index switches between resident networks (configs[4] without the overlay); `output_side` adds
    the bicubic resize in HBM (configs[3] on one GPU).
    """
    from types import SimpleNamespace  # pylint: disable=import-outside-toplevel

    from gance_amd import projection_file_blend, synthetic  # pylint: disable=import-outside-toplevel
    from gance_amd.data_into_network_visualization import visualization_inputs  # pylint: disable=import-outside-toplevel

    num_frames = 1800
    engines = [
        hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=seed), resolution, max_batch=batch, device=device.index, profile=True)
        for seed in range(num_networks)
    ]
    # the network's last conv launch, whatever form the engine runs it in (names: engine.hip)
    last_conv_pattern = re.compile(r"^conv[A-Z]*%d[+_].*_%dx%d_" % (2 * int(np.log2(resolution)) - 4, resolution, resolution))
    audio, latents = synthetic.benchmark_blend_inputs(num_frames)
    frames = torch.empty((batch, resolution, resolution, 3), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream(device)
    rows = engines[0].num_layers
    # what synthesize_device_frames_network_major needs of a MultiNetwork
    resident = SimpleNamespace(_network_at=lambda index: SimpleNamespace(engine=engines[index]))
    switches = 0

    def run_once():
        nonlocal switches
        t0 = time.perf_counter()
        blend = visualization_inputs.alpha_blend_projection_file_device(
            latents, 0.25, True, (-5, 5), 12, audio, 512, num_networks, device=device.index
        )
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        if num_networks == 1 and output_side is None:
            dlat = blend.dlatents[:, :rows, :].contiguous()
            for start in range(0, num_frames, batch):
                count = min(batch, num_frames - start)
                engines[0].synthesize_w_device(dlat[start : start + count].data_ptr(), count, frames.data_ptr(), 0, stream.cuda_stream)
        else:  # the RMS-driven index switches networks; every network is resident, frames come back in order
            ordered = projection_file_blend.synthesize_device_frames_network_major(
                blend.dlatents, blend.network_indices, resident, output_side, batch
            )
            assert ordered.shape[0] == num_frames
            chosen = blend.network_indices.cpu().numpy()
            switches = int((chosen[1:] != chosen[:-1]).sum())
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        blend.blend.close()
        return t1 - t0, t2 - t1

    run_once()  # warm-up (LDS attribute setup, allocator), every launch bracketed: it names the last conv launch
    last_conv = next((s.name for s in engines[0].steps() if last_conv_pattern.match(s.name)), "conv")
    for engine in engines:
        engine.set_profiling(True, only_step=last_conv)
    audio_s, synth_s = run_once()
    timed = [s for engine in engines for s in engine.steps() if s.name.startswith(last_conv)]
    # audio -> latents: algorithmic bytes = the samples read once + the per-frame latent rows written once
    # (two distinct rows per frame) + the [N][L] float64 spectrogram written and read once per stage (5 stages)
    audio_bytes = num_frames * 512 * 4 + num_frames * 18 * 512 * 4 + 5 * 2 * num_frames * 512 * 8
    result = {
        "metric": "projection-file-blend frames/sec at 1024x1024 (BASELINE.json configs[2]: 30 s WAV -> FFT + fft-roll -> alpha-blended latents -> synthesis), host audio -> frames in HBM",
        "value": round(num_frames / (audio_s + synth_s), 3), "unit": "frames/s", "n_gpus": 1,
        "frames": num_frames, "audio_to_latents_ms": round(audio_s * 1e3, 3), "synthesis_ms": round(synth_s * 1e3, 3),
        "frames_per_call": batch, "networks_resident": num_networks, "network_switches": switches,
        "output_side_length": output_side or resolution,
        "dtype": "f64 (audio) / f32 (synthesis)", "data": "synthetic",
        "roofline": {
            "bound": "mfma",
            "kernel": "%s (%s)" % (kernel_of_step(last_conv), last_conv),
            "launches_averaged": len(timed),
            "achieved": round(executed_fraction(last_conv) * timed[0].flops / (sum(s.ms for s in timed) / len(timed) * 1e-3) / 1e12, 3) if timed else None,
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed_fraction(last_conv) * timed[0].flops / (sum(s.ms for s in timed) / len(timed) * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4) if timed else None,
            "audio_stage": {
                "bound": "hbm (launch-latency in practice: six kernels over < 60 MB)", "algorithmic_bytes": audio_bytes,
                "achieved": round(audio_bytes / audio_s / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(audio_bytes / audio_s / 1e9 / HBM_PEAK_GBS, 6),
            },
        },
    }
    if with_cpu_baseline:
        from oracle import audio_ref  # pylint: disable=import-outside-toplevel

        t0 = time.perf_counter()
        audio_ref.alpha_blend_projection_file(latents, 0.25, True, (-5, 5), 12, audio, 512, list(range(num_networks)))
        cpu_s = time.perf_counter() - t0
        result["cpu_baseline"] = {
            "value": round(cpu_s * 1e3, 1), "unit": "ms for audio -> latents of the same 1800 frames (lower is better)", "cores": 1, "kind": "port",
            "sample": "oracle/audio_ref.alpha_blend_projection_file (numpy / scipy restatement of the reference's chain, its Python loops vectorised), the whole 30 s workload once, single thread; the GPU stage beside it: %.1f ms" % (audio_s * 1e3),
        }
    for engine in engines:
        engine.close()
    return result


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


def blend_workload(args, device) -> int:
    """`--workload blend`: only the configs[2] line (with --networks / --output-side: configs[4] / configs[3] on one GPU)."""
    print(json.dumps(blend_measurement(args.resolution, args.batch, args.networks, args.output_side, device, not args.no_cpu_baseline)), flush=True)
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
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)

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
    executed_flops = sum(s.flops * (4.0 / 9.0 if s.name.startswith("convW") else 1.0) for s in conv_steps)
    winograd_launches = sum(1 for s in conv_steps if s.name.startswith("convW"))
    total_ms = sum(s.ms for s in steps_info)
    if args.print_steps and rank == 0:
        for info in steps_info:
            tflops = info.flops / (info.ms * 1e-3) / 1e12 if info.ms > 0 else 0.0
            gbs = info.bytes / (info.ms * 1e-3) / 1e9 if info.ms > 0 else 0.0
            print(f"  {info.name:34s} {info.ms * 1e3:9.1f} us {tflops:8.2f} TFLOP/s {gbs:9.1f} GB/s", file=sys.stderr)
        print(f"  sum of launches {total_ms:.3f} ms", file=sys.stderr)

    if rank == 0:
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
            "dtype": "f32",
            "data": "synthetic (random-init weights seed 0, RandomState(1) z vectors)" + (" --all-terms: noise strengths and biases non-zero" if args.all_terms else "") + (" REHEARSAL: ranks share one GPU over gloo, not a measurement" if rehearsal else ""),
            "config": {
                "workload": "BASELINE.json configs[1]: FFHQ config-f %dx%d random-init, batched random-z synthesis (mapping + truncation psi=1.2 + synthesis + uint8 NHWC), frames resident in HBM" % (resolution, resolution),
                "frames_per_step_per_gpu": batch,
                "parallelism": f"frame-sharded x{world_size}" + (" + RCCL gather to rank 0" if world_size > 1 else ""),
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "%s (%s)" % (kernel_of_step(dominant.name), dominant.name),
                "launches_averaged": len(timed),
                # what the matrix cores execute per second: for a Winograd launch 4/9 of the direct-form figure beside it
                "achieved": round(executed_fraction(dominant.name) * dominant.flops / (dominant.ms * 1e-3) / 1e12, 3),
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(executed_fraction(dominant.name) * dominant.flops / (dominant.ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "algorithmic_direct_form": round(dominant.flops / (dominant.ms * 1e-3) / 1e12, 3),
                "traffic": measured_traffic(dominant.name, resolution, batch),
                "traffic_note": "HBM bytes per launch from profiles/traffic_latest.json (2*FETCH_SIZE + WRITE_SIZE, separate --pmc passes); algorithmic bytes per launch = %d" % int(dominant.bytes),
                "all_conv_launches": {
                    # executed matrix-core flops of all conv launches over their summed durations (Winograd launches
                    # count 4/9 of their direct-form flops: that is what runs)
                    "achieved": round(executed_flops / (conv_ms * 1e-3) / 1e12, 3),
                    "frac": round(executed_flops / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                    "algorithmic_direct_form": round(conv_flops / (conv_ms * 1e-3) / 1e12, 3),
                    "share_of_step_time": round(conv_ms / total_ms, 4),
                    "winograd_launches": winograd_launches,
                    "note": "%d of the %d conv launches run in Winograd F(2x2,3x3) form and execute 4/9 of their direct-form flops; "
                    "algorithmic_direct_form prices every launch as a direct convolution and may exceed the peak" % (winograd_launches, len(conv_steps)),
                },
                # the whole step (mapping, styles, ToRGB, uint8 included) against the matrix peak: executed flops of a
                # step over its wall time; and the same in direct-form flops, the work a direct implementation would do
                "whole_path_frac": round(executed_flops / (elapsed / args.steps) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                "whole_path_direct_form_frac": round(
                    (fps / world_size) * ALGORITHMIC_GFLOP_PER_FRAME_1024 * (resolution / 1024) ** 2 / 1e3 / FP32_MFMA_PEAK_TFLOPS, 4
                ),
            },
        }
        if world_size == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(resolution, variables)
    engine.close()
    if rank == 0:
        if world_size == 1 and not args.no_extras:
            # measured beside the contract line, same process, same GPU (each is also reachable alone: --workload blend)
            result["extras"] = {
                "config_3_blend": blend_measurement(resolution, batch, 1, None, device, not args.no_cpu_baseline),
                "config_5_three_resident_networks": blend_measurement(resolution, batch, 3, None, device, False),
                "config_4_output_side_2160_one_gpu": blend_measurement(resolution, batch, 1, 2160, device, False),
                "one_frame_latency": one_frame_latency(resolution, device),
            }
        print(json.dumps(result), flush=True)

    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
