"""Parts of the product stream at 2160^2 on their own (one GPU): the bicubic resize of a 64-frame chunk and the pinned
device-to-host copy of its result. `python tools/gpu_stream_parts.py` prints milliseconds and GB/s."""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import gance_amd.torch_ops  # noqa: E402,F401  pylint: disable=unused-import,wrong-import-position

device = torch.device("cuda", 0)
frames = torch.randint(0, 255, (64, 1024, 1024, 3), dtype=torch.uint8, device=device)
for side in (2160, 1536):
    out = torch.ops.gance.resize_bicubic(frames, side)
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
        out = torch.ops.gance.resize_bicubic(frames, side)
    end.record()
    torch.cuda.synchronize()
    ms = start.elapsed_time(end) / 5
    print(f"resize 64 x 1024^2 -> {side}^2: {ms:.2f} ms per chunk, {(frames.numel() + out.numel()) / ms / 1e6:.0f} GB/s of algorithmic traffic")
    host = torch.empty(out.shape, dtype=torch.uint8, pin_memory=True)
    host.copy_(out, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        host.copy_(out, non_blocking=True)
    torch.cuda.synchronize()
    seconds = (time.perf_counter() - t0) / 3
    print(f"pinned D2H of {out.numel() / 1e6:.0f} MB: {seconds * 1e3:.1f} ms = {out.numel() / seconds / 1e9:.1f} GB/s")
