"""
Developer script (not collected by pytest): layer-by-layer comparison of the HIP engine with the
oracle, to localise a parity failure. Run on a GPU box:
    python tools/gpu_debug_layers.py [resolution] [batch]
"""

import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))

from gance_amd import hip_lib  # noqa: E402
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402
from oracle import stylegan2_ref  # noqa: E402


def main() -> int:
    resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    spec = sg2_spec.make_spec(resolution)
    variables = sg2_spec.make_random_variables(resolution, seed=3, perturb=True)
    rng = np.random.RandomState(5)
    dlatents = rng.randn(batch, spec.num_layers, 512).astype(np.float32)
    engine = hip_lib.Engine(variables, resolution, max_batch=batch, profile=True)
    worst = 0.0
    for n in range(1, len(spec.convs) + 1):
        got = engine.debug_activation_after(dlatents, n)
        with torch.no_grad():
            want = stylegan2_ref.g_synthesis(
                torch.from_numpy(dlatents).double(), variables, resolution, stop_after=n
            ).numpy()
        err = np.abs(got - want).max()
        scale = np.abs(want).max()
        worst = max(worst, err / scale)
        print(f"conv layer {n:2d} {spec.convs[n-1].scope:18s} shape {got.shape} max|x| {scale:9.4f} max err {err:.3e} rel {err/scale:.3e}")
    t0 = time.time()
    u8, img = engine.synthesize_w(dlatents, want_float=True)
    print("full synth wall", time.time() - t0)
    with torch.no_grad():
        want = stylegan2_ref.g_synthesis(torch.from_numpy(dlatents).double(), variables, resolution)
    want_u8 = stylegan2_ref.convert_images_to_uint8(want)
    err = np.abs(img - want.numpy()).max()
    print(f"image max|y| {np.abs(want.numpy()).max():.4f} max abs err {err:.3e}")
    diff = np.abs(u8.astype(int) - want_u8.astype(int))
    print(f"u8 max LSB diff {diff.max()} differing {100.0 * (diff > 0).mean():.4f}%")
    for step in engine.steps():
        tf = step.flops / (step.ms * 1e-3) / 1e12 if step.ms > 0 else 0.0
        print(f"  {step.name:32s} {step.ms*1e3:9.1f} us  {tf:7.2f} TFLOP/s")
    return 0 if worst < 1e-4 and err < 1e-3 else 1


if __name__ == "__main__":
    sys.exit(main())
