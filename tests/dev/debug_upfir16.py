"""Dev check of the fused up kernel's 16-channel geometry: where does it differ from the fp64 oracle?
    python tests/dev/debug_upfir16.py [resolution] [batch]   (GANCE_TUNE_UPFIR16=0 for the 32-channel geometry)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from gance_amd import hip_lib  # noqa: E402
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402
from oracle import stylegan2_ref as ref  # noqa: E402

resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 3
spec = sg2_spec.make_spec(resolution)
variables = sg2_spec.make_random_variables(resolution, seed=3, perturb=True)
dlatents = np.random.RandomState(5).randn(batch, spec.num_layers, 512).astype(np.float32)
engine = hip_lib.Engine(variables, resolution, max_batch=batch, up_form="fused")
wants = []
with torch.no_grad():
    ref.g_synthesis(torch.from_numpy(dlatents).double(), variables, resolution, collect=wants)
for n, conv in enumerate(spec.convs, start=1):
    if not (conv.up and 2 ** conv.res_log2 >= 32):
        continue
    got = engine.debug_activation_after(dlatents, n)
    want = wants[n - 1].numpy()
    err = np.abs(got - want)
    scale = np.abs(want).max()
    print(f"layer {n} {conv.scope}: rel err {err.max() / scale:.3e}")
    bad = err > 1e-4 * scale
    print("  bad fraction", bad.mean())
    print("  per sample", bad.reshape(batch, -1).mean(axis=1))
    per_c = bad.mean(axis=(0, 2, 3))
    print("  per channel (first 32)", np.round(per_c[:32], 3))
    per_row = bad.mean(axis=(0, 1, 3))
    print("  rows with errors:", np.nonzero(per_row > 0)[0][:64], "count", int((per_row > 0).sum()), "of", len(per_row))
    print("  row error fraction (first 40)", np.round(per_row[:40], 2))
    per_col = bad.mean(axis=(0, 1, 2))
    print("  cols with errors:", np.nonzero(per_col > 0)[0][:64], "count", int((per_col > 0).sum()), "of", len(per_col))
    print("  col error fraction (first 40)", np.round(per_col[:40], 2))
    c = int(np.argmax(per_c))
    print(f"  channel {c} sample 0, rows 0..11 cols 0..7: got\n", np.round(got[0, c, :12, :8], 3), "\n want\n", np.round(want[0, c, :12, :8], 3))
engine.close()
