"""
Which kernel form every conv layer runs in, by batch size: the engine's own decisions (engine.hip: conv_form_of,
up_runs_fused, plan_layer), read back from the launch names of real calls at every batch size 1 ... max_batch.
    python tools/gpu_form_table.py [resolution] [max_batch] > profiles/rNN_form_table.txt
Legend: conv<N> direct form, convW F(2x2,3x3), convV F(4x4,3x3), +rgb ToRGB channel sum in the epilogue, +torgb fused ToRGB +
uint8; convT two-pass up layer (+ fir pass), convTF / convTFp one fused up kernel (p: input pre-scaled by its style; /16: the
16-channel two-blocks-per-CU geometry); (xK): split-K factor K of a direct-form launch is not in the name -- see `finish` rows.
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gance_amd import hip_lib  # noqa: E402
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402

resolution = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
max_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
variables = sg2_spec.make_random_variables(resolution, seed=0)
engine = hip_lib.Engine(variables, resolution, max_batch=max_batch, profile=True)
rng = np.random.RandomState(0)
table = {}  # layer tag -> list of form per batch
order = []
for batch in range(1, max_batch + 1):
    engine.synthesize_z(rng.randn(batch, 512).astype(np.float32))
    seen = {}
    for step in engine.steps():
        if not step.name.startswith("conv"):
            continue
        kind, _, rest = step.name.partition("_")
        digits = "".join(ch for ch in kind if ch.isdigit())
        tag = f"{int(digits):2d} {rest.split('/')[0]}"  # (the name's "/16" / "/16x" suffix goes into the form)
        form = kind.replace(digits, "", 1) + ("/" + step.name.rsplit("/", 1)[1] if "/" in step.name else "")  # (/16, /16x, /s3, /s3r)
        seen[tag] = form
        if tag not in table:
            table[tag] = {}
            order.append(tag)
    finishes = {s.name.split("_")[0].replace("finish", "") for s in engine.steps() if s.name.startswith("finish")}
    for tag, form in seen.items():
        layer = tag.split()[0]
        table[tag][batch] = form + (" +finish (split-K)" if layer in finishes else "")
engine.close()
print(f"kernel form of every conv layer of the {resolution}x{resolution} generator by frames per engine call (1 ... {max_batch}); from tools/gpu_form_table.py")
for tag in order:
    runs, start, current = [], 1, table[tag].get(1)
    for batch in range(2, max_batch + 2):
        form = table[tag].get(batch) if batch <= max_batch else None
        if form != current:
            runs.append(f"B {start}" + (f"-{batch - 1}" if batch - 1 > start else "") + f": {current}")
            start, current = batch, form
    print(f"  {tag:32s} " + " | ".join(runs))
