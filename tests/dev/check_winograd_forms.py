import numpy as np, torch, sys
sys.path.insert(0, '.')
from gance_amd import hip_lib
from gance_amd.stylegan2 import spec as sg2_spec
from oracle import stylegan2_ref as ref
for resolution, batch in [(512, 3), (256, 7), (128, 11)]:
    spec = sg2_spec.make_spec(resolution)
    variables = sg2_spec.make_random_variables(resolution, seed=11, perturb=True)
    dl = np.random.RandomState(3).randn(batch, spec.num_layers, 512).astype(np.float32)
    outs = {}
    for form in ("direct", "winograd", "auto"):
        e = hip_lib.Engine(variables, resolution, max_batch=batch, conv_form=form)
        frames, image = e.synthesize_w(dl, want_float=True)
        e.close()
        outs[form] = (frames, image)
    want = ref.synthesize_w(dl[:1], variables, resolution).numpy()
    for form in outs:
        err = np.abs(outs[form][1][:1] - want).max()
        d = np.abs(outs[form][1] - outs["direct"][1]).max()
        u8 = (outs[form][0] != outs["direct"][0]).mean()
        print(f"res {resolution} batch {batch} {form}: max|img-oracle| frame0 {err:.2e}, max|img-direct| all frames {d:.2e}, u8 differing share {u8:.2e}")
