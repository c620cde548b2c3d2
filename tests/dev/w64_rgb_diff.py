"""Dev check: float image of a small network with the fused ToRGB product on / off (GANCE_TUNE_W64_RGB), error pattern."""
import os, subprocess, sys
import numpy as np

RES = int(os.environ.get("RES", "512"))
if len(sys.argv) > 1:
    from gance_amd import hip_lib
    from gance_amd.stylegan2 import spec as sg2_spec
    spec = sg2_spec.make_spec(RES)
    variables = sg2_spec.make_random_variables(RES, seed=3, perturb=True)
    dl = np.random.RandomState(5).randn(2, spec.num_layers, 512).astype(np.float32)
    eng = hip_lib.Engine(variables, RES, max_batch=2)
    u8, f32 = eng.synthesize_w(dl, want_float=True)
    np.save(sys.argv[1], f32)
    sys.exit(0)
for v in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"gpurun_out/w64rgb_{v}.npy"], check=True, env=dict(os.environ, GANCE_TUNE_W64_RGB=v))
a = np.load("gpurun_out/w64rgb_0.npy"); b = np.load("gpurun_out/w64rgb_1.npy")
d = b - a
print("shape", a.shape, "max|a|", np.abs(a).max(), "max|d|", np.abs(d).max())
for c in range(3): print("colour", c, np.abs(d[:, c]).max(), np.abs(d[:, c]).mean())
for py in range(2):
    for px in range(2): print("parity", py, px, np.abs(d[:, :, py::2, px::2]).max())
print("rows 0..15 max:", [float(np.abs(d[0, :, y]).max().round(4)) for y in range(16)])
print("cols 0..35 max:", [float(np.abs(d[0, :, :, x]).max().round(4)) for x in range(36)])
print("corr d vs a per colour", [float(np.corrcoef(d[0, c].ravel(), a[0, c].ravel())[0, 1]) for c in range(3)])
print("sample d/a", (d[0, :, 100, 100:104]), a[0, :, 100, 100:104])
