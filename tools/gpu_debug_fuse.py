"""Dev check: fused last-layer ToRGB against the separate kernels, same engine inputs."""
import os, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

if len(sys.argv) > 1:
    import torch
    from gance_amd import hip_lib
    from gance_amd.stylegan2 import spec
    variables = spec.make_random_variables(1024, seed=0, perturb=True)
    engine = hip_lib.Engine(variables, 1024, max_batch=2, device=0)
    z = np.random.RandomState(1).randn(2, 512).astype(np.float32)
    u8, f32 = engine.synthesize_z(z, want_float=True)
    np.save(sys.argv[1], f32)
    np.save(sys.argv[1] + ".u8.npy", u8)
    sys.exit(0)
for flag in ("0", "1"):
    subprocess.run([sys.executable, __file__, f"/tmp/fuse_{flag}.npy"], env=dict(os.environ, GANCE_TUNE_FUSE_RGB=flag), check=True)
a, b = np.load("/tmp/fuse_0.npy"), np.load("/tmp/fuse_1.npy")
print("shape", a.shape, "max abs diff", np.abs(a - b).max())
diff = np.abs(a - b)
print("per sample/channel max", diff.reshape(2, 3, -1).max(axis=2))
bad = np.argwhere(diff > 1e-3)
print("bad count", len(bad), "first", bad[:10])
print("rows with bad", np.unique(bad[:, 2])[:20], "cols", np.unique(bad[:, 3])[:40])
ua, ub = np.load("/tmp/fuse_0.npy.u8.npy"), np.load("/tmp/fuse_1.npy.u8.npy")
print("u8 diff count", int((ua != ub).sum()), "max", int(np.abs(ua.astype(int) - ub.astype(int)).max()))
print("sample values", a[0, :, 5, 5], b[0, :, 5, 5], a[0, :, 500, 700], b[0, :, 500, 700])
