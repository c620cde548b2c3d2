"""Dev check: Winograd conv path against the direct path, same engine inputs (fp32 image and uint8 frames)."""
import os, subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

if len(sys.argv) > 1:
    from gance_amd import hip_lib
    from gance_amd.stylegan2 import spec
    variables = spec.make_random_variables(1024, seed=0, perturb=True)
    engine = hip_lib.Engine(variables, 1024, max_batch=4, device=0)
    z = np.random.RandomState(1).randn(4, 512).astype(np.float32)
    u8, f32 = engine.synthesize_z(z, want_float=True)
    np.save(sys.argv[1], f32)
    np.save(sys.argv[1] + ".u8.npy", u8)
    sys.exit(0)
for flag in ("0", "2"):
    subprocess.run([sys.executable, __file__, f"/tmp/wino_{flag}.npy"], env=dict(os.environ, GANCE_TUNE_WINOGRAD=flag), check=True)
a, b = np.load("/tmp/wino_0.npy"), np.load("/tmp/wino_2.npy")
print("image range", a.min(), a.max(), "max abs diff direct vs winograd", np.abs(a - b).max(), "rms", np.sqrt(np.mean((a - b) ** 2)))
ua, ub = np.load("/tmp/wino_0.npy.u8.npy"), np.load("/tmp/wino_2.npy.u8.npy")
print("u8 differing", int((ua != ub).sum()), "of", ua.size, "max", int(np.abs(ua.astype(int) - ub.astype(int)).max()))
