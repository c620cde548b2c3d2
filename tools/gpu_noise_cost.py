"""Dev experiment: whole-step time with noise / bias terms active (perturb=True: what a trained network has) vs the plain random init."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
from gance_amd import hip_lib
from gance_amd.stylegan2 import spec

batch = 16
z = torch.from_numpy(np.random.RandomState(1).randn(batch, 512).astype(np.float32)).cuda()
out = torch.empty((batch, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for perturb in (False, True, False, True):
    variables = spec.make_random_variables(1024, seed=0, perturb=perturb)
    engine = hip_lib.Engine(variables, 1024, max_batch=batch, device=0, profile=False)
    for _ in range(3):
        engine.synthesize_z_device(z.data_ptr(), batch, 1.2, out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        engine.synthesize_z_device(z.data_ptr(), batch, 1.2, out.data_ptr(), 0, stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"perturb={perturb}: {dt*1e3:.3f} ms/step  {batch/dt:.1f} fps", flush=True)
    engine.close()
