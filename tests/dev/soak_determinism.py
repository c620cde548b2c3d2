"""Dev soak (GPU): the same batch synthesised many times must give bit-identical frames (no race in the
persistent / ring-buffered kernels), at the bench's batch size."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
from gance_amd import hip_lib
from gance_amd.stylegan2 import spec

batch, rounds = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 150
variables = spec.make_random_variables(1024, seed=0, perturb=True)
engine = hip_lib.Engine(variables, 1024, max_batch=batch, device=0)
z = torch.from_numpy(np.random.RandomState(1).randn(batch, 512).astype(np.float32)).cuda()
ref = torch.empty((batch, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
out = torch.empty_like(ref)
stream = torch.cuda.current_stream().cuda_stream
engine.synthesize_z_device(z.data_ptr(), batch, 1.2, ref.data_ptr(), 0, stream)
torch.cuda.synchronize()
bad = 0
t0 = time.perf_counter()
for i in range(rounds):
    engine.synthesize_z_device(z.data_ptr(), batch, 1.2, out.data_ptr(), 0, stream)
    if not torch.equal(out, ref):
        bad += 1
        print("round", i, "differs in", int((out != ref).sum()), "bytes", flush=True)
torch.cuda.synchronize()
print(f"{rounds} rounds of {batch} frames in {time.perf_counter() - t0:.1f} s: {bad} differing rounds")
engine.close()
sys.exit(1 if bad else 0)
