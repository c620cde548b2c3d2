import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from gance_amd import hip_lib
from gance_amd.stylegan2 import spec as S
res = 1024
variables = S.make_random_variables(res, 0)
def run(n_engines, batch, steps=8, warm=2):
    engines = [hip_lib.Engine(variables, res, max_batch=batch) for _ in range(n_engines)]
    streams = [torch.cuda.Stream() for _ in range(n_engines)]
    z = [torch.randn(batch, 512, device='cuda') for _ in range(n_engines)]
    out = [torch.empty((batch, res, res, 3), dtype=torch.uint8, device='cuda') for _ in range(n_engines)]
    def step():
        for e, s, zz, o in zip(engines, streams, z, out):
            e.synthesize_z_device(zz.data_ptr(), batch, 1.2, o.data_ptr(), 0, s.cuda_stream)
    for _ in range(warm): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    fps = n_engines * batch * steps / dt
    for e in engines: e.close()
    return fps
for cfg in [(1, 16), (2, 8), (2, 16), (4, 8), (3, 8)]:
    print(cfg, round(run(*cfg), 1), flush=True)
