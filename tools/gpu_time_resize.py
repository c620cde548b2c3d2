"""Dev timing: bicubic resize 1024 -> 2160 of a 32-frame batch, and the ordered-output index_copy_."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from gance_amd import hip_lib

frames = torch.randint(0, 256, (32, 1024, 1024, 3), dtype=torch.uint8, device="cuda")
out = torch.empty((32, 2160, 2160, 3), dtype=torch.uint8, device="cuda")
big = torch.empty((256, 2160, 2160, 3), dtype=torch.uint8, device="cuda")
members = torch.arange(32, device="cuda") * 3
stream = torch.cuda.current_stream().cuda_stream
for name, fn in [
    ("resize 32 frames", lambda: hip_lib.resize_bicubic_u8_device(frames.data_ptr(), 32, 1024, out.data_ptr(), 2160, stream)),
    ("index_copy_ 32 frames", lambda: big.index_copy_(0, members, out)),
    ("empty+resize", lambda: hip_lib.resize_bicubic_u8_device(frames.data_ptr(), 32, 1024, torch.empty_like(out).data_ptr(), 2160, stream)),
]:
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
