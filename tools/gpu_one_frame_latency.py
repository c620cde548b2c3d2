"""One-frame latency of the host-buffer entries at 1024^2 (the reference's call pattern): create_image_matrix / create_image_vector."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gance_amd import hip_lib  # noqa: E402
from gance_amd.stylegan2 import spec as sg2_spec  # noqa: E402


def main() -> None:
    resolution = 1024
    engine = hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=0), resolution, max_batch=1)
    rng = np.random.RandomState(0)
    result = {}
    for name, make, call in (
        ("create_image_matrix_ms", lambda: rng.randn(1, engine.num_layers, 512).astype(np.float32), engine.synthesize_w),
        ("create_image_vector_ms", lambda: rng.randn(1, 512).astype(np.float32), engine.synthesize_z),
    ):
        for _ in range(5):
            call(make())
        times = []
        for _ in range(50):
            data = make()
            start = time.perf_counter()
            call(data)
            times.append(time.perf_counter() - start)
        result[name] = {"median": round(1e3 * float(np.median(times)), 4), "min": round(1e3 * min(times), 4), "p90": round(1e3 * float(np.percentile(times, 90)), 4)}
    engine.close()
    print(json.dumps(result))
    # the per-launch table of ONE frame (every launch bracketed by HIP events: eager launches, no graph): where the
    # 2.3 ms of a one-frame call go (the small layers run split-K 8 ... 128 ways at this batch)
    profiled = hip_lib.Engine(sg2_spec.make_random_variables(resolution, seed=0), resolution, max_batch=1, profile=True)
    for _ in range(3):
        profiled.synthesize_w(rng.randn(1, profiled.num_layers, 512).astype(np.float32))
    total = 0.0
    for step in profiled.steps():
        total += step.ms
        print(f"  {step.name:34s} {step.ms * 1e3:9.1f} us", file=sys.stderr)
    print(f"  sum of launches {total:.3f} ms", file=sys.stderr)
    profiled.close()


if __name__ == "__main__":
    main()
