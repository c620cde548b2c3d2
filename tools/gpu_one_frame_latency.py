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


if __name__ == "__main__":
    main()
