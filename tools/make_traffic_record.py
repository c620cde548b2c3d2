"""
Write profiles/traffic_latest.json -- the HBM traffic per launch that bench.py reports as `roofline.traffic` -- from the
PMC summaries of a profile round (tools/gpu_profile_round.sh), instead of assembling it by hand:

    python tools/make_traffic_record.py <tag> [gpurun_out]      e.g. r03_a

Inputs (all written by the round script): <dir>/<tag>_pmc_fetch_size.csv, <tag>_pmc_write_size_clock.csv,
<tag>_pmc_sq.csv (tools/pmc_summary.py tables: one row per kernel and launch slot) and <tag>_steps.txt
(bench.py --print-steps: the engine's per-launch table, whose names and algorithmic GB/s give the bytes a launch must move).
Every conv launch of the step that a kernel row can be matched to is recorded (kernel#launchN = the N-th launch of that
kernel in a step, in step order). Corrections, as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KB) counts 64 B
per 128-B request on wide streaming reads -> x2; WRITE_SIZE (KB) is exact. The record carries a digest of the HIP
sources so that bench.py can tell when the kernels have changed since the PMC pass.
"""
import csv
import json
import re
import sys
from pathlib import Path

REPO_ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO_ROOT))
import bench  # noqa: E402


def table(path: Path):
    with open(path) as handle:
        return {row["kernel"]: row for row in csv.DictReader(handle)}


def main() -> int:
    tag = sys.argv[1]
    directory = Path(sys.argv[2]) if len(sys.argv) > 2 else REPO_ROOT / "gpurun_out"
    fetch, write, sq = (table(directory / f"{tag}_pmc_{name}.csv") for name in ("fetch_size", "write_size_clock", "sq"))
    bench_line = json.loads((directory / f"{tag}_bench.json").read_text().strip().splitlines()[-1])
    batch = bench_line["config"]["frames_per_step_per_gpu"]
    resolution = int(re.search(r"(\d+)x\1", bench_line["config"]["workload"]).group(1))
    launches, seen = {}, {}
    for line in (directory / f"{tag}_steps.txt").read_text().splitlines():
        match = re.match(r"\s+(conv\S+)\s+([\d.]+) us\s+([\d.]+) TFLOP/s\s+([\d.]+) GB/s", line)
        if not match:
            continue
        name, micros, gbs = match.group(1), float(match.group(2)), float(match.group(4))
        kernel = bench.kernel_of_step(name)
        slot = seen.get(kernel, 0)
        seen[kernel] = slot + 1
        row_name = next((key for key in (f"{kernel}#launch{slot}", kernel) if key in fetch and key in write and key in sq), None)
        if row_name is None or (row_name == kernel and seen[kernel] > 1):
            continue
        fetch_kb, write_kb = float(fetch[row_name]["FETCH_SIZE"]), float(write[row_name]["WRITE_SIZE"])
        busy = float(sq[row_name]["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024.0 * float(write[row_name]["GRBM_GUI_ACTIVE"]) / 8.0)
        key = name.rsplit("_", 1)[0]  # conv..._RxR: bench.py matches launches by prefix
        launches[key.replace("convTFp", "convTF", 1)] = {
            "kernel": row_name,
            "fetch_size_kb": fetch_kb,
            "write_size_kb": write_kb,
            "hbm_bytes_per_launch": int((2.0 * fetch_kb + write_kb) * 1024),
            "algorithmic_bytes_per_launch": int(gbs * 1e9 * micros * 1e-6),
            "mfma_busy_fraction": round(busy, 3),
            "effective_clock_ghz": round(float(write[row_name]["GRBM_GUI_ACTIVE"]) / 8.0 / (float(write[row_name]["dur_us"]) * 1e3), 3),  # (the counter sums the 8 XCDs)
            "duration_us": float(write[row_name]["dur_us"]),
        }
    record = {
        "source": f"profiles/{tag}_pmc_fetch_size.csv + {tag}_pmc_write_size_clock.csv + {tag}_pmc_sq.csv (separate rocprofv3 --pmc passes; tools/make_traffic_record.py)",
        "workload": {"resolution": resolution, "frames_per_step_per_gpu": batch},
        "kernel_sources_digest": bench.kernel_sources_digest(),
        "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide streaming reads -> x2 (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "launches": launches,
    }
    (REPO_ROOT / "profiles" / "traffic_latest.json").write_text(json.dumps(record, indent=2) + "\n")
    print(f"profiles/traffic_latest.json: {len(launches)} launches from {tag}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
