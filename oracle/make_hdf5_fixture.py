"""
ORACLE tooling (build container only): write a small, REAL HDF5 projection file with h5py, in the layout the
reference's writer produces (gance/projection/projector_file_writer.py:71-88 group names, :91-169 attribute
fields written through `f.attrs.update(attributes.to_dict())` :728-734, :814-834 one gzip-9 + shuffle dataset
per frame named `<group>_<index>`, final latents of shape (1, 18, 512) float32).

h5py is not installed in the interpreter the product runs on; the image carries one under /opt/conda:

    /opt/conda/bin/python3.9 oracle/make_hdf5_fixture.py

Writes tests/golden/projection_v2.hdf5 (6 projected frames), tests/golden/projection_v1.hdf5 (the version-1
attribute names the reader renames, projection_file_reader.py:116-119) and tests/golden/projection_hdf5_expected.npz
(the arrays and attributes that were written, for the pure-Python reader's test).
"""

import json
from pathlib import Path

import h5py
import numpy as np

GOLDEN_DIR = Path(__file__).resolve().parent.parent / "tests" / "golden"
COMPRESSION_LEVEL = 9  # projector_file_writer.py:88


def create_dataset(group, name: str, data: np.ndarray) -> None:
    """`_create_dataset_wrapper` (projector_file_writer.py:814-834)."""
    group.create_dataset(
        f"/{group.name}/{name}", shape=data.shape, dtype=data.dtype, data=data, compression="gzip",
        compression_opts=COMPRESSION_LEVEL, shuffle=True,
    )


def attributes(version: int, frames: int, complete: bool) -> dict:
    """The fields of ProjectionAttributes (projector_file_writer.py:91-169), as `to_dict()` hands them to h5py."""
    out = {
        "version_number": version,
        "complete": complete,
        "original_target_path": "/videos/some video (take 2).mp4",
        "original_width_height": (1920, 1080),
        "projection_width_height": (1024, 1024),
        "target_md5_hash": "0123456789abcdef0123456789abcdef",
        "steps_in_projection": 1000,
        "noises_shapes": np.nan,
        "latents_histories_enabled": False,
        "noises_histories_enabled": False,
        "images_histories_enabled": False,
        "original_fps": 59.94,
        "projection_fps": 15.0,
        "original_frame_count": 4 * frames,
        "projection_frame_count": frames,
    }
    if version == 1:
        out["original_model_path"] = "/networks/old name.pkl"
        out["model_md5_hash"] = "fedcba9876543210fedcba9876543210"
    else:
        out["original_network_path"] = "/networks/network-snapshot-000123.pkl"
        out["network_md5_hash"] = "fedcba9876543210fedcba9876543210"
    return out


def write(path: Path, version: int, frames: int, rng: np.random.RandomState) -> dict:
    latents = [np.tile(rng.randn(1, 1, 512).astype(np.float32), (1, 18, 1)) for _ in range(frames)]
    targets = [rng.randint(0, 256, size=(24, 24, 3)).astype(np.uint8) for _ in range(frames)]
    finals = [rng.randint(0, 256, size=(24, 24, 3)).astype(np.uint8) for _ in range(frames)]
    attrs = attributes(version, frames, complete=False)
    with h5py.File(name=str(path), mode="w") as f:
        f.attrs.update(attrs)  # written once incomplete, updated at the end, like the reference (:728-734, :798-803)
        groups = {name: f.create_group(name) for name in ("target_images", "final_latents", "final_images")}
        for name in ("latents_histories", "images_histories", "noises_histories"):
            f.create_group(name)
        # frames are appended out of lexicographic order on purpose: readers sort by the trailing integer (:60-63)
        for index in list(range(frames)):
            create_dataset(groups["target_images"], f"target_images_{index}", targets[index])
            create_dataset(groups["final_latents"], f"final_latents_{index}", latents[index])
            create_dataset(groups["final_images"], f"final_images_{index}", finals[index])
            f.flush()
        attrs = attributes(version, frames, complete=True)
        f.attrs.update(attrs)
    return {"latents": np.concatenate(latents), "targets": np.stack(targets), "finals": np.stack(finals), "attrs": attrs}


def main() -> None:
    rng = np.random.RandomState(2024)
    expected = {}
    for version, frames, name in ((2, 12, "projection_v2"), (1, 3, "projection_v1")):
        result = write(GOLDEN_DIR / f"{name}.hdf5", version, frames, rng)
        expected[f"{name}_latents"] = result["latents"]
        expected[f"{name}_targets"] = result["targets"]
        expected[f"{name}_finals"] = result["finals"]
        expected[f"{name}_attrs"] = np.array(
            json.dumps({k: (None if isinstance(v, float) and np.isnan(v) else (list(v) if isinstance(v, tuple) else v)) for k, v in result["attrs"].items()})
        )
        print(f"wrote {name}.hdf5 ({(GOLDEN_DIR / (name + '.hdf5')).stat().st_size} bytes)")
    np.savez_compressed(GOLDEN_DIR / "projection_hdf5_expected.npz", **expected)


if __name__ == "__main__":
    main()
