"""CPU tests of the projection-file reader on the .npz container."""

from pathlib import Path

import numpy as np
import pytest

from gance_amd import synthetic
from gance_amd.projection import projection_file_reader as pfr


@pytest.fixture
def projection_path(tmp_path: Path) -> Path:
    latents = synthetic.synthetic_final_latents(6, 512, seed=3).reshape(18, 6, 512).transpose(1, 0, 2)
    images = np.random.RandomState(0).randint(0, 256, (6, 8, 8, 3)).astype(np.uint8)
    path = tmp_path / "projection.npz"
    pfr.write_projection_npz(path, latents, projection_fps=30.0, target_images=images)
    return path


def test_reader_round_trip(projection_path: Path) -> None:
    with pfr.load_projection_file(projection_path) as reader:
        attributes = reader.projection_attributes
        assert attributes.complete and attributes.projection_fps == 30.0 and attributes.projection_frame_count == 6
        matrices = list(reader.final_latents)
        assert len(matrices) == 6 and matrices[0].shape == (18, 512) and matrices[0].dtype == np.float32
        assert len(list(reader.target_images)) == 6
        label = pfr.final_latents_matrices_label(reader)
    assert label.data.shape == (18, 6 * 512) and label.vector_length == 512  # test_projection_file.py shapes
    assert label.label == "synthetic.mp4 proj by synthetic.pkl"
    assert np.array_equal(label.data, synthetic.synthetic_final_latents(6, 512, seed=3))
    assert np.array_equal(pfr.load_final_latents_matrices_label(projection_path).data, label.data)
    pfr.verify_projection_file_assumptions(projection_path)


def test_version_one_attribute_rename() -> None:
    attributes = pfr.ProjectionAttributes.from_dict(
        {"version_number": 1, "complete": np.bool_(True), "original_model_path": "a.pkl", "model_md5_hash": "ff", "projection_fps": 15.0}
    )
    assert attributes.version_number == 2 and attributes.original_network_path == "a.pkl" and attributes.network_md5_hash == "ff"
    assert attributes.complete is True


def test_broken_assumption_is_detected(tmp_path: Path) -> None:
    latents = np.random.RandomState(0).randn(2, 18, 512).astype(np.float32)  # rows differ
    path = tmp_path / "bad.npz"
    pfr.write_projection_npz(path, latents, projection_fps=30.0)
    with pytest.raises(AssertionError):
        pfr.verify_projection_file_assumptions(path)


@pytest.mark.parametrize("name,frames", [("projection_v2", 12), ("projection_v1", 3)])
def test_real_hdf5_projection_files_are_read(golden_dir: Path, name: str, frames: int) -> None:
    """
    The reference's own container: files WRITTEN WITH h5py in the reference writer's layout
    (oracle/make_hdf5_fixture.py: gzip-9 + shuffle chunked datasets named <group>_<index>, attributes through
    f.attrs.update, version-1 attribute names in the second file), read through ProjectionFileReader -- with h5py
    if this interpreter has it, otherwise with gance_amd/projection/hdf5_lite.py.
    """
    import json  # pylint: disable=import-outside-toplevel

    expected = np.load(golden_dir / "projection_hdf5_expected.npz")
    want_attributes = json.loads(str(expected[f"{name}_attrs"]))
    path = golden_dir / f"{name}.hdf5"
    with pfr.load_projection_file(path) as reader:
        attributes = reader.projection_attributes
        assert attributes.version_number == 2 and attributes.complete is True
        assert attributes.projection_fps == 15.0 and attributes.projection_frame_count == frames
        assert attributes.original_frame_count == 4 * frames and attributes.original_fps == 59.94
        assert attributes.original_target_path == want_attributes["original_target_path"]
        assert attributes.original_network_path == want_attributes.get("original_network_path", want_attributes.get("original_model_path"))
        assert attributes.network_md5_hash == "fedcba9876543210fedcba9876543210"
        assert list(attributes.original_width_height) == [1920, 1080]
        latents = list(reader.final_latents)
        assert len(latents) == frames and latents[0].shape == (18, 512) and latents[0].dtype == np.float32
        assert np.array_equal(np.stack(latents), expected[f"{name}_latents"])  # in frame order: _10 after _9, not after _1
        assert np.array_equal(np.stack(list(reader.target_images)), expected[f"{name}_targets"])
        assert np.array_equal(np.stack(list(reader.final_images)), expected[f"{name}_finals"])
        label = pfr.final_latents_matrices_label(reader)
    assert label.data.shape == (18, frames * 512) and label.vector_length == 512
    assert label.label == f"{Path(attributes.original_target_path).name} proj by {Path(attributes.original_network_path).name}"
    pfr.verify_projection_file_assumptions(path)


def test_hdf5_reader_refuses_what_it_does_not_implement(tmp_path: Path) -> None:
    from gance_amd.projection import hdf5_lite  # pylint: disable=import-outside-toplevel

    path = tmp_path / "truncated.hdf5"
    path.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes([3]) + bytes(100))  # superblock version 3 (libver="latest")
    with pytest.raises(hdf5_lite.UnsupportedHdf5, match="superblock version 3"):
        hdf5_lite.File(path)
    path.write_bytes(b"not an hdf5 file at all")
    with pytest.raises(hdf5_lite.UnsupportedHdf5):
        hdf5_lite.File(path)
