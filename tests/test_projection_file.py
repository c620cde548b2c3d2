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


def test_hdf5_without_h5py_fails_loudly(tmp_path: Path) -> None:
    try:
        import h5py  # noqa: F401  pylint: disable=unused-import,import-outside-toplevel
        pytest.skip("h5py is installed here")
    except ImportError:
        pass
    path = tmp_path / "projection.hdf5"
    path.write_bytes(b"\x89HDF\r\n\x1a\n")
    with pytest.raises(RuntimeError, match="needs h5py"):
        pfr.load_projection_file(path)
