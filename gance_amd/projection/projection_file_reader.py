"""
Read a projection file: the on-disk input of the path.

Mirrors the parts of gance/projection/projection_file_reader.py the blend uses
(`ProjectionFileReader` :102-233, `final_latents_matrices_label` :287-300,
`load_projection_file` :457, `verify_projection_file_assumptions` :236-260) and the schema
constants of gance/projection/projector_file_writer.py:71-88,91-169.

Two containers:
* HDF5, the reference's own format (groups `final_latents/final_latents_{i}` of shape (1, W, L)
  float32, gzip-9 + shuffle, `target_images/...`, attributes as file attrs, v1 -> v2 attribute rename),
  read with h5py when it is importable and otherwise with this package's own pure-Python reader of the HDF5
  subset the reference's writer produces (hdf5_lite.py; real h5py-written fixtures in tests/golden/);
* `.npz`, this implementation's container for environments without h5py: arrays `final_latents`
  [F][W][L] float32, optional `target_images` / `final_images` [F][H][W][3] uint8 and a JSON
  string `attributes` with the same attribute names.
"""

import json
from dataclasses import dataclass, field, fields
from pathlib import Path
from types import TracebackType
from typing import Any, Dict, Iterator, Optional, Tuple, Type

import numpy as np

from gance_amd.gance_types import ImageSourceType, RGBInt8ImageType
from gance_amd.vector_sources.vector_types import ConcatenatedMatrices, MatricesLabel, SingleMatrix

LATEST_VERSION = 2
TARGET_IMAGES_GROUP_NAME = "target_images"
FINAL_LATENTS_GROUP_NAME = "final_latents"
FINAL_IMAGE_GROUP_NAME = "final_images"


@dataclass
class ProjectionAttributes:  # pylint: disable=too-many-instance-attributes
    """Metadata of a projection (projector_file_writer.py:91-169); only some fields drive the blend."""

    version_number: int = LATEST_VERSION
    complete: bool = False
    original_target_path: str = ""
    original_width_height: Tuple[int, int] = (0, 0)
    projection_width_height: Tuple[int, int] = (0, 0)
    target_md5_hash: str = ""
    original_network_path: str = ""
    network_md5_hash: str = ""
    steps_in_projection: int = 0
    noises_shapes: Any = None
    latents_histories_enabled: bool = False
    noises_histories_enabled: bool = False
    images_histories_enabled: bool = False
    original_fps: Optional[float] = None
    projection_fps: Optional[float] = None
    original_frame_count: Optional[int] = None
    projection_frame_count: Optional[int] = None
    extra: Dict[str, Any] = field(default_factory=dict)

    @classmethod
    def from_dict(cls, attributes: Dict[str, Any]) -> "ProjectionAttributes":
        """Build from file attributes, applying the v1 -> v2 rename (projection_file_reader.py:116-119)."""
        attributes = dict(attributes)
        if attributes.get("version_number") == 1:
            attributes["original_network_path"] = attributes.pop("original_model_path", "")
            attributes["network_md5_hash"] = attributes.pop("model_md5_hash", "")
            attributes["version_number"] = LATEST_VERSION
        known = {f.name for f in fields(cls)} - {"extra"}
        values = {}
        for key, value in attributes.items():
            if isinstance(value, np.generic):
                value = value.item()
            if isinstance(value, bytes):
                value = value.decode()
            if key in known:
                values[key] = value
        out = cls(**values)
        out.extra = {k: v for k, v in attributes.items() if k not in known}
        out.complete = bool(out.complete)
        return out


def _trailing_int(name: str) -> int:
    return int(name.split("_")[-1])  # `final_latents_12` -> 12 (projection_file_reader.py:60-63)


class ProjectionFileReader:
    """Everything the blend needs from a projection, as lazy iterators."""

    def __init__(self, projection_file_path: Path) -> None:
        self._path = Path(projection_file_path)
        self._h5 = None
        self._npz = None
        if self._path.suffix == ".npz":
            self._npz = np.load(str(self._path), allow_pickle=False)
            self._projection_attributes = ProjectionAttributes.from_dict(json.loads(str(self._npz["attributes"])))
        else:
            # h5py when it is installed; otherwise the pure-Python reader of the subset of HDF5 the reference's
            # writer produces (gance_amd/projection/hdf5_lite.py, pinned by real h5py-written fixtures)
            try:
                import h5py  # pylint: disable=import-outside-toplevel

                self._h5 = h5py.File(name=str(self._path), mode="r")
            except ImportError:
                from gance_amd.projection import hdf5_lite  # pylint: disable=import-outside-toplevel

                self._h5 = hdf5_lite.File(self._path, mode="r")
            self._projection_attributes = ProjectionAttributes.from_dict(dict(self._h5.attrs))

    @property
    def projection_attributes(self) -> ProjectionAttributes:
        """Metadata."""
        return self._projection_attributes

    def _datasets(self, group_name: str) -> Iterator[np.ndarray]:
        if self._npz is not None:
            if group_name in self._npz.files:
                yield from self._npz[group_name]
            return
        group = self._h5[group_name]
        for name in sorted(group.keys(), key=_trailing_int):
            yield np.array(group[name])

    @property
    def final_latents(self) -> Iterator[SingleMatrix]:
        """One (W, L) matrix per projected frame; HDF5 datasets are (1, W, L) (projection_types.py:22-28)."""
        for item in self._datasets(FINAL_LATENTS_GROUP_NAME):
            yield SingleMatrix(item[0] if item.ndim == 3 else item)

    @property
    def target_images(self) -> ImageSourceType:
        """The frames that were projected."""
        return (RGBInt8ImageType(image) for image in self._datasets(TARGET_IMAGES_GROUP_NAME))

    @property
    def final_images(self) -> ImageSourceType:
        """The network's rendering of each final latent."""
        return (RGBInt8ImageType(image) for image in self._datasets(FINAL_IMAGE_GROUP_NAME))

    def close(self) -> None:
        """Release the file."""
        if self._h5 is not None:
            self._h5.close()
        if self._npz is not None:
            self._npz.close()

    def __enter__(self) -> "ProjectionFileReader":
        return self

    def __exit__(
        self,
        exc_type: Optional[Type[BaseException]],
        exc_value: Optional[BaseException],
        traceback: Optional[TracebackType],
    ) -> None:
        self.close()


def load_projection_file(projection_file_path: Path) -> ProjectionFileReader:
    """Open a projection file (projection_file_reader.py:457-464)."""
    return ProjectionFileReader(projection_file_path)


def final_latents_matrices_label(reader: ProjectionFileReader) -> MatricesLabel:
    """All final latents concatenated along the last axis: (W, F*L) (projection_file_reader.py:263-300)."""
    matrices = list(reader.final_latents)
    label = (
        f"{Path(reader.projection_attributes.original_target_path).name} "
        f"proj by {Path(reader.projection_attributes.original_network_path).name}"
    )
    if not matrices:
        raise StopIteration(f"Iterator labeled: {label} was empty!")
    return MatricesLabel(
        data=ConcatenatedMatrices(np.concatenate(matrices, axis=-1)), vector_length=matrices[0].shape[-1], label=label
    )


def load_final_latents_matrices_label(projection_file_path: Path) -> MatricesLabel:
    """Convenience wrapper (projection_file_reader.py:374-384)."""
    with load_projection_file(projection_file_path) as reader:
        return final_latents_matrices_label(reader)


def verify_projection_file_assumptions(projection_file_path: Path) -> None:
    """Every row of every final latent matrix is identical (projection_file_reader.py:236-260)."""
    with load_projection_file(projection_file_path) as reader:
        for matrix in reader.final_latents:
            for row in matrix:
                assert np.array_equal(matrix[0], row)


def write_projection_npz(  # pylint: disable=too-many-arguments
    path: Path,
    final_latents: np.ndarray,
    projection_fps: float,
    complete: bool = True,
    target_images: Optional[np.ndarray] = None,
    original_target_path: str = "synthetic.mp4",
    original_network_path: str = "synthetic.pkl",
) -> None:
    """Write the `.npz` container (synthetic projections for tests and benchmarks)."""
    latents = np.asarray(final_latents, dtype=np.float32)
    attributes = {
        "version_number": LATEST_VERSION,
        "complete": bool(complete),
        "original_target_path": original_target_path,
        "original_network_path": original_network_path,
        "projection_fps": float(projection_fps),
        "projection_frame_count": int(latents.shape[0]),
        "original_fps": float(projection_fps),
        "original_frame_count": int(latents.shape[0]),
    }
    arrays = {"attributes": np.array(json.dumps(attributes)), FINAL_LATENTS_GROUP_NAME: latents}
    if target_images is not None:
        arrays[TARGET_IMAGES_GROUP_NAME] = np.asarray(target_images, dtype=np.uint8)
    np.savez(str(path), **arrays)
