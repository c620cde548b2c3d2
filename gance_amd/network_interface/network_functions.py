"""
Load networks and turn vectors / matrices into images: the drop-in boundary.

Same public names, argument meaning and error behaviour as
gance/network_interface/network_functions.py (line numbers cited per symbol), re-designed for one
MI355X per process:

* no worker subprocess, no Queues, no TF session: a network is a `hip_lib.Engine` (weights and
  workspace resident in HBM), freed deterministically by `stop_function` / `unload`;
* `MultiNetwork` keeps EVERY distinct network resident (120 MB each against 288 GB), so switching
  the index per frame costs nothing and the reference's sort-frames-to-disk trick
  (network_visualization.py:653-674) is not needed;
* every image function also has a batched form (`create_images_*`).

There is no CPU fallback: without libgance_hip.so or without a GPU, loading raises.
"""

import json
import logging
import os
import typing
from functools import wraps
from pathlib import Path
from typing import Callable, Dict, List, NamedTuple, Optional, Union

import numpy as np
import pydantic
import torch
from pydantic import BaseModel, FilePath
from typing_extensions import Protocol

from gance_amd import hip_lib, network_file
from gance_amd.gance_types import RGBInt8ImageType
from gance_amd.logger_common import LOGGER
from gance_amd.vector_sources.vector_types import SingleMatrix, SingleVector, is_vector

NETWORK_SUFFIX = ".pkl"  # network_functions.py:38
TRUNCATION_PSI = 1.2  # network_functions.py:124,155
# frames per engine call of the batched entry points: the batch the kernels are tuned for (64: 1350 frames/s class at 1024^2; 16: -12 %),
# about 0.8 GB of activation workspace per frame of capacity at 1024^2 (51 of the 288 GB), shared by every resident network
DEFAULT_MAX_BATCH = 64
DEFAULT_DEVICE = 0


def sorted_networks_in_directory(networks_directory: Path) -> List[Path]:
    """`.pkl` files of a directory, sorted by name (network_functions.py:41-48)."""
    return sorted(networks_directory.glob(f"*{NETWORK_SUFFIX}"))


class ImageFunction(Protocol):  # pylint: disable=too-few-public-methods
    """Vector or matrix in, image out (network_functions.py:51-63)."""

    def __call__(self: "ImageFunction", data: Union[SingleVector, SingleMatrix]) -> RGBInt8ImageType:
        """(L,) or (W, L) of any float dtype -> C-contiguous uint8 (H, W, 3) RGB owned by the caller."""


class NetworkInterface(NamedTuple):
    """The necessary parts of a network (network_functions.py:66-78)."""

    expected_vector_length: int
    create_image_vector: ImageFunction
    create_image_matrix: ImageFunction
    create_image_generic: ImageFunction


class NetworkInterfaceInProcess(NamedTuple):
    """A network interface plus the function that frees it (network_functions.py:207-214)."""

    network_interface: NetworkInterface
    stop_function: Callable[[], None]


class LoadedNetwork:
    """One generator resident in HBM and its image functions (single and batched)."""

    def __init__(self, network_path: Path, max_batch: int = DEFAULT_MAX_BATCH, device: Optional[int] = None) -> None:
        """`device` None: the process's current GPU (one process per GPU: `torch.cuda.set_device(LOCAL_RANK)` decides)."""
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else DEFAULT_DEVICE
        loaded = network_file.load_network(network_path)  # RuntimeError if not loadable
        self.network_path = network_path
        self.engine = hip_lib.Engine(loaded.variables, loaded.resolution, max_batch=max_batch, device=device)
        self.max_batch = max_batch

    @property
    def expected_vector_length(self) -> int:
        """`network.input_shape[1]` (network_functions.py:191)."""
        return self.engine.vector_length

    def create_images_vector(self, data: np.ndarray, randomize_noise: bool = True, noise_seed: Optional[int] = None) -> np.ndarray:
        """
        z (B, L) -> uint8 (B, H, W, 3): mapping, truncation psi = 1.2, synthesis (network_functions.py:144-158).
        The reference does not pass `randomize_noise` on this path, so the upstream default True applies: fresh noise per
        call, one plane per layer and per frame as upstream's `tf.random_normal([N, 1, H, W])` (`noise_seed` makes it
        repeatable: frame k of `data` then reads the planes of (noise_seed, k) however the frames are batched). A network
        whose noise strengths are all zero -- every random-init network -- gives the same image either way.
        """
        data = np.asarray(data)
        out = []
        if randomize_noise and noise_seed is None:
            noise_seed = int.from_bytes(os.urandom(8), "little")
        for start in range(0, len(data), self.max_batch):
            if randomize_noise:
                self.engine.randomize_noise(noise_seed, count=min(self.max_batch, len(data) - start), first_sample=start)
            else:
                self.engine.restore_noise()
            out.append(self.engine.synthesize_z(data[start : start + self.max_batch], truncation_psi=TRUNCATION_PSI))
        return np.concatenate(out) if len(out) > 1 else out[0]

    def create_images_matrix(self, data: np.ndarray) -> np.ndarray:
        """dlatents (B, W, L) -> uint8 (B, H, W, 3): synthesis only, stored noise (network_functions.py:160-169: randomize_noise=False)."""
        self.engine.restore_noise()
        data = np.asarray(data)
        out = [self.engine.synthesize_w(data[start : start + self.max_batch]) for start in range(0, len(data), self.max_batch)]
        return np.concatenate(out) if len(out) > 1 else out[0]

    def create_image_vector(self, data: SingleVector) -> RGBInt8ImageType:
        """One z vector -> one image; the callee adds the batch axis (network_functions.py:127-134)."""
        return RGBInt8ImageType(self.create_images_vector(np.reshape(data, (1, *np.shape(data))))[0])

    def create_image_matrix(self, data: SingleMatrix) -> RGBInt8ImageType:
        """One (W, L) latent matrix -> one image."""
        return RGBInt8ImageType(self.create_images_matrix(np.reshape(data, (1, *np.shape(data))))[0])

    def create_image_generic(self, data: Union[SingleVector, SingleMatrix]) -> RGBInt8ImageType:
        """Dispatch on the number of axes (network_functions.py:171-183, vector_types.py:58-68)."""
        if is_vector(data):
            LOGGER.info(f"Generic -> Vector, shape: {data.shape}")
            return self.create_image_vector(data)
        LOGGER.info(f"Generic -> matrix, shape: {data.shape}")
        return self.create_image_matrix(data)

    def interface(self) -> NetworkInterface:
        """The reference's NamedTuple view of this network."""
        return NetworkInterface(
            expected_vector_length=self.expected_vector_length,
            create_image_vector=self.create_image_vector,
            create_image_matrix=self.create_image_matrix,
            create_image_generic=self.create_image_generic,
        )

    def stop(self) -> None:
        """Free the network's HBM. Safe to call twice."""
        self.engine.close()


def create_network_interface(network_path: Path, call_init_function: bool = True) -> NetworkInterface:  # pylint: disable=unused-argument
    """
    Load a network and expose it (network_functions.py:195-204). `call_init_function` is accepted
    for signature compatibility; there is no TF session to initialise.
    """
    return LoadedNetwork(network_path).interface()


def create_network_interface_process(network_path: Path) -> NetworkInterfaceInProcess:
    """
    Load a network and return its interface with a `stop_function` that frees it
    (network_functions.py:232-340). The reference needs a child process because TF1 cannot unload;
    an Engine frees its HBM on `close()`, so this stays in-process.
    :raises RuntimeError: if the file cannot be loaded (re-raised in the caller like the
    reference's startup error, network_functions.py:272-278).
    """
    loaded = LoadedNetwork(network_path)
    return NetworkInterfaceInProcess(network_interface=loaded.interface(), stop_function=loaded.stop)


@typing.no_type_check
def _raise_exception_if_unloaded(function):
    """Guard for MultiNetwork members that need loaded networks (network_functions.py:451-481)."""

    @wraps(function)
    def wrapper(*args, **kwargs):
        self = args[0]
        if self._loaded is None or self._expected_vector_length is None:  # pylint: disable=protected-access
            raise ValueError("Multinetwork is not initialized! Call load or use context manager.")
        return function(*args, **kwargs)

    return wrapper


class MultiNetwork:
    """
    Switch between networks by index during a run (network_functions.py:484-640). All distinct
    network files are resident at once; an index change is a dictionary lookup.
    """

    def __init__(
        self: "MultiNetwork", network_paths: List[Path], load: bool = False, max_batch: int = DEFAULT_MAX_BATCH, device: Optional[int] = None
    ) -> None:
        """
        `max_batch` (not in the reference): frames per engine call of the batched entry points; `device` (not in the
        reference): the GPU the networks live on, default the process's current one.
        """
        self._max_batch = max_batch
        self._device = device
        self._network_paths: List[Path] = network_paths
        self._loaded: Optional[Dict[Path, LoadedNetwork]] = None
        self._expected_vector_length: Optional[int] = None
        if load:
            self.load()

    @property  # type: ignore
    @_raise_exception_if_unloaded
    def expected_vector_length(self: "MultiNetwork") -> int:
        """Vector length reported by the first network (network_functions.py:609-614)."""
        return self._expected_vector_length

    def __enter__(self: "MultiNetwork") -> Optional["MultiNetwork"]:
        """Load; `None` if the networks cannot be loaded onto the GPU (network_functions.py:516-529)."""
        try:
            self.load()
        except RuntimeError:
            logging.warning("Couldn't load network into GPU, proceeding without network.")
            return None
        return self

    @typing.no_type_check
    def __exit__(self, exec_type, exec_value, exec_traceback) -> None:
        if self._loaded is not None:
            self.unload()

    def load(self: "MultiNetwork") -> None:
        """Load every distinct network file into HBM (network_functions.py:604-614)."""
        loaded: Dict[Path, LoadedNetwork] = {}
        try:
            for path in self._network_paths:
                if path not in loaded:
                    LOGGER.info(f"Loading network: {path}")
                    loaded[path] = LoadedNetwork(path, max_batch=self._max_batch, device=self._device)
        except Exception:
            for network in loaded.values():
                network.stop()
            raise
        self._loaded = loaded
        self._expected_vector_length = loaded[self._network_paths[0]].expected_vector_length

    @_raise_exception_if_unloaded
    def unload(self: "MultiNetwork") -> None:
        """Free every network (network_functions.py:616-623). The object can be loaded again."""
        for network in self._loaded.values():
            network.stop()
        self._loaded = None
        self._expected_vector_length = None

    @property
    def max_batch(self: "MultiNetwork") -> int:
        """Frames per engine call the resident networks were created for (not in the reference)."""
        return self._max_batch

    def _network_at(self: "MultiNetwork", index: int) -> LoadedNetwork:
        return self._loaded[self._network_paths[index]]

    @_raise_exception_if_unloaded
    def indexed_create_image_vector(self: "MultiNetwork", index: int, data: SingleVector) -> RGBInt8ImageType:
        """Image of network `index` for a z vector (network_functions.py:565-576)."""
        return self._network_at(index).create_image_vector(data)

    @_raise_exception_if_unloaded
    def indexed_create_image_matrix(self: "MultiNetwork", index: int, data: SingleMatrix) -> RGBInt8ImageType:
        """Image of network `index` for a latent matrix (network_functions.py:578-589)."""
        return self._network_at(index).create_image_matrix(data)

    @_raise_exception_if_unloaded
    def indexed_create_image_generic(
        self: "MultiNetwork", index: int, data: Union[SingleVector, SingleMatrix]
    ) -> RGBInt8ImageType:
        """Image of network `index` for either input kind (network_functions.py:591-602)."""
        return self._network_at(index).create_image_generic(data)

    @_raise_exception_if_unloaded
    def indexed_create_images_generic(self: "MultiNetwork", index: int, data: np.ndarray) -> np.ndarray:
        """Batched form: (B, L) z vectors or (B, W, L) latent matrices -> (B, H, W, 3) uint8."""
        network = self._network_at(index)
        return network.create_images_vector(data) if np.ndim(data) == 2 else network.create_images_matrix(data)

    @property
    def network_indices(self: "MultiNetwork") -> List[int]:
        """Candidate indices (network_functions.py:625-632)."""
        return list(range(len(self._network_paths)))

    @property
    def network_paths(self: "MultiNetwork") -> List[Path]:
        """The network files (network_functions.py:634-640)."""
        return self._network_paths


class NetworksFile(BaseModel):
    """A `.json` file listing network paths (network_functions.py:685-690)."""

    networks: List[FilePath]


def parse_network_paths(
    networks_directory: Optional[str], networks: Optional[List[str]], networks_json: Optional[str]
) -> List[Path]:
    """
    Resolve the CLI's three ways of naming networks into one list (network_functions.py:643-682).
    :raises ValueError: bad JSON, unreadable JSON, or no network at all.
    """
    all_networks: List[Path] = []
    if networks_directory is not None:
        all_networks += sorted_networks_in_directory(networks_directory=Path(networks_directory))
    if networks is not None:
        all_networks += [Path(network) for network in networks]
    if networks_json is not None:
        LOGGER.info(f"Loading network JSON: {networks_json}")
        try:
            with open(networks_json) as file:
                all_networks += [Path(path) for path in NetworksFile(**json.load(file)).networks]
        except pydantic.ValidationError as error:
            raise ValueError("Ran into formatting problem with networks JSON.") from error
        except Exception as error:
            raise ValueError("Couldn't open networks JSON.") from error
    if not all_networks:
        raise ValueError("No networks given, cannot continue.")
    LOGGER.info("Discovered networks: ")
    for path in all_networks:
        LOGGER.info(f"\t{path}")
    return all_networks
