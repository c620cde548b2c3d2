"""
Latents -> frames: the caller loop of the hot path.

`vector_synthesis` keeps the reference's signature and lazy-iterator contract
(gance/data_into_network_visualization/network_visualization.py:462-690) for the synthesis side;
the matplotlib debug visualisations (`enable_2d` / `enable_3d`, :54-400, :542-596) are out of scope
and yield `visualization_images=None`.

Differences by design: frames are synthesised in batches, and because `MultiNetwork` keeps every
network resident the reference's "sort frames by network, spill each to a gzip-HDF5 temp file, reload
in order" detour (:653-674) disappears -- frames of a batch are grouped by network index in memory
and emitted in frame order.
"""

from typing import Iterator, List, NamedTuple, Optional

import numpy as np

from gance_amd.data_into_network_visualization.visualization_common import VisualizationInput
from gance_amd.gance_types import ImageSourceType, RGBInt8ImageType
from gance_amd.logger_common import LOGGER
from gance_amd.network_interface.network_functions import MultiNetwork
from gance_amd.vector_sources.vector_sources_common import sub_vectors


class SynthesisOutput(NamedTuple):
    """The two image sources of a synthesis run (network_visualization.py:403-409)."""

    synthesized_images: Optional[ImageSourceType]
    visualization_images: Optional[ImageSourceType]


def _batched_frames(
    samples: np.ndarray, indices: List[int], networks: MultiNetwork, batch: int
) -> Iterator[RGBInt8ImageType]:
    """Synthesize `samples[f]` on network `indices[f]`, `batch` frames at a time, in frame order."""
    total = len(samples)
    for start in range(0, total, batch):
        stop = min(total, start + batch)
        chunk_indices = np.asarray(indices[start:stop])
        frames: List[Optional[np.ndarray]] = [None] * (stop - start)
        for network_index in np.unique(chunk_indices):
            members = np.nonzero(chunk_indices == network_index)[0]
            images = networks.indexed_create_images_generic(int(network_index), samples[start:stop][members])
            for slot, image in zip(members, images):
                frames[slot] = image
        for offset, frame in enumerate(frames):
            LOGGER.info(f"Rendered frame #{start + offset}")
            yield RGBInt8ImageType(frame)


def vector_synthesis(  # pylint: disable=too-many-arguments,unused-argument
    data: VisualizationInput,
    networks: Optional[MultiNetwork],
    default_vector_length: Optional[int] = 1024,
    visualization_height: Optional[int] = None,
    enable_3d: bool = False,
    enable_2d: bool = True,
    frames_to_visualize: Optional[int] = None,
    network_index_window_width: Optional[int] = None,
    force_optimize_synthesis_order: bool = True,
    unload_networks_when_complete: bool = False,
) -> SynthesisOutput:
    """
    For every vector (1-D `combined`) or matrix (2-D `combined`) in `data.combined`, synthesize
    the frame on the network `data.network_indices` selects. Frames come back lazily, at the
    network's native size, as uint8 (H, W, 3) RGB.
    :raises ValueError: nothing to render (no networks and no visualisation requested), as in the
    reference (:513-514).
    """
    if not enable_3d and not enable_2d and networks is None:
        raise ValueError("Nothing to render!")
    if networks is None:
        LOGGER.warning("matplotlib visualisations are out of scope here; nothing to synthesize without networks")
        return SynthesisOutput(synthesized_images=None, visualization_images=None)

    vector_length = networks.expected_vector_length
    samples = sub_vectors(data=data.combined.data, vector_length=vector_length)  # (N, L) or (N, W, L)
    indices = [int(index) for index in data.network_indices.result.data]
    if frames_to_visualize is not None:
        samples = samples[:frames_to_visualize]
    indices = indices[: len(samples)]

    def frames() -> Iterator[RGBInt8ImageType]:
        yield from _batched_frames(samples, indices, networks, networks.max_batch)  # full engine calls: the capacity the networks were loaded with
        if unload_networks_when_complete:
            networks.unload()

    return SynthesisOutput(synthesized_images=frames(), visualization_images=None)
