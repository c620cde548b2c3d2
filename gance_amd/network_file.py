"""
Network files of this implementation.

The reference loads a TF1 StyleGAN2 pickle and takes element [2], `Gs`
(gance/network_interface/network_functions.py:108-110). Unpickling that needs TensorFlow 1.x and
the un-vendored dnnlib, neither of which exists here, so this implementation stores a generator as
a plain pickle of numpy arrays under the SAME TF variable names (gance_amd/stylegan2/spec.py): a
legacy importer only has to copy arrays by name (gance_amd/legacy_import.py). The file suffix
stays `.pkl` so `sorted_networks_in_directory` / `parse_network_paths` behave identically.
"""

import pickle
from pathlib import Path
from typing import Dict, NamedTuple

import numpy as np

from gance_amd.stylegan2 import spec as sg2_spec

FORMAT = "gance_amd.stylegan2.v1"


class NetworkFile(NamedTuple):
    """A generator on disk."""

    resolution: int
    variables: Dict[str, np.ndarray]


def save_network(path: Path, resolution: int, variables: Dict[str, np.ndarray]) -> None:
    """Write a generator (raw, un-scaled TF-named variables)."""
    spec = sg2_spec.make_spec(resolution)
    sg2_spec.pack_variables(variables, spec)  # validates names and shapes
    with open(str(path), "wb") as file:
        pickle.dump({"format": FORMAT, "resolution": int(resolution), "variables": dict(variables)}, file, protocol=4)


def load_network(path: Path) -> NetworkFile:
    """
    Read a generator.
    :raises RuntimeError: if the file is not in this implementation's format (e.g. a legacy TF
    pickle), the error class the reference's callers already handle (network_functions.py:523-529).
    """
    from gance_amd import legacy_import  # pylint: disable=import-outside-toplevel

    # Network files are downloaded artefacts: both formats are read through ONE restricted unpickler
    # (exact allow-list of numpy / container globals, legacy_import.py); plain pickle.load never runs.
    try:
        content = legacy_import.restricted_load(path)
        if isinstance(content, dict) and content.get("format") == FORMAT:
            return NetworkFile(int(content["resolution"]), content["variables"])
        resolution, variables = legacy_import.legacy_network_from_content(content)
    except Exception as error:  # pylint: disable=broad-except
        raise RuntimeError(
            f"{path} is neither a gance_amd network file (format tag {FORMAT!r}) nor an importable "
            f"TF1 StyleGAN2 (G, D, Gs) pickle: {error}"
        ) from error
    return NetworkFile(resolution, variables)


def write_random_network(path: Path, resolution: int, seed: int = 0) -> None:
    """Random-init generator as a network file."""
    save_network(path, resolution, sg2_spec.make_random_variables(resolution, seed=seed))
