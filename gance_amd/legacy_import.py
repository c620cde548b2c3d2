"""
Import a legacy TF1 StyleGAN2 network pickle without TensorFlow.

The reference does `pickle.load(f)[2]` (gance/network_interface/network_functions.py:108-110): the
pickle holds the tuple (G, D, Gs) of `dnnlib.tflib.network.Network` objects, and unpickling one
normally imports dnnlib, re-executes the stored build-function source and needs a TF session.
None of that exists here. A `Network` pickles as a plain state dict
(published dnnlib/tflib/network.py `__getstate__`: version, name, static_kwargs, components,
build_module_src, build_func_name, variables = [(local name, ndarray), ...]), so a restricted
unpickler that maps the class to an inert holder is enough to pull the weights out:

    Gs.components["synthesis"].variables  ->  G_synthesis/<local name>
    Gs.components["mapping"].variables    ->  G_mapping/<local name>
    Gs.variables (dlatent_avg)            ->  dlatent_avg

PARITY UNPINNED: no legacy pickle exists in the reference tree (test/assets/__init__.py:13-14 are
git-ignored), so this is verified structurally only (names, shapes, a synthetic pickle written
through the same state layout in tests/test_legacy_import.py). Only the config-f `skip`
generator (the architecture this engine implements) is accepted; anything else raises.
"""

import io
import pickle
from pathlib import Path
from typing import Any, Dict

import numpy as np

from gance_amd.stylegan2 import spec as sg2_spec

# Exact (module, name) pairs a network pickle may reference: numpy array / scalar reconstruction, plain
# containers and value types. Nothing callable with side effects (`builtins.eval`, `os.system`, ...) resolves.
_NUMPY_CORE = ("numpy.core.multiarray", "numpy._core.multiarray")
_ALLOWED_GLOBALS = frozenset(
    [(module, name) for module in _NUMPY_CORE for name in ("_reconstruct", "scalar")]
    + [("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")]
    + [("collections", "OrderedDict"), ("_codecs", "encode"), ("copyreg", "_reconstructor"), ("copy_reg", "_reconstructor")]
    + [
        (module, name)
        for module in ("builtins", "__builtin__")
        for name in ("set", "frozenset", "dict", "list", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "object", "slice", "range")
    ]
)


class LegacyNetworkState:
    """Inert stand-in for dnnlib.tflib.network.Network: keeps the pickled state, runs nothing."""

    def __init__(self) -> None:
        self.state: Dict[str, Any] = {}

    def __setstate__(self, state: Dict[str, Any]) -> None:
        self.state = dict(state)

    @property
    def name(self) -> str:
        return str(self.state.get("name", ""))

    @property
    def variables(self) -> Dict[str, np.ndarray]:
        return {str(name): np.asarray(value) for name, value in self.state.get("variables", [])}

    @property
    def components(self) -> Dict[str, "LegacyNetworkState"]:
        return dict(self.state.get("components", {}))


class _EasyDict(dict):
    """dnnlib.EasyDict is a dict with attribute access; a plain dict is enough to hold it."""


class _RestrictedUnpickler(pickle.Unpickler):
    """Resolves only what a Network state needs (exact allow-list); nothing else a pickle names is imported."""

    def find_class(self, module: str, name: str) -> Any:
        if module.split(".")[0] == "dnnlib":
            if name == "Network":
                return LegacyNetworkState
            if name == "EasyDict":
                return _EasyDict
            raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a network pickle")
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to load {module}.{name} from a network pickle")


def restricted_load(network_path: Path) -> Any:
    """
    Unpickle a network file with the allow-list above: this implementation's own format (a dict of numpy
    arrays) and legacy (G, D, Gs) pickles both go through it; a payload that names anything else
    (`builtins.eval`, `os.system`, a reduce into arbitrary modules) raises pickle.UnpicklingError.
    """
    with open(str(network_path), "rb") as file:
        return _RestrictedUnpickler(io.BytesIO(file.read())).load()


def _resolution_of(names) -> int:
    sides = [int(name.split("/")[0].split("x")[0]) for name in names if "x" in name.split("/")[0] and name.split("/")[0].split("x")[0].isdigit()]
    if not sides:
        raise ValueError("no NxN scopes among the synthesis variables: not a StyleGAN2 synthesis network")
    return max(sides)


def extract_generator_variables(gs: LegacyNetworkState) -> Dict[str, np.ndarray]:
    """Gs state -> {TF variable name: float32 array} in this implementation's naming."""
    components = gs.components
    if "synthesis" not in components or "mapping" not in components:
        raise ValueError(f"Gs has components {sorted(components)}; expected 'synthesis' and 'mapping'")
    variables: Dict[str, np.ndarray] = {}
    for local, value in components["synthesis"].variables.items():
        variables[f"G_synthesis/{local}"] = value
    for local, value in components["mapping"].variables.items():
        variables[f"G_mapping/{local}"] = value
    for local, value in gs.variables.items():
        variables[local] = value
    return variables


def load_legacy_network(network_path: Path):
    """
    Read (G, D, Gs)[2] out of a TF1 StyleGAN2 pickle.
    :return: (resolution, {name: float32 array}) holding exactly the variables of
    `gance_amd.stylegan2.spec.variable_shapes`.
    :raises ValueError: the pickle is not a config-f skip-architecture StyleGAN2 generator.
    """
    return legacy_network_from_content(restricted_load(network_path))


def legacy_network_from_content(content: Any):
    """`load_legacy_network` on an already (restricted-)unpickled object."""
    if not isinstance(content, (tuple, list)) or len(content) < 3 or not isinstance(content[2], LegacyNetworkState):
        raise ValueError("expected a pickled (G, D, Gs) tuple of dnnlib Networks")
    raw = extract_generator_variables(content[2])
    resolution = _resolution_of(name[len("G_synthesis/"):] for name in raw if name.startswith("G_synthesis/"))
    spec = sg2_spec.make_spec(resolution)
    variables: Dict[str, np.ndarray] = {}
    for name, shape in sg2_spec.variable_shapes(spec).items():
        if name not in raw:
            raise ValueError(f"legacy network has no variable {name!r}: not the config-f skip generator")
        value = np.asarray(raw[name], dtype=np.float32)
        if tuple(value.shape) != tuple(shape):
            raise ValueError(f"legacy variable {name!r} has shape {value.shape}, config-f expects {shape}")
        variables[name] = np.array(value, dtype=np.float32, order="C").reshape(shape)
    return resolution, variables
