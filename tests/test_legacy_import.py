"""
CPU tests of the TF-free legacy pickle importer. No legacy pickle exists in the
reference tree, so a synthetic one is written through the published Network state layout by
stand-in `dnnlib` classes that are removed again before the importer runs: the importer must work
with NO dnnlib importable, like on the GPU box.
"""

import pickle
import sys
import types
from pathlib import Path

import numpy as np
import pytest

from gance_amd import legacy_import, network_file
from gance_amd.stylegan2 import spec as sg2_spec


def _write_legacy_pickle(path: Path, variables, drop=None, extra_component_vars=None) -> None:
    """Pickle (G, D, Gs) the way dnnlib does: Network objects whose state is a plain dict."""
    dnnlib = types.ModuleType("dnnlib")
    tflib = types.ModuleType("dnnlib.tflib")
    network_module = types.ModuleType("dnnlib.tflib.network")

    class Network:  # pylint: disable=too-few-public-methods
        def __init__(self, name, variables_, components=None):
            self._state = {
                "version": 4, "name": name, "static_kwargs": {"resolution": 0}, "components": components or {},
                "build_module_src": "raise RuntimeError('must never be executed')", "build_func_name": name,
                "variables": list(variables_.items()),
            }

        def __getstate__(self):
            return self._state

        def __setstate__(self, state):
            raise AssertionError("the real class must not be used to load")

    Network.__module__ = "dnnlib.tflib.network"
    Network.__qualname__ = "Network"
    network_module.Network = Network
    dnnlib.tflib = tflib
    tflib.network = network_module
    sys.modules.update({"dnnlib": dnnlib, "dnnlib.tflib": tflib, "dnnlib.tflib.network": network_module})
    try:
        synthesis = {k[len("G_synthesis/"):]: v for k, v in variables.items() if k.startswith("G_synthesis/") and k != drop}
        synthesis["lod"] = np.float32(0)  # present in real pickles, not a generator weight
        synthesis.update(extra_component_vars or {})
        mapping = {k[len("G_mapping/"):]: v for k, v in variables.items() if k.startswith("G_mapping/")}
        gs = Network("Gs", {"dlatent_avg": variables["dlatent_avg"]},
                     {"synthesis": Network("G_synthesis", synthesis), "mapping": Network("G_mapping", mapping)})
        with open(str(path), "wb") as file:
            pickle.dump((Network("G", {}), Network("D", {}), gs), file, protocol=2)
    finally:
        for name in ("dnnlib", "dnnlib.tflib", "dnnlib.tflib.network"):
            sys.modules.pop(name, None)


def test_legacy_pickle_round_trip_without_dnnlib(tmp_path: Path) -> None:
    variables = sg2_spec.make_random_variables(32, seed=4, perturb=True)
    path = tmp_path / "legacy.pkl"
    _write_legacy_pickle(path, variables)
    assert "dnnlib" not in sys.modules
    resolution, loaded = legacy_import.load_legacy_network(path)
    assert resolution == 32 and set(loaded) == set(variables)
    for name, value in variables.items():
        assert np.array_equal(loaded[name], value) and loaded[name].dtype == np.float32
    # and through the normal entry point used by LoadedNetwork
    via_file = network_file.load_network(path)
    assert via_file.resolution == 32 and np.array_equal(via_file.variables["dlatent_avg"], variables["dlatent_avg"])


def test_other_architectures_and_foreign_pickles_are_refused(tmp_path: Path) -> None:
    variables = sg2_spec.make_random_variables(16, seed=1)
    missing = tmp_path / "missing.pkl"
    _write_legacy_pickle(missing, variables, drop="G_synthesis/16x16/ToRGB/weight")
    with pytest.raises(ValueError, match="no variable 'G_synthesis/16x16/ToRGB/weight'"):
        legacy_import.load_legacy_network(missing)
    with pytest.raises(RuntimeError, match="neither a gance_amd network file"):
        network_file.load_network(missing)
    wrong_shape = dict(variables)
    wrong_shape["G_synthesis/8x8/Conv1/weight"] = np.zeros((3, 3, 256, 512), dtype=np.float32)
    shaped = tmp_path / "shape.pkl"
    _write_legacy_pickle(shaped, wrong_shape)
    with pytest.raises(ValueError, match="config-f expects"):
        legacy_import.load_legacy_network(shaped)
    evil = tmp_path / "evil.pkl"
    with open(str(evil), "wb") as file:
        pickle.dump((1, 2, Path("/tmp")), file)  # pathlib is not on the allow-list
    with pytest.raises(pickle.UnpicklingError, match="refusing to load"):
        legacy_import.load_legacy_network(evil)


@pytest.mark.parametrize(
    "payload",
    [
        b"cbuiltins\neval\n(S'__import__(\"os\").getpid()'\ntR.",
        b"cos\nsystem\n(S'true'\ntR.",
        b"cposix\nsystem\n(S'true'\ntR.",
        b"cbuiltins\nexec\n(S'pass'\ntR.",
        b"cbuiltins\ngetattr\n(cbuiltins\nobject\nS'__subclasses__'\ntR.",
        b"cnumpy\nload\n(S'/etc/hostname'\ntR.",
    ],
)
def test_code_executing_payloads_are_refused_through_load_network(tmp_path: Path, payload: bytes) -> None:
    """A network file is a downloaded artefact: nothing outside the exact allow-list may resolve, on EITHER format's path."""
    path = tmp_path / "evil.pkl"
    path.write_bytes(payload)
    with pytest.raises(pickle.UnpicklingError):
        legacy_import.restricted_load(path)
    with pytest.raises(RuntimeError, match="refusing to load"):
        network_file.load_network(path)


def test_native_network_file_loads_through_the_restricted_unpickler(tmp_path: Path) -> None:
    path = tmp_path / "native.pkl"
    network_file.write_random_network(path, 8, seed=3)
    loaded = network_file.load_network(path)
    assert loaded.resolution == 8
    want = sg2_spec.make_random_variables(8, seed=3)
    assert set(loaded.variables) == set(want)
    for name, value in want.items():
        assert np.array_equal(loaded.variables[name], value)
