"""
CPU tests of the drop-in boundary's host logic. They read like the reference's own tests
(test/test_network_functions.py:29-97: a fake network whose image functions return a constant
array, to pin MultiNetwork's load / unload / guard semantics without a GPU) plus the path parsing
and type helpers.
"""

import json
from pathlib import Path

import numpy as np
import pytest

from gance_amd import network_file
from gance_amd.data_into_network_visualization import network_visualization
from gance_amd.data_into_network_visualization.visualization_common import DataLabel, ResultLayers, VisualizationInput
from gance_amd.network_interface import network_functions
from gance_amd.stylegan2 import spec as sg2_spec
from gance_amd.vector_sources import vector_sources_common as vsc
from gance_amd.vector_sources.vector_types import MatricesLabel, VectorsLabel, is_vector

EXPECTED_VECTOR_LENGTH = 512


class FakeLoadedNetwork:
    """Stands in for LoadedNetwork: constant 10x10x3 images tagged with the network's identity."""

    instances = []

    def __init__(self, network_path: Path, max_batch: int = 8, device: int = 0) -> None:
        if "broken" in network_path.name:
            raise RuntimeError("cannot load")
        self.network_path = network_path
        self.stopped = False
        self.tag = int(network_path.stem.split("_")[-1])
        FakeLoadedNetwork.instances.append(self)

    expected_vector_length = EXPECTED_VECTOR_LENGTH

    def _image(self) -> np.ndarray:
        return np.full((10, 10, 3), self.tag, dtype=np.uint8)

    def create_image_vector(self, data):
        assert is_vector(data)
        return self._image()

    def create_image_matrix(self, data):
        assert not is_vector(data)
        return self._image()

    def create_image_generic(self, data):
        return self._image()

    def create_images_vector(self, data):
        return np.stack([self._image() for _ in data])

    create_images_matrix = create_images_vector

    def stop(self) -> None:
        self.stopped = True


@pytest.fixture
def fake_networks(monkeypatch):
    FakeLoadedNetwork.instances = []
    monkeypatch.setattr(network_functions, "LoadedNetwork", FakeLoadedNetwork)
    return [Path("net_0.pkl"), Path("net_1.pkl"), Path("net_0.pkl")]


def test_multinetwork_guards_before_load(fake_networks) -> None:
    multi = network_functions.MultiNetwork(network_paths=fake_networks)
    with pytest.raises(ValueError, match="Multinetwork is not initialized"):
        multi.indexed_create_image_vector(0, np.zeros(EXPECTED_VECTOR_LENGTH))
    with pytest.raises(ValueError):
        _ = multi.expected_vector_length
    with pytest.raises(ValueError):
        multi.unload()
    assert multi.network_indices == [0, 1, 2]
    assert multi.network_paths == fake_networks


def test_multinetwork_context_manager_loads_and_stops(fake_networks) -> None:
    with network_functions.MultiNetwork(network_paths=fake_networks) as multi:
        assert multi.expected_vector_length == EXPECTED_VECTOR_LENGTH
        assert len(FakeLoadedNetwork.instances) == 2  # the same path twice is loaded once
        vector = np.zeros(EXPECTED_VECTOR_LENGTH)
        matrix = np.zeros((18, EXPECTED_VECTOR_LENGTH))
        assert multi.indexed_create_image_vector(0, vector)[0, 0, 0] == 0
        assert multi.indexed_create_image_matrix(1, matrix)[0, 0, 0] == 1
        assert multi.indexed_create_image_generic(2, matrix)[0, 0, 0] == 0  # index 2 is net_0 again
        assert multi.indexed_create_images_generic(1, np.zeros((3, 18, EXPECTED_VECTOR_LENGTH))).shape == (3, 10, 10, 3)
    assert all(network.stopped for network in FakeLoadedNetwork.instances)
    with pytest.raises(ValueError):
        multi.indexed_create_image_vector(0, np.zeros(EXPECTED_VECTOR_LENGTH))


def test_multinetwork_enter_returns_none_when_loading_fails(monkeypatch) -> None:
    monkeypatch.setattr(network_functions, "LoadedNetwork", FakeLoadedNetwork)
    FakeLoadedNetwork.instances = []
    with network_functions.MultiNetwork(network_paths=[Path("net_0.pkl"), Path("broken_1.pkl")]) as multi:
        assert multi is None  # callers check for None (synthesize_images.py:172-174)
    assert FakeLoadedNetwork.instances[0].stopped  # the half-loaded set was released


def test_load_true_in_constructor(fake_networks) -> None:
    multi = network_functions.MultiNetwork(network_paths=fake_networks, load=True)
    assert multi.expected_vector_length == EXPECTED_VECTOR_LENGTH
    multi.unload()


def test_parse_network_paths(tmp_path: Path) -> None:
    for name in ("b.pkl", "a.pkl", "c.txt"):
        (tmp_path / name).write_bytes(b"x")
    assert network_functions.parse_network_paths(str(tmp_path), None, None) == [tmp_path / "a.pkl", tmp_path / "b.pkl"]
    assert network_functions.parse_network_paths(None, ["x.pkl", "y.pkl"], None) == [Path("x.pkl"), Path("y.pkl")]
    listing = tmp_path / "networks.json"
    listing.write_text(json.dumps({"networks": [str(tmp_path / "b.pkl")]}))
    assert network_functions.parse_network_paths(None, None, str(listing)) == [tmp_path / "b.pkl"]
    listing.write_text(json.dumps({"networks": [str(tmp_path / "missing.pkl")]}))
    with pytest.raises(ValueError, match="formatting problem"):
        network_functions.parse_network_paths(None, None, str(listing))
    with pytest.raises(ValueError, match="Couldn't open"):
        network_functions.parse_network_paths(None, None, str(tmp_path / "nope.json"))
    with pytest.raises(ValueError, match="No networks given"):
        network_functions.parse_network_paths(None, None, None)
    assert network_functions.NETWORK_SUFFIX == ".pkl"


def test_network_file_round_trip_and_rejection(tmp_path: Path) -> None:
    variables = sg2_spec.make_random_variables(8, seed=0)
    path = tmp_path / "tiny.pkl"
    network_file.save_network(path, 8, variables)
    loaded = network_file.load_network(path)
    assert loaded.resolution == 8 and set(loaded.variables) == set(variables)
    assert np.array_equal(loaded.variables["G_synthesis/4x4/Conv/weight"], variables["G_synthesis/4x4/Conv/weight"])
    bad = tmp_path / "legacy.pkl"
    bad.write_bytes(b"not a pickle")
    with pytest.raises(RuntimeError):
        network_file.load_network(bad)
    import pickle

    bad.write_bytes(pickle.dumps(("G", "D", "Gs")))
    with pytest.raises(RuntimeError, match="format tag"):
        network_file.load_network(bad)
    with pytest.raises(ValueError):
        network_file.save_network(tmp_path / "short.pkl", 8, {k: v for k, v in list(variables.items())[:-1]} | {"dlatent_avg": np.zeros(3)})


def test_vector_helpers_match_reference_goldens(golden_dir) -> None:
    """The product's plumbing helpers against the reference-captured fixture (and its own tests' shapes)."""
    golden = np.load(golden_dir / "vector_helpers.npz")
    data = golden["data"]
    assert np.array_equal(vsc.rotate_vectors_over_time(data, 64, golden["rolls"]), golden["rotated"])
    assert np.array_equal(vsc.duplicate_to_vector_count(data, 64, 18), golden["duplicated_x3"])
    assert np.array_equal(vsc.promote_to_matrix_duplicate(data[:64], 4), golden["promoted"])
    assert np.array_equal(vsc.sub_vectors(golden["matrices"], 32), golden["sub_vectors_matrix"])
    assert np.array_equal(vsc.sub_vectors(data, 64), golden["sub_vectors_vector"])
    assert np.array_equal(vsc.demote_to_vector_select(golden["matrices"], 0), golden["demoted"])
    assert vsc.sub_vectors(np.zeros((18, 5120)), 512).shape == (10, 18, 512)  # test_vector_sources_common.py:66-83
    assert vsc.sub_vectors(np.zeros(5120), 512).shape == (10, 512)
    assert vsc.underlying_length(np.zeros(512)) == 512 and vsc.underlying_length(np.zeros((18, 1024))) == 1024
    assert is_vector(np.zeros(5)) and not is_vector(np.zeros((2, 5)))  # test_vector_sources_common.py:86-97
    with pytest.raises(ValueError, match="Cannot duplicate"):
        vsc.duplicate_to_vector_count(data, 64, 20)
    with pytest.raises(ValueError, match="Undefined behavior"):
        vsc.promote_to_matrix_duplicate(np.zeros((2, 4)), 3)


def test_vector_synthesis_orders_frames_and_switches_networks(fake_networks) -> None:
    """Frames come back in frame order whatever the per-frame network index is."""
    L, frames = EXPECTED_VECTOR_LENGTH, 21
    combined = np.zeros((18, frames * L))
    indices = np.array([(f * 7) % 3 for f in range(frames)])
    data = VisualizationInput(
        a_vectors=VectorsLabel(np.zeros(frames * L), L, "a"),
        b_vectors=MatricesLabel(combined, L, "b"),
        combined=MatricesLabel(combined, L, "c"),
        network_indices=ResultLayers(result=DataLabel(indices, "idx")),
    )
    with network_functions.MultiNetwork(network_paths=fake_networks) as multi:
        output = network_visualization.vector_synthesis(data=data, networks=multi, enable_2d=False, enable_3d=False)
        assert output.visualization_images is None
        tags = [int(frame[0, 0, 0]) for frame in output.synthesized_images]
        limited = network_visualization.vector_synthesis(data=data, networks=multi, enable_2d=False, frames_to_visualize=5)
        assert len(list(limited.synthesized_images)) == 5
    expected = [0 if index in (0, 2) else 1 for index in indices]  # index 2 is the net_0 file again
    assert tags == expected
    with pytest.raises(ValueError, match="Nothing to render"):
        network_visualization.vector_synthesis(data=data, networks=None, enable_2d=False, enable_3d=False)
