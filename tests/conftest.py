"""Shared pytest configuration: the `gpu` marker and the repo root on sys.path."""

import sys
from pathlib import Path

import pytest

REPO_ROOT = Path(__file__).resolve().parent.parent
if str(REPO_ROOT) not in sys.path:
    sys.path.insert(0, str(REPO_ROOT))

GOLDEN_DIR = Path(__file__).resolve().parent / "golden"


def pytest_configure(config: pytest.Config) -> None:
    """Register markers."""
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir() -> Path:
    """Directory of committed golden fixtures."""
    return GOLDEN_DIR
