"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/ptts.h declares.  No compute calls (there is no GPU in the build container)."""

import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib_path():
    from pocket_tts_amd import _lib

    return _lib.build()


def test_header_symbols_are_exported(lib_path):
    header = (REPO / "include" / "ptts.h").read_text()
    declared = set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(str(lib_path))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in ptts.h but not exported: {missing}"


def test_python_prototypes_cover_header(lib_path):
    from pocket_tts_amd import _lib

    header = (REPO / "include" / "ptts.h").read_text()
    declared = set(re.findall(r"\b(ptts_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.load()
    assert lib.ptts_abi_version() == 1


def test_config_struct_matches_header():
    from pocket_tts_amd import _lib
    from pocket_tts_amd.config import named_config
    from pocket_tts_amd.engine import make_ptts_config

    pc = make_ptts_config(named_config("en100m"))
    assert ctypes.sizeof(_lib.PttsConfig) == 23 * 4
    assert (pc.d_model, pc.num_heads, pc.num_layers, pc.ff_dim, pc.ldim) == (1024, 16, 6, 4096, 32)
    assert (pc.m_dim, pc.m_heads, pc.m_layers, pc.m_ff, pc.m_context) == (512, 8, 2, 2048, 250)
    assert list(pc.ratios) == [6, 5, 4] and pc.upsample_stride == 16


def test_engine_requires_gpu():
    import torch

    from pocket_tts_amd.config import named_config
    from pocket_tts_amd.engine import Engine

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        Engine(named_config("tiny"), {}, "cuda:0")
    with pytest.raises(RuntimeError):
        Engine(named_config("tiny"), {}, "cpu")
