import ast
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = Path(__file__).parent / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def load_golden(case: str):
    z = np.load(GOLDEN / f"golden_{case}.npz", allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = ast.literal_eval(str(d["meta"]))
    return d


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(case):
        if case not in cache:
            cache[case] = load_golden(case)
        return cache[case]

    return get


_W = {}


def synth_weights(cfg_name: str, seed: int = 0):
    """Synthetic weights (numpy) for a named config, cached per session."""
    from pocket_tts_amd.config import named_config
    from pocket_tts_amd.weights import generate_state_dict

    key = (cfg_name, seed)
    if key not in _W:
        cfg = named_config(cfg_name)
        _W[key] = (cfg, generate_state_dict(cfg, seed))
    return _W[key]
