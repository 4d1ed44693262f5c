"""Host text pipeline against input/output pairs recorded from the reference's own functions
(tests/golden/gen_golden_e2e.py -> e2e_text.json).  CPU only."""

import json
from pathlib import Path

import pytest

from pocket_tts_amd.text import estimate_max_gen_len, prepare_text_prompt, split_into_best_sentences

G = Path(__file__).parent / "golden"
FX = json.loads((G / "e2e_text.json").read_text())


@pytest.fixture(scope="module")
def sp():
    import sentencepiece

    return sentencepiece.SentencePieceProcessor(str(G / "e2e_sp.model"))


def test_prepare_text_prompt_matches_reference():
    for c in FX["cases"]:
        p, guess = prepare_text_prompt(c["text"], c["pad"], c["semi"])
        assert p == c["prepared"] and guess == c["guess"], c["text"]


def test_empty_text_raises_valueerror():
    assert FX["empty_raises"]
    with pytest.raises(ValueError):
        prepare_text_prompt("   ", False, False)
    with pytest.raises(ValueError):
        prepare_text_prompt("", True, True)


def test_split_into_best_sentences_matches_reference(sp):
    enc = lambda s: sp.encode(s, out_type=int)  # noqa: E731
    for c in FX["cases"]:
        assert enc(c["prepared"]) == c["tokens"]
        assert split_into_best_sentences(enc, sp, c["text"], 12, c["pad"], c["semi"]) == c["chunks12"], c["text"]
        assert split_into_best_sentences(enc, sp, c["text"], 50, c["pad"], c["semi"]) == c["chunks50"], c["text"]


def test_estimate_max_gen_len_matches_reference():
    for n, v in FX["gen_len"].items():
        assert estimate_max_gen_len(int(n), 12.5) == v
