"""Host-side text preparation (CPU string work; not accelerated, reproduced for the drop-in surface).

Behaviour follows the reference's `prepare_text_prompt` (tts_model.py:913-942),
`split_into_best_sentences` (tts_model.py:978-1044) and `_estimate_max_gen_len` (tts_model.py:907-910);
tests/test_text.py checks these against input/output pairs recorded from the reference functions.
"""

from __future__ import annotations

import logging
import math

logger = logging.getLogger(__name__)

TOKENS_PER_SECOND_ESTIMATE = 3.0  # tts_model.py:63
GEN_SECONDS_PADDING = 2.0  # tts_model.py:64


def estimate_max_gen_len(token_count: int, frame_rate: float) -> int:
    return math.ceil((token_count / TOKENS_PER_SECOND_ESTIMATE + GEN_SECONDS_PADDING) * frame_rate)


def prepare_text_prompt(text: str, pad_with_spaces_for_short_inputs: bool, remove_semicolons: bool):
    """-> (normalised text, frames_after_eos guess).  Raises ValueError on empty text."""
    text = text.strip()
    if not text:
        raise ValueError("Text prompt cannot be empty")
    for a, b in (("\n", " "), ("\r", " "), ("  ", " ")):
        text = text.replace(a, b)
    if remove_semicolons:
        text = text.replace(";", ",")
    frames_after_eos_guess = 3 if len(text.split()) <= 4 else 1
    if not text[0].isupper():
        text = text[0].upper() + text[1:]
    if text[-1].isalnum():
        text += "."
    if pad_with_spaces_for_short_inputs and len(text.split()) < 5:
        text = " " * 8 + text
    return text, frames_after_eos_guess


def _boundaries(tokens: list, marks: list) -> list:
    """Indices where a new segment starts: the first non-mark token after a run of mark tokens."""
    cut = [0]
    in_run = False
    for i, tok in enumerate(tokens):
        if tok in marks:
            in_run = True
        elif in_run:
            cut.append(i)
            in_run = False
        else:
            in_run = False
    cut.append(len(tokens))
    return cut


def _segments(tokens: list, cut: list, sp) -> list:
    return [(cut[i + 1] - cut[i], sp.decode(tokens[cut[i]: cut[i + 1]])) for i in range(len(cut) - 1)]


def split_into_best_sentences(encode, sp, text: str, max_tokens: int, pad_with_spaces_for_short_inputs: bool,
                              remove_semicolons: bool) -> list:
    """Greedy packing of sentences (sub-split on , ; : when a sentence alone exceeds `max_tokens`) into
    chunks of at most `max_tokens` tokens.  `encode(str) -> list[int]`, `sp.decode(list[int]) -> str`."""
    text, _ = prepare_text_prompt(text, pad_with_spaces_for_short_inputs, remove_semicolons)
    text = text.strip()
    tokens = encode(text)
    end_marks = encode(".!...?")[1:]
    pieces = _segments(tokens, _boundaries(tokens, end_marks), sp)
    soft_marks = encode(",;:")[1:]
    refined = []
    for n, sentence in pieces:
        if n > max_tokens:
            sub_tokens = encode(sentence.strip())
            sub = _segments(sub_tokens, _boundaries(sub_tokens, soft_marks), sp)
            if len(sub) > 1:
                refined.extend(sub)
                continue
        refined.append((n, sentence))
    chunks, cur, cur_n = [], "", 0
    for n, sentence in refined:
        if cur == "":
            cur, cur_n = sentence, n
        elif cur_n + n > max_tokens:
            chunks.append(cur.strip())
            cur, cur_n = sentence, n
        else:
            cur += " " + sentence
            cur_n += n
    if cur != "":
        chunks.append(cur.strip())
    for ch in chunks:
        n = len(encode(ch.strip()))
        if n > max_tokens:
            logger.warning("Chunk has %d tokens (max %d), generation may skip words: '%.50s...'", n, max_tokens, ch)
    return chunks
