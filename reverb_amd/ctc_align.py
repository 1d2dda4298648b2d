"""Token -> word merging with CTC-peak timestamps, and the global model time offset.

Host-side integer-millisecond logic with the behaviour of the reference's
`asr/wenet/bin/ctc_align.py` (`ctc_align` :24-113, `adjust_model_time_offset` :116-138), pinned
by the reference-generated known answer C2 of SURVEY.md Appendix C (tests/test_host_format.py):

  * a piece containing the sentencepiece space mark opens a word (its first character is dropped);
    a `<...>` piece is a word of its own;
  * a word starts 100 ms before its first token's frame (clamped at 0) unless the previous token
    is closer than 100 ms, then at the midpoint frame; it ends at its last token's frame, or at the
    midpoint to the next token when that one is closer than 100 ms;
  * word confidence is the maximum of its tokens' confidences (0 when none are given).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

SPACE_MARK = "▁"
GAP_MS = 100


def piece_of(token_id: int, tokenizer) -> str:
    return tokenizer.detokenize([token_id])[1][0]


def _looks_special(text: str) -> bool:
    lo, hi = text.find("<"), text.find(">")
    return lo != -1 and hi != -1 and lo < hi


def _blank_word(text: str) -> bool:
    return text in ("", SPACE_MARK)


def ctc_align(hypothesis: Sequence[int], time_stamp: Sequence[int], confidence_scores: Optional[Sequence[float]],
              tokenizer, frame_shift_ms: int, time_shift_ms: int) -> List[Dict[str, Any]]:
    assert len(hypothesis) == len(time_stamp)
    n = len(hypothesis)
    pieces = [piece_of(t, tokenizer) for t in hypothesis]

    def begin_ms(i: int) -> int:
        ms = max(time_stamp[i] * frame_shift_ms - GAP_MS, 0)
        if i > 0 and (time_stamp[i] - time_stamp[i - 1]) * frame_shift_ms < GAP_MS:
            ms = (time_stamp[i - 1] + time_stamp[i]) // 2 * frame_shift_ms
        return ms

    def finish_ms(i: int) -> int:
        ms = time_stamp[i] * frame_shift_ms
        if i < n - 1 and (time_stamp[i + 1] - time_stamp[i]) * frame_shift_ms < GAP_MS:
            ms = (time_stamp[i + 1] + time_stamp[i]) // 2 * frame_shift_ms
        return ms

    def best_conf(first: int, last: int):
        return max(confidence_scores[first:last + 1]) if confidence_scores else 0

    words: List[Dict[str, Any]] = []
    text, ids, t_begin, first_tok = "", [], -1, -1
    for i, piece in enumerate(pieces):
        following = pieces[i + 1] if i + 1 < n else SPACE_MARK
        text += piece[len(SPACE_MARK):] if SPACE_MARK in piece else piece
        ids.append(hypothesis[i])
        if t_begin == -1:
            t_begin, first_tok = begin_ms(i), i

        if not _blank_word(text) and _looks_special(text):          # a <tag> closes immediately
            t_end = finish_ms(i)
            assert t_begin < t_end
            assert len(ids) == 1
            words.append({"word": text, "unit_id": ids[0], "start_time_ms": t_begin + time_shift_ms,
                          "end_time_ms": t_end + time_shift_ms, "confidence": best_conf(first_tok, i),
                          "unit_ids": ids})
            text, ids, t_begin, first_tok = "", [], -1, 0

        if SPACE_MARK in following or _looks_special(following):    # next piece opens a new word
            t_end = finish_ms(i)
            if not _blank_word(text):
                assert len(ids) > 0
                assert t_begin <= t_end
                assert not _looks_special(text)
                words.append({"word": text, "unit_id": -1, "start_time_ms": t_begin + time_shift_ms,
                              "end_time_ms": t_end + time_shift_ms, "confidence": best_conf(first_tok, i),
                              "unit_ids": ids})
            text, ids, t_begin, first_tok = "", [], -1, 0
    return words


def adjust_model_time_offset(hypothesis: List[Dict[str, Any]], adjustment):
    """Shift every word earlier by up to `adjustment` ms without crossing the (already shifted)
    previous word.  Like the reference, an adjustment of 0 returns None."""
    if adjustment == 0:
        return None
    shifted = []
    for i, word in enumerate(hypothesis):
        assert word["start_time_ms"] >= 0
        assert word["start_time_ms"] <= word["end_time_ms"]
        if i == 0:
            move = min(adjustment, word["start_time_ms"])
        else:
            before = hypothesis[i - 1]
            assert word["start_time_ms"] >= before["end_time_ms"], f"ERROR! {word} >= {before}"
            move = min(adjustment, word["start_time_ms"] - before["end_time_ms"])
        assert move >= 0
        word["start_time_ms"] -= move
        word["end_time_ms"] -= move
        shifted.append(word)
    return shifted


def hyps_to_ctm(audio_name: str, path):
    """CTM lines `<audio> 0 <start s> <duration s> <word> <confidence>` (cli/utils.py:4-14)."""
    for w in path:
        start_s = w["start_time_ms"] / 1000
        dur_s = w["end_time_ms"] / 1000 - start_s
        yield f"{audio_name} 0 {start_s:.2f} {dur_s:.2f} {w['word']} {w['confidence']:.2f}"


def hyps_to_txt(path):
    """Plain words (cli/utils.py:16-21)."""
    for w in path:
        yield w["word"]
