"""Host-side formatting: token->word alignment, time offset, CTM/TXT rendering and the tokenizer
table, pinned by known answers generated from the unmodified reference."""
import pytest

from reverb_amd.ctc_align import adjust_model_time_offset, ctc_align, hyps_to_ctm, hyps_to_txt
from reverb_amd.reverb import get_output
from reverb_amd.search import DecodeResult
from reverb_amd.tokenizer import RevBpeTokenizer

TABLE = {"<blank>": 0, "▁he": 1, "llo": 2, "▁wor": 3, "ld": 4, "<laugh>": 5, "▁a": 6}


def tok():
    return RevBpeTokenizer(None, dict(TABLE))


def test_c2_reference_known_answer():
    hyp, times, conf = [1, 2, 5, 3, 4, 6], [3, 5, 20, 30, 31, 60], [0.9, 0.8, 0.5, 0.7, 0.95, 0.6]
    path = ctc_align(hyp, times, conf, tok(), 40, 20510)
    got = [(w["word"], w["start_time_ms"], w["end_time_ms"], w["confidence"], w["unit_id"]) for w in path]
    assert got == [("hello", 20530, 20710, 0.9, -1), ("<laugh>", 21210, 21310, 0.5, 5),
                   ("world", 21610, 21750, 0.95, -1), ("a", 22810, 22910, 0.6, -1)]
    path = adjust_model_time_offset(path, 230)
    assert list(hyps_to_ctm("a.wav", path)) == ["a.wav 0 20.30 0.18 hello 0.90", "a.wav 0 20.98 0.10 <laugh> 0.50",
                                                "a.wav 0 21.38 0.14 world 0.95", "a.wav 0 22.58 0.10 a 0.60"]
    assert " ".join(hyps_to_txt(path)) == "hello <laugh> world a"


def test_offset_zero_returns_none_like_reference():
    assert adjust_model_time_offset([], 0) is None


def test_midpoint_rules_and_first_word_clamp():
    # tokens 1 frame apart (<100 ms): start/end use the midpoint frame; the first word is clamped at 0
    path = ctc_align([1, 6, 6], [1, 2, 3], None, tok(), 40, 0)
    assert [(w["word"], w["start_time_ms"], w["end_time_ms"], w["confidence"]) for w in path] == \
        [("he", 0, 40, 0), ("a", 40, 80, 0), ("a", 80, 120, 0)]
    out = adjust_model_time_offset(path, 230)
    assert [(w["start_time_ms"], w["end_time_ms"]) for w in out] == [(0, 40), (40, 80), (80, 120)]


def test_length_mismatch_asserts():
    with pytest.raises(AssertionError):
        ctc_align([1, 2], [3], None, tok(), 40, 0)


def test_get_output_chunk_time_shift_and_formats():
    hyps = [DecodeResult([1, 2], times=[3, 5], tokens_confidence=[0.9, 0.8]),
            DecodeResult([], times=[], tokens_confidence=[]),
            DecodeResult([6], times=[10], tokens_confidence=[0.7])]
    ctm = get_output("ctm", tok(), "x.wav", hyps, 230, 2051, 10, 40)
    # third chunk is shifted by 2 * 2051 * 10 ms (cli/reverb.py:320-325)
    assert ctm.split("\n") == ["x.wav 0 0.00 0.18 hello 0.90", "x.wav 0 41.09 0.10 a 0.70"]
    assert get_output("txt", tok(), "x.wav", hyps, 230, 2051, 10, 40) == "hello a"
    with pytest.raises(ValueError):
        get_output("srt", tok(), "x.wav", hyps, 230, 2051, 10, 40)


def test_tokenizer_detokenize():
    t = tok()
    assert t.detokenize([1, 2])[1] == ["▁he", "llo"]
    assert t.detokenize([1, 2, 3, 4])[0] == "hello world"
    assert t.vocab_size() == 7


@pytest.mark.parametrize("name", ["tiny_ln", "tiny_ln_r2l", "tiny_bn", "small_ln", "r268_chunk"])
def test_formatter_reproduces_the_reference_strings_of_the_goldens(name):
    """The reference's own DecodeResults (tests/golden/*.json, written by oracle/gen_golden.py from the unmodified
    reference) through OUR get_output -> ctc_align -> adjust_model_time_offset -> hyps_to_ctm / hyps_to_txt must give
    the strings the reference's get_output (cli/reverb.py:298-327) gave for them -- including the cases where the
    reference raises because a result has fewer times than tokens."""
    from reverb_amd import synth
    from tests.golden_util import Case
    case = Case(name)
    units = synth.make_units(case.cfg["output_dim"])
    t = RevBpeTokenizer(None, {u: i for i, u in enumerate(units)})
    for mode in ("ctc_prefix_beam_search", "attention_rescoring"):
        want = case.js["outputs"][mode]
        hyps = [DecodeResult(g["tokens"], g["score"], g["confidence"], g["tokens_confidence"], g["times"]) for g in case.golden(mode)]
        if "error" in want:
            with pytest.raises(AssertionError):
                get_output("ctm", t, "golden.wav", hyps, 230, case.chunk, 10, 40)
            continue
        for fmt in ("ctm", "txt"):
            assert get_output(fmt, t, "golden.wav", hyps, 230, case.chunk, 10, 40) == want[fmt], (mode, fmt)
