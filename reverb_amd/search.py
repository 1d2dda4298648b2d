"""Result record of a decode call -- same fields as the reference's
`wenet.transformer.search.DecodeResult` (asr/wenet/transformer/search.py:29-58) so callers that
read `.tokens/.score/.confidence/.tokens_confidence/.times/.nbest/.nbest_scores/.nbest_times`
keep working."""
from __future__ import annotations

from typing import List, Optional


class DecodeResult:
    __slots__ = ("tokens", "score", "confidence", "tokens_confidence", "times", "nbest", "nbest_scores",
                 "nbest_times", "ctc_frames", "end_times")

    def __init__(self, tokens: List[int], score: float = 0.0, confidence: float = 0.0,
                 tokens_confidence: Optional[List[float]] = None, times: Optional[List[int]] = None,
                 nbest: Optional[List[List[int]]] = None, nbest_scores: Optional[List[float]] = None,
                 nbest_times: Optional[List[List[int]]] = None):
        self.tokens = tokens
        self.score = score
        self.confidence = confidence
        self.tokens_confidence = tokens_confidence
        self.times = times
        self.nbest = nbest
        self.nbest_scores = nbest_scores
        self.nbest_times = nbest_times
        # extension: frame of each greedy token's first emission (the reference leaves greedy
        # results without times, which makes its own CTM formatter raise)
        self.ctc_frames = None
        # extension: end frame per token of a joint_decoding result (BeamSearchTimeSync computes them, search.py:490-494
        # keeps the start frames only)
        self.end_times = None

    def __repr__(self):
        return (f"DecodeResult(tokens={list(self.tokens)!r}, score={self.score!r}, "
                f"confidence={self.confidence!r}, times={self.times!r})")
