#!/usr/bin/env python
"""fp8 policy sweep on the bench hour (r640_1h golden): which GEMM groups / blocks can run in fp8 at what token error rate.
Each configuration = (RVB_FP8_GROUPS bit mask: 1 macaron FFN, 2 qkv, 4 pointwise conv 1, 8 pointwise conv 2, 16 FFN;
RVB_FP8_FIRST, RVB_FP8_LAST block range).  Prints TER (greedy / rescored) against the unmodified reference and ms per step.
    python scripts/fp8_sweep.py [mask:first:last ...]"""
import os
import sys
import time

os.environ.setdefault("RVB_LAB", "1")                 # the policy is selected through lab switches: bind librvb_test.so

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np                                    # noqa: E402
from golden_util import LongCase                      # noqa: E402
from util import edit_distance                        # noqa: E402
from reverb_amd.engine import Engine                  # noqa: E402

MODES = ["ctc_greedy_search", "attention_rescoring"]
CONFIGS = [(31, 0, 17), (17, 0, 17), (29, 0, 17), (14, 0, 17), (31, 2, 15), (17, 2, 15), (31, 4, 17), (19, 0, 17), (0, 0, 17)]
if len(sys.argv) > 1:
    CONFIGS = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]]
case = LongCase(os.environ.get("CASE", "r640_1h"))
n = len(case.js["lens"])
pcm = case.pcm
sd = case.sd
for mask, first, last in CONFIGS:
    os.environ.update(RVB_FP8_GROUPS=str(mask), RVB_FP8_FIRST=str(first), RVB_FP8_LAST=str(last))
    eng = Engine(case.cfg, sd, dtype="fp8", device=0, max_chunks=n, chunk_frames=case.chunk, cat_embs=case.cat)
    eng.upload_pcm(pcm)
    nf = eng.fbank()
    eng.decode_resident(nf, MODES, case.chunk, case.beam, case.ctc_weight, case.reverse_weight)      # calibration pass (bf16)
    res = eng.decode_resident(nf, MODES, case.chunk, case.beam, case.ctc_weight, case.reverse_weight)
    t0 = time.perf_counter()
    for _ in range(2):
        eng.fbank()
        eng.decode_resident(nf, MODES[1:], case.chunk, case.beam, case.ctc_weight, case.reverse_weight)
    ms = (time.perf_counter() - t0) / 2 * 1e3
    line = f"groups {mask:2d} blocks {first:2d}-{last:2d}: {ms:7.2f} ms/step"
    for m in MODES:
        err = sum(edit_distance(r.tokens, g["tokens"]) for r, g in zip(res[m], case.golden(m)))
        tot = sum(len(g["tokens"]) for g in case.golden(m))
        line += f"   {m} TER {err}/{tot} = {100.0 * err / tot:.2f} %"
    print(line, flush=True)
    eng.close()
