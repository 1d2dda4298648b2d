#!/usr/bin/env python
"""BASELINE.json configs[4]: the joint pipeline -- Reverb-ASR attention_rescoring + diarization + word->speaker assignment
(`recognize_wav.py`, `infer_pyannote3.0.py`, `assign_words2speakers.py` in the reference) -- on one recording per GPU, with
the ASR encoder's GEMMs in fp8 (`--dtype fp8`).  Same output contract as bench.py; a step is
`reverb_amd.bin.transcribe_diarize.run` on PCM held in host memory: ASR and diarization run as two host threads on their own
engines and streams (the diarization's one-CU merge loop and host numpy hide under the ASR encoder), then the join.
`sequential_ms_per_step` is the same work without the overlap.

  python bench_joint.py --steps 3 --warmup 1 [--dtype fp8|bf16] [--hours 1]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class _Asr:
    """What transcribe_diarize.run reads of a ReverbASR object, built on a bare engine (no model directory on disk)."""

    def __init__(self, engine, tokenizer):
        self.engine, self.tokenizer = engine, tokenizer
        self.input_frame_length, self.output_frame_length = 10, 40

    def decode_resident(self, *a, **k):
        return self.engine.decode_resident(*a, **k)


def run(device_index=0, steps=3, warmup=1, hours=3.0, dtype="fp8", model="r640", state=None):
    import torch
    from reverb_amd import diarization as D, synth, synth_diar as SD
    from reverb_amd.bin import transcribe_diarize as TD
    from reverb_amd.engine import Engine
    from reverb_amd.tokenizer import RevBpeTokenizer
    seconds = hours * 3600.0
    n = int(round(seconds * 16000))
    n_chunks = -(-(1 + (n - 400) // 160) // 2051)
    cfg, sd = state if state is not None else synth.calibrated_state_dict(model, 0)
    eng = Engine(cfg, sd, dtype=dtype, device=device_index, max_chunks=n_chunks, chunk_frames=2051)
    del sd
    units = synth.make_units(cfg["output_dim"])
    asr = _Asr(eng, RevBpeTokenizer(None, {u: i for i, u in enumerate(units)}))
    dcfg = SD.make_diar_config()
    # "fp8 MFMA GEMMs" on the diarization side = the MFMA-bound half of the ResNet34 trunk (stages 3-4) on e4m3 operands
    diar_dtype = "fp8" if dtype == "fp8" else "bf16"
    pipe = D.SpeakerDiarization(dcfg, SD.make_segmentation_sd(dcfg, 0), SD.make_embedding_sd(dcfg, 0), None, dtype=diar_dtype).to(device_index)
    base = SD.synth_conversation(120.0)
    pcm = np.tile(base, n // len(base) + 1)[:n]
    pcm = (pcm.astype(np.int32) + np.random.default_rng(7).integers(-3, 4, size=n)).clip(-32768, 32767).astype(np.int16)
    audio = ("bench", pcm)

    def timed(k, overlap):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = TD.run(audio, asr, pipe, None, overlap=overlap)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, out + (dict(pipe.timings),)

    for _ in range(max(warmup, 1)):          # the first pass also calibrates the fp8 activation scales
        TD.run(audio, asr, pipe, None, overlap=True)
    dt, (ctm, ann, stm, tm, _) = timed(steps, True)
    ds, (_, _, stm_seq, tms, pt) = timed(max(1, steps // 2), False)
    assert stm_seq == stm, "overlapped and sequential runs must give the same speaker-attributed transcript"
    # What shards across GPUs and what every rank repeats (reverb_amd/dist.py), from the SEQUENTIAL step so that the parts
    # add up: chunks and windows are independent (decode_sharded / diarize_sharded: 1/N of these per rank) -- the ASR step
    # and both diarization networks; speaker counting, clustering (one merge loop over all embeddings), reconstruction and
    # the word -> speaker join need the whole recording and run on every rank.  An N-GPU step is about
    # sharded_s / N + replicated_s + two result gathers.
    networks = sum(pt.get(k, 0.0) for k in ("upload", "segmentation", "host_masks", "embedding"))
    sharded_s = tms["asr"] + networks
    replicated_s = pt.get("clustering", 0.0) + pt.get("reconstruction", 0.0) + tms["join"]
    out = {
        "metric": "RTFx (audio-sec/wall-sec) joint ASR + diarization + word->speaker pipeline",
        "value": round(seconds * steps / dt, 2), "unit": "audio-sec/wall-sec", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": round(dt / steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": f"joint pipeline on one {hours:g} h 16 kHz recording: Reverb-ASR attention_rescoring ({model} synthetic "
                               f"weights, encoder GEMMs in {dtype}), pyannote-style diarization ({diar_dtype}"
                               f"{': ResNet34 stages 3-4 on e4m3 operands' if diar_dtype == 'fp8' else ''}), words -> speakers",
                   "words": len(stm), "speakers": len(ann.labels()), "turns": len(ann)},
        "sequential_ms_per_step": round(ds / max(1, steps // 2) * 1e3, 2),
        "last_step_s": {k: round(v, 4) for k, v in tm.items()},
        "sequential_last_step_s": {k: round(v, 4) for k, v in tms.items()},
        "sharded_s": round(sharded_s, 4), "replicated_s": round(replicated_s, 4),
        "sharded_parts_s": {"asr": round(tms["asr"], 4), "diarization_networks": round(networks, 4)},
        "replicated_parts_s": {"clustering": round(pt.get("clustering", 0.0), 4), "reconstruction": round(pt.get("reconstruction", 0.0), 4),
                               "join": round(tms["join"], 4)},
        "projected_8gpu_step_s": round(sharded_s / 8.0 + replicated_s, 4),
    }
    if diar_dtype == "fp8":
        st, _, clipped = pipe.engine.emb_fp8()
        out["diarization_fp8"] = {"state": int(st), "clipped_values": int(clipped)}
    eng.close()
    pipe.engine.close()
    pipe._engine = None
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--hours", type=float, default=3.0, help="BASELINE configs[4]: an Earnings21-shaped 3 h recording")
    p.add_argument("--dtype", default="fp8", choices=["fp8", "bf16"])
    p.add_argument("--model", default="r640")
    a = p.parse_args()
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench_joint.py needs an MI355X: no CPU fallback")
    out = run(int(os.environ.get("LOCAL_RANK", "0")), a.steps, a.warmup, a.hours, a.dtype, a.model)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
