"""Where the wall time of one bench step goes on the HOST side: the calls of Engine.decode_resident one by one, each bracketed
with perf_counter (ctypes calls block only where librvb waits for the device).  Run after a warm-up step; prints milliseconds.

    python scripts/step_timeline.py [--model r640] [--hours 1.0] [--dtype bf16] [--steps 3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model", default="r640")
    p.add_argument("--hours", type=float, default=1.0)
    p.add_argument("--dtype", default="bf16")
    p.add_argument("--steps", type=int, default=3)
    a = p.parse_args()
    import torch
    from reverb_amd import synth
    from reverb_amd._lib import check
    from reverb_amd.engine import Engine
    chunk = 2051
    n_samples = int(round(a.hours * 3600 * 16000))
    n_frames = 1 + (n_samples - 400) // 160
    n_chunks = -(-n_frames // chunk)
    cfg, sd = synth.calibrated_state_dict(a.model, 0)
    eng = Engine(cfg, sd, dtype=a.dtype, device=0, max_chunks=n_chunks, chunk_frames=chunk)
    pcm = eng.pinned_pcm(n_samples)
    pcm[:] = synth.synth_audio(a.hours * 3600, seed=1234)
    eng.upload_pcm(pcm)
    lens = np.full(n_chunks, chunk, np.int32)
    lens[-1] = n_frames - (n_chunks - 1) * chunk

    def step(rec):
        t = [time.perf_counter()]
        nf = eng.fbank(); t.append(time.perf_counter())
        eng.encode(None, lens, 10, 0.0, first_chunk=0, T0=chunk); t.append(time.perf_counter())
        check(eng.lib.rvb_prepare_rescoring(eng.handle, 0), "rvb_prepare_rescoring")
        check(eng.lib.rvb_ctc_prefix_beam(eng.handle, eng.beam), "rvb_ctc_prefix_beam"); t.append(time.perf_counter())
        check(eng.lib.rvb_attention_rescore(eng.handle, 0.1, 0.0), "rvb_attention_rescore"); t.append(time.perf_counter())
        res = eng._rescore_fetch()
        t.append(time.perf_counter())
        torch.cuda.synchronize(); t.append(time.perf_counter())
        if rec is not None:
            rec.append(np.diff(t) * 1e3)
        return res

    step(None)
    rec = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(rec)
    wall = (time.perf_counter() - t0) / a.steps * 1e3
    m = np.mean(rec, axis=0)
    names = ["fbank (sync)", "encode (enqueue 2 slices)", "prefix beam (waits for slices, host search)",
             "rescore (trie, decoder, D2H)", "fetch results", "final sync"]
    for n, v in zip(names, m):
        print(f"{n:48s} {v:8.2f} ms")
    print(f"{'step':48s} {wall:8.2f} ms")
    for k in ("search_host", "rescore_trie_host", "rescore_decoder_wall", "rescore_scores_host"):
        t = eng.timing(k)
        print(f"  {k:46s} {t['ms'] / max(t['launches'], 1):8.2f} ms per call ({t['launches']} calls)")
    eng.close()


if __name__ == "__main__":
    main()
