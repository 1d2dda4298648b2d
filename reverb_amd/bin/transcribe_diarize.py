#!/usr/bin/env python
"""Joint ASR + diarization of one recording (BASELINE config 5; the reference does it in three commands:
`recognize_wav.py --output_format ctm`, `infer_pyannote3.0.py`, `assign_words2speakers.py rttm ctm stm`):

    python -m reverb_amd.bin.transcribe_diarize a.wav --asr-model DIR --pipeline-model DIR --out-dir out

writes out/a.ctm, out/a.rttm and out/a.stm (one `<uri> 1 <speaker> <start> <end> <word>` line per word).
With torchrun (one process per GPU) the ASR chunks and the diarization windows are sharded across the ranks
(reverb_amd/dist.py) and rank 0 writes the files.
"""
import argparse
import os
import time

import numpy as np


def run(audio, asr, pipe, out_dir, mode="attention_rescoring", device=None, world=1, rank=0, overlap=True, linkage_workgroups=4,
        **decode_kw):
    """overlap (single process): both diarization networks run first (they fill the GPU); then the diarization's host part --
    speaker counting, the clustering (a 150 ms per hour merge loop on ONE compute unit), reconstruction -- runs in a second
    host thread underneath the ASR encoder, which fills the other 255 compute units on its own HIP stream (ctypes releases
    the GIL inside librvb)."""
    from reverb_amd.bin.assign_words2speakers import make_turns, speakers_for_words
    from reverb_amd.reverb import get_output
    from reverb_amd import audio as audio_reader
    if isinstance(audio, tuple):        # (name, int16 mono PCM at 16 kHz) already in memory
        stem, pcm16 = audio
        wave, rate = np.ascontiguousarray(pcm16, np.int16).reshape(1, -1), 16000
        audio = {"waveform": wave[0], "sample_rate": 16000, "uri": stem}
    else:
        stem = os.path.splitext(os.path.basename(audio))[0]
        wave, rate = audio_reader.load(audio, channel=0)
    timings = {}
    t0 = time.perf_counter()
    chunk = asr.engine.cfg.chunk_frames
    kw = dict(beam_size=10, ctc_weight=0.1, reverse_weight=0.0)
    kw.update(decode_kw)
    if world > 1:
        from reverb_amd.dist import decode_sharded, diarize_sharded
        if rate != 16000:
            raise NotImplementedError("sharded decoding slices 16 kHz PCM; resample the file first")
        hyps = decode_sharded(asr.engine, wave[0], [mode], chunk, kw["beam_size"], kw["ctc_weight"], kw["reverse_weight"], device)[mode]

    def asr_local():
        ta = time.perf_counter()
        asr.engine.upload_pcm(wave[0], rate)
        nf = asr.engine.fbank()
        h = asr.decode_resident(nf, [mode], chunk, kw["beam_size"], kw["ctc_weight"], kw["reverse_weight"])[mode]
        c = get_output("ctm", asr.tokenizer, stem, h, 230, chunk, asr.input_frame_length, asr.output_frame_length)
        timings["asr"] = time.perf_counter() - ta
        return c

    def diar_local():
        td = time.perf_counter()
        a = pipe(audio)
        timings["diarization"] = time.perf_counter() - td
        return a

    if world > 1:
        ctm = get_output("ctm", asr.tokenizer, stem, hyps, 230, chunk, asr.input_frame_length, asr.output_frame_length)
        timings["asr"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        mono = np.clip(np.rint(wave.astype(np.float32).mean(axis=0)), -32768, 32767).astype(np.int16)
        ann = diarize_sharded(pipe, mono, device, uri=stem)
        timings["diarization"] = time.perf_counter() - t1
    elif overlap:
        from concurrent.futures import ThreadPoolExecutor
        td = time.perf_counter()
        pcm_d, uri = pipe._load(audio)
        classes, emb = pipe.networks(pcm_d)
        timings["diarization_networks"] = time.perf_counter() - td
        # the clustering's merge loop is persistent: with its default 16 workgroups it takes 16 CUs and an XCD's L2 away from the
        # ASR encoder it runs underneath (1 h: joint step 493-499 ms; 472 ms with 4 workgroups, 477 with one); recordings too long
        # for 4 workgroups' LDS (more than ~18 000 embeddings) get as many as they need (rvd_set_linkage_workgroups)
        pipe.engine.set_linkage_workgroups(int(linkage_workgroups))
        try:
            with ThreadPoolExecutor(1) as ex:
                fd = ex.submit(pipe.finish, classes, emb, uri)
                ctm = asr_local()
                ann = fd.result()
        finally:
            pipe.engine.set_linkage_workgroups(0)
        timings["diarization"] = time.perf_counter() - td
    else:
        ctm = asr_local()
        ann = diar_local()
    t2 = time.perf_counter()
    turns = make_turns(ann) if len(ann) else []
    words = [(float(p_[2]), float(p_[3]), p_[4]) for p_ in (line.split(" ") for line in ctm.splitlines()) if len(p_) >= 6]
    who = speakers_for_words([w[0] for w in words], [w[1] for w in words], turns)
    stm_lines = [f"{stem} 1 {spk} {start:.3f} {(start + dur):.3f} {token}" for (start, dur, token), spk in zip(words, who)]
    timings["join"] = time.perf_counter() - t2
    if rank == 0 and out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, stem + ".ctm"), "w") as f:
            f.write(ctm + ("\n" if ctm else ""))
        with open(os.path.join(out_dir, stem + ".rttm"), "w") as f:
            ann.write_rttm(f)
        with open(os.path.join(out_dir, stem + ".stm"), "w") as f:
            f.write("\n".join(stm_lines) + ("\n" if stm_lines else ""))
    timings["total"] = time.perf_counter() - t0
    return ctm, ann, stm_lines, timings


def main(argv=None):
    p = argparse.ArgumentParser(description="ASR + diarization + word->speaker assignment")
    p.add_argument("audios", nargs="+")
    p.add_argument("--asr-model", required=True, help="Reverb-ASR model directory (config.yaml, *.pt, tk.units.txt, ...)")
    p.add_argument("--pipeline-model", required=True, help="diarization pipeline directory (config.yaml, segmentation.pt, embedding.pt)")
    p.add_argument("--out-dir", required=True)
    p.add_argument("--mode", default="attention_rescoring")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"], help="fp8: the ASR encoder's feed-forward GEMMs and stages 3-4 of the embedding ResNet34 on e4m3 operands")
    p.add_argument("--sequential", action="store_true", help="do not overlap ASR and diarization")
    args = p.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = None
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        dist.init_process_group("nccl", device_id=device)
    from reverb_amd.diarization import Pipeline
    from reverb_amd.reverb import load_model
    asr = load_model(args.asr_model, gpu=local, dtype=args.dtype, max_chunks=256)
    pipe = Pipeline.from_pretrained(args.pipeline_model, dtype=args.dtype).to(f"cuda:{local}")
    for audio in args.audios:
        _, ann, stm, t = run(audio, asr, pipe, args.out_dir, args.mode, device, world, rank, overlap=not args.sequential)
        if rank == 0:
            print(f"{audio}: {len(stm)} words, {len(ann.labels())} speakers, asr {t['asr']:.3f} s, diarization {t['diarization']:.3f} s")
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
