"""TEST INFRASTRUCTURE: golden vectors of `joint_decoding` (time-synchronous joint CTC / attention beam search) from the
reference's own `BeamSearchTimeSync` class, UNMODIFIED, through oracle/ref_shim.py.

The reference's entry point cannot run: `joint_decoding` (transformer/search.py:450-496) hands the class a 2-D
`encoder_outs[b, :len, :]`, and `BeamSearchTimeSync.reset` (espnet/beam_search_timesync.py:150-154) takes `size(1)` of it --
the model dimension -- for the memory length; the first decoder step raises a size mismatch (checked below and recorded in
the json).  `asr/wer_evaluation/RESULTS.md:24` publishes a WER for the mode, so it worked with the call shape `reset`
expects: memory (1, len, d).  This script therefore repeats the body of `joint_decoding` line for line with that one
argument changed (`encoder_outs[b:b+1, :len, :]`) and drives the unmodified class; everything else -- sos = 10000, weights,
pre-beam ratio, length bonus, which outputs become the DecodeResult -- is the reference's.

Models: the `tiny_v10k` / `small_v10k` bodies with the Reverb vocabulary size (10001: sos = 10000 is hard-coded).
Writes tests/golden/joint_{tiny,small}.json.        python -m oracle.gen_golden_joint
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

from oracle import fbank_ref, ref_shim            # noqa: E402
from reverb_amd import synth                      # noqa: E402
from oracle.gen_golden import build_reference_model, calibrate_beta, chunk_feats   # noqa: E402

ref_shim.install()
import torch                                      # noqa: E402

CASES = [
    dict(name="joint_tiny", dims="tiny_v10k", norm="layer_norm", seed=6, seconds=14.0, chunk=700, cat=[1.0, 0.0],
         runs=[dict(beam=4, ctc_weight=0.3, pre_beam_ratio=1.5, length_bonus=0.5),
               dict(beam=3, ctc_weight=0.6, pre_beam_ratio=2.0, length_bonus=0.2),
               # a random decoder charges ~9 nats per token: a large length bonus / CTC weight keeps the hypotheses long
               dict(beam=4, ctc_weight=0.3, pre_beam_ratio=1.5, length_bonus=7.0),
               dict(beam=5, ctc_weight=0.9, pre_beam_ratio=1.5, length_bonus=1.0),
               # round 4: a pre-beam of 18 candidates per frame (the CTC kernel kept at most 16 until then), and a blank penalty
               # (`ctc_logprobs(encoder_out, blank_penalty, blank_id)`, asr_model.py:318-329 -> search.py:466)
               dict(beam=12, ctc_weight=0.5, pre_beam_ratio=1.5, length_bonus=2.0),
               dict(beam=4, ctc_weight=0.3, pre_beam_ratio=1.5, length_bonus=3.0, blank_penalty=2.0)]),
    dict(name="joint_small", dims="small_v10k", norm="layer_norm", seed=9, seconds=20.6, chunk=2051, cat=[0.3, 0.7],
         runs=[dict(beam=4, ctc_weight=0.3, pre_beam_ratio=1.5, length_bonus=0.5),
               dict(beam=4, ctc_weight=0.5, pre_beam_ratio=1.5, length_bonus=5.0)]),
]


def joint_decoding_3d(model, encoder_outs, encoder_lens, ctc_probs, ctc_weight, beam_size, pre_beam_ratio, length_bonus, cat_embs):
    """transformer/search.py:450-496 with the memory handed over as (1, len, d)."""
    from wenet.espnet.beam_search_timesync import BeamSearchTimeSync
    decoder = model.decoder.left_decoder
    weights = dict(decoder=1.0 - ctc_weight, ctc=ctc_weight, length_bonus=length_bonus)
    rows = []
    for b in range(encoder_outs.shape[0]):
        n = int(encoder_lens[b])
        bs = BeamSearchTimeSync(sos=10000, beam_size=beam_size, ctc_probs=ctc_probs[b, :n, :], decoder=decoder, weights=weights,
                                pre_beam_ratio=pre_beam_ratio)
        hyps, scores, starts, ends, confs = bs(x=encoder_outs[b:b + 1, :n, :], cat_embs=cat_embs)
        rows.append(dict(tokens=[h.item() for h in hyps[0][1:]], score=scores[0].item(),
                         times=[t.item() for t in starts[0][0, 1:]], end_times=[t.item() for t in ends[0][0, 1:]],
                         tokens_confidence=[math.exp(c.item()) for c in confs[0][1:]],
                         nbest=[[t.item() for t in h[1:]] for h in hyps], nbest_scores=[s.item() for s in scores]))
    return rows


def main():
    torch.set_num_threads(8)
    infos = {"tasks": ["transcribe"], "langs": ["en"]}
    for case in CASES:
        cfg = synth.make_config(case["dims"], case["norm"])
        assert cfg["output_dim"] == 10001
        x, lens = chunk_feats(fbank_ref.fbank(synth.synth_audio(case["seconds"], seed=1234 + case["seed"])), case["chunk"])
        cat = torch.tensor(case["cat"])
        model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, 0.0))
        beta = calibrate_beta(model, x, lens, cat)
        model, _ = build_reference_model(cfg, synth.make_state_dict(cfg, case["seed"], synth.CTC_GAMMA, beta))
        out = dict(case=case, beta=beta, gamma=synth.CTC_GAMMA, lens=lens.tolist(), runs=[])
        with torch.no_grad():
            try:
                model.decode(["joint_decoding"], torch.from_numpy(x), torch.from_numpy(lens), 4, ctc_weight=0.3, cat_embs=cat, blank_id=0,
                             infos=infos)
                out["reference_entry_point"] = "ran"
            except Exception as ex:
                out["reference_entry_point"] = f"{type(ex).__name__}: {str(ex)[:160]}"
            enc, mask = model.encoder(torch.from_numpy(x), torch.from_numpy(lens), -1, -1, cat_embs=cat)
            probs = model.ctc_logprobs(enc)
            elens = mask.squeeze(1).sum(1)
            out["encoder_lens"] = elens.tolist()
            for run in case["runs"]:
                bp = run.get("blank_penalty", 0.0)
                rows = joint_decoding_3d(model, enc, elens, model.ctc_logprobs(enc, bp, 0) if bp else probs, run["ctc_weight"], run["beam"],
                                         run["pre_beam_ratio"], run["length_bonus"], cat)
                out["runs"].append(dict(run, chunks=rows))
                print(case["name"], run, [len(r["tokens"]) for r in rows], [round(r["score"], 3) for r in rows])
        print("  reference entry point:", out["reference_entry_point"])
        with open(os.path.join(GOLDEN, case["name"] + ".json"), "w") as f:
            json.dump(out, f)


if __name__ == "__main__":
    main()
