"""TEST/BENCH INFRASTRUCTURE.  Measures the constants frozen in reverb_amd/synth.py:
  * CMVN statistics of the synthetic audio (mean / istd of its log-mel)
  * the CTC blank-bias calibration beta per (dims, seed) (SURVEY.md 8d "Synthetic model")
Run:  python -m oracle.calibrate r268 r640
"""
import sys

import numpy as np
import torch

from oracle import fbank_ref, model_ref as M
from reverb_amd import synth


def beta_for(name: str, seed: int = 0) -> float:
    cfg = synth.make_config(name)
    sd = M.to_torch_sd(synth.make_state_dict(cfg, seed, synth.CTC_GAMMA, 0.0))
    feats = fbank_ref.fbank(synth.synth_audio(20.6, seed=1234))[:2051]
    x = torch.from_numpy(feats).unsqueeze(0)
    with torch.no_grad():
        enc, mask = M.encoder_forward(sd, cfg, x, torch.tensor([2051]), torch.tensor([1.0, 0.0]))
        logits = torch.nn.functional.linear(enc[0], sd["ctc.ctc_lo.weight"], sd["ctc.ctc_lo.bias"])
    nb = logits.clone()
    nb[:, 0] = -1e30
    return float(torch.quantile(nb.max(-1).values - logits[:, 0], 0.84))


if __name__ == "__main__":
    for name in sys.argv[1:] or ["tiny", "small"]:
        print(f'    ("{name}", 0): {beta_for(name):.4f},', flush=True)
