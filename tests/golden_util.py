"""Rebuild the exact inputs of a golden case (weights, audio, chunked features) from its json."""
import json
import os

import numpy as np

from oracle import fbank_ref
from reverb_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["tiny_ln", "tiny_ln_r2l", "tiny_bn", "small_ln", "r268_chunk"]
MODES = ["ctc_greedy_search", "ctc_prefix_beam_search", "attention_rescoring"]


class Case:
    def __init__(self, name):
        self.name = name
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.js = json.load(f)
        self.arrays = np.load(os.path.join(GOLDEN, name + ".npz"))
        c = self.js["case"]
        self.c = c
        self.cfg = synth.make_config(c["dims"], c["norm"])
        self.beam, self.ctc_weight, self.reverse_weight, self.cat = c["beam"], c["ctc_weight"], c["reverse_weight"], c["cat"]
        self.chunk = c["chunk"]
        self._sd = None
        self._feats = None

    @property
    def sd(self):
        if self._sd is None:
            self._sd = synth.make_state_dict(self.cfg, self.c["seed"], self.js["gamma"], self.js["beta"])
        return self._sd

    @property
    def pcm(self):
        return synth.synth_audio(self.c["seconds"], seed=1234 + self.c["seed"])

    def chunked_feats(self):
        """(x [nch, chunk, 80], lens) exactly as oracle/gen_golden.py fed the reference."""
        if self._feats is None:
            feats = fbank_ref.fbank(self.pcm)
            n = feats.shape[0]
            tail = self.c.get("tail_frames")
            if tail is not None:
                n = (n // self.chunk) * self.chunk + tail
                feats = feats[:n]
            nch = -(-n // self.chunk)
            x = np.zeros((nch, self.chunk, 80), np.float32)
            lens = np.zeros(nch, np.int32)
            for i in range(nch):
                part = feats[i * self.chunk:(i + 1) * self.chunk]
                x[i, :len(part)] = part
                lens[i] = len(part)
            assert lens.tolist() == self.js["lens"]
            self._feats = (x, lens)
        return self._feats

    def golden(self, mode):
        return self.js["modes"][mode]


LONG_CASES = ["small_66", "r640_chunk", "r640_1h"]
_SD_CACHE = {}


class LongCase(Case):
    """Long-form golden (oracle/gen_golden.py:run_long_case): the unmodified reference decoded chunk by chunk with batch 1;
    per chunk the greedy tokens and the rescoring winner (tokens, times, score, confidence), plus a strided encoder sample
    and the per-frame top-1 CTC log-prob / argmax of the first and the last chunk.  r640_1h is bench.py's workload."""

    def __init__(self, name):
        super().__init__(name)
        assert self.c.get("long")

    @property
    def sd(self):
        key = (self.c["dims"], self.c["seed"], self.js["gamma"], self.js["beta"])
        if key not in _SD_CACHE:
            _SD_CACHE.clear()            # one big state dict at a time (r640: 2.7 GB)
            _SD_CACHE[key] = synth.make_state_dict(self.cfg, self.c["seed"], self.js["gamma"], self.js["beta"])
        return _SD_CACHE[key]

    def chunk_feats(self, c):
        """(x [1, chunk, 80], lens [1]) of chunk c from the oracle fbank of only the samples that chunk covers."""
        f0 = c * self.chunk
        n = self.js["lens"][c]
        pcm = self.pcm[f0 * 160:(f0 + n - 1) * 160 + 400]
        feats = fbank_ref.fbank(pcm)
        assert feats.shape[0] == n
        x = np.zeros((1, self.chunk, 80), np.float32)
        x[0, :n] = feats
        return x, np.array([n], np.int32)


class CausalCase:
    """Golden of a causal-convolution model (oracle/gen_golden_causal.py): config, weights and features rebuilt from the json."""

    def __init__(self, name):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.js = json.load(f)
        c = self.c = self.js["case"]
        self.cfg = synth.make_config(c["dims"], c["norm"], causal=c["causal"], use_dynamic_chunk=c["use_dynamic_chunk"],
                                     cnn_module_kernel=c["cnn_module_kernel"], pass_cat_emb=c.get("pass_cat_emb", True))
        self.sd = synth.make_state_dict(self.cfg, c["seed"], self.js["gamma"], self.js["beta"])
        self.feats = fbank_ref.fbank(synth.synth_audio(c["seconds"], seed=1234 + c["seed"]))
        npz = os.path.join(GOLDEN, name + ".npz")
        self.arrays = np.load(npz) if os.path.exists(npz) else None
        self.beam, self.ctc_weight, self.reverse_weight, self.cat, self.chunk = (c["beam"], c["ctc_weight"], c["reverse_weight"],
                                                                                c["cat"], c["chunk"])
        nch = -(-self.feats.shape[0] // self.chunk)
        self.x = np.zeros((nch, self.chunk, 80), np.float32)
        self.lens = np.zeros(nch, np.int32)
        for i in range(nch):
            part = self.feats[i * self.chunk:(i + 1) * self.chunk]
            self.x[i, :len(part)] = part
            self.lens[i] = len(part)
        assert self.lens.tolist() == self.js["lens"]


class JointCase:
    """Golden of `joint_decoding` (oracle/gen_golden_joint.py: the reference's BeamSearchTimeSync class on (1, len, d) memories)."""

    def __init__(self, name):
        with open(os.path.join(GOLDEN, name + ".json")) as f:
            self.js = json.load(f)
        c = self.c = self.js["case"]
        self.cfg = synth.make_config(c["dims"], c["norm"])
        self.sd = synth.make_state_dict(self.cfg, c["seed"], self.js["gamma"], self.js["beta"])
        feats = fbank_ref.fbank(synth.synth_audio(c["seconds"], seed=1234 + c["seed"]))
        self.chunk, self.cat = c["chunk"], c["cat"]
        nch = -(-feats.shape[0] // self.chunk)
        self.x = np.zeros((nch, self.chunk, 80), np.float32)
        self.lens = np.zeros(nch, np.int32)
        for i in range(nch):
            part = feats[i * self.chunk:(i + 1) * self.chunk]
            self.x[i, :len(part)] = part
            self.lens[i] = len(part)
        assert self.lens.tolist() == self.js["lens"]
