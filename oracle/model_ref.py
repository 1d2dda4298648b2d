"""TEST INFRASTRUCTURE ONLY (oracle) -- never imported by the product path.

CPU restatement (plain torch fp32 ops, functional, driven by a state dict) of the
Reverb-ASR forward pass on the `recognize_wav` hot path.  Every function cites
the reference file:line it follows (paths relative to /root/reference/asr/wenet).

PINNED: `oracle/gen_golden.py` runs the unmodified reference (imported through
`oracle/ref_shim.py`) and this restatement on the same synthetic weights and
inputs; `tests/test_oracle_vs_golden.py` checks this file against the committed
golden outputs (tests/golden/*.npz) on every CPU run, and
`tests/test_oracle_vs_reference.py` against the live reference when
/root/reference is present.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this module -- as the checker / the timed CPU baseline, never as the
thing shipped.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def to_torch_sd(sd) -> SD:
    out = {}
    for k, v in sd.items():
        out[k] = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
    return out


def _lin(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _ln(sd: SD, p: str, x: torch.Tensor, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def make_pad_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """utils/mask.py:200-233 -- True at padded positions."""
    return torch.arange(max_len)[None, :] >= lengths[:, None].to(torch.long)


def sinusoid_pe(length: int, d: int) -> torch.Tensor:
    """transformer/embedding.py:48-56 (PositionalEncoding table, first `length` rows)."""
    pe = torch.zeros(length, d)
    position = torch.arange(0, length, dtype=torch.float32).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


# --------------------------------------------------------------------------- encoder
def global_cmvn(sd: SD, x: torch.Tensor) -> torch.Tensor:
    """transformer/cmvn.py:36-47."""
    return (x - sd["encoder.global_cmvn.mean"]) * sd["encoder.global_cmvn.istd"]


def conv2d_subsampling4(sd: SD, x: torch.Tensor, mask: torch.Tensor):
    """transformer/subsampling.py:201-226 + RelPositionalEncoding.forward embedding.py:132-146.
    x (B,T,80) -> (B,T',d), pos_emb (1,T',d), mask (B,1,T')."""
    x = x.unsqueeze(1)
    x = F.relu(F.conv2d(x, sd["encoder.embed.conv.0.weight"], sd["encoder.embed.conv.0.bias"], stride=2))
    x = F.relu(F.conv2d(x, sd["encoder.embed.conv.2.weight"], sd["encoder.embed.conv.2.bias"], stride=2))
    b, c, t, f = x.shape
    x = _lin(sd, "encoder.embed.out.0", x.transpose(1, 2).contiguous().view(b, t, c * f))
    d = x.shape[-1]
    x = x * math.sqrt(d)
    pos_emb = sinusoid_pe(t, d).unsqueeze(0)
    return x, pos_emb, mask[:, :, 2::2][:, :, 2::2]


def rel_pos_mhsa(sd: SD, p: str, x: torch.Tensor, mask: torch.Tensor, pos_emb: torch.Tensor, h: int):
    """transformer/attention.py:317-399 (no rel_shift) + forward_attention :81-127."""
    B, T, d = x.shape
    dk = d // h
    q = _lin(sd, p + ".linear_q", x).view(B, T, h, dk)
    k = _lin(sd, p + ".linear_k", x).view(B, T, h, dk).transpose(1, 2)
    v = _lin(sd, p + ".linear_v", x).view(B, T, h, dk).transpose(1, 2)
    pp = F.linear(pos_emb, sd[p + ".linear_pos.weight"]).view(pos_emb.shape[0], -1, h, dk).transpose(1, 2)
    q_u = (q + sd[p + ".pos_bias_u"]).transpose(1, 2)
    q_v = (q + sd[p + ".pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(q_u, k.transpose(-2, -1))
    bd = torch.matmul(q_v, pp.transpose(-2, -1))
    scores = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)                      # (B,1,1,T) True = masked key
    scores = scores.masked_fill(m, -float("inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, d)
    return _lin(sd, p + ".linear_out", o)


def conv_module(sd: SD, p: str, x: torch.Tensor, mask_pad: torch.Tensor, norm: str, causal: bool = False, cache=None):
    """transformer/convolution.py:89-144.  causal (lorder = K-1, :55-57): the module INPUT is padded on the left with
    lorder zero frames, or with `cache` (B, d, cache_t) when one is given (:113-121); returns (out, new_cache) with
    new_cache = the last lorder input frames, or None for the symmetric convolution."""
    x = x.transpose(1, 2).clone()
    x.masked_fill_(~mask_pad, 0.0)
    w = sd[p + ".depthwise_conv.weight"]
    lorder = w.shape[-1] - 1 if causal else 0
    new_cache = None
    if lorder > 0:
        x = F.pad(x, (lorder, 0)) if cache is None else torch.cat((cache, x), dim=2)
        new_cache = x[:, :, -lorder:]
    x = F.conv1d(x, sd[p + ".pointwise_conv1.weight"], sd[p + ".pointwise_conv1.bias"])
    x = F.glu(x, dim=1)
    x = F.conv1d(x, w, sd[p + ".depthwise_conv.bias"], padding=0 if causal else (w.shape[-1] - 1) // 2, groups=w.shape[0])
    if norm == "layer_norm":
        x = _ln(sd, p + ".norm", x.transpose(1, 2), 1e-5).transpose(1, 2)
    else:  # BatchNorm1d in eval mode
        x = F.batch_norm(x, sd[p + ".norm.running_mean"], sd[p + ".norm.running_var"],
                         sd[p + ".norm.weight"], sd[p + ".norm.bias"], False, 0.0, 1e-5)
    x = F.silu(x)
    x = F.conv1d(x, sd[p + ".pointwise_conv2.weight"], sd[p + ".pointwise_conv2.bias"])
    x.masked_fill_(~mask_pad, 0.0)
    return (x.transpose(1, 2), new_cache) if causal else x.transpose(1, 2)


def ffn(sd: SD, p: str, x: torch.Tensor, act) -> torch.Tensor:
    """transformer/positionwise_feed_forward.py:47-55."""
    return _lin(sd, p + ".w_2", act(_lin(sd, p + ".w_1", x)))


def lsl_mix(sd: SD, p: str, x: torch.Tensor, cat_embs: torch.Tensor) -> torch.Tensor:
    """encoder_layer.py:378-390 / decoder_layer.py:319-330 (1-D cat_embs)."""
    y = None
    i = 0
    while (p + f".language_layers.{i}.weight") in sd:
        t = cat_embs[i] * _lin(sd, p + f".language_layers.{i}", x)
        y = t if y is None else y + t
        i += 1
    return y


def conformer_layer(sd: SD, p: str, x, mask, pos_emb, mask_pad, h, norm, cat_embs, is_lsl, causal: bool = False):
    """encoder_layer.py:164-244 (regular) and :305-402 (language-specific)."""
    x = x + 0.5 * ffn(sd, p + ".feed_forward_macaron", _ln(sd, p + ".norm_ff_macaron", x, 1e-5), F.silu)
    x = x + rel_pos_mhsa(sd, p + ".self_attn", _ln(sd, p + ".norm_mha", x, 1e-5), mask, pos_emb, h)
    c = conv_module(sd, p + ".conv_module", _ln(sd, p + ".norm_conv", x, 1e-5), mask_pad, norm, causal)
    x = x + (c[0] if causal else c)
    residual = x
    z = _ln(sd, p + ".norm_ff", x, 1e-5)
    if is_lsl:
        y = lsl_mix(sd, p, z, cat_embs)
        x = residual + 0.5 * ffn(sd, p + ".feed_forward", y, F.silu)
        x = _ln(sd, p + ".norm_final", x, 1e-5)
        return x + y
    x = residual + 0.5 * ffn(sd, p + ".feed_forward", z, F.silu)
    return _ln(sd, p + ".norm_final", x, 1e-5)


def subsequent_chunk_mask(size: int, chunk_size: int, num_left_chunks: int = -1) -> torch.Tensor:
    """utils/mask.py:86-123."""
    ret = torch.zeros(size, size, dtype=torch.bool)
    for i in range(size):
        start = 0 if num_left_chunks < 0 else max((i // chunk_size - num_left_chunks) * chunk_size, 0)
        ret[i, start:min((i // chunk_size + 1) * chunk_size, size)] = True
    return ret


def encoder_forward(sd: SD, cfg: dict, feats: torch.Tensor, feats_lens: torch.Tensor,
                    cat_embs: torch.Tensor, taps: Optional[dict] = None, decoding_chunk_size: int = -1,
                    num_decoding_left_chunks: int = -1):
    """transformer/encoder.py:117-149; the attention mask is add_optional_chunk_mask's (utils/mask.py:126-197,
    decoding branches): full context unless the model has use_dynamic_chunk and decoding_chunk_size > 0, or a
    static_chunk_size.  feats (B,T0,80) raw log-mel; returns (B,T',d), mask (B,1,T')."""
    ec = cfg["encoder_conf"]
    h, nb, norm = ec["attention_heads"], ec["num_blocks"], ec.get("cnn_module_norm", "batch_norm")
    T0 = feats.shape[1]
    masks = ~make_pad_mask(feats_lens, T0).unsqueeze(1)
    x = global_cmvn(sd, feats)
    x, pos_emb, masks = conv2d_subsampling4(sd, x, masks)
    if taps is not None:
        taps["embed"] = x.clone()
    has_lsl = "encoder.encoders.0.language_layers.0.weight" in sd
    att_mask = masks
    L = x.shape[1]
    if ec.get("use_dynamic_chunk", False):
        if decoding_chunk_size > 0:
            att_mask = masks & subsequent_chunk_mask(L, decoding_chunk_size, num_decoding_left_chunks).unsqueeze(0)
    elif int(ec.get("static_chunk_size", 0)) > 0:
        att_mask = masks & subsequent_chunk_mask(L, int(ec["static_chunk_size"]), num_decoding_left_chunks).unsqueeze(0)
    for i in range(nb):
        is_lsl = has_lsl and i in (0, nb - 1)
        x = conformer_layer(sd, f"encoder.encoders.{i}", x, att_mask, pos_emb, masks, h, norm, cat_embs, is_lsl,
                            bool(ec.get("causal", False)))
        if taps is not None:
            taps[f"layer{i}"] = x.clone()
    x = _ln(sd, "encoder.after_norm", x, 1e-5)
    return x, masks


# --------------------------------------------------------------------------- streaming encoder
def encoder_forward_chunk(sd: SD, cfg: dict, xs: torch.Tensor, offset: int, required_cache_size: int, att_cache,
                          cat_embs: torch.Tensor, cnn_cache=None):
    """BaseEncoder.forward_chunk (transformer/encoder.py:231-341).  xs (1, time, 80); att_cache: None or a list with
    one (k, v) pair of (1, h, cache_t1, dk) tensors per layer; cnn_cache (causal convolution modules only,
    convolution.py:113-121): None or a list with one (1, d, lorder) tensor per layer.  Returns (ys (1, chunk, d),
    new att_cache) -- and the new cnn_cache as a third value when the model is causal."""
    ec = cfg["encoder_conf"]
    h, nb, norm = ec["attention_heads"], ec["num_blocks"], ec.get("cnn_module_norm", "batch_norm")
    causal = bool(ec.get("causal", False))
    new_cnn = []
    assert xs.shape[0] == 1
    x = global_cmvn(sd, xs)
    fake = torch.ones(1, 1, xs.shape[1], dtype=torch.bool)
    x, _, _ = conv2d_subsampling4(sd, x, fake)                      # x * sqrt(d); pos_emb is recomputed below
    d = x.shape[-1]
    chunk = x.shape[1]
    cache_t1 = 0 if att_cache is None else att_cache[0][0].shape[2]
    key_size = cache_t1 + chunk
    pos_emb = sinusoid_pe(offset + chunk, d)[offset - cache_t1:offset + chunk].unsqueeze(0)     # encoder.py:305-306
    if required_cache_size < 0:
        start = 0
    elif required_cache_size == 0:
        start = key_size
    else:
        start = max(key_size - required_cache_size, 0)
    has_lsl = "encoder.encoders.0.language_layers.0.weight" in sd
    dk = d // h
    new_cache = []
    for i in range(nb):
        p = f"encoder.encoders.{i}"
        is_lsl = has_lsl and i in (0, nb - 1)
        x = x + 0.5 * ffn(sd, p + ".feed_forward_macaron", _ln(sd, p + ".norm_ff_macaron", x, 1e-5), F.silu)
        # attention.py:317-399 with the cache branch :361-369 and no mask (fake (0,0,0) att_mask, :112)
        z = _ln(sd, p + ".norm_mha", x, 1e-5)
        a = p + ".self_attn"
        q = _lin(sd, a + ".linear_q", z).view(1, chunk, h, dk)
        k = _lin(sd, a + ".linear_k", z).view(1, chunk, h, dk).transpose(1, 2)
        v = _lin(sd, a + ".linear_v", z).view(1, chunk, h, dk).transpose(1, 2)
        if att_cache is not None:
            k = torch.cat([att_cache[i][0], k], dim=2)
            v = torch.cat([att_cache[i][1], v], dim=2)
        new_cache.append((k[:, :, start:], v[:, :, start:]))
        pp = F.linear(pos_emb, sd[a + ".linear_pos.weight"]).view(1, -1, h, dk).transpose(1, 2)
        q_u = (q + sd[a + ".pos_bias_u"]).transpose(1, 2)
        q_v = (q + sd[a + ".pos_bias_v"]).transpose(1, 2)
        scores = (torch.matmul(q_u, k.transpose(-2, -1)) + torch.matmul(q_v, pp.transpose(-2, -1))) / math.sqrt(dk)
        o = torch.matmul(torch.softmax(scores, dim=-1), v).transpose(1, 2).contiguous().view(1, chunk, d)
        x = x + _lin(sd, a + ".linear_out", o)
        c = conv_module(sd, p + ".conv_module", _ln(sd, p + ".norm_conv", x, 1e-5), torch.ones(1, 1, chunk, dtype=torch.bool), norm,
                        causal, None if cnn_cache is None else cnn_cache[i])
        if causal:
            new_cnn.append(c[1])
            c = c[0]
        x = x + c
        residual = x
        z = _ln(sd, p + ".norm_ff", x, 1e-5)
        if is_lsl:
            y = lsl_mix(sd, p, z, cat_embs)
            x = _ln(sd, p + ".norm_final", residual + 0.5 * ffn(sd, p + ".feed_forward", y, F.silu), 1e-5) + y
        else:
            x = _ln(sd, p + ".norm_final", residual + 0.5 * ffn(sd, p + ".feed_forward", z, F.silu), 1e-5)
    if causal:
        return _ln(sd, "encoder.after_norm", x, 1e-5), new_cache, new_cnn
    return _ln(sd, "encoder.after_norm", x, 1e-5), new_cache


def encoder_forward_chunk_by_chunk(sd: SD, cfg: dict, xs: torch.Tensor, decoding_chunk_size: int,
                                   num_decoding_left_chunks: int, cat_embs: torch.Tensor, return_cnn_cache: bool = False):
    """BaseEncoder.forward_chunk_by_chunk (encoder.py:343-402).  xs (1, T, 80) -> (1, T', d), final cache length
    (and, on request, the final cnn cache as the reference stacks it: (layers, 1, d, lorder), encoder.py:332,340)."""
    assert decoding_chunk_size > 0
    subsampling, context = 4, 7
    stride = subsampling * decoding_chunk_size
    window = (decoding_chunk_size - 1) * subsampling + context
    required = decoding_chunk_size * num_decoding_left_chunks
    cache, cnn, outs, offset = None, None, [], 0
    for cur in range(0, xs.shape[1] - context + 1, stride):
        r = encoder_forward_chunk(sd, cfg, xs[:, cur:min(cur + window, xs.shape[1])], offset, required, cache, cat_embs, cnn)
        y, cache = r[0], r[1]
        cnn = r[2] if len(r) > 2 else None
        outs.append(y)
        offset += y.shape[1]
    n_cache = 0 if cache is None else cache[0][0].shape[2]
    if return_cnn_cache:
        return torch.cat(outs, 1), n_cache, (None if cnn is None else torch.stack(cnn, 0))
    return torch.cat(outs, 1), n_cache


def ctc_logprobs(sd: SD, enc: torch.Tensor, blank_penalty: float = 0.0, blank_id: int = 0):
    """asr_model.py:318-329 + ctc.py:106-114."""
    logits = _lin(sd, "ctc.ctc_lo", enc)
    if blank_penalty > 0.0:
        logits[:, :, blank_id] -= blank_penalty
    return logits.log_softmax(dim=2)


# --------------------------------------------------------------------------- decoder
def mha(sd: SD, p: str, q_in, kv_in, mask, h: int):
    """transformer/attention.py:129-197 (MultiHeadedAttention)."""
    B, L, d = q_in.shape
    dk = d // h
    q = _lin(sd, p + ".linear_q", q_in).view(B, -1, h, dk).transpose(1, 2)
    k = _lin(sd, p + ".linear_k", kv_in).view(B, -1, h, dk).transpose(1, 2)
    v = _lin(sd, p + ".linear_v", kv_in).view(B, -1, h, dk).transpose(1, 2)
    scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(dk)
    m = mask.unsqueeze(1).eq(0)
    scores = scores.masked_fill(m, -float("inf"))
    attn = torch.softmax(scores, dim=-1).masked_fill(m, 0.0)
    o = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, -1, d)
    return _lin(sd, p + ".linear_out", o)


def decoder_forward(sd: SD, cfg: dict, side: str, memory, memory_mask, ys_in, ys_lens, cat_embs):
    """decoder.py:116-169 (TransformerDecoder.forward) with the language-specific first/last
    layers of decoder.py:308-383; layers decoder_layer.py:62-133 and :251-340."""
    dc = cfg["decoder_conf"]
    h = dc["attention_heads"]
    p = f"decoder.{side}"
    L = ys_in.shape[1]
    tgt_mask = ~make_pad_mask(ys_lens, L).unsqueeze(1)                       # (B,1,L)
    sub = torch.tril(torch.ones(L, L, dtype=torch.bool)).unsqueeze(0)        # mask.py:52-85
    tgt_mask = tgt_mask & sub
    d = sd[p + ".embed.0.weight"].shape[1]
    x = F.embedding(ys_in, sd[p + ".embed.0.weight"]) * math.sqrt(d) + sinusoid_pe(L, d).unsqueeze(0)
    j = 0
    while (p + f".decoders.{j}.norm1.weight") in sd:
        q = p + f".decoders.{j}"
        is_lsl = (q + ".language_layers.0.weight") in sd
        eps = 1e-12 if is_lsl else 1e-5                                      # decoder_layer.py:241-243
        t = _ln(sd, q + ".norm1", x, eps)
        x = x + mha(sd, q + ".self_attn", t, t, tgt_mask, h)
        x = x + mha(sd, q + ".src_attn", _ln(sd, q + ".norm2", x, eps), memory, memory_mask, h)
        z = _ln(sd, q + ".norm3", x, eps)
        if is_lsl:
            z = lsl_mix(sd, q, z, cat_embs)
        x = x + ffn(sd, q + ".feed_forward", z, F.relu)
        j += 1
    x = _ln(sd, p + ".after_norm", x, 1e-5)
    return _lin(sd, p + ".output_layer", x)


def decoder_step(sd: SD, cfg: dict, side: str, memory, memory_mask, tok, pos: int, cache, cat_embs):
    """One autoregressive position of decoder_forward for every row (decoder.py:191-234 computes the same thing
    from cached layer outputs; here the cache holds each layer's self-attention keys/values, which is the same
    function of the prefix).  tok (R,) int64, cache: list of [k, v] per layer or None.  -> logits (R, V), cache."""
    dc = cfg["decoder_conf"]
    h = dc["attention_heads"]
    p = f"decoder.{side}"
    d = sd[p + ".embed.0.weight"].shape[1]
    dk = d // h
    R = tok.shape[0]
    x = F.embedding(tok, sd[p + ".embed.0.weight"]) * math.sqrt(d) + sinusoid_pe(pos + 1, d)[pos].unsqueeze(0)   # (R, d)
    new_cache = []
    j = 0
    while (p + f".decoders.{j}.norm1.weight") in sd:
        q = p + f".decoders.{j}"
        is_lsl = (q + ".language_layers.0.weight") in sd
        eps = 1e-12 if is_lsl else 1e-5
        t = _ln(sd, q + ".norm1", x, eps)
        k_new = _lin(sd, q + ".self_attn.linear_k", t).view(R, h, 1, dk)
        v_new = _lin(sd, q + ".self_attn.linear_v", t).view(R, h, 1, dk)
        if cache is not None:
            k_all, v_all = torch.cat([cache[j][0], k_new], 2), torch.cat([cache[j][1], v_new], 2)
        else:
            k_all, v_all = k_new, v_new
        new_cache.append([k_all, v_all])
        qq = _lin(sd, q + ".self_attn.linear_q", t).view(R, h, 1, dk)
        att = torch.softmax(torch.matmul(qq, k_all.transpose(-2, -1)) / math.sqrt(dk), dim=-1)
        o = torch.matmul(att, v_all).transpose(1, 2).reshape(R, d)
        x = x + _lin(sd, q + ".self_attn.linear_out", o)
        x = x + mha(sd, q + ".src_attn", _ln(sd, q + ".norm2", x, eps).unsqueeze(1), memory, memory_mask, h).squeeze(1)
        z = _ln(sd, q + ".norm3", x, eps)
        if is_lsl:
            z = lsl_mix(sd, q, z, cat_embs)
        x = x + ffn(sd, q + ".feed_forward", z, F.relu)
        j += 1
    x = _ln(sd, p + ".after_norm", x, 1e-5)
    return _lin(sd, p + ".output_layer", x), new_cache


def reverse_hyps(hyps: torch.Tensor, hyps_lens: torch.Tensor, eos: int) -> torch.Tensor:
    """asr_model.py:896-953: right-to-left decoder input built from sos-prefixed hyps."""
    r_lens = hyps_lens - 1
    r = hyps[:, 1:]
    max_len = int(torch.max(r_lens))
    idx_range = torch.arange(0, max_len)
    seq_mask = r_lens.unsqueeze(1) > idx_range
    index = ((r_lens.unsqueeze(1) - 1) - idx_range) * seq_mask
    r = torch.gather(r, 1, index)
    r = torch.where(seq_mask, r, eos)
    return torch.cat([hyps[:, 0:1], r], dim=1)


def forward_attention_decoder(sd: SD, cfg: dict, hyps, hyps_lens, encoder_out, reverse_weight, cat_embs):
    """asr_model.py:868-978."""
    n = hyps.shape[0]
    eos = cfg["output_dim"] - 1
    mem = encoder_out.repeat(n, 1, 1)
    mem_mask = torch.ones(n, 1, mem.shape[1], dtype=torch.bool)
    out = decoder_forward(sd, cfg, "left_decoder", mem, mem_mask, hyps, hyps_lens, cat_embs)
    out = F.log_softmax(out, dim=-1)
    r_out = torch.tensor(0.0)
    if reverse_weight > 0.0:
        r_hyps = reverse_hyps(hyps, hyps_lens, eos)
        r_out = decoder_forward(sd, cfg, "right_decoder", mem, mem_mask, r_hyps, hyps_lens, cat_embs)
    r_out = F.log_softmax(r_out, dim=-1)
    return out, r_out
