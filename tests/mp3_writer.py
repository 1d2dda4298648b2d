"""TEST INFRASTRUCTURE: an independent MPEG audio Layer III ENCODER, just enough of one to drive every path of the decoder in
reverb_amd/csrc/mp3.cpp with streams whose content is known (the counterpart of tests/flac_writer.py).  Written from the
published algorithm (ISO/IEC 11172-3 Annex C: polyphase analysis filterbank, MDCT with the four window shapes, aliasing
butterflies; 13818-3 for the lower sampling frequencies); no psychoacoustics -- one quantiser step per granule found by
bisection against a bit budget -- but real syntax: MPEG-1 and LSF headers, side information, scale factors (scfsi, LSF
partitions), all Huffman tables incl. linbits and both count1 tables, the bit reservoir (main_data_begin), M/S and intensity
stereo, long / start / short / stop / mixed blocks, CRC-16, a Xing/Info + LAME tag frame.
"""
import math

import numpy as np

PI = math.pi
SFB_L = {44100: [0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576],
         48000: [0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576],
         32000: [0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576],
         22050: [0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576],
         24000: [0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576],
         16000: [0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576]}
SFB_S = {44100: [0, 4, 8, 12, 16, 22, 30, 40, 52, 66, 84, 106, 136, 192], 48000: [0, 4, 8, 12, 16, 22, 28, 38, 50, 64, 80, 100, 126, 192],
         32000: [0, 4, 8, 12, 16, 22, 30, 42, 58, 78, 104, 138, 180, 192], 22050: [0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192],
         24000: [0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192], 16000: [0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192]}
SFB_L[8000] = [0, 12, 24, 36, 48, 60, 72, 88, 108, 132, 160, 192, 232, 280, 336, 400, 476, 566, 568, 570, 572, 574, 576]
SFB_S[8000] = [0, 8, 16, 24, 36, 52, 72, 96, 124, 160, 162, 164, 166, 192]
for _r, _b in ((11025, 16000), (12000, 16000)):
    SFB_L[_r], SFB_S[_r] = SFB_L[_b], SFB_S[_b]
LINBITS = [0] * 16 + [1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13]
SLEN = [(0, 0), (0, 1), (0, 2), (0, 3), (3, 0), (1, 1), (1, 2), (1, 3), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (3, 3), (4, 2), (4, 3)]
PRETAB = [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0]


class Bits:
    def __init__(self):
        self.b = []

    def put(self, v, n):
        assert 0 <= v < (1 << n) or n == 0, (v, n)
        for k in range(n - 1, -1, -1):
            self.b.append((v >> k) & 1)

    def __len__(self):
        return len(self.b)

    def to_bytes(self):
        b = self.b + [0] * (-len(self.b) % 8)
        return bytes(int("".join(map(str, b[i:i + 8])), 2) for i in range(0, len(b), 8))


# ------------------------------------------------------------------------------------------------ filterbank, MDCT (float64)
def analysis_filterbank(x, window_d):
    """x: samples (float), length a multiple of 32 -> subband samples [n/32][32] (11172-3 C.1.3); C[i] = D[i] / 32"""
    Cw = np.asarray(window_d, np.float64) / 32.0
    i = np.arange(64)[None, :]
    k = np.arange(32)[:, None]
    M = np.cos((2 * k + 1) * (i - 16) * PI / 64)
    X = np.zeros(512)
    out = np.zeros((len(x) // 32, 32))
    for t in range(len(x) // 32):
        X[32:] = X[:-32]
        X[:32] = x[32 * t:32 * t + 32][::-1]
        Z = Cw * X
        Y = Z.reshape(8, 64).sum(0)
        out[t] = M @ Y
    return out


def mdct_windows():
    w = np.zeros((4, 36))
    i = np.arange(36)
    w[0] = np.sin(PI / 36 * (i + 0.5))
    w[1, :18] = w[0, :18]; w[1, 18:24] = 1.0; w[1, 24:30] = np.sin(PI / 12 * (np.arange(24, 30) - 18 + 0.5))
    w[3, 6:12] = np.sin(PI / 12 * (np.arange(6, 12) - 6 + 0.5)); w[3, 12:18] = 1.0; w[3, 18:] = w[0, 18:]
    w[2, :12] = np.sin(PI / 12 * (np.arange(12) + 0.5))
    return w


W = mdct_windows()
_i36, _k18 = np.arange(36)[None, :], np.arange(18)[:, None]
COS36 = np.cos(PI / 72 * (2 * _i36 + 1 + 18) * (2 * _k18 + 1))
_i12, _k6 = np.arange(12)[None, :], np.arange(6)[:, None]
COS12 = np.cos(PI / 24 * (2 * _i12 + 1 + 6) * (2 * _k6 + 1))
_c = np.array([-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037])
CS, CA = 1 / np.sqrt(1 + _c * _c), _c / np.sqrt(1 + _c * _c)


def mdct_granule(prev_sb, cur_sb, block_type, mixed=False):
    """prev_sb, cur_sb: [18][32] subband samples of the previous / this granule -> 576 MDCT lines in the order the bit stream
    stores them (short blocks: band by band, window by window is done by the caller through `to_stream_order`): here the
    decoder's input order before reordering is NOT applied -- lines are [subband][18] with short windows interleaved 3k + w."""
    xr = np.zeros(576)
    for sb in range(32):
        z = np.concatenate([prev_sb[:, sb], cur_sb[:, sb]]).copy()
        if sb & 1:
            z[1::2] = -z[1::2]                  # frequency inversion (the granule length is even: parity of the index survives)
        bt = 0 if (mixed and sb < 2) else block_type      # mixed_block_flag: normal window in the two lowest subbands
        if bt != 2:
            xr[18 * sb:18 * sb + 18] = COS36 @ (z * W[bt]) / 9.0
        else:
            for w in range(3):
                seg = z[6 + 6 * w:18 + 6 * w] * W[2, :12]
                xr[18 * sb + w:18 * sb + 18:3] = COS12 @ seg / 3.0
    nsb = 32 if block_type != 2 else (2 if mixed else 0)
    for sb in range(1, nsb):                     # aliasing butterflies, the encoder's direction
        for i in range(8):
            a, b = xr[18 * sb - 1 - i], xr[18 * sb + i]
            xr[18 * sb - 1 - i] = a * CS[i] + b * CA[i]
            xr[18 * sb + i] = b * CS[i] - a * CA[i]
    return xr


def to_stream_order(xr, sr, block_type, mixed):
    """decoder-side line order (3k + w inside a short band) -> the order of the bit stream (band, window, line)"""
    if block_type != 2:
        return xr.copy()
    S = SFB_S[sr]
    out = xr.copy()
    for b in range(3 if mixed else 0, 13):
        width, base = S[b + 1] - S[b], 3 * S[b]
        for w in range(3):
            for k in range(width):
                out[base + w * width + k] = xr[base + 3 * k + w]
    return out


def from_stream_order(xs, sr, block_type, mixed):
    """inverse of to_stream_order"""
    if block_type != 2:
        return xs.copy()
    S = SFB_S[sr]
    out = xs.copy()
    for b in range(3 if mixed else 0, 13):
        width, base = S[b + 1] - S[b], 3 * S[b]
        for w in range(3):
            for k in range(width):
                out[base + 3 * k + w] = xs[base + w * width + k]
    return out


# ------------------------------------------------------------------------------------------------ Huffman coding
def _tables():
    from mp3_tables import T
    return T


def table_max(t):
    T = _tables()
    if t == 0:
        return 0
    n = T[t if t < 16 else (16 if t < 24 else 24)][0]
    return n - 1 + ((1 << LINBITS[t]) - 1 if LINBITS[t] else 0)


def code_pairs(bits, vals, t):
    """Huffman-code the (x, y) pairs of `vals` with table t"""
    if t == 0:
        assert not np.any(vals)
        return
    T = _tables()
    n, hb, hl = T[t if t < 16 else (16 if t < 24 else 24)]
    lb = LINBITS[t]
    for i in range(0, len(vals), 2):
        x, y = int(vals[i]), int(vals[i + 1])
        ax, ay = abs(x), abs(y)
        ex, ey = min(ax, 15) if lb else ax, min(ay, 15) if lb else ay
        assert ex < n and ey < n, (t, x, y)
        bits.put(hb[ex * n + ey], hl[ex * n + ey])
        if lb and ax >= 15:
            bits.put(ax - 15, lb)
        if ax:
            bits.put(1 if x < 0 else 0, 1)
        if lb and ay >= 15:
            bits.put(ay - 15, lb)
        if ay:
            bits.put(1 if y < 0 else 0, 1)


def code_count1(bits, vals, tsel):
    T = _tables()
    _, hb, hl = T[32 + tsel]
    for i in range(0, len(vals), 4):
        q = [int(v) for v in vals[i:i + 4]]
        s = (abs(q[0]) << 3) | (abs(q[1]) << 2) | (abs(q[2]) << 1) | abs(q[3])
        bits.put(hb[s], hl[s])
        for v in q:
            if v:
                bits.put(1 if v < 0 else 0, 1)


def split_regions(ix):
    """-> (big_values pairs count, count1 quadruples count) of a quantised granule"""
    n = 576
    while n > 1 and ix[n - 1] == 0 and ix[n - 2] == 0:
        n -= 2
    c1 = 0
    while n > 3 and np.abs(ix[n - 4:n]).max() <= 1:
        n -= 4
        c1 += 1
    return n // 2, c1


class GranuleCfg:
    def __init__(self, block_type=0, mixed=False, scalefac_scale=0, preflag=0, subblock_gain=(0, 0, 0)):
        self.block_type, self.mixed, self.scalefac_scale, self.preflag = block_type, mixed, scalefac_scale, preflag
        self.subblock_gain = tuple(subblock_gain)


def pick_table(rng, mx, prefer=None):
    if mx == 0:
        return 0
    ok = [t for t in range(1, 32) if t not in (4, 14) and table_max(t) >= mx]
    # near-minimal tables most of the time (fewer bits), any valid one sometimes (coverage)
    small = [t for t in ok if table_max(t) <= max(2 * mx, 3)] or ok
    return int(rng.choice(small if rng.random() < 0.8 else ok))


# ------------------------------------------------------------------------------------------------ granule coding
NR_OF_SFB = [[[6, 5, 5, 5], [9, 9, 9, 9], [6, 9, 9, 9]], [[6, 5, 7, 3], [9, 9, 12, 6], [6, 9, 12, 6]], [[11, 10, 0, 0], [18, 18, 0, 0], [15, 18, 0, 0]],
             [[7, 7, 7, 0], [12, 12, 12, 0], [6, 15, 12, 0]], [[6, 6, 6, 3], [12, 9, 9, 6], [6, 12, 9, 6]], [[8, 8, 5, 0], [15, 12, 9, 0], [6, 18, 9, 0]]]


def band_layout(sr, lsf, cfg):
    """[(first line in stream order, width, kind 'l'/'s', sfb, window)] of the granule"""
    Lb, Sb = SFB_L[sr], SFB_S[sr]
    out = []
    if cfg.block_type != 2:
        return [(Lb[b], Lb[b + 1] - Lb[b], "l", b, -1) for b in range(22)]
    if cfg.mixed:
        out += [(Lb[b], Lb[b + 1] - Lb[b], "l", b, -1) for b in range(6 if lsf else 8)]
    for b in range(3 if cfg.mixed else 0, 13):
        wd = Sb[b + 1] - Sb[b]
        out += [(3 * Sb[b] + w * wd, wd, "s", b, w) for w in range(3)]
    return out


def band_scale(cfg, kind, sfb, win, sf_l, sf_s):
    """the decoder's scale of a band, without global_gain"""
    mult = 1.0 if cfg.scalefac_scale else 0.5
    if kind == "l":
        return 2.0 ** (-mult * (sf_l[sfb] + (PRETAB[sfb] if cfg.preflag else 0))) if sfb < 21 else 1.0
    return 2.0 ** (-2.0 * cfg.subblock_gain[win]) * (2.0 ** (-mult * sf_s[win][sfb]) if sfb < 12 else 1.0)


def quantize(xr_stream, sr, lsf, cfg, sf_l, sf_s, global_gain):
    ix = np.zeros(576, np.int64)
    gg = 2.0 ** (0.25 * (global_gain - 210))
    for start, wd, kind, sfb, win in band_layout(sr, lsf, cfg):
        sc = gg * band_scale(cfg, kind, sfb, win, sf_l, sf_s)
        v = np.abs(xr_stream[start:start + wd]) / sc
        ix[start:start + wd] = np.sign(xr_stream[start:start + wd]) * np.floor(v ** 0.75 + 0.4054)
    return ix


def dequantize(ix, sr, lsf, cfg, sf_l, sf_s, global_gain):
    xr = np.zeros(576)
    gg = 2.0 ** (0.25 * (global_gain - 210))
    for start, wd, kind, sfb, win in band_layout(sr, lsf, cfg):
        sc = gg * band_scale(cfg, kind, sfb, win, sf_l, sf_s)
        v = ix[start:start + wd]
        xr[start:start + wd] = np.sign(v) * np.abs(v).astype(np.float64) ** (4.0 / 3.0) * sc
    return xr


def code_spectrum(rng, ix, sr, cfg, version25=False, lsf=False, force=None):
    """-> (Bits of the Huffman data, side-info fields dict)"""
    bv, c1 = split_regions(ix)
    Lb = SFB_L[sr]
    f = dict(big_values=bv, table_select=[0, 0, 0], region0_count=0, region1_count=0, count1table=int(rng.integers(0, 2)))
    ws = cfg.block_type != 0
    if ws:
        if version25:
            r1 = Lb[((5 if (cfg.block_type == 2 and not cfg.mixed) else 7)) + 1]
        else:
            r1 = 36 if (not lsf or cfg.block_type == 2) else 54
        bounds = [min(r1, 2 * bv), 2 * bv, 2 * bv]
    else:
        # region sizes: any split at long-band edges (the decoder reads the counts)
        nb = max(b for b in range(23) if Lb[b] <= 2 * bv) if bv else 0
        r0 = int(rng.integers(0, min(15, max(nb - 1, 0)) + 1))
        r1c = int(rng.integers(0, min(7, max(nb - r0 - 2, 0)) + 1))
        f["region0_count"], f["region1_count"] = r0, r1c
        bounds = [min(Lb[r0 + 1], 2 * bv), min(Lb[min(r0 + r1c + 2, 22)], 2 * bv), 2 * bv]
    bits = Bits()
    lo = 0
    for r in range(3):
        hi = bounds[r]
        seg = ix[lo:hi]
        if r == 2 and ws:
            assert hi == lo
        t = pick_table(rng, int(np.abs(seg).max()) if len(seg) else 0)
        if force and len(seg):
            cands = [u for u in force if table_max(u) >= int(np.abs(seg).max())]
            if cands:
                t = int(rng.choice(cands))
        if r < 2 or not ws:
            f["table_select"][r] = t
        code_pairs(bits, seg, t)
        lo = hi
    code_count1(bits, ix[2 * bv:2 * bv + 4 * c1], f["count1table"])
    return bits, f


def scalefac_bits_mpeg1(cfg, sf_l, sf_s, scfsi, gr, compress):
    s1, s2 = SLEN[compress]
    b = Bits()
    if cfg.block_type == 2:
        if cfg.mixed:
            for k in range(8): b.put(sf_l[k], s1)
            for k in range(3, 6):
                for w in range(3): b.put(sf_s[w][k], s1)
        else:
            for k in range(0, 6):
                for w in range(3): b.put(sf_s[w][k], s1)
        for k in range(6, 12):
            for w in range(3): b.put(sf_s[w][k], s2)
    else:
        for grp, (a, z) in enumerate(((0, 6), (6, 11), (11, 16), (16, 21))):
            if gr == 1 and scfsi[grp]:
                continue
            for k in range(a, z): b.put(sf_l[k], s1 if grp < 2 else s2)
    return b


def lsf_partition(cfg, intensity_right, slen, blocknumber):
    btn = 0 if cfg.block_type != 2 else (2 if cfg.mixed else 1)
    return NR_OF_SFB[blocknumber][btn]


def scalefac_bits_lsf(cfg, sf_l, sf_s, slen, blocknumber):
    """the scale factors in transmission order with the partition's field widths"""
    nr = lsf_partition(cfg, False, slen, blocknumber)
    if cfg.block_type != 2:
        seq = [sf_l[k] for k in range(21)]
    elif not cfg.mixed:
        seq = [sf_s[w][k] for k in range(12) for w in range(3)]
    else:
        seq = [sf_l[k] for k in range(6)] + [sf_s[w][k] for k in range(3, 12) for w in range(3)]
    b = Bits()
    q = 0
    for i in range(4):
        for _ in range(nr[i]):
            v = seq[q] if q < len(seq) else 0
            b.put(v, slen[i])
            q += 1
    return b


def random_scalefactors(rng, cfg, widths_l, widths_s):
    """widths: bit widths per band -> random values that fit"""
    sf_l = [int(rng.integers(0, 1 << w)) if w else 0 for w in widths_l] + [0, 0]
    sf_s = [[int(rng.integers(0, 1 << w)) if w else 0 for w in widths_s] + [0] for _ in range(3)]
    return sf_l, sf_s


# ------------------------------------------------------------------------------------------------ frames
BR1 = [0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320]
BR2 = [0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160]


def crc16(data_bits):
    crc = 0xffff
    for bit in data_bits:
        top = (crc >> 15) & 1
        crc = (crc << 1) & 0xffff
        if bit ^ top:
            crc ^= 0x8005
    return crc


def header_bits(sr, bitrate, padding, mode, mode_ext, crc):
    ver = 3 if sr >= 32000 else (2 if sr >= 16000 else 0)
    base = {3: [44100, 48000, 32000], 2: [22050, 24000, 16000], 0: [11025, 12000, 8000]}[ver]
    b = Bits()
    b.put(0x7ff, 11); b.put(ver, 2); b.put(1, 2); b.put(0 if crc else 1, 1)
    b.put((BR1 if ver == 3 else BR2).index(bitrate), 4); b.put(base.index(sr), 2); b.put(padding, 1); b.put(0, 1)
    b.put(mode, 2); b.put(mode_ext, 2); b.put(0, 1); b.put(1, 1); b.put(0, 2)
    return b


class Encoder:
    """encode(pcm [channels][n]) -> bytes.  `plan(g)` -> GranuleCfg for granule g (same for both channels)."""

    def __init__(self, sr, channels, bitrate, window_d, seed=0, mode=None, mode_ext=0, crc=False, plan=None, bit_share=None,
                 tables=None, intensity=None, scfsi=False, info_frame=None):
        self.sr, self.nch, self.bitrate, self.D = sr, channels, bitrate, np.asarray(window_d, np.float64)
        self.rng = np.random.default_rng(seed)
        self.lsf = sr < 32000
        self.v25 = sr < 16000
        self.mode = (3 if channels == 1 else 0) if mode is None else mode
        self.mode_ext, self.crc = mode_ext, crc
        self.plan = plan or (lambda g: GranuleCfg())
        self.bit_share = bit_share or (lambda f: 1.0)      # fraction of the mean frame budget frame f may use (drives the reservoir)
        self.tables = tables
        self.intensity = intensity                         # (first intensity band (long) / (short), is_pos) or None
        self.scfsi = scfsi
        self.info_frame = info_frame                       # (delay, padding) -> a Xing/Info + LAME tag frame in front
        self.side_bytes = (17 if channels == 1 else 32) if not self.lsf else (9 if channels == 1 else 17)
        self.spf = 576 if self.lsf else 1152
        self.log = []
        self.units = {}          # (granule, channel) -> what was transmitted: quantised lines, scale factors, global gain

    def frame_bytes(self, padding):
        return (72 if self.lsf else 144) * self.bitrate * 1000 // self.sr + padding

    def encode(self, pcm):
        pcm = np.atleast_2d(np.asarray(pcm, np.float64))
        nch, n = pcm.shape
        assert nch == self.nch
        ngr_f = 1 if self.lsf else 2
        nfr = -(-n // self.spf)
        x = np.zeros((nch, nfr * self.spf))
        x[:, :n] = pcm
        sbs = [analysis_filterbank(x[c], self.D).reshape(-1, 18, 32) for c in range(nch)]
        ng = nfr * ngr_f
        cfgs = [self.plan(g) for g in range(ng)]
        # spectra in stream order, per granule and channel
        spec = np.zeros((ng, nch, 576))
        for c in range(nch):
            prev = np.zeros((18, 32))
            for g in range(ng):
                xr = mdct_granule(prev, sbs[c][g], cfgs[g].block_type, cfgs[g].mixed)
                spec[g, c] = to_stream_order(xr, self.sr, cfgs[g].block_type, cfgs[g].mixed)
                prev = sbs[c][g]
        ms = self.nch == 2 and self.mode == 1 and (self.mode_ext & 2)
        ist = self.nch == 2 and self.mode == 1 and (self.mode_ext & 1) and self.intensity is not None
        out = bytearray()
        if self.info_frame is not None:
            out += self._info_frame(nfr)
        pipe_cur = 0          # bytes of main data written into the pipe of main-data areas so far
        pipe_area = 0         # start of the current frame's area in the pipe
        frames = []
        acc = 0.0
        for f in range(nfr):
            # padding keeps the average bit rate (44.1 kHz family)
            exact = (72 if self.lsf else 144) * self.bitrate * 1000 / self.sr
            acc += exact - int(exact)
            padding = 0
            if acc >= 1.0:
                padding, acc = 1, acc - 1.0
            fb = self.frame_bytes(padding)
            cap = fb - 4 - (2 if self.crc else 0) - self.side_bytes
            start = max(pipe_cur, pipe_area - (255 if self.lsf else 511))
            avail_bits = (pipe_area + cap - start) * 8
            budget = int(min(avail_bits, cap * 8 * self.bit_share(f)))
            side, main = self._encode_frame(f, spec, cfgs, ms, ist, budget, ngr_f)
            mb = main.to_bytes()
            assert start + len(mb) <= pipe_area + cap, (f, len(mb), cap, pipe_area - start)
            frames.append(dict(fb=fb, cap=cap, padding=padding, mdb=pipe_area - start, side=side, main=mb, start=start))
            self.log.append(dict(frame=f, main_data_begin=pipe_area - start, main_bytes=len(mb), cap=cap))
            pipe_cur = start + len(mb)
            pipe_area += cap
        # lay the main data into the pipe, then cut the pipe into the frames' areas
        pipe = bytearray(pipe_area)
        for fr in frames:
            pipe[fr["start"]:fr["start"] + len(fr["main"])] = fr["main"]
        pos = 0
        for fr in frames:
            hb = header_bits(self.sr, self.bitrate, fr["padding"], self.mode, self.mode_ext if self.mode == 1 else 0, self.crc)
            sb = self._side_bits(fr["mdb"], fr["side"])
            assert len(sb) == self.side_bytes * 8, (len(sb), self.side_bytes)
            out += hb.to_bytes()
            if self.crc:
                c = crc16(hb.b[16:] + sb.b)
                out += bytes([c >> 8, c & 255])
            out += sb.to_bytes()
            out += pipe[pos:pos + fr["cap"]]
            pos += fr["cap"]
        return bytes(out)

    # -- one frame: choose global_gain per granule / channel so that the frame fits `budget` bits
    def _encode_frame(self, f, spec, cfgs, ms, ist, budget, ngr_f):
        rng = self.rng
        nch = self.nch
        units = [(g, c) for g in range(f * ngr_f, (f + 1) * ngr_f) for c in range(nch)]
        per_unit = budget // len(units)
        side, main = {}, Bits()
        prev_sf = {}
        for (g, c) in units:
            cfg = cfgs[g]
            xs = spec[g].copy()
            if ms:
                m, s = (xs[0] + xs[1]) / math.sqrt(2), (xs[0] - xs[1]) / math.sqrt(2)
                xs = np.stack([m, s])
            is_right = bool(ist and c == 1)
            # scale factors: random values in fields of random widths (MPEG-1: scalefac_compress; LSF: a partition)
            if not self.lsf:
                compress = 15 if is_right else int(rng.integers(0, 16))      # intensity positions up to 6 need 3-bit fields everywhere
                s1, s2 = SLEN[compress]
                wl = [s1] * 11 + [s2] * 10
                wsb = [s1] * 6 + [s2] * 6
                sf_l, sf_s = random_scalefactors(rng, cfg, wl, wsb)
                scfsi = [0, 0, 0, 0]
                if self.scfsi and (g % 2) == 1 and cfg.block_type != 2 and cfgs[g - 1].block_type != 2 and (g - 1, c) in prev_sf:
                    scfsi = [int(rng.integers(0, 2)) for _ in range(4)]
                    for grp, (a, z) in enumerate(((0, 6), (6, 11), (11, 16), (16, 21))):
                        if scfsi[grp]:
                            sf_l[a:z] = prev_sf[(g - 1, c)][a:z]
                if cfg.block_type == 2 and cfg.mixed:
                    pass
                lsf_fields = None
            else:
                blocknumber = int(rng.integers(0, 3)) if not is_right else int(rng.integers(3, 6))
                if blocknumber == 0: slen = [int(rng.integers(0, 5)), int(rng.integers(0, 5)), int(rng.integers(0, 4)), int(rng.integers(0, 4))]
                elif blocknumber == 1: slen = [int(rng.integers(0, 5)), int(rng.integers(0, 5)), int(rng.integers(0, 4)), 0]
                elif blocknumber == 2: slen = [int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0, 0]
                elif blocknumber == 3: slen = [int(rng.integers(0, 5)), int(rng.integers(0, 6)), int(rng.integers(0, 6)), 0]
                elif blocknumber == 4: slen = [int(rng.integers(0, 4)), int(rng.integers(0, 4)), int(rng.integers(0, 4)), 0]
                else: slen = [int(rng.integers(0, 4)), int(rng.integers(0, 3)), 0, 0]
                nr = lsf_partition(cfg, is_right, slen, blocknumber)
                widths = [slen[i] for i in range(4) for _ in range(nr[i])]
                vals = [int(rng.integers(0, 1 << w)) if w else 0 for w in widths]
                sf_l, sf_s = [0] * 23, [[0] * 13 for _ in range(3)]
                if cfg.block_type != 2:
                    for k in range(21): sf_l[k] = vals[k]
                elif not cfg.mixed:
                    q = 0
                    for k in range(12):
                        for w in range(3): sf_s[w][k] = vals[q]; q += 1
                else:
                    q = 0
                    for k in range(6): sf_l[k] = vals[q]; q += 1
                    for k in range(3, 12):
                        for w in range(3): sf_s[w][k] = vals[q]; q += 1
                if blocknumber == 0: sfc = ((slen[0] * 5 + slen[1]) << 4) + (slen[2] << 2) + slen[3]
                elif blocknumber == 1: sfc = 400 + ((slen[0] * 5 + slen[1]) << 2) + slen[2]
                elif blocknumber == 2: sfc = 500 + slen[0] * 3 + slen[1]
                elif blocknumber == 3: sfc = (slen[0] * 36 + slen[1] * 6 + slen[2]) * 2 + self.intensity_scale
                elif blocknumber == 4: sfc = (180 + (slen[0] << 4) + (slen[1] << 2) + slen[2]) * 2 + self.intensity_scale
                else: sfc = (244 + slen[0] * 3 + slen[1]) * 2 + self.intensity_scale
                cfg = GranuleCfg(cfg.block_type, cfg.mixed, cfg.scalefac_scale, 1 if blocknumber == 2 else 0, cfg.subblock_gain)
                lsf_fields = (slen, blocknumber, sfc, widths)
            x = xs[c].copy()
            # intensity stereo: above the bound the left channel carries L + R (scaled), the right channel zeros and positions
            if ist:
                bound_l, bound_s, pos = self.intensity
                for start, wd, kind, sfb, win in band_layout(self.sr, self.lsf, cfg):
                    if sfb >= (bound_l if kind == "l" else bound_s):
                        if c == 1:
                            x[start:start + wd] = 0.0
                            p = pos(kind, sfb, win)
                            if kind == "l" and sfb < 21: sf_l[sfb] = p
                            if kind == "s" and sfb < 12: sf_s[win][sfb] = p
                        else:
                            x[start:start + wd] = self._intensity_mono(spec[g], start, wd, pos(kind, min(sfb, 20 if kind == "l" else 11), win))
                if c == 1 and self.lsf:
                    # the chosen positions must fit the partition's widths and must not be the "illegal" maximum: the caller's `pos`
                    # returns small values; give every field 3 bits
                    # partition set 3 (7 + 7 + 7 long bands / 12 + 12 + 12 short): every band has a field -- a band whose field is 0 bits
                    # wide has position 0 = its own "illegal" value and is NOT intensity-coded (13818-3: is_pos == 2^slen - 1)
                    slen, blocknumber = [3, 3, 3, 0], 3
                    sfc = (3 * 36 + 3 * 6 + 3) * 2 + self.intensity_scale
                    nr = lsf_partition(cfg, True, slen, blocknumber)
                    lsf_fields = (slen, blocknumber, sfc, [slen[i] for i in range(4) for _ in range(nr[i])])
                    for k in range(21): sf_l[k] = min(sf_l[k], 6)
                    for w in range(3):
                        for k in range(12): sf_s[w][k] = min(sf_s[w][k], 6)
                    cfg = GranuleCfg(cfg.block_type, cfg.mixed, cfg.scalefac_scale, 0, cfg.subblock_gain)
            if not self.lsf:
                sfb_bits = scalefac_bits_mpeg1(cfg, sf_l, sf_s, scfsi, g % 2, compress)
            else:
                sfb_bits = scalefac_bits_lsf(cfg, sf_l, sf_s, lsf_fields[0], lsf_fields[1])
            # global gain by bisection: the finest step whose Huffman data fits
            lo, hi = 0, 255
            best = None
            state = rng.bit_generator.state
            while lo <= hi:
                gg = (lo + hi) // 2
                ix = quantize(x, self.sr, self.lsf, cfg, sf_l, sf_s, gg)
                ok = int(np.abs(ix).max()) <= 8191 + 14
                if ok:
                    rng.bit_generator.state = state
                    hb, fields = code_spectrum(rng, ix, self.sr, cfg, self.v25, self.lsf, self.tables)
                    ok = len(hb) + len(sfb_bits) <= min(per_unit, 4095)
                if ok:
                    best = (gg, ix, hb, fields)
                    hi = gg - 1
                else:
                    lo = gg + 1
            assert best is not None, "budget too small even for silence"
            gg, ix, hb, fields = best
            fields.update(part2_3_length=len(hb) + len(sfb_bits), global_gain=gg, cfg=cfg, scfsi=scfsi if not self.lsf else None,
                          scalefac_compress=compress if not self.lsf else lsf_fields[2])
            side[(g, c)] = fields
            main.b += sfb_bits.b + hb.b
            prev_sf[(g, c)] = list(sf_l)
            self.log.append(dict(g=g, c=c, gg=gg, bits=fields["part2_3_length"], tables=fields["table_select"], bt=cfg.block_type))
            self.units[(g, c)] = dict(ix=ix.copy(), cfg=cfg, sf_l=list(sf_l), sf_s=[list(r) for r in sf_s], gg=gg)
            # what the decoder will reconstruct (kept for tests that want the exact spectrum)
        return side, main

    intensity_scale = 0

    def _intensity_mono(self, spec_gc, start, wd, p):
        """the one spectrum an intensity band transmits: the decoder outputs L = x kl, R = x kr"""
        kl, kr = intensity_factors(p, self.lsf, self.intensity_scale)
        L, R = spec_gc[0][start:start + wd], spec_gc[1][start:start + wd]
        return (L * kl + R * kr) / (kl * kl + kr * kr)          # least squares

    def _side_bits(self, mdb, side):
        b = Bits()
        nch = self.nch
        keys = sorted(side)
        if not self.lsf:
            b.put(mdb, 9); b.put(0, 5 if nch == 1 else 3)
            for c in range(nch):
                sc = side[(keys[-1][0], c)]["scfsi"]
                for v in sc: b.put(v, 1)
        else:
            b.put(mdb, 8); b.put(0, 1 if nch == 1 else 2)
        for (g, c) in keys:
            s = side[(g, c)]
            cfg = s["cfg"]
            b.put(s["part2_3_length"], 12); b.put(s["big_values"], 9); b.put(s["global_gain"], 8)
            b.put(s["scalefac_compress"], 9 if self.lsf else 4)
            ws = cfg.block_type != 0
            b.put(1 if ws else 0, 1)
            if ws:
                b.put(cfg.block_type, 2); b.put(1 if cfg.mixed else 0, 1)
                b.put(s["table_select"][0], 5); b.put(s["table_select"][1], 5)
                for w in range(3): b.put(cfg.subblock_gain[w], 3)
            else:
                for r in range(3): b.put(s["table_select"][r], 5)
                b.put(s["region0_count"], 4); b.put(s["region1_count"], 3)
            if not self.lsf:
                b.put(cfg.preflag, 1)
            b.put(cfg.scalefac_scale, 1); b.put(s["count1table"], 1)
        return b

    def _info_frame(self, nfr):
        delay, padding = self.info_frame
        fb = self.frame_bytes(0)
        hb = header_bits(self.sr, self.bitrate, 0, self.mode, 0, False)
        body = bytearray(fb - 4)
        x = self.side_bytes
        body[x:x + 4] = b"Info"
        body[x + 4:x + 8] = (1).to_bytes(4, "big")                  # flags: frame count
        body[x + 8:x + 12] = nfr.to_bytes(4, "big")
        p = x + 12
        body[p:p + 9] = b"LAME3.99r"
        v = (delay << 12) | padding
        body[p + 21:p + 24] = v.to_bytes(3, "big")
        return hb.to_bytes() + bytes(body)


def intensity_factors(p, lsf, scale):
    if not lsf:
        if p == 6:
            return 1.0, 0.0
        r = math.tan(p * PI / 12)
        return r / (1 + r), 1 / (1 + r)
    io = 2.0 ** (-0.5 if scale else -0.25)
    if p == 0:
        return 1.0, 1.0
    return (io ** ((p + 1) // 2), 1.0) if p & 1 else (1.0, io ** (p // 2))


def expected_spectra(enc, n_granules):
    """What the decoder must hold after requantisation and stereo processing, from what the encoder transmitted: per granule
    [channels][576] in the decoder's line order (short windows interleaved), float64."""
    out = []
    ms = enc.nch == 2 and enc.mode == 1 and (enc.mode_ext & 2)
    ist = enc.nch == 2 and enc.mode == 1 and (enc.mode_ext & 1) and enc.intensity is not None
    for g in range(n_granules):
        xs = []
        for c in range(enc.nch):
            u = enc.units[(g, c)]
            xs.append(dequantize(u["ix"], enc.sr, enc.lsf, u["cfg"], u["sf_l"], u["sf_s"], u["gg"]))
        xs = np.stack(xs)
        cfg = enc.units[(g, 0)]["cfg"]
        if enc.nch == 2 and (ms or ist):
            cfg1 = enc.units[(g, 1)]["cfg"]
            u1 = enc.units[(g, 1)]
            res = xs.copy()
            for start, wd, kind, sfb, win in band_layout(enc.sr, enc.lsf, cfg1):
                sl = slice(start, start + wd)
                inten = False
                if ist:
                    bound_l, bound_s, pos = enc.intensity
                    if sfb >= (bound_l if kind == "l" else bound_s):
                        p = u1["sf_l"][min(sfb, 20)] if kind == "l" else u1["sf_s"][win][min(sfb, 11)]
                        kl, kr = intensity_factors(p, enc.lsf, enc.intensity_scale)
                        res[0, sl], res[1, sl] = xs[0, sl] * kl, xs[0, sl] * kr
                        inten = True
                if not inten and ms:
                    res[0, sl], res[1, sl] = (xs[0, sl] + xs[1, sl]) / math.sqrt(2), (xs[0, sl] - xs[1, sl]) / math.sqrt(2)
            xs = res
        out.append(np.stack([from_stream_order(xs[c], enc.sr, cfg.block_type, cfg.mixed) for c in range(enc.nch)]))
    return out
