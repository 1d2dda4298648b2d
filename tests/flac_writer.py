"""TEST INFRASTRUCTURE: a small FLAC *encoder* (RFC 9639) used to exercise librvb's decoder (csrc/audio.cpp) on every
feature of the format -- the image has no FLAC tool or library.  It is written the other way round from the decoder
(numpy residuals, bit strings, bitwise CRCs, hashlib's MD5), so a shared misreading of a field would have to be made
twice; the decoder is additionally pinned on the complete example file of RFC 9639 appendix D.1
(tests/test_audio_decode.py), whose CRC-8, CRC-16 and MD5 it verifies.

    encode(samples[C][N] ints, bps, rate, block=..., stereo=..., predictor=..., ...) -> bytes
"""
import hashlib

import numpy as np

FIXED = [[], [1], [2, -1], [3, -3, 1], [4, -6, 4, -1]]
RATE_CODES = {88200: 1, 176400: 2, 192000: 3, 8000: 4, 16000: 5, 22050: 6, 24000: 7, 32000: 8, 44100: 9, 48000: 10, 96000: 11}
BPS_CODES = {8: 1, 12: 2, 16: 4, 20: 5, 24: 6, 32: 7}


def bits(value, width):
    """two's complement bit string of `width` bits"""
    if width == 0:
        return ""
    return format(int(value) & ((1 << width) - 1), "0%db" % width)


def crc(data, poly, width):
    reg, top, mask = 0, 1 << (width - 1), (1 << width) - 1
    for byte in data:
        reg ^= byte << (width - 8)
        for _ in range(8):
            reg = ((reg << 1) ^ poly) & mask if reg & top else (reg << 1) & mask
    return reg


def coded_number(n):
    if n < 0x80:
        return bytes([n])
    out, lead, limit = [], 0xC0, 0x20
    while True:
        out.append(0x80 | (n & 0x3F))
        n >>= 6
        if n < limit:
            return bytes([lead | n] + out[::-1])
        lead, limit = (lead >> 1) | 0x80, limit >> 1


def rice_bits(res, k):
    u = np.where(res >= 0, 2 * res, -2 * res - 1).astype(object)
    return "".join("0" * int(v >> k) + "1" + bits(v & ((1 << k) - 1), k) for v in u)


def rice_cost(res, k):
    u = np.where(res >= 0, 2 * res, -2 * res - 1)
    return int((u >> k).sum()) + (k + 1) * len(res)


def residual_bits(res, order, block, partition_order=None, escape_first=False, force_rice2=False):
    """res: the block - order residuals.  Chooses the partition order and the Rice parameters by cost."""
    best = None
    orders = range(0, 9) if partition_order is None else [partition_order]
    for po in orders:
        parts = 1 << po
        if block % parts or (block >> po) < order or (po > 0 and (block >> po) == 0):
            continue
        per, pos, chunks, wide = block >> po, 0, [], force_rice2
        for part in range(parts):
            cnt = per - (order if part == 0 else 0)
            r = res[pos:pos + cnt]
            pos += cnt
            if escape_first and part == 0:
                chunks.append(("esc", r))
                continue
            k = min(range(0, 31), key=lambda kk: rice_cost(r, kk)) if len(r) else 0
            wide |= k >= 15
            chunks.append((k, r))
        out = [bits(1 if wide else 0, 2), bits(po, 4)]
        pbits = 5 if wide else 4
        for k, r in chunks:
            if k == "esc":
                width = max([int(v).bit_length() + 1 for v in r] + [1]) if len(r) and np.any(r != 0) else 0
                out.append(bits((1 << pbits) - 1, pbits) + bits(width, 5) + "".join(bits(v, width) for v in r))
            else:
                out.append(bits(k, pbits) + rice_bits(r, k))
        s = "".join(out)
        if best is None or len(s) < len(best):
            best = s
    assert best is not None, "no valid partition order"
    return best


def lpc_coefficients(x, order, precision):
    """least-squares predictor of `order` taps, quantised to `precision` bits with the largest usable shift"""
    xf = x.astype(np.float64)
    rows = np.stack([xf[order - 1 - j:len(xf) - 1 - j] for j in range(order)], axis=1)
    coef, *_ = np.linalg.lstsq(rows, xf[order:], rcond=None)
    peak = max(float(np.abs(coef).max()), 1e-9)
    shift = int(np.clip(precision - 1 - int(np.floor(np.log2(peak))) - 1, 0, 15))
    q = np.clip(np.round(coef * (1 << shift)), -(1 << (precision - 1)), (1 << (precision - 1)) - 1).astype(np.int64)
    return q, shift


def predict(x, coefs, shift):
    order = len(coefs)
    acc = np.zeros(len(x) - order, dtype=object)
    xo = x.astype(object)
    for j, c in enumerate(coefs):
        acc = acc + int(c) * xo[order - 1 - j:len(x) - 1 - j]
    return np.array([int(v) >> shift for v in acc], dtype=object) if len(acc) else acc


def subframe_bits(x, bps, predictor="auto", wasted="auto", **rice):
    """x: integer samples of one channel of one block (numpy object / int64 array)"""
    x = np.asarray(x, dtype=object)
    n = len(x)
    w = 0
    if wasted == "auto" and np.any(x != 0):
        while all(int(v) & ((1 << (w + 1)) - 1) == 0 for v in x):
            w += 1
    elif isinstance(wasted, int):
        w = wasted
    if w:
        x = np.array([int(v) >> w for v in x], dtype=object)
        bps -= w
    wbits = "1" + "0" * (w - 1) + "1" if w else "0"
    if predictor == "auto":
        if np.all(x == x[0]):
            predictor = "constant"
        else:
            costs = []
            for k in range(0, min(4, n - 1) + 1):
                r = x[k:] - (predict(x, FIXED[k], 0) if k else 0)
                costs.append((int(np.abs(r.astype(np.float64)).sum()), k))
            predictor = ("fixed", min(costs)[1])
    if predictor == "constant":
        assert np.all(x == x[0])
        return "0" + bits(0, 6) + wbits + bits(x[0], bps)
    if predictor == "verbatim":
        return "0" + bits(1, 6) + wbits + "".join(bits(v, bps) for v in x)
    kind, order = predictor[0], predictor[1]
    warm = "".join(bits(v, bps) for v in x[:order])
    if kind == "fixed":
        res = x[order:] - (predict(x, FIXED[order], 0) if order else 0)
        head = "0" + bits(8 + order, 6) + wbits + warm
    else:
        precision = predictor[2] if len(predictor) > 2 else 12
        q, shift = lpc_coefficients(np.asarray(x, dtype=np.float64), order, precision)
        res = x[order:] - predict(x, q, shift)
        head = "0" + bits(31 + order, 6) + wbits + warm + bits(precision - 1, 4) + bits(shift, 5) + "".join(bits(c, precision) for c in q)
    res = np.array([int(v) for v in res], dtype=np.int64)
    return head + residual_bits(res, order, n, **rice)


def frame_bytes(number, chans, bps, rate, stereo="independent", variable=False, rate_code=None, explicit_bps=True,
                predictor="auto", wasted="auto", **rice):
    """chans: list of per-channel integer arrays of one block.  `number`: frame number (fixed block size) or first sample."""
    n = len(chans[0])
    if n == 192:
        bs_code, bs_extra = 1, b""
    elif n in (576, 1152, 2304, 4608):
        bs_code, bs_extra = 2 + (576, 1152, 2304, 4608).index(n), b""
    elif n in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768):
        bs_code, bs_extra = 8 + (256, 512, 1024, 2048, 4096, 8192, 16384, 32768).index(n), b""
    elif n <= 256:
        bs_code, bs_extra = 6, bytes([n - 1])
    else:
        bs_code, bs_extra = 7, (n - 1).to_bytes(2, "big")
    if rate_code is None:
        rate_code = RATE_CODES.get(rate, 0)
    if rate_code == 12:
        sr_extra = bytes([rate // 1000])
    elif rate_code == 13:
        sr_extra = rate.to_bytes(2, "big")
    elif rate_code == 14:
        sr_extra = (rate // 10).to_bytes(2, "big")
    else:
        sr_extra = b""
    a = [np.asarray(c, dtype=object) for c in chans]
    if stereo == "independent":
        ch_code, subs = len(a) - 1, [(c, bps) for c in a]
    elif stereo == "left_side":
        ch_code, subs = 8, [(a[0], bps), (a[0] - a[1], bps + 1)]
    elif stereo == "side_right":
        ch_code, subs = 9, [(a[0] - a[1], bps + 1), (a[1], bps)]
    else:
        mid = np.array([(int(l) + int(r)) >> 1 for l, r in zip(a[0], a[1])], dtype=object)
        ch_code, subs = 10, [(mid, bps), (a[0] - a[1], bps + 1)]
    ss_code = BPS_CODES[bps] if explicit_bps else 0
    head = bytes([0xFF, 0xF8 | (1 if variable else 0), (bs_code << 4) | rate_code, (ch_code << 4) | (ss_code << 1)])
    head += coded_number(number) + bs_extra + sr_extra
    head += bytes([crc(head, 0x07, 8)])
    body = "".join(subframe_bits(x, b, predictor=predictor, wasted=wasted, **rice) for x, b in subs)
    body += "0" * (-len(body) % 8)
    frame = head + int(body, 2).to_bytes(len(body) // 8, "big")
    return frame + crc(frame, 0x8005, 16).to_bytes(2, "big")


def pcm_md5(samples, bps):
    width = (bps + 7) // 8
    inter = np.asarray(samples, dtype=object).T.reshape(-1)
    return hashlib.md5(b"".join(int(v).to_bytes(width, "little", signed=True) for v in inter)).digest()


def encode(samples, bps, rate, block=4096, stereo="independent", predictor="auto", md5=True, total_known=True, id3=False,
           padding=True, variable_blocks=None, **kw):
    """samples: [channels][frames] integers within the signed `bps`-bit range.  variable_blocks: a list of block sizes
    (variable block-size stream, frames numbered by their first sample)."""
    s = np.asarray(samples, dtype=object)
    nch, total = s.shape
    sizes = variable_blocks if variable_blocks else [block] * (total // block) + ([total % block] if total % block else [])
    assert sum(sizes) == total
    frames, at = [], 0
    for i, n in enumerate(sizes):
        frames.append(frame_bytes(at if variable_blocks else i, [s[c, at:at + n] for c in range(nch)], bps, rate, stereo=stereo,
                                  variable=bool(variable_blocks), predictor=predictor, **kw))
        at += n
    min_b = max_b = block
    if variable_blocks:
        min_b, max_b = max(16, min(sizes)), max(max(sizes), 16)
    packed = (rate << 44) | ((nch - 1) << 41) | ((bps - 1) << 36) | (total if total_known else 0)
    info = min_b.to_bytes(2, "big") + max_b.to_bytes(2, "big") + min(map(len, frames)).to_bytes(3, "big") + \
        max(map(len, frames)).to_bytes(3, "big") + packed.to_bytes(8, "big") + (pcm_md5(s, bps) if md5 else bytes(16))
    out = b""
    if id3:
        out += b"ID3\x04\x00\x00" + bytes([0, 0, 0, 21]) + b"TIT2" + bytes([0, 0, 0, 7, 0, 0]) + b"\x03reverb" + bytes(4)
    out += b"fLaC" + bytes([0 if padding else 0x80]) + (34).to_bytes(3, "big") + info
    if padding:
        vendor = b"reverb_amd test writer"
        comment = len(vendor).to_bytes(4, "little") + vendor + (0).to_bytes(4, "little")
        out += bytes([4]) + len(comment).to_bytes(3, "big") + comment
        out += bytes([0x81]) + (16).to_bytes(3, "big") + bytes(16)
    return out + b"".join(frames)
