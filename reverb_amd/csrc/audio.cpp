// Host-side audio file decoding for librvb: what `torchaudio.load(audio_file, normalize=False)` hands to
// ReverbASR.compute_feats (asr/wenet/cli/reverb.py:128) for the lossless containers -- RIFF/WAVE (and RF64), AIFF / AIFF-C, FLAC.
//
// The reference keeps the decoder's NATIVE sample format (normalize=False) and then calls `.to(torch.float)`
// (reverb.py:130), so the numbers the fbank sees depend on the container:
//   WAVE 16-bit PCM, A-law, mu-law   int16                      -> sample value
//   WAVE 8-bit PCM                   uint8                      -> 0 .. 255 (offset binary, as stored)
//   WAVE 24-bit / 32-bit PCM         int32 (24-bit left-justified: value << 8)
//   WAVE IEEE float 32 / 64          float32 / float64          -> as stored (64-bit rounded to float)
//   AIFF / AIFF-C                    as the WAVE form of the same samples (big-endian or `sowt`); 8-bit AIFF is signed
//                                    and arrives as uint8 with the offset added (value + 128)
//   FLAC <= 16 bits per sample       int16, left-justified      -> value << (16 - bps)
//   FLAC  > 16 bits per sample       int32, left-justified      -> value << (32 - bps)
//   MPEG audio Layer III (.mp3)      float32                    -> the decoder's output, full scale = 1.0 (csrc/mp3.cpp; round 6)
// (the FFmpeg-backed loader of the torchaudio the reference pins: s16 / s32 / u8 / flt / dbl planar or packed
// sample formats, nothing rescaled).  MP3 / Vorbis are lossy float decoders and are not built: rvb_audio_probe
// names the container in its error.
//
// FLAC is decoded from the format definition (RFC 9639): STREAMINFO, frame header with its CRC-8, the four subframe
// kinds (constant, verbatim, fixed predictor of order 0..4, LPC of order 1..32), partitioned Rice residuals with both
// parameter widths and the escape code, wasted bits, the three stereo decorrelations, the frame CRC-16, and the MD5
// of the decoded PCM that STREAMINFO carries -- every real FLAC file is its own known-answer test, and the decoder
// checks it.  A frame's length is known only once it is decoded, but frames do not depend on each other: long streams are
// cut into runs of frames for a few host threads (flac_decode).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rvb.h"

#include <type_traits>
#include "mp3.h"

namespace rvb {
void set_error(const std::string& msg);  // engine.hip: thread-local last error
}

namespace {

enum { OK = 0, E_ARG = -1, E_UNSUPPORTED = -5, E_DATA = -6 };

struct Fail {
  int code;
  std::string msg;
};
[[noreturn]] void fail(int code, const std::string& m) { throw Fail{code, m}; }

// ---------------------------------------------------------------------------------------------- MD5 (RFC 1321)
struct Md5 {
  uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
  uint8_t buf[64];
  uint64_t total = 0;
  size_t fill = 0;
  static uint32_t rol(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }
  void block(const uint8_t* p) {
    static const int S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};
    // floor(|sin(i + 1)| * 2^32), RFC 1321 section 3.4; a constant table (no lazily filled static: two decoder threads may hash at once)
    static constexpr uint32_t K[64] = {
        0xd76aa478u, 0xe8c7b756u, 0x242070dbu, 0xc1bdceeeu, 0xf57c0fafu, 0x4787c62au, 0xa8304613u, 0xfd469501u, 0x698098d8u, 0x8b44f7afu,
        0xffff5bb1u, 0x895cd7beu, 0x6b901122u, 0xfd987193u, 0xa679438eu, 0x49b40821u, 0xf61e2562u, 0xc040b340u, 0x265e5a51u, 0xe9b6c7aau,
        0xd62f105du, 0x02441453u, 0xd8a1e681u, 0xe7d3fbc8u, 0x21e1cde6u, 0xc33707d6u, 0xf4d50d87u, 0x455a14edu, 0xa9e3e905u, 0xfcefa3f8u,
        0x676f02d9u, 0x8d2a4c8au, 0xfffa3942u, 0x8771f681u, 0x6d9d6122u, 0xfde5380cu, 0xa4beea44u, 0x4bdecfa9u, 0xf6bb4b60u, 0xbebfbc70u,
        0x289b7ec6u, 0xeaa127fau, 0xd4ef3085u, 0x04881d05u, 0xd9d4d039u, 0xe6db99e5u, 0x1fa27cf8u, 0xc4ac5665u, 0xf4292244u, 0x432aff97u,
        0xab9423a7u, 0xfc93a039u, 0x655b59c3u, 0x8f0ccc92u, 0xffeff47du, 0x85845dd1u, 0x6fa87e4fu, 0xfe2ce6e0u, 0xa3014314u, 0x4e0811a1u,
        0xf7537e82u, 0xbd3af235u, 0x2ad7d2bbu, 0xeb86d391u};
    uint32_t w[16];
    for (int i = 0; i < 16; ++i) w[i] = (uint32_t)p[4 * i] | (uint32_t)p[4 * i + 1] << 8 | (uint32_t)p[4 * i + 2] << 16 | (uint32_t)p[4 * i + 3] << 24;
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define RVB_MD5_STEP(F, G)                               \
  {                                                      \
    const uint32_t f = (F), t = d;                       \
    d = c;                                               \
    c = b;                                               \
    b = b + rol(a + f + K[i] + w[(G)&15], S[i]);         \
    a = t;                                               \
  }
    for (int i = 0; i < 16; ++i) RVB_MD5_STEP((b & c) | (~b & d), i)
    for (int i = 16; i < 32; ++i) RVB_MD5_STEP((d & b) | (~d & c), 5 * i + 1)
    for (int i = 32; i < 48; ++i) RVB_MD5_STEP(b ^ c ^ d, 3 * i + 5)
    for (int i = 48; i < 64; ++i) RVB_MD5_STEP(c ^ (b | ~d), 7 * i)
#undef RVB_MD5_STEP
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
  void update(const uint8_t* p, size_t n) {
    total += n;
    if (fill) {
      const size_t take = std::min(n, 64 - fill);
      std::memcpy(buf + fill, p, take);
      fill += take; p += take; n -= take;
      if (fill < 64) return;
      block(buf);
      fill = 0;
    }
    for (; n >= 64; p += 64, n -= 64) block(p);
    if (n) { std::memcpy(buf, p, n); fill = n; }
  }
  void finish(uint8_t out[16]) {
    const uint64_t bits = total * 8;
    const uint8_t one = 0x80, zero = 0;
    update(&one, 1);
    while (fill != 56) update(&zero, 1);
    uint8_t len[8];
    for (int i = 0; i < 8; ++i) len[i] = (uint8_t)(bits >> (8 * i));
    update(len, 8);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) out[4 * i + j] = (uint8_t)(h[i] >> (8 * j));
  }
};

// ---------------------------------------------------------------------------------------------- CRCs
struct CrcTables {
  uint8_t c8[256];
  uint16_t c16[256];
  CrcTables() {
    for (int i = 0; i < 256; ++i) {
      uint8_t a = (uint8_t)i;
      uint16_t b = (uint16_t)(i << 8);
      for (int k = 0; k < 8; ++k) {
        a = (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x07 : 0));            // x^8 + x^2 + x + 1
        b = (uint16_t)((b << 1) ^ ((b & 0x8000) ? 0x8005 : 0));       // x^16 + x^15 + x^2 + 1
      }
      c8[i] = a;
      c16[i] = b;
    }
  }
};
const CrcTables& crc_tables() {
  static const CrcTables t;
  return t;
}
uint8_t crc8(const uint8_t* p, size_t n) {
  const CrcTables& t = crc_tables();
  uint8_t c = 0;
  for (size_t i = 0; i < n; ++i) c = t.c8[c ^ p[i]];
  return c;
}
uint16_t crc16(const uint8_t* p, size_t n) {
  const CrcTables& t = crc_tables();
  uint16_t c = 0;
  for (size_t i = 0; i < n; ++i) c = (uint16_t)((c << 8) ^ t.c16[(c >> 8) ^ p[i]]);
  return c;
}

// ---------------------------------------------------------------------------------------------- bit reader (MSB first)
struct Bits {
  const uint8_t* p;
  size_t n, pos = 0;     // pos: next byte to load
  uint64_t acc = 0;      // the low `have` bits are pending
  int have = 0;
  Bits(const uint8_t* d, size_t len) : p(d), n(len) {}
  void refill() {
    while (have <= 56 && pos < n) { acc = (acc << 8) | p[pos++]; have += 8; }
  }
  uint32_t u(int bits) {                 // 0 .. 32 bits
    if (bits == 0) return 0;
    if (have < bits) {
      refill();
      if (have < bits) fail(E_DATA, "FLAC: the stream ends inside a frame");
    }
    have -= bits;
    return (uint32_t)((acc >> have) & ((bits == 32) ? 0xffffffffull : ((1ull << bits) - 1)));
  }
  int64_t s(int bits) {                  // signed, 1 .. 33 bits
    if (bits <= 32) {
      const uint32_t v = u(bits);
      const uint64_t sign = 1ull << (bits - 1);
      return (int64_t)((v ^ sign)) - (int64_t)sign;
    }
    const uint64_t hi = u(bits - 32), lo = u(32);
    const uint64_t v = (hi << 32) | lo, sign = 1ull << (bits - 1);
    return (int64_t)(v ^ sign) - (int64_t)sign;
  }
  uint32_t unary() {                     // number of 0 bits before the next 1
    uint32_t q = 0;
    for (;;) {
      if (have == 0) {
        refill();
        if (have == 0) fail(E_DATA, "FLAC: the stream ends inside a residual");
      }
      const uint64_t window = acc & ((have == 64) ? ~0ull : ((1ull << have) - 1));
      if (window == 0) { q += (uint32_t)have; have = 0; continue; }
      const int lead = __builtin_clzll(window) - (64 - have);
      q += (uint32_t)lead;
      have -= lead + 1;
      return q;
    }
  }
  void align() { have -= have & 7; }
  size_t byte_pos() const { return pos - (size_t)(have >> 3); }   // valid when aligned
};

// ---------------------------------------------------------------------------------------------- FLAC
struct StreamInfo {
  int min_block = 0, max_block = 0, rate = 0, channels = 0, bps = 0;
  int64_t total = 0;
  uint8_t md5[16] = {0};
  bool has_md5 = false;
};

struct FlacStream {
  StreamInfo si;
  size_t first_frame = 0;
};

uint32_t be(const uint8_t* p, int n) {
  uint32_t v = 0;
  for (int i = 0; i < n; ++i) v = (v << 8) | p[i];
  return v;
}

size_t skip_id3(const uint8_t* d, size_t n) {
  size_t off = 0;
  while (n - off >= 10 && d[off] == 'I' && d[off + 1] == 'D' && d[off + 2] == '3') {
    const size_t sz = ((size_t)(d[off + 6] & 0x7f) << 21) | ((size_t)(d[off + 7] & 0x7f) << 14) | ((size_t)(d[off + 8] & 0x7f) << 7) | (d[off + 9] & 0x7f);
    off += 10 + sz + ((d[off + 5] & 0x10) ? 10 : 0);
    if (off > n) fail(E_DATA, "ID3v2 tag runs past the end of the file");
  }
  return off;
}

FlacStream flac_open(const uint8_t* d, size_t n) {
  size_t off = skip_id3(d, n);
  if (n - off < 4 || std::memcmp(d + off, "fLaC", 4) != 0) fail(E_DATA, "FLAC: missing fLaC marker");
  off += 4;
  FlacStream fs;
  bool seen = false;
  for (;;) {
    if (n - off < 4) fail(E_DATA, "FLAC: truncated metadata");
    const bool last = d[off] & 0x80;
    const int type = d[off] & 0x7f;
    const size_t len = be(d + off + 1, 3);
    off += 4;
    if (len > n - off) fail(E_DATA, "FLAC: metadata block runs past the end of the file");
    if (type == 127) fail(E_DATA, "FLAC: invalid metadata block type 127");
    if (!seen) {
      if (type != 0 || len != 34) fail(E_DATA, "FLAC: the first metadata block must be a 34-byte STREAMINFO");
      const uint8_t* p = d + off;
      StreamInfo& s = fs.si;
      s.min_block = (int)be(p, 2);
      s.max_block = (int)be(p + 2, 2);
      const uint64_t x = ((uint64_t)be(p + 10, 4) << 32) | be(p + 14, 4);
      s.rate = (int)(x >> 44);
      s.channels = (int)((x >> 41) & 7) + 1;
      s.bps = (int)((x >> 36) & 31) + 1;
      s.total = (int64_t)(x & 0xfffffffffull);
      std::memcpy(s.md5, p + 18, 16);
      for (int i = 0; i < 16; ++i) s.has_md5 |= s.md5[i] != 0;
      if (s.bps < 4) fail(E_DATA, "FLAC: STREAMINFO declares fewer than 4 bits per sample");
      if (s.rate == 0) fail(E_DATA, "FLAC: STREAMINFO declares sample rate 0 (not an audio stream)");
      if (s.min_block < 16 || s.max_block < s.min_block) fail(E_DATA, "FLAC: invalid block size bounds in STREAMINFO");
      seen = true;
    }
    off += len;
    if (last) break;
  }
  fs.first_frame = off;
  return fs;
}

struct Frame {
  int block = 0, rate = 0, chan_mode = 0, channels = 0, bps = 0;
  bool variable = false;
  uint64_t number = 0;
};

const int kFixed[5][4] = {{0, 0, 0, 0}, {1, 0, 0, 0}, {2, -1, 0, 0}, {3, -3, 1, 0}, {4, -6, 4, -1}};

void read_residual(Bits& br, int64_t* out, int block, int order) {
  const int method = (int)br.u(2);
  if (method > 1) fail(E_DATA, "FLAC: reserved residual coding method");
  const int pbits = method ? 5 : 4, escape = method ? 31 : 15;
  const int po = (int)br.u(4);
  const int parts = 1 << po;
  if ((block & (parts - 1)) != 0 && po > 0) fail(E_DATA, "FLAC: block size is not a multiple of the partition count");
  const int per = block >> po;
  if (per < order) fail(E_DATA, "FLAC: the first residual partition is shorter than the predictor order");
  int i = order;
  for (int part = 0; part < parts; ++part) {
    const int cnt = per - (part == 0 ? order : 0);
    const int k = (int)br.u(pbits);
    if (k == escape) {
      const int raw = (int)br.u(5);
      for (int j = 0; j < cnt; ++j) out[i++] = raw ? br.s(raw) : 0;
    } else {
      for (int j = 0; j < cnt; ++j) {
        const uint64_t q = br.unary();
        const uint64_t v = (q << k) | br.u(k);
        out[i++] = (int64_t)(v >> 1) ^ -(int64_t)(v & 1);
      }
    }
  }
}

void read_subframe(Bits& br, int64_t* s, int block, int bps) {
  if (br.u(1)) fail(E_DATA, "FLAC: subframe padding bit is set");
  const int type = (int)br.u(6);
  int wasted = 0;
  if (br.u(1)) wasted = (int)br.unary() + 1;
  if (wasted >= bps) fail(E_DATA, "FLAC: more wasted bits than bits per sample");
  bps -= wasted;
  if (type == 0) {
    const int64_t v = br.s(bps);
    for (int i = 0; i < block; ++i) s[i] = v;
  } else if (type == 1) {
    for (int i = 0; i < block; ++i) s[i] = br.s(bps);
  } else if (type >= 8 && type <= 12) {
    const int order = type - 8;
    if (order > block) fail(E_DATA, "FLAC: predictor order exceeds the block size");
    for (int i = 0; i < order; ++i) s[i] = br.s(bps);
    read_residual(br, s, block, order);
    // predictor sums in wrapping (unsigned) arithmetic: a valid stream stays far inside 64 bits, a corrupt one must not
    // run into undefined behaviour before its CRC-16 is looked at
    const int* c = kFixed[order];
    for (int i = order; i < block; ++i) {
      uint64_t pred = 0;
      for (int j = 0; j < order; ++j) pred += (uint64_t)(int64_t)c[j] * (uint64_t)s[i - 1 - j];
      s[i] = (int64_t)((uint64_t)s[i] + pred);
    }
  } else if (type >= 32) {
    const int order = type - 31;
    if (order > block) fail(E_DATA, "FLAC: predictor order exceeds the block size");
    for (int i = 0; i < order; ++i) s[i] = br.s(bps);
    const int prec = (int)br.u(4) + 1;
    if (prec == 16) fail(E_DATA, "FLAC: reserved LPC precision");
    const int shift = (int)br.s(5);
    if (shift < 0) fail(E_DATA, "FLAC: negative LPC shift");
    int64_t coef[32];
    for (int j = 0; j < order; ++j) coef[j] = br.s(prec);
    read_residual(br, s, block, order);
    for (int i = order; i < block; ++i) {
      uint64_t pred = 0;
      for (int j = 0; j < order; ++j) pred += (uint64_t)coef[j] * (uint64_t)s[i - 1 - j];
      s[i] = (int64_t)((uint64_t)s[i] + (uint64_t)((int64_t)pred >> shift));
    }
  } else {
    fail(E_DATA, "FLAC: reserved subframe type");
  }
  if (wasted)
    for (int i = 0; i < block; ++i) s[i] = (int64_t)((uint64_t)s[i] << wasted);
}

// Decodes the frame at d[off..]; returns the offset of the byte after its CRC-16.  ch[c] receives `block` samples.
size_t read_frame(const uint8_t* d, size_t n, size_t off, const StreamInfo& si, Frame& f, std::vector<std::vector<int64_t>>& ch) {
  if (n - off < 6) fail(E_DATA, "FLAC: truncated frame header");
  Bits br(d + off, n - off);
  if (br.u(14) != 0x3ffe) fail(E_DATA, "FLAC: lost frame synchronisation at byte " + std::to_string(off));
  if (br.u(1)) fail(E_DATA, "FLAC: reserved frame header bit is set");
  f.variable = br.u(1);
  const int bs_code = (int)br.u(4), sr_code = (int)br.u(4), ch_code = (int)br.u(4), ss_code = (int)br.u(3);
  if (br.u(1)) fail(E_DATA, "FLAC: reserved frame header bit is set");
  {   // coded number: UTF-8 style, up to 36 bits
    const uint32_t b0 = br.u(8);
    int extra;
    uint64_t v;
    if (b0 < 0x80) { extra = 0; v = b0; }
    else if ((b0 & 0xe0) == 0xc0) { extra = 1; v = b0 & 0x1f; }
    else if ((b0 & 0xf0) == 0xe0) { extra = 2; v = b0 & 0x0f; }
    else if ((b0 & 0xf8) == 0xf0) { extra = 3; v = b0 & 0x07; }
    else if ((b0 & 0xfc) == 0xf8) { extra = 4; v = b0 & 0x03; }
    else if ((b0 & 0xfe) == 0xfc) { extra = 5; v = b0 & 0x01; }
    else if (b0 == 0xfe) { extra = 6; v = 0; }
    else fail(E_DATA, "FLAC: invalid coded frame number");
    for (int i = 0; i < extra; ++i) {
      const uint32_t b = br.u(8);
      if ((b & 0xc0) != 0x80) fail(E_DATA, "FLAC: invalid coded frame number");
      v = (v << 6) | (b & 0x3f);
    }
    f.number = v;
  }
  switch (bs_code) {
    case 0: fail(E_DATA, "FLAC: reserved block size code");
    case 1: f.block = 192; break;
    case 6: f.block = (int)br.u(8) + 1; break;
    case 7: f.block = (int)br.u(16) + 1; break;
    default: f.block = bs_code < 6 ? 576 << (bs_code - 2) : 256 << (bs_code - 8);
  }
  static const int kRates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
  if (sr_code < 12) f.rate = sr_code ? kRates[sr_code] : si.rate;
  else if (sr_code == 12) f.rate = (int)br.u(8) * 1000;
  else if (sr_code == 13) f.rate = (int)br.u(16);
  else if (sr_code == 14) f.rate = (int)br.u(16) * 10;
  else fail(E_DATA, "FLAC: invalid sample rate code");
  static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
  if (kBits[ss_code] < 0) fail(E_DATA, "FLAC: reserved sample size code");
  f.bps = ss_code ? kBits[ss_code] : si.bps;
  if (ch_code > 10) fail(E_DATA, "FLAC: reserved channel assignment");
  f.chan_mode = ch_code < 8 ? 0 : ch_code - 7;          // 1 left/side, 2 side/right, 3 mid/side
  f.channels = ch_code < 8 ? ch_code + 1 : 2;
  const size_t hdr = br.byte_pos();
  if (br.u(8) != crc8(d + off, hdr)) fail(E_DATA, "FLAC: frame header CRC-8 mismatch at byte " + std::to_string(off));
  if (f.channels != si.channels || f.bps != si.bps || f.rate != si.rate)
    fail(E_UNSUPPORTED, "FLAC: a frame changes the channel count, sample size or sample rate of STREAMINFO");
  if (f.block > si.max_block) fail(E_DATA, "FLAC: frame block size exceeds STREAMINFO's maximum");
  for (int c = 0; c < f.channels; ++c) {
    if ((int)ch[c].size() < f.block) ch[c].resize(f.block);
    const bool side = (f.chan_mode == 1 && c == 1) || (f.chan_mode == 2 && c == 0) || (f.chan_mode == 3 && c == 1);
    read_subframe(br, ch[c].data(), f.block, f.bps + (side ? 1 : 0));
  }
  br.align();
  const size_t body = br.byte_pos();
  if (br.u(16) != crc16(d + off, body)) fail(E_DATA, "FLAC: frame CRC-16 mismatch at byte " + std::to_string(off));
  int64_t* a = ch[0].data();
  int64_t* b = f.channels > 1 ? ch[1].data() : nullptr;
  if (f.chan_mode == 1) {
    for (int i = 0; i < f.block; ++i) b[i] = (int64_t)((uint64_t)a[i] - (uint64_t)b[i]);
  } else if (f.chan_mode == 2) {
    for (int i = 0; i < f.block; ++i) a[i] = (int64_t)((uint64_t)a[i] + (uint64_t)b[i]);
  } else if (f.chan_mode == 3) {
    for (int i = 0; i < f.block; ++i) {
      const uint64_t side = (uint64_t)b[i], mid = ((uint64_t)a[i] << 1) | (side & 1);
      a[i] = (int64_t)(mid + side) >> 1;
      b[i] = (int64_t)(mid - side) >> 1;
    }
  }
  return off + body + 2;
}

// The interleaved little-endian PCM the MD5 signature is defined over
void pack_pcm(std::vector<uint8_t>& pcm, const std::vector<std::vector<int64_t>>& ch, int channels, int block, int bytes) {
  const size_t at = pcm.size();
  pcm.resize(at + (size_t)block * channels * bytes);
  uint8_t* q = pcm.data() + at;
  if (bytes == 2 && channels == 1) {
    const int64_t* s = ch[0].data();
    for (int i = 0; i < block; ++i) { const uint16_t v = (uint16_t)s[i]; q[2 * i] = (uint8_t)v; q[2 * i + 1] = (uint8_t)(v >> 8); }
    return;
  }
  for (int i = 0; i < block; ++i)
    for (int c = 0; c < channels; ++c) {
      const uint64_t v = (uint64_t)ch[c][i];
      for (int k = 0; k < bytes; ++k) *q++ = (uint8_t)(v >> (8 * k));
    }
}

// One contiguous run of frames.  A run other than the first starts at a byte offset FOUND by scanning for a plausible
// frame header, so everything about it is provisional until the run before it ends exactly there.
struct Run {
  size_t pos = 0, end_pos = 0;          // first byte; one past the last frame decoded
  int64_t sample = 0, end_sample = 0;   // absolute index of the first sample; one past the last
  size_t stop = 0;                      // decode frames while pos < stop (the next run's start; n for the last run)
  std::vector<uint8_t> pcm;             // interleaved PCM for the MD5 (when asked for)
  bool failed = false;
  Fail error{0, ""};
};

template <typename Sink>
void decode_run(const uint8_t* d, size_t n, const StreamInfo& si, Run& r, bool keep_pcm, int64_t total, Sink& sink, Md5* live = nullptr) {
  std::vector<std::vector<int64_t>> ch(si.channels);
  const int bytes = (si.bps + 7) / 8;
  Frame f;
  size_t off = r.pos;
  int64_t done = r.sample;
  try {
    while (off < r.stop) {
      if (total > 0 && done >= total) break;             // trailing bytes (tags) after the declared samples
      if (n - off >= 3 && d[off] == 'T' && d[off + 1] == 'A' && d[off + 2] == 'G') break;   // ID3v1
      off = read_frame(d, n, off, si, f, ch);
      if (keep_pcm) pack_pcm(r.pcm, ch, si.channels, f.block, bytes);
      if (keep_pcm && live) {               // front-to-back walk: hash as we go, nothing is kept
        live->update(r.pcm.data(), r.pcm.size());
        r.pcm.clear();
      }
      sink(done, f.block, ch);
      done += f.block;
    }
  } catch (const Fail& e) {
    r.failed = true;
    r.error = e;
  } catch (const std::exception& e) {        // allocation failure inside a worker thread: reported, not propagated
    r.failed = true;
    r.error = Fail{-4, std::string("FLAC: ") + e.what()};
  }
  r.end_pos = off;
  r.end_sample = done;
}

// The header fields of a frame that may start at d[off]: false unless the sync code, the reserved bits, the CRC-8 and
// STREAMINFO's stream parameters all agree (read_frame repeats this with error messages).
bool plausible_header(const uint8_t* d, size_t n, size_t off, const StreamInfo& si, Frame& f) {
  if (n - off < 16 || d[off] != 0xff || (d[off + 1] & 0xfe) != 0xf8) return false;
  try {
    std::vector<std::vector<int64_t>> none;
    Bits br(d + off, std::min<size_t>(n - off, 16));
    br.u(15);
    f.variable = br.u(1);
    const int bs_code = (int)br.u(4), sr_code = (int)br.u(4), ch_code = (int)br.u(4), ss_code = (int)br.u(3);
    if (br.u(1) || bs_code == 0 || sr_code == 15 || ch_code > 10 || ss_code == 3) return false;
    const uint32_t b0 = br.u(8);
    int extra = 0;
    uint64_t v = b0;
    if (b0 >= 0x80) {
      if (b0 == 0xff || b0 < 0xc0) return false;
      extra = __builtin_clz(~(b0 << 24)) - 1;
      v = b0 & (0x7fu >> (extra + 1));
    }
    for (int i = 0; i < extra; ++i) {
      const uint32_t b = br.u(8);
      if ((b & 0xc0) != 0x80) return false;
      v = (v << 6) | (b & 0x3f);
    }
    f.number = v;
    f.block = bs_code == 1 ? 192 : bs_code == 6 ? (int)br.u(8) + 1 : bs_code == 7 ? (int)br.u(16) + 1
              : bs_code < 6 ? 576 << (bs_code - 2) : 256 << (bs_code - 8);
    static const int kRates[12] = {0, 88200, 176400, 192000, 8000, 16000, 22050, 24000, 32000, 44100, 48000, 96000};
    const int rate = sr_code < 12 ? (sr_code ? kRates[sr_code] : si.rate) : sr_code == 12 ? (int)br.u(8) * 1000
                     : sr_code == 13 ? (int)br.u(16) : (int)br.u(16) * 10;
    static const int kBits[8] = {0, 8, 12, -1, 16, 20, 24, 32};
    const int bps = ss_code ? kBits[ss_code] : si.bps;
    const int channels = ch_code < 8 ? ch_code + 1 : 2;
    const size_t hdr = br.byte_pos();
    if (br.u(8) != crc8(d + off, hdr)) return false;
    return rate == si.rate && bps == si.bps && channels == si.channels && f.block <= si.max_block;
  } catch (const Fail&) {
    return false;
  }
}

enum { FLAG_NO_MD5 = 1 };

template <typename Sink>
int64_t flac_decode(const uint8_t* d, size_t n, rvb_audio_info* info, Sink&& sink, bool want_samples, int flags) {
  const FlacStream fs = flac_open(d, n);
  const StreamInfo& si = fs.si;
  info->container = RVB_AUDIO_FLAC;
  info->channels = si.channels;
  info->sample_rate = si.rate;
  info->bits_per_sample = si.bps;
  info->sample_format = si.bps <= 16 ? RVB_SAMPLE_I16 : RVB_SAMPLE_I32;
  info->frames = si.total;
  info->md5_checked = 0;
  if (!want_samples && si.total > 0) return si.total;
  const bool check_md5 = si.has_md5 && !(flags & FLAG_NO_MD5);

  // Frames decode independently of each other, so a long stream is cut into runs of frames for a few host threads.  A
  // run's start is found by scanning forward from an even split of the bytes for a header that passes every check a header
  // can pass; where it sits in the output follows from its frame number (fixed block size) or sample number.  The claim is
  // settled afterwards: each run must end exactly where the next one began, byte and sample; if one does not (a header
  // look-alike inside audio data, 1 in ~10^8 bytes), or a run fails, the stream is decoded again front to back, which
  // also produces the error message of the first bad frame.
  int threads = (flags >> 8) & 0xff;
  if (threads == 0) threads = (int)std::min<size_t>(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 16u),
                                                    (n - fs.first_frame) / (512u << 10));
  const bool fixed_ok = si.min_block == si.max_block;
  std::vector<Run> runs;
  if (threads > 1 && si.total > 0) {
    Frame f0;
    if (plausible_header(d, n, fs.first_frame, si, f0) && (f0.variable || fixed_ok)) {
      runs.resize(1);
      runs[0].pos = fs.first_frame;
      for (int t = 1; t < threads; ++t) {
        size_t p = fs.first_frame + (n - fs.first_frame) / threads * t;
        Frame f;
        const size_t limit = std::min(n, p + (size_t)(1u << 20));
        while (p < limit && !plausible_header(d, n, p, si, f)) ++p;
        if (p >= limit || f.variable != f0.variable) { runs.clear(); break; }
        if (p <= runs.back().pos) continue;
        Run r;
        r.pos = p;
        r.sample = f.variable ? (int64_t)f.number : (int64_t)f.number * si.max_block;
        if (r.sample <= runs.back().sample || r.sample >= si.total) { runs.clear(); break; }
        runs.push_back(std::move(r));
      }
    }
  }
  bool settled = false;
  if (runs.size() > 1) {
    for (size_t t = 0; t < runs.size(); ++t) runs[t].stop = t + 1 < runs.size() ? runs[t + 1].pos : n;
    std::vector<std::thread> pool;
    for (size_t t = 1; t < runs.size(); ++t)
      pool.emplace_back([&, t] { decode_run(d, n, si, runs[t], check_md5, si.total, sink); });
    decode_run(d, n, si, runs[0], check_md5, si.total, sink);
    for (auto& th : pool) th.join();
    settled = true;
    for (size_t t = 0; t < runs.size() && settled; ++t) {
      settled = !runs[t].failed;
      if (t + 1 < runs.size()) settled = settled && runs[t].end_pos == runs[t + 1].pos && runs[t].end_sample == runs[t + 1].sample;
    }
  }
  Md5 md5;
  if (!settled) {
    runs.assign(1, Run());
    runs[0].pos = fs.first_frame;
    runs[0].stop = n;
    decode_run(d, n, si, runs[0], check_md5, si.total, sink, &md5);
    if (runs[0].failed) throw runs[0].error;
  }
  const int64_t done = runs.back().end_sample;
  if (si.total > 0 && done != si.total)
    fail(E_DATA, "FLAC: STREAMINFO declares " + std::to_string(si.total) + " samples per channel, the frames hold " + std::to_string(done));
  if (check_md5) {
    for (const Run& r : runs) md5.update(r.pcm.data(), r.pcm.size());
    uint8_t got[16];
    md5.finish(got);
    if (std::memcmp(got, si.md5, 16) != 0) fail(E_DATA, "FLAC: MD5 of the decoded audio differs from STREAMINFO's signature");
    info->md5_checked = 1;
  }
  info->frames = done;
  info->decode_threads = (int32_t)runs.size();
  return done;
}

// ---------------------------------------------------------------------------------------------- RIFF / WAVE
struct Wave {
  int tag = 0, channels = 0, rate = 0, align = 0, bits = 0;
  const uint8_t* data = nullptr;
  size_t bytes = 0;
};

uint32_t le(const uint8_t* p, int n) {
  uint32_t v = 0;
  for (int i = n - 1; i >= 0; --i) v = (v << 8) | p[i];
  return v;
}

Wave wave_open(const uint8_t* d, size_t n) {
  const bool rf64 = n >= 12 && std::memcmp(d, "RF64", 4) == 0;          // EBU Tech 3306: RIFF with 64-bit sizes in a ds64 chunk
  if (n < 12 || (!rf64 && std::memcmp(d, "RIFF", 4) != 0) || std::memcmp(d + 8, "WAVE", 4) != 0) fail(E_DATA, "not a RIFF/WAVE file");
  Wave w;
  bool fmt = false;
  uint64_t data64 = 0;
  size_t pos = 12;
  while (pos <= n && n - pos >= 8) {
    const uint8_t* id = d + pos;
    size_t size = le(d + pos + 4, 4);
    pos += 8;
    if (std::memcmp(id, "data", 4) == 0) {
      if (!fmt) fail(E_DATA, "WAVE: data chunk before the fmt chunk");
      w.data = d + pos;
      if (rf64 && size == 0xffffffffu && data64 != 0) size = (size_t)std::min<uint64_t>(data64, n - pos);
      w.bytes = std::min(size, n - pos);       // streamed writers leave 0 / 0xffffffff here: take what is there
      if (size == 0) w.bytes = n - pos;
      return w;
    }
    if (rf64 && std::memcmp(id, "ds64", 4) == 0 && size >= 24 && size <= n - pos)
      data64 = (uint64_t)le(d + pos + 8, 4) | (uint64_t)le(d + pos + 12, 4) << 32;      // riffSize, dataSize, sampleCount
    if (size > n - pos) fail(E_DATA, "WAVE: chunk runs past the end of the file");
    if (std::memcmp(id, "fmt ", 4) == 0) {
      if (size < 16) fail(E_DATA, "WAVE: fmt chunk shorter than 16 bytes");
      const uint8_t* p = d + pos;
      w.tag = (int)le(p, 2);
      w.channels = (int)le(p + 2, 2);
      w.rate = (int)le(p + 4, 4);
      w.align = (int)le(p + 12, 2);
      w.bits = (int)le(p + 14, 2);
      if (w.tag == 0xfffe) {                   // WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
        if (size < 40) fail(E_DATA, "WAVE: extensible fmt chunk shorter than 40 bytes");
        w.tag = (int)le(p + 24, 2);
      }
      fmt = true;
    }
    pos += size;
    if ((size & 1) && pos < n) ++pos;     // the pad byte of an odd-sized chunk may be missing at the end of the file
  }
  fail(E_DATA, "WAVE: missing fmt or data chunk");
}

// AIFF / AIFF-C (Apple, 1989 / 1991): big-endian chunks; COMM = channels, frames, bits, sample rate as an 80-bit extended
// float (+ a compression id in AIFF-C: NONE / twos big-endian, sowt little-endian, fl32 / FL32, fl64); SSND = offset,
// block size, samples.  Mapped onto the WAVE decoder's sample formats (endianness handled by a byte-swapped copy).
struct Aiff {
  Wave w;
  bool big_endian = true;
};
double ext80(const uint8_t* p) {
  const int exp = ((p[0] & 0x7f) << 8) | p[1];
  uint64_t mant = 0;
  for (int i = 0; i < 8; ++i) mant = (mant << 8) | p[2 + i];
  if (exp == 0 && mant == 0) return 0.0;
  const double v = std::ldexp((double)mant, exp - 16383 - 63);
  return (p[0] & 0x80) ? -v : v;
}
Aiff aiff_open(const uint8_t* d, size_t n) {
  const bool aifc = std::memcmp(d + 8, "AIFC", 4) == 0;
  Aiff a;
  bool comm = false;
  uint32_t frames = 0;
  size_t pos = 12;
  while (pos <= n && n - pos >= 8) {
    const uint8_t* id = d + pos;
    const size_t size = be(d + pos + 4, 4);
    pos += 8;
    if (size > n - pos && std::memcmp(id, "SSND", 4) != 0) fail(E_DATA, "AIFF: chunk runs past the end of the file");
    if (std::memcmp(id, "COMM", 4) == 0) {
      if (size < 18) fail(E_DATA, "AIFF: COMM chunk shorter than 18 bytes");
      const uint8_t* p = d + pos;
      a.w.channels = (int)be(p, 2);
      frames = be(p + 2, 4);
      a.w.bits = (int)be(p + 6, 2);
      a.w.rate = (int)std::lround(ext80(p + 8));
      a.w.tag = 1;
      if (aifc) {
        if (size < 22) fail(E_DATA, "AIFF-C: COMM chunk without a compression type");
        const uint8_t* c = p + 18;
        if (std::memcmp(c, "NONE", 4) == 0 || std::memcmp(c, "twos", 4) == 0) a.big_endian = true;
        else if (std::memcmp(c, "sowt", 4) == 0) a.big_endian = false;
        else if (std::memcmp(c, "fl32", 4) == 0 || std::memcmp(c, "FL32", 4) == 0) a.w.tag = 3;
        else if (std::memcmp(c, "fl64", 4) == 0 || std::memcmp(c, "FL64", 4) == 0) a.w.tag = 3;
        else if (std::memcmp(c, "alaw", 4) == 0 || std::memcmp(c, "ALAW", 4) == 0) { a.w.tag = 6; a.w.bits = 8; }
        else if (std::memcmp(c, "ulaw", 4) == 0 || std::memcmp(c, "ULAW", 4) == 0) { a.w.tag = 7; a.w.bits = 8; }
        else fail(E_UNSUPPORTED, std::string("AIFF-C: compression type '") + std::string((const char*)c, 4) + "' is not decoded here");
      }
      a.w.bits = (a.w.bits + 7) / 8 * 8;       // 12-bit samples sit left-justified in 16 bits, 20-bit in 24
      comm = true;
    } else if (std::memcmp(id, "SSND", 4) == 0) {
      if (!comm) fail(E_DATA, "AIFF: SSND chunk before the COMM chunk");
      if (size < 8 || n - pos < 8) fail(E_DATA, "AIFF: SSND chunk shorter than its header");
      const size_t off = be(d + pos, 4);
      const size_t start = pos + 8 + off;
      if (start > n) fail(E_DATA, "AIFF: sample data offset past the end of the file");
      a.w.data = d + start;
      const size_t want = (size_t)frames * (size_t)a.w.channels * (size_t)(a.w.bits / 8);
      a.w.bytes = std::min(want, n - start);
      return a;
    }
    pos += size;
    if ((size & 1) && pos < n) ++pos;     // the pad byte of an odd-sized chunk may be missing at the end of the file
  }
  fail(E_DATA, "AIFF: missing COMM or SSND chunk");
}

int16_t alaw(uint8_t a) {
  a ^= 0x55;
  int t = (a & 0x0f) << 4;
  const int seg = (a & 0x70) >> 4;
  if (seg == 0) t += 8;
  else if (seg == 1) t += 0x108;
  else t = (t + 0x108) << (seg - 1);
  return (int16_t)((a & 0x80) ? t : -t);
}
int16_t ulaw(uint8_t u) {
  u = (uint8_t)~u;
  int t = ((u & 0x0f) << 3) + 0x84;
  t <<= (u & 0x70) >> 4;
  return (int16_t)((u & 0x80) ? (0x84 - t) : (t - 0x84));
}

struct WaveFormat {
  int sample_format, bytes;
};
WaveFormat wave_format(const Wave& w) {
  if (w.channels < 1 || w.rate < 1) fail(E_DATA, "WAVE: fmt chunk declares no channels or sample rate 0");
  switch (w.tag) {
    case 1:
      if (w.bits == 8) return {RVB_SAMPLE_U8, 1};
      if (w.bits == 16) return {RVB_SAMPLE_I16, 2};
      if (w.bits == 24) return {RVB_SAMPLE_I32, 3};
      if (w.bits == 32) return {RVB_SAMPLE_I32, 4};
      break;
    case 3:
      if (w.bits == 32) return {RVB_SAMPLE_F32, 4};
      if (w.bits == 64) return {RVB_SAMPLE_F64, 8};
      break;
    case 6:
    case 7:
      if (w.bits == 8) return {RVB_SAMPLE_I16, 1};
      break;
    default:
      break;
  }
  fail(E_UNSUPPORTED, "WAVE: format tag " + std::to_string(w.tag) + " with " + std::to_string(w.bits) + " bits per sample is not decoded here");
}

// value of sample (frame i, channel c) as the double that `.to(torch.float)` would round
inline double wave_sample(const Wave& w, const WaveFormat& wf, size_t i, int c) {
  const uint8_t* p = w.data + (i * (size_t)w.channels + (size_t)c) * (size_t)wf.bytes;
  switch (w.tag) {
    case 1:
      if (wf.bytes == 1) return (double)p[0];
      if (wf.bytes == 2) return (double)(int16_t)le(p, 2);
      if (wf.bytes == 3) return (double)(int32_t)(le(p, 3) << 8);
      return (double)(int32_t)le(p, 4);
    case 3:
      if (wf.bytes == 4) { float f; std::memcpy(&f, p, 4); return (double)f; }
      { double g; std::memcpy(&g, p, 8); return g; }
    case 6: return (double)alaw(p[0]);
    default: return (double)ulaw(p[0]);
  }
}

int sniff(const uint8_t* d, size_t n) {
  if (n >= 12 && (std::memcmp(d, "RIFF", 4) == 0 || std::memcmp(d, "RF64", 4) == 0) && std::memcmp(d + 8, "WAVE", 4) == 0) return RVB_AUDIO_WAVE;
  if (n >= 12 && std::memcmp(d, "FORM", 4) == 0 && (std::memcmp(d + 8, "AIFF", 4) == 0 || std::memcmp(d + 8, "AIFC", 4) == 0)) return RVB_AUDIO_AIFF;
  const size_t off = n >= 10 && std::memcmp(d, "ID3", 3) == 0 ? skip_id3(d, n) : 0;
  if (n - off >= 4 && std::memcmp(d + off, "fLaC", 4) == 0) return RVB_AUDIO_FLAC;
  if (n - off >= 4 && std::memcmp(d + off, "OggS", 4) == 0) fail(E_UNSUPPORTED, "Ogg container (Vorbis / Opus): lossy codecs are not decoded here");
  if (off > 0 || (n >= 2 && d[0] == 0xff && (d[1] & 0xe0) == 0xe0)) return RVB_AUDIO_MP3;       // MPEG audio: csrc/mp3.cpp says which layers it takes
  if (n >= 12 && std::memcmp(d, "FORM", 4) == 0) fail(E_UNSUPPORTED, "IFF container other than AIFF / AIFF-C is not decoded here");
  fail(E_DATA, "unrecognised audio container (RIFF/WAVE, RF64, AIFF, FLAC and MPEG audio Layer III are decoded)");
}

template <typename Out>
int64_t decode(const uint8_t* d, size_t n, int channel, Out* out, int64_t capacity, int flags, rvb_audio_info* info, bool i16) {
  const int kind = sniff(d, n);
  const bool want = out != nullptr;
  if (kind == RVB_AUDIO_WAVE || kind == RVB_AUDIO_AIFF) {
    Wave w;
    std::vector<uint8_t> swapped;              // AIFF: big-endian samples, byte-swapped once into the WAVE layout
    bool signed8 = false;
    if (kind == RVB_AUDIO_WAVE) {
      w = wave_open(d, n);
    } else {
      const Aiff a = aiff_open(d, n);
      w = a.w;
      signed8 = w.tag == 1 && w.bits == 8;
      const size_t width = (size_t)w.bits / 8;
      if (want && width > 1 && a.big_endian) {
        swapped.resize(w.bytes / width * width);
        for (size_t i = 0; i + width <= swapped.size(); i += width)
          for (size_t k = 0; k < width; ++k) swapped[i + k] = w.data[i + width - 1 - k];
        w.data = swapped.data();
        w.bytes = swapped.size();
      }
    }
    WaveFormat wf = wave_format(w);
    // 8-bit AIFF is two's complement; FFmpeg's pcm_s8 decoder hands it on as offset binary (uint8, + 128), the same
    // native format as 8-bit WAVE
    info->container = kind;
    info->channels = w.channels;
    info->sample_rate = w.rate;
    info->bits_per_sample = w.bits;
    info->sample_format = wf.sample_format;
    info->frames = (int64_t)(w.bytes / ((size_t)wf.bytes * (size_t)w.channels));
    info->md5_checked = 0;
    info->decode_threads = 1;
    if (!want) return info->frames;
    if (i16 && wf.sample_format != RVB_SAMPLE_I16) fail(E_UNSUPPORTED, "rvb_audio_decode_i16: the file's native sample format is not int16");
    if (channel >= w.channels) fail(E_ARG, "audio decode: channel index out of range");
    const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? w.channels : channel + 1;
    if ((int64_t)(c1 - c0) * info->frames > capacity) fail(E_ARG, "audio decode: output buffer too small");
    for (int c = c0; c < c1; ++c) {
      Out* o = out + (size_t)(c - c0) * (size_t)info->frames;
      if (signed8)
        for (size_t i = 0; i < (size_t)info->frames; ++i) o[i] = (Out)((int)(int8_t)w.data[i * (size_t)w.channels + (size_t)c] + 128);
      else
        for (size_t i = 0; i < (size_t)info->frames; ++i) o[i] = (Out)wave_sample(w, wf, i, c);
    }
    return info->frames;
  }
  if (kind == RVB_AUDIO_MP3) {
    // MPEG audio Layer III: float32, as torchaudio's (FFmpeg's float) decoder hands it on; encoder delay / padding of a LAME-style
    // tag trimmed as FFmpeg trims them (mp3.cpp probe())
    try {
      rvb::mp3::Info mi;
      int64_t frames;
      if (!want) {
        mi = rvb::mp3::probe(d, n);
        frames = mi.samples;
      } else {
        if (i16) fail(E_UNSUPPORTED, "rvb_audio_decode_i16: the file's native sample format is not int16");
        std::vector<float> tmp;
        float* dst;
        if (std::is_same<Out, float>::value) dst = (float*)out;
        else { tmp.resize((size_t)std::max<int64_t>(capacity, 1)); dst = tmp.data(); }
        frames = rvb::mp3::decode(d, n, channel, dst, capacity, &mi, nullptr, (flags >> 8) & 0xff);
      }
      info->container = RVB_AUDIO_MP3;
      info->channels = mi.channels;
      info->sample_rate = mi.sample_rate;
      info->bits_per_sample = 0;                     // a lossy stream declares none
      info->sample_format = RVB_SAMPLE_F32;
      info->frames = mi.samples;
      info->md5_checked = 0;
      info->decode_threads = mi.decode_threads;
      return frames;
    } catch (const rvb::mp3::Error& e) {
      fail(e.code, e.msg);
    }
  }
  // FLAC: integer samples left-justified in the 16- or 32-bit word of the native sample format
  const FlacStream fs = flac_open(d, n);
  StreamInfo si = fs.si;
  {
    // STREAMINFO's 36-bit sample count is attacker-controlled and callers size buffers from it before a single frame is
    // decoded: a frame is at least 8 bytes (header 5, one subframe byte, CRC-16) and holds at most max_block (<= 65535) samples
    const int64_t per_frame = si.max_block > 0 ? si.max_block : 65535;
    const int64_t most = ((int64_t)(n - std::min(n, fs.first_frame)) / 8 + 1) * per_frame;
    if (si.total > most) fail(E_DATA, "FLAC: STREAMINFO declares more samples than the file can hold");
  }
  if (want) {
    if (i16 && si.bps > 16) fail(E_UNSUPPORTED, "rvb_audio_decode_i16: the file's native sample format is not int16");
    if (channel >= si.channels) fail(E_ARG, "audio decode: channel index out of range");
  }
  const int shift = (si.bps <= 16 ? 16 : 32) - si.bps;
  const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? si.channels : channel + 1;
  // total == 0 ("unknown"): a first pass counts the frames so that the planar layout is known
  auto nothing = [](int64_t, int, const std::vector<std::vector<int64_t>>&) {};
  int64_t frames = si.total;
  if (frames == 0) frames = flac_decode(d, n, info, nothing, true, want ? (flags | FLAG_NO_MD5) : flags);
  else if (!want) return flac_decode(d, n, info, nothing, false, flags);
  if (!want) return frames;
  if ((int64_t)(c1 - c0) * frames > capacity) fail(E_ARG, "audio decode: output buffer too small");
  return flac_decode(d, n, info, [&](int64_t at, int block, const std::vector<std::vector<int64_t>>& ch) {
    if (at + block > frames) fail(E_DATA, "FLAC: more samples than STREAMINFO declares");
    for (int c = c0; c < c1; ++c) {
      Out* o = out + (size_t)(c - c0) * (size_t)frames + (size_t)at;
      const int64_t* s = ch[c].data();
      for (int i = 0; i < block; ++i) o[i] = (Out)(int32_t)((uint64_t)s[i] << shift);
    }
  }, true, flags);
}

template <typename F>
int64_t guarded(const char* where, F&& f) {
  try {
    return f();
  } catch (const Fail& e) {
    rvb::set_error(std::string(where) + ": " + e.msg);
    return e.code;
  } catch (const std::bad_alloc&) {
    rvb::set_error(std::string(where) + ": out of host memory");
    return -4;
  } catch (const std::exception& e) {
    rvb::set_error(std::string(where) + ": " + e.what());
    return E_DATA;
  }
}

}  // namespace

extern "C" {

int rvb_audio_probe(const void* data, int64_t nbytes, rvb_audio_info* info) {
  if (!data || nbytes < 0 || !info) { rvb::set_error("rvb_audio_probe: bad argument"); return E_ARG; }
  std::memset(info, 0, sizeof(*info));
  const int64_t r = guarded("rvb_audio_probe", [&] { return decode<float>((const uint8_t*)data, (size_t)nbytes, -1, nullptr, 0, 0, info, false); });
  return r < 0 ? (int)r : OK;
}

int64_t rvb_audio_decode_f32(const void* data, int64_t nbytes, int channel, float* out, int64_t capacity, int flags, rvb_audio_info* info) {
  if (!data || nbytes < 0 || !info || !out || capacity < 0) { rvb::set_error("rvb_audio_decode_f32: bad argument"); return E_ARG; }
  std::memset(info, 0, sizeof(*info));
  return guarded("rvb_audio_decode_f32", [&] { return decode<float>((const uint8_t*)data, (size_t)nbytes, channel, out, capacity, flags, info, false); });
}

int64_t rvb_audio_decode_i16(const void* data, int64_t nbytes, int channel, int16_t* out, int64_t capacity, int flags, rvb_audio_info* info) {
  if (!data || nbytes < 0 || !info || !out || capacity < 0) { rvb::set_error("rvb_audio_decode_i16: bad argument"); return E_ARG; }
  std::memset(info, 0, sizeof(*info));
  return guarded("rvb_audio_decode_i16", [&] { return decode<int16_t>((const uint8_t*)data, (size_t)nbytes, channel, out, capacity, flags, info, true); });
}

}  // extern "C"
