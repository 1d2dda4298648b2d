// MPEG audio Layer III decoder (csrc/mp3.cpp): what torchaudio.load hands the reference for its `.mp3` inputs
// (/root/reference/README.md:61-106, asr/wenet/cli/reverb.py:128).  Host code, no GPU.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

namespace rvb {
namespace mp3 {

struct Error {
  int code;          // RVB_E_* (negative)
  std::string msg;
};

struct Info {
  int version = 0;            // 1 = MPEG-1, 2 = MPEG-2 (LSF), 3 = MPEG-2.5
  int channels = 0;
  int sample_rate = 0;
  int64_t audio_frames = 0;   // Layer III frames that carry audio (a leading Xing / Info / VBRI frame is not one)
  int samples_per_frame = 0;  // 1152 (MPEG-1) or 576
  int has_info_frame = 0;     // first frame is a Xing / Info (or VBRI) header frame
  int start_skip = 0;         // samples dropped at the start (encoder delay from the LAME tag + the decoder's 529), 0 without a tag
  int64_t samples = 0;        // samples per channel that decode() returns
  int bitrate_kbps = 0;       // of the first audio frame
  int decode_threads = 1;     // host threads decode() used (runs of frames; the result does not depend on it)
};

// what a decode pass saw (tests read it through rvb_test_mp3_decode)
struct Stats {
  int64_t granules = 0;           // granule x channel units decoded
  int64_t huff_exact = 0;         // ... whose Huffman data ended exactly on part2_3_length
  int64_t huff_short = 0;         // ... that stopped before it (stuffing bits: legal, rare)
  int64_t huff_overrun = 0;       // ... whose last count1 quadruple ran past it (discarded, as every decoder does)
  int64_t crc_checked = 0, crc_failed = 0;
  int64_t reservoir_missing = 0;  // frames whose main_data_begin points before the data we hold (their granules are silent)
  int64_t short_granules = 0, mixed_granules = 0, ms_granules = 0, intensity_granules = 0;
  int64_t max_main_data_begin = 0;
};

// Scan the stream (headers only).  Throws Error.
Info probe(const uint8_t* d, size_t n);

// Decode to float PCM, full scale = 1.0, planar [channel][samples]: `channel` < 0 = all channels, else that one.
// `out` may be null (count only).  `threads`: 0 = by stream length (one per ~400 frames, at most 16).  Returns samples per channel.
// Throws Error.
int64_t decode(const uint8_t* d, size_t n, int channel, float* out, int64_t capacity, Info* info, Stats* stats, int threads = 0);

// test access: Huffman table t (1..33 as numbered by the standard; 32 / 33 = count1 A / B): entries, codes, lengths; -1 = no such table
int huffman_table(int t, const uint16_t** codes, const uint8_t** lens, int* linbits_of_select /* [32] or null */);
// test access: the 512 coefficients of the synthesis window (ISO 11172-3 Table B.3, D[i])
const float* synthesis_window();
// test access: one granule's hybrid synthesis (alias reduction, IMDCT, overlap-add, frequency inversion) and the polyphase
// synthesis filterbank of 18 x 32 subband samples, on caller state (so a test can drive them with its own MDCT / analysis)
void hybrid_granule(float xr[576], float overlap[576], int block_type, int mixed, int lsf_long_bands_mixed, float out[576] /* [18][32] */);
void polyphase_granule(const float sb[576] /* [18][32] */, float vbuf[1024], int* voff, float pcm[576]);

}  // namespace mp3
}  // namespace rvb
