// MPEG-1 / MPEG-2 (LSF) / MPEG-2.5 audio Layer III decoder, written from the published algorithm (ISO/IEC 11172-3 section 2.4,
// ISO/IEC 13818-3 section 2.4.3.2 for the lower sampling frequencies) -- the reference reaches MP3 input through
// torchaudio.load (asr/wenet/cli/reverb.py:128; every usage example of /root/reference/README.md:61-106 is an .mp3).
//
// Stages per frame: header (+ CRC-16 when present) -> side information -> main data through the bit reservoir -> per granule and
// channel: scale factors, Huffman-coded spectrum (big_values regions with linbits, count1 quadruples), requantisation, then per
// granule: MS / intensity stereo, short-block reordering, alias reduction, IMDCT (36 / 3 x 12) with the four window shapes and
// overlap-add, frequency inversion, and the 32-band polyphase synthesis filterbank (DCT matrixing + the 512-tap window D[i]).
// Arithmetic in float with double-precision tables; output full scale = 1.0, not clipped (what a float decoder hands torchaudio).
//
// What pins it (tests/test_mp3.py): the Huffman tables are complete prefix codes (a single wrong number breaks that), the
// synthesis window was assembled from two independently remembered listings that agree on all 257 distinct values; a real
// encoder's stream (tests/golden/mathjax_invalid_keypress.mp3: 44.1 kHz joint stereo, long / start / short / stop blocks, bit
// reservoir) decodes with every granule's Huffman data ending exactly on its part2_3_length, every CRC-less frame in sync;
// IMDCT / filterbank against direct fp64 forms and against their own analysis (TDAC / near-perfect reconstruction); and an
// independent Layer III ENCODER in tests/mp3_writer.py (MPEG-1 and LSF, mono / L-R / M-S / intensity, all block types, mixed
// blocks, the reservoir) round-trips signals through this decoder at the quantiser's accuracy.
// Not pinned: torchaudio's own output for the same file (FFmpeg's trimming of encoder delay / padding is followed, see probe()).
#include "mp3.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace rvb {
namespace mp3 {
namespace {

enum { E_ARG = -1, E_UNSUPPORTED = -5, E_DATA = -6 };
[[noreturn]] void fail(int code, const std::string& m) { throw Error{code, "MP3: " + m}; }

// Huffman code tables of ISO/IEC 11172-3 Table B.7 as (code value, code length) per symbol, index = x * ylen + y (count1 tables:
// index = v*8 + w*4 + x*2 + y).  Every table is a complete prefix code (Kraft sum exactly 1, no codeword a prefix of another):
// tests/test_mp3.py checks that through rvb_test_mp3_huffman, which is what pins these numbers.

static const uint16_t HB1[4] = {
    1, 1, 1, 0,
};
static const uint8_t HL1[4] = {
    1, 3, 2, 3,
};
static const uint16_t HB2[9] = {
    1, 2, 1, 3, 1, 1, 3, 2, 0,
};
static const uint8_t HL2[9] = {
    1, 3, 6, 3, 3, 5, 5, 5, 6,
};
static const uint16_t HB3[9] = {
    3, 2, 1, 1, 1, 1, 3, 2, 0,
};
static const uint8_t HL3[9] = {
    2, 2, 6, 3, 2, 5, 5, 5, 6,
};
static const uint16_t HB5[16] = {
    1, 2, 6, 5, 3, 1, 4, 4, 7, 5, 7, 1, 6, 1, 1, 0,
};
static const uint8_t HL5[16] = {
    1, 3, 6, 7, 3, 3, 6, 7, 6, 6, 7, 8, 7, 6, 7, 8,
};
static const uint16_t HB6[16] = {
    7, 3, 5, 1, 6, 2, 3, 2, 5, 4, 4, 1, 3, 3, 2, 0,
};
static const uint8_t HL6[16] = {
    3, 3, 5, 7, 3, 2, 4, 5, 4, 4, 5, 6, 6, 5, 6, 7,
};
static const uint16_t HB7[36] = {
    1, 2, 10, 19, 16, 10, 3, 3, 7, 10, 5, 3, 11, 4, 13, 17, 8, 4, 12, 11, 18, 15, 11, 2,
    7, 6, 9, 14, 3, 1, 6, 4, 5, 3, 2, 0,
};
static const uint8_t HL7[36] = {
    1, 3, 6, 8, 8, 9, 3, 4, 6, 7, 7, 8, 6, 5, 7, 8, 8, 9, 7, 7, 8, 9, 9, 9, 7, 7, 8, 9, 9, 10, 8, 8,
    9, 10, 10, 10,
};
static const uint16_t HB8[36] = {
    3, 4, 6, 18, 12, 5, 5, 1, 2, 16, 9, 3, 7, 3, 5, 14, 7, 3, 19, 17, 15, 13, 10, 4,
    13, 5, 8, 11, 5, 1, 12, 4, 4, 1, 1, 0,
};
static const uint8_t HL8[36] = {
    2, 3, 6, 8, 8, 9, 3, 2, 4, 8, 8, 8, 6, 4, 6, 8, 8, 9, 8, 8, 8, 9, 9, 10, 8, 7, 8, 9, 10, 10, 9, 8,
    9, 9, 11, 11,
};
static const uint16_t HB9[36] = {
    7, 5, 9, 14, 15, 7, 6, 4, 5, 5, 6, 7, 7, 6, 8, 8, 8, 5, 15, 6, 9, 10, 5, 1,
    11, 7, 9, 6, 4, 1, 14, 4, 6, 2, 6, 0,
};
static const uint8_t HL9[36] = {
    3, 3, 5, 6, 8, 9, 3, 3, 4, 5, 6, 8, 4, 4, 5, 6, 7, 8, 6, 5, 6, 7, 7, 8, 7, 6, 7, 7, 8, 9, 8, 7,
    8, 8, 9, 9,
};
static const uint16_t HB10[64] = {
    1, 2, 10, 23, 35, 30, 12, 17, 3, 3, 8, 12, 18, 21, 12, 7, 11, 9, 15, 21, 32, 40, 19, 6,
    14, 13, 22, 34, 46, 23, 18, 7, 20, 19, 33, 47, 27, 22, 9, 3, 31, 22, 41, 26, 21, 20, 5, 3,
    14, 13, 10, 11, 16, 6, 5, 1, 9, 8, 7, 8, 4, 4, 2, 0,
};
static const uint8_t HL10[64] = {
    1, 3, 6, 8, 9, 9, 9, 10, 3, 4, 6, 7, 8, 9, 8, 8, 6, 6, 7, 8, 9, 10, 9, 9, 7, 7, 8, 9, 10, 10, 9, 10,
    8, 8, 9, 10, 10, 10, 10, 10, 9, 9, 10, 10, 11, 11, 10, 11, 8, 8, 9, 10, 10, 10, 11, 11, 9, 8, 9, 10, 10, 11, 11, 11,
};
static const uint16_t HB11[64] = {
    3, 4, 10, 24, 34, 33, 21, 15, 5, 3, 4, 10, 32, 17, 11, 10, 11, 7, 13, 18, 30, 31, 20, 5,
    25, 11, 19, 59, 27, 18, 12, 5, 35, 33, 31, 58, 30, 16, 7, 5, 28, 26, 32, 19, 17, 15, 8, 14,
    14, 12, 9, 13, 14, 9, 4, 1, 11, 4, 6, 6, 6, 3, 2, 0,
};
static const uint8_t HL11[64] = {
    2, 3, 5, 7, 8, 9, 8, 9, 3, 3, 4, 6, 8, 8, 7, 8, 5, 5, 6, 7, 8, 9, 8, 8, 7, 6, 7, 9, 8, 10, 8, 9,
    8, 8, 8, 9, 9, 10, 9, 10, 8, 8, 9, 10, 10, 11, 10, 11, 8, 7, 7, 8, 9, 10, 10, 10, 8, 7, 8, 9, 10, 10, 10, 10,
};
static const uint16_t HB12[64] = {
    9, 6, 16, 33, 41, 39, 38, 26, 7, 5, 6, 9, 23, 16, 26, 11, 17, 7, 11, 14, 21, 30, 10, 7,
    17, 10, 15, 12, 18, 28, 14, 5, 32, 13, 22, 19, 18, 16, 9, 5, 40, 17, 31, 29, 17, 13, 4, 2,
    27, 12, 11, 15, 10, 7, 4, 1, 27, 12, 8, 12, 6, 3, 1, 0,
};
static const uint8_t HL12[64] = {
    4, 3, 5, 7, 8, 9, 9, 9, 3, 3, 4, 5, 7, 7, 8, 8, 5, 4, 5, 6, 7, 8, 7, 8, 6, 5, 6, 6, 7, 8, 8, 8,
    7, 6, 7, 7, 8, 8, 8, 9, 8, 7, 8, 8, 8, 9, 8, 9, 8, 7, 7, 8, 8, 9, 9, 10, 9, 8, 8, 9, 9, 9, 9, 10,
};
static const uint16_t HB13[256] = {
    1, 5, 14, 21, 34, 51, 46, 71, 42, 52, 68, 52, 67, 44, 43, 19, 3, 4, 12, 19, 31, 26, 44, 33,
    31, 24, 32, 24, 31, 35, 22, 14, 15, 13, 23, 36, 59, 49, 77, 65, 29, 40, 30, 40, 27, 33, 42, 16,
    22, 20, 37, 61, 56, 79, 73, 64, 43, 76, 56, 37, 26, 31, 25, 14, 35, 16, 60, 57, 97, 75, 114, 91,
    54, 73, 55, 41, 48, 53, 23, 24, 58, 27, 50, 96, 76, 70, 93, 84, 77, 58, 79, 29, 74, 49, 41, 17,
    47, 45, 78, 74, 115, 94, 90, 79, 69, 83, 71, 50, 59, 38, 36, 15, 72, 34, 56, 95, 92, 85, 91, 90,
    86, 73, 77, 65, 51, 44, 43, 42, 43, 20, 30, 44, 55, 78, 72, 87, 78, 61, 46, 54, 37, 30, 20, 16,
    53, 25, 41, 37, 44, 59, 54, 81, 66, 76, 57, 54, 37, 18, 39, 11, 35, 33, 31, 57, 42, 82, 72, 80,
    47, 58, 55, 21, 22, 26, 38, 22, 53, 25, 23, 38, 70, 60, 51, 36, 55, 26, 34, 23, 27, 14, 9, 7,
    34, 32, 28, 39, 49, 75, 30, 52, 48, 40, 52, 28, 18, 17, 9, 5, 45, 21, 34, 64, 56, 50, 49, 45,
    31, 19, 12, 15, 10, 7, 6, 3, 48, 23, 20, 39, 36, 35, 53, 21, 16, 23, 13, 10, 6, 1, 4, 2,
    16, 15, 17, 27, 25, 20, 29, 11, 17, 12, 16, 8, 1, 1, 0, 1,
};
static const uint8_t HL13[256] = {
    1, 4, 6, 7, 8, 9, 9, 10, 9, 10, 11, 11, 12, 12, 13, 13, 3, 4, 6, 7, 8, 8, 9, 9, 9, 9, 10, 10, 11, 12, 12, 12,
    6, 6, 7, 8, 9, 9, 10, 10, 9, 10, 10, 11, 11, 12, 13, 13, 7, 7, 8, 9, 9, 10, 10, 10, 10, 11, 11, 11, 11, 12, 13, 13,
    8, 7, 9, 9, 10, 10, 11, 11, 10, 11, 11, 12, 12, 13, 13, 14, 9, 8, 9, 10, 10, 10, 11, 11, 11, 11, 12, 11, 13, 13, 14, 14,
    9, 9, 10, 10, 11, 11, 11, 11, 11, 12, 12, 12, 13, 13, 14, 14, 10, 9, 10, 11, 11, 11, 12, 12, 12, 12, 13, 13, 13, 14, 16, 16,
    9, 8, 9, 10, 10, 11, 11, 12, 12, 12, 12, 13, 13, 14, 15, 15, 10, 9, 10, 10, 11, 11, 11, 13, 12, 13, 13, 14, 14, 14, 16, 15,
    10, 10, 10, 11, 11, 12, 12, 13, 12, 13, 14, 13, 14, 15, 16, 17, 11, 10, 10, 11, 12, 12, 12, 12, 13, 13, 13, 14, 15, 15, 15, 16,
    11, 11, 11, 12, 12, 13, 12, 13, 14, 14, 15, 15, 15, 16, 16, 16, 12, 11, 12, 13, 13, 13, 14, 14, 14, 14, 14, 15, 16, 15, 16, 16,
    13, 12, 12, 13, 13, 13, 15, 14, 14, 17, 15, 15, 15, 17, 16, 16, 12, 12, 13, 14, 14, 14, 15, 14, 15, 15, 16, 16, 19, 18, 19, 16,
};
static const uint16_t HB15[256] = {
    7, 12, 18, 53, 47, 76, 124, 108, 89, 123, 108, 119, 107, 81, 122, 63, 13, 5, 16, 27, 46, 36, 61, 51,
    42, 70, 52, 83, 65, 41, 59, 36, 19, 17, 15, 24, 41, 34, 59, 48, 40, 64, 50, 78, 62, 80, 56, 33,
    29, 28, 25, 43, 39, 63, 55, 93, 76, 59, 93, 72, 54, 75, 50, 29, 52, 22, 42, 40, 67, 57, 95, 79,
    72, 57, 89, 69, 49, 66, 46, 27, 77, 37, 35, 66, 58, 52, 91, 74, 62, 48, 79, 63, 90, 62, 40, 38,
    125, 32, 60, 56, 50, 92, 78, 65, 55, 87, 71, 51, 73, 51, 70, 30, 109, 53, 49, 94, 88, 75, 66, 122,
    91, 73, 56, 42, 64, 44, 21, 25, 90, 43, 41, 77, 73, 63, 56, 92, 77, 66, 47, 67, 48, 53, 36, 20,
    71, 34, 67, 60, 58, 49, 88, 76, 67, 106, 71, 54, 38, 39, 23, 15, 109, 53, 51, 47, 90, 82, 58, 57,
    48, 72, 57, 41, 23, 27, 62, 9, 86, 42, 40, 37, 70, 64, 52, 43, 70, 55, 42, 25, 29, 18, 11, 11,
    118, 68, 30, 55, 50, 46, 74, 65, 49, 39, 24, 16, 22, 13, 14, 7, 91, 44, 39, 38, 34, 63, 52, 45,
    31, 52, 28, 19, 14, 8, 9, 3, 123, 60, 58, 53, 47, 43, 32, 22, 37, 24, 17, 12, 15, 10, 2, 1,
    71, 37, 34, 30, 28, 20, 17, 26, 21, 16, 10, 6, 8, 6, 2, 0,
};
static const uint8_t HL15[256] = {
    3, 4, 5, 7, 7, 8, 9, 9, 9, 10, 10, 11, 11, 11, 12, 13, 4, 3, 5, 6, 7, 7, 8, 8, 8, 9, 9, 10, 10, 10, 11, 11,
    5, 5, 5, 6, 7, 7, 8, 8, 8, 9, 9, 10, 10, 11, 11, 11, 6, 6, 6, 7, 7, 8, 8, 9, 9, 9, 10, 10, 10, 11, 11, 11,
    7, 6, 7, 7, 8, 8, 9, 9, 9, 9, 10, 10, 10, 11, 11, 11, 8, 7, 7, 8, 8, 8, 9, 9, 9, 9, 10, 10, 11, 11, 11, 12,
    9, 7, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 11, 11, 12, 12, 9, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10, 11, 11, 11, 12,
    9, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 11, 11, 12, 12, 12, 9, 8, 9, 9, 9, 9, 10, 10, 10, 11, 11, 11, 11, 12, 12, 12,
    10, 9, 9, 9, 10, 10, 10, 10, 10, 11, 11, 11, 11, 12, 13, 12, 10, 9, 9, 9, 10, 10, 10, 10, 11, 11, 11, 11, 12, 12, 12, 13,
    11, 10, 9, 10, 10, 10, 11, 11, 11, 11, 11, 11, 12, 12, 13, 13, 11, 10, 10, 10, 10, 11, 11, 11, 11, 12, 12, 12, 12, 12, 13, 13,
    12, 11, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 13, 13, 12, 13, 12, 11, 11, 11, 11, 11, 11, 12, 12, 12, 12, 12, 13, 13, 13, 13,
};
static const uint16_t HB16[256] = {
    1, 5, 14, 44, 74, 63, 110, 93, 172, 149, 138, 242, 225, 195, 376, 17, 3, 4, 12, 20, 35, 62, 53, 47,
    83, 75, 68, 119, 201, 107, 207, 9, 15, 13, 23, 38, 67, 58, 103, 90, 161, 72, 127, 117, 110, 209, 206, 16,
    45, 21, 39, 69, 64, 114, 99, 87, 158, 140, 252, 212, 199, 387, 365, 26, 75, 36, 68, 65, 115, 101, 179, 164,
    155, 264, 246, 226, 395, 382, 362, 9, 66, 30, 59, 56, 102, 185, 173, 265, 142, 253, 232, 400, 388, 378, 445, 16,
    111, 54, 52, 100, 184, 178, 160, 133, 257, 244, 228, 217, 385, 366, 715, 10, 98, 48, 91, 88, 165, 157, 148, 261,
    248, 407, 397, 372, 380, 889, 884, 8, 85, 84, 81, 159, 156, 143, 260, 249, 427, 401, 392, 383, 727, 713, 708, 7,
    154, 76, 73, 141, 131, 256, 245, 426, 406, 394, 384, 735, 359, 710, 352, 11, 139, 129, 67, 125, 247, 233, 229, 219,
    393, 743, 737, 720, 885, 882, 439, 4, 243, 120, 118, 115, 227, 223, 396, 746, 742, 736, 721, 712, 706, 223, 436, 6,
    202, 224, 222, 218, 216, 389, 386, 381, 364, 888, 443, 707, 440, 437, 1728, 4, 747, 211, 210, 208, 370, 379, 734, 723,
    714, 1735, 883, 877, 876, 3459, 865, 2, 377, 369, 102, 187, 726, 722, 358, 711, 709, 866, 1734, 871, 3458, 870, 434, 0,
    12, 10, 7, 11, 10, 17, 11, 9, 13, 12, 10, 7, 5, 3, 1, 3,
};
static const uint8_t HL16[256] = {
    1, 4, 6, 8, 9, 9, 10, 10, 11, 11, 11, 12, 12, 12, 13, 9, 3, 4, 6, 7, 8, 9, 9, 9, 10, 10, 10, 11, 12, 11, 12, 8,
    6, 6, 7, 8, 9, 9, 10, 10, 11, 10, 11, 11, 11, 12, 12, 9, 8, 7, 8, 9, 9, 10, 10, 10, 11, 11, 12, 12, 12, 13, 13, 10,
    9, 8, 9, 9, 10, 10, 11, 11, 11, 12, 12, 12, 13, 13, 13, 9, 9, 8, 9, 9, 10, 11, 11, 12, 11, 12, 12, 13, 13, 13, 14, 10,
    10, 9, 9, 10, 11, 11, 11, 11, 12, 12, 12, 12, 13, 13, 14, 10, 10, 9, 10, 10, 11, 11, 11, 12, 12, 13, 13, 13, 13, 15, 15, 10,
    10, 10, 10, 11, 11, 11, 12, 12, 13, 13, 13, 13, 14, 14, 14, 10, 11, 10, 10, 11, 11, 12, 12, 13, 13, 13, 13, 14, 13, 14, 13, 11,
    11, 11, 10, 11, 12, 12, 12, 12, 13, 14, 14, 14, 15, 15, 14, 10, 12, 11, 11, 11, 12, 12, 13, 14, 14, 14, 14, 14, 14, 13, 14, 11,
    12, 12, 12, 12, 12, 13, 13, 13, 13, 15, 14, 14, 14, 14, 16, 11, 14, 12, 12, 12, 13, 13, 14, 14, 14, 16, 15, 15, 15, 17, 15, 11,
    13, 13, 11, 12, 14, 14, 13, 14, 14, 15, 16, 15, 17, 15, 14, 11, 9, 8, 8, 9, 9, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 8,
};
static const uint16_t HB24[256] = {
    15, 13, 46, 80, 146, 262, 248, 434, 426, 669, 653, 649, 621, 517, 1032, 88, 14, 12, 21, 38, 71, 130, 122, 216,
    209, 198, 327, 345, 319, 297, 279, 42, 47, 22, 41, 74, 68, 128, 120, 221, 207, 194, 182, 340, 315, 295, 541, 18,
    81, 39, 75, 70, 134, 125, 116, 220, 204, 190, 178, 325, 311, 293, 271, 16, 147, 72, 69, 135, 127, 118, 112, 210,
    200, 188, 352, 323, 306, 285, 540, 14, 263, 66, 129, 126, 119, 114, 214, 202, 192, 180, 341, 317, 301, 281, 262, 12,
    249, 123, 121, 117, 113, 215, 206, 195, 185, 347, 330, 308, 291, 272, 520, 10, 435, 115, 111, 109, 211, 203, 196, 187,
    353, 332, 313, 298, 283, 531, 381, 17, 427, 212, 208, 205, 201, 193, 186, 177, 169, 320, 303, 286, 268, 514, 377, 16,
    335, 199, 197, 191, 189, 181, 174, 333, 321, 305, 289, 275, 521, 379, 371, 11, 668, 184, 183, 179, 175, 344, 331, 314,
    304, 290, 277, 530, 383, 373, 366, 10, 652, 346, 171, 168, 164, 318, 309, 299, 287, 276, 263, 513, 375, 368, 362, 6,
    648, 322, 316, 312, 307, 302, 292, 284, 269, 261, 512, 376, 370, 364, 359, 4, 620, 300, 296, 294, 288, 282, 273, 266,
    515, 380, 374, 369, 365, 361, 357, 2, 1033, 280, 278, 274, 267, 264, 259, 382, 378, 372, 367, 363, 360, 358, 356, 0,
    43, 20, 19, 17, 15, 13, 11, 9, 7, 6, 4, 7, 5, 3, 1, 3,
};
static const uint8_t HL24[256] = {
    4, 4, 6, 7, 8, 9, 9, 10, 10, 11, 11, 11, 11, 11, 12, 9, 4, 4, 5, 6, 7, 8, 8, 9, 9, 9, 10, 10, 10, 10, 10, 8,
    6, 5, 6, 7, 7, 8, 8, 9, 9, 9, 9, 10, 10, 10, 11, 7, 7, 6, 7, 7, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 7,
    8, 7, 7, 8, 8, 8, 8, 9, 9, 9, 10, 10, 10, 10, 11, 7, 9, 7, 8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10, 7,
    9, 8, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10, 11, 7, 10, 8, 8, 8, 9, 9, 9, 9, 10, 10, 10, 10, 10, 11, 11, 8,
    10, 9, 9, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 11, 11, 8, 10, 9, 9, 9, 9, 9, 9, 10, 10, 10, 10, 10, 11, 11, 11, 8,
    11, 9, 9, 9, 9, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 8, 11, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 8,
    11, 10, 10, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 8, 11, 10, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 8,
    12, 10, 10, 10, 10, 10, 10, 11, 11, 11, 11, 11, 11, 11, 11, 8, 8, 7, 7, 7, 7, 7, 7, 7, 7, 7, 7, 8, 8, 8, 8, 4,
};
static const uint16_t HB32[16] = {
    1, 5, 4, 5, 6, 5, 4, 4, 7, 3, 6, 0, 7, 2, 3, 1,
};
static const uint8_t HL32[16] = {
    1, 4, 4, 5, 4, 6, 5, 6, 4, 5, 5, 6, 5, 6, 6, 6,
};
static const uint16_t HB33[16] = {
    15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0,
};
static const uint8_t HL33[16] = {
    4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4,
};

// The synthesis window D[i] of ISO/IEC 11172-3 Table B.3 in units of 2^-16 (every coefficient of the standard's listing is a multiple
// of it).  Assembled from two independently remembered forms -- the standard's decimal listing and the integer polyphase arrays
// of a widely used decoder -- which agree on all 257 distinct values; D[512 - i] = -D[i] except at i = 64, 128, 192 (= +D[i]).
// tests/test_mp3.py: analysis with C[i] = D[i] / 32 followed by this synthesis reconstructs noise to -84 dB.
static const int32_t DWIN[512] = {
    0, -1, -1, -1, -1, -1, -1, -2, -2, -2, -2, -3, -3, -4, -4, -5,
    -5, -6, -7, -7, -8, -9, -10, -11, -13, -14, -16, -17, -19, -21, -24, -26,
    -29, -31, -35, -38, -41, -45, -49, -53, -58, -63, -68, -73, -79, -85, -91, -97,
    -104, -111, -117, -125, -132, -139, -147, -154, -161, -169, -176, -183, -190, -196, -202, -208,
    213, 218, 222, 225, 227, 228, 228, 227, 224, 221, 215, 208, 200, 189, 177, 163,
    146, 127, 106, 83, 57, 29, -2, -36, -72, -111, -153, -197, -244, -294, -347, -401,
    -459, -519, -581, -645, -711, -779, -848, -919, -991, -1064, -1137, -1210, -1283, -1356, -1428, -1498,
    -1567, -1634, -1698, -1759, -1817, -1870, -1919, -1962, -2001, -2032, -2057, -2075, -2085, -2087, -2080, -2063,
    2037, 2000, 1952, 1893, 1822, 1739, 1644, 1535, 1414, 1280, 1131, 970, 794, 605, 402, 185,
    -45, -288, -545, -814, -1095, -1388, -1692, -2006, -2330, -2663, -3004, -3351, -3705, -4063, -4425, -4788,
    -5153, -5517, -5879, -6237, -6589, -6935, -7271, -7597, -7910, -8209, -8491, -8755, -8998, -9219, -9416, -9585,
    -9727, -9838, -9916, -9959, -9966, -9935, -9863, -9750, -9592, -9389, -9139, -8840, -8492, -8092, -7640, -7134,
    6574, 5959, 5288, 4561, 3776, 2935, 2037, 1082, 70, -998, -2122, -3300, -4533, -5818, -7154, -8540,
    -9975, -11455, -12980, -14548, -16155, -17799, -19478, -21189, -22929, -24694, -26482, -28289, -30112, -31947, -33791, -35640,
    -37489, -39336, -41176, -43006, -44821, -46617, -48390, -50137, -51853, -53534, -55178, -56778, -58333, -59838, -61289, -62684,
    -64019, -65290, -66494, -67629, -68692, -69679, -70590, -71420, -72169, -72835, -73415, -73908, -74313, -74630, -74856, -74992,
    75038, 74992, 74856, 74630, 74313, 73908, 73415, 72835, 72169, 71420, 70590, 69679, 68692, 67629, 66494, 65290,
    64019, 62684, 61289, 59838, 58333, 56778, 55178, 53534, 51853, 50137, 48390, 46617, 44821, 43006, 41176, 39336,
    37489, 35640, 33791, 31947, 30112, 28289, 26482, 24694, 22929, 21189, 19478, 17799, 16155, 14548, 12980, 11455,
    9975, 8540, 7154, 5818, 4533, 3300, 2122, 998, -70, -1082, -2037, -2935, -3776, -4561, -5288, -5959,
    6574, 7134, 7640, 8092, 8492, 8840, 9139, 9389, 9592, 9750, 9863, 9935, 9966, 9959, 9916, 9838,
    9727, 9585, 9416, 9219, 8998, 8755, 8491, 8209, 7910, 7597, 7271, 6935, 6589, 6237, 5879, 5517,
    5153, 4788, 4425, 4063, 3705, 3351, 3004, 2663, 2330, 2006, 1692, 1388, 1095, 814, 545, 288,
    45, -185, -402, -605, -794, -970, -1131, -1280, -1414, -1535, -1644, -1739, -1822, -1893, -1952, -2000,
    2037, 2063, 2080, 2087, 2085, 2075, 2057, 2032, 2001, 1962, 1919, 1870, 1817, 1759, 1698, 1634,
    1567, 1498, 1428, 1356, 1283, 1210, 1137, 1064, 991, 919, 848, 779, 711, 645, 581, 519,
    459, 401, 347, 294, 244, 197, 153, 111, 72, 36, 2, -29, -57, -83, -106, -127,
    -146, -163, -177, -189, -200, -208, -215, -221, -224, -227, -228, -228, -227, -225, -222, -218,
    213, 208, 202, 196, 190, 183, 176, 169, 161, 154, 147, 139, 132, 125, 117, 111,
    104, 97, 91, 85, 79, 73, 68, 63, 58, 53, 49, 45, 41, 38, 35, 31,
    29, 26, 24, 21, 19, 17, 16, 14, 13, 11, 10, 9, 8, 7, 7, 6,
    5, 5, 4, 4, 3, 3, 2, 2, 2, 2, 1, 1, 1, 1, 1, 1,
};

struct HuffTab { int n; const uint16_t* code; const uint8_t* len; int ylen; };
// by table number of the standard (0, 4, 14 do not exist: 0 = all zeros)
const HuffTab HT[34] = {
    {0, nullptr, nullptr, 0}, {4, HB1, HL1, 2}, {9, HB2, HL2, 3}, {9, HB3, HL3, 3}, {0, nullptr, nullptr, 0}, {16, HB5, HL5, 4}, {16, HB6, HL6, 4},
    {36, HB7, HL7, 6}, {36, HB8, HL8, 6}, {36, HB9, HL9, 6}, {64, HB10, HL10, 8}, {64, HB11, HL11, 8}, {64, HB12, HL12, 8}, {256, HB13, HL13, 16},
    {0, nullptr, nullptr, 0}, {256, HB15, HL15, 16},
    {256, HB16, HL16, 16}, {256, HB16, HL16, 16}, {256, HB16, HL16, 16}, {256, HB16, HL16, 16}, {256, HB16, HL16, 16}, {256, HB16, HL16, 16},
    {256, HB16, HL16, 16}, {256, HB16, HL16, 16},
    {256, HB24, HL24, 16}, {256, HB24, HL24, 16}, {256, HB24, HL24, 16}, {256, HB24, HL24, 16}, {256, HB24, HL24, 16}, {256, HB24, HL24, 16},
    {256, HB24, HL24, 16}, {256, HB24, HL24, 16},
    {16, HB32, HL32, 0}, {16, HB33, HL33, 0}};
const int LINBITS[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3, 4, 6, 8, 10, 13, 4, 5, 6, 7, 8, 9, 11, 13};

// decoding form of a table: the first FAST bits index `fast` (symbol | len << 12, or 0xffff = longer than FAST bits), longer codes are
// found by a scan of the (few) long codewords
constexpr int FAST = 10;
struct HuffDec {
  std::vector<uint16_t> fast;
  std::vector<uint32_t> slow_code;   // code << (32 - len)
  std::vector<uint16_t> slow_sym;    // symbol | len << 12... len up to 19 does not fit 4 bits: kept separately
  std::vector<uint8_t> slow_len;
  bool built = false;
};
HuffDec g_dec[34];
void build_dec(int t) {
  HuffDec& d = g_dec[t];
  const HuffTab& h = HT[t];
  d.fast.assign(1u << FAST, 0xffff);
  for (int s = 0; s < h.n; ++s) {
    const int len = h.len[s];
    const unsigned code = h.code[s];
    if (len <= FAST) {
      const unsigned base = code << (FAST - len);
      for (unsigned k = 0; k < (1u << (FAST - len)); ++k) d.fast[base + k] = (uint16_t)(s | (len << 8 << 4));      // s < 256: bits 0-7; len bits 12-15
    } else {
      d.slow_code.push_back(code << (32 - len));
      d.slow_sym.push_back((uint16_t)s);
      d.slow_len.push_back((uint8_t)len);
    }
  }
  d.built = true;
}
struct Tables {
  double pow43[8207];                 // |is|^(4/3), is up to 15 + 2^13 - 1
  double cs[8], ca[8];
  double win[4][36];                  // IMDCT windows by block type
  double cos36[36][18], cos12[12][6];
  double dct32[32][32];               // cos((2k+1) m pi / 64)
  float dwin[512];
  double is_ratio[7][2];              // MPEG-1 intensity: left, right factors for is_pos 0..6
  Tables() {
    for (int i = 0; i < 8207; ++i) pow43[i] = std::pow((double)i, 4.0 / 3.0);
    const double c[8] = {-0.6, -0.535, -0.33, -0.185, -0.095, -0.041, -0.0142, -0.0037};
    for (int i = 0; i < 8; ++i) { const double s = std::sqrt(1.0 + c[i] * c[i]); cs[i] = 1.0 / s; ca[i] = c[i] / s; }
    const double PI = 3.14159265358979323846;
    for (int i = 0; i < 36; ++i) {
      win[0][i] = std::sin(PI / 36 * (i + 0.5));
      win[1][i] = i < 18 ? std::sin(PI / 36 * (i + 0.5)) : (i < 24 ? 1.0 : (i < 30 ? std::sin(PI / 12 * (i - 18 + 0.5)) : 0.0));
      win[3][i] = i < 6 ? 0.0 : (i < 12 ? std::sin(PI / 12 * (i - 6 + 0.5)) : (i < 18 ? 1.0 : std::sin(PI / 36 * (i + 0.5))));
      win[2][i] = i < 12 ? std::sin(PI / 12 * (i + 0.5)) : 0.0;
    }
    for (int i = 0; i < 36; ++i) for (int k = 0; k < 18; ++k) cos36[i][k] = std::cos(PI / 72 * (2 * i + 1 + 18) * (2 * k + 1));
    for (int i = 0; i < 12; ++i) for (int k = 0; k < 6; ++k) cos12[i][k] = std::cos(PI / 24 * (2 * i + 1 + 6) * (2 * k + 1));
    for (int m = 0; m < 32; ++m) for (int k = 0; k < 32; ++k) dct32[m][k] = std::cos(PI / 64 * (2 * k + 1) * m);
    for (int i = 0; i < 512; ++i) dwin[i] = (float)(DWIN[i] / 65536.0);
    for (int p = 0; p < 7; ++p) {
      if (p == 6) { is_ratio[p][0] = 1.0; is_ratio[p][1] = 0.0; continue; }      // tan(pi/2): everything to the left channel
      const double r = std::tan(p * PI / 12);
      is_ratio[p][0] = r / (1 + r); is_ratio[p][1] = 1 / (1 + r);
    }
    for (int t = 1; t < 34; ++t) if (HT[t].n) build_dec(t);
  }
};
const Tables& tables() { static const Tables t; return t; }

// scale factor band boundaries: long (23 entries) and short (14), by sampling frequency index 0-8 (44.1, 48, 32 | 22.05, 24, 16 | 11.025, 12, 8)
const int SFB_L[9][23] = {
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 52, 62, 74, 90, 110, 134, 162, 196, 238, 288, 342, 418, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 42, 50, 60, 72, 88, 106, 128, 156, 190, 230, 276, 330, 384, 576},
    {0, 4, 8, 12, 16, 20, 24, 30, 36, 44, 54, 66, 82, 102, 126, 156, 194, 240, 296, 364, 448, 550, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 114, 136, 162, 194, 232, 278, 332, 394, 464, 540, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 6, 12, 18, 24, 30, 36, 44, 54, 66, 80, 96, 116, 140, 168, 200, 238, 284, 336, 396, 464, 522, 576},
    {0, 12, 24, 36, 48, 60, 72, 88, 108, 132, 160, 192, 232, 280, 336, 400, 476, 566, 568, 570, 572, 574, 576}};
const int SFB_S[9][14] = {
    {0, 4, 8, 12, 16, 22, 30, 40, 52, 66, 84, 106, 136, 192}, {0, 4, 8, 12, 16, 22, 28, 38, 50, 64, 80, 100, 126, 192},
    {0, 4, 8, 12, 16, 22, 30, 42, 58, 78, 104, 138, 180, 192}, {0, 4, 8, 12, 18, 24, 32, 42, 56, 74, 100, 132, 174, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 136, 180, 192}, {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
    {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192}, {0, 4, 8, 12, 18, 26, 36, 48, 62, 80, 104, 134, 174, 192},
    {0, 8, 16, 24, 36, 52, 72, 96, 124, 160, 162, 164, 166, 192}};
const int PRETAB[22] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 3, 3, 3, 2, 0};
const int SLEN1[16] = {0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4};
const int SLEN2[16] = {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3};
// ISO 13818-3 table of scale factor partitions: [blocknumber][blocktypenumber][partition]
const int NR_OF_SFB[6][3][4] = {{{6, 5, 5, 5}, {9, 9, 9, 9}, {6, 9, 9, 9}},   {{6, 5, 7, 3}, {9, 9, 12, 6}, {6, 9, 12, 6}},
                                {{11, 10, 0, 0}, {18, 18, 0, 0}, {15, 18, 0, 0}}, {{7, 7, 7, 0}, {12, 12, 12, 0}, {6, 15, 12, 0}},
                                {{6, 6, 6, 3}, {12, 9, 9, 6}, {6, 12, 9, 6}},     {{8, 8, 5, 0}, {15, 12, 9, 0}, {6, 18, 9, 0}}};

// ---------------------------------------------------------------------------------------------- bits (MSB first)
struct BitReader {
  const uint8_t* d;
  size_t nbits, pos;
  BitReader(const uint8_t* p, size_t bytes) : d(p), nbits(bytes * 8), pos(0) {}
  unsigned peek(int n) const {         // up to 24 bits; bits past the end read as zero
    uint32_t v = 0;
    const size_t byte = pos >> 3;
    for (int k = 0; k < 4; ++k) v = (v << 8) | ((byte + k) * 8 < nbits ? d[byte + k] : 0);
    return (v << (pos & 7)) >> (32 - n);
  }
  unsigned get(int n) {
    if (n == 0) return 0;
    unsigned v;
    if (n <= 24) v = peek(n);
    else { v = peek(n - 16) << 16; pos += n - 16; v |= peek(16); pos -= n - 16; }
    pos += n;
    return v;
  }
};

// ---------------------------------------------------------------------------------------------- frame header
struct Header {
  int version;        // 1, 2, 3 (= 2.5)
  int layer;          // 1, 2, 3
  bool crc;
  int bitrate;        // kbit/s
  int sr, sr_index;   // Hz; 0-8 as SFB_* are indexed
  int padding, mode, mode_ext, channels;
  int frame_bytes, side_bytes, spf;
  bool lsf() const { return version != 1; }
};
bool parse_header(const uint8_t* p, Header* h) {
  const uint32_t v = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
  if ((v >> 21) != 0x7ff) return false;
  const int vb = (v >> 19) & 3, lb = (v >> 17) & 3, bi = (v >> 12) & 15, si = (v >> 10) & 3;
  if (vb == 1 || lb == 0 || bi == 15 || si == 3) return false;
  h->version = vb == 3 ? 1 : (vb == 2 ? 2 : 3);
  h->layer = 4 - lb;
  h->crc = ((v >> 16) & 1) == 0;
  static const int BR1[15] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320};
  static const int BR2[15] = {0, 8, 16, 24, 32, 40, 48, 56, 64, 80, 96, 112, 128, 144, 160};
  static const int SR[3] = {44100, 48000, 32000};
  h->sr = SR[si] >> (h->version - 1);
  h->sr_index = (h->version - 1) * 3 + si;
  h->padding = (v >> 9) & 1;
  h->mode = (v >> 6) & 3;
  h->mode_ext = (v >> 4) & 3;
  h->channels = h->mode == 3 ? 1 : 2;
  h->bitrate = 0; h->frame_bytes = 0; h->side_bytes = 0; h->spf = 0;
  if (h->layer != 3) return true;               // recognised, refused by the caller
  h->bitrate = (h->version == 1 ? BR1 : BR2)[bi];
  h->spf = h->version == 1 ? 1152 : 576;
  h->side_bytes = h->version == 1 ? (h->channels == 1 ? 17 : 32) : (h->channels == 1 ? 9 : 17);
  if (bi == 0) return true;                     // free format: frame_bytes stays 0, refused by the caller
  h->frame_bytes = (h->version == 1 ? 144 : 72) * h->bitrate * 1000 / h->sr + h->padding;
  return true;
}
bool same_stream(const Header& a, const Header& b) { return a.version == b.version && a.layer == b.layer && a.sr == b.sr && a.channels == b.channels; }

size_t skip_id3v2(const uint8_t* d, size_t n) {
  size_t off = 0;
  while (n - off >= 10 && d[off] == 'I' && d[off + 1] == 'D' && d[off + 2] == '3') {
    const size_t sz = ((size_t)(d[off + 6] & 0x7f) << 21) | ((size_t)(d[off + 7] & 0x7f) << 14) | ((size_t)(d[off + 8] & 0x7f) << 7) | (d[off + 9] & 0x7f);
    off += 10 + sz + ((d[off + 5] & 0x10) ? 10 : 0);
    if (off > n) fail(E_DATA, "ID3v2 tag runs past the end of the file");
  }
  return off;
}

// first frame: the position where a Layer III header is followed by another header of the same stream (or the file ends)
size_t find_first_frame(const uint8_t* d, size_t n, Header* h) {
  size_t off = skip_id3v2(d, n);
  const size_t limit = std::min(n, off + (size_t)(1 << 16));       // garbage tolerated in front of the first frame
  for (; off + 4 <= limit; ++off) {
    if (d[off] != 0xff || (d[off + 1] & 0xe0) != 0xe0) continue;
    Header a;
    if (!parse_header(d + off, &a)) continue;
    if (a.layer != 3) fail(E_UNSUPPORTED, "MPEG audio Layer " + std::string(a.layer == 1 ? "I" : "II") + " is not decoded here (Layer III only)");
    if (a.frame_bytes == 0) fail(E_UNSUPPORTED, "free-format bit rate is not decoded here");
    const size_t next = off + (size_t)a.frame_bytes;
    Header b;
    if (next + 4 <= n) {
      if (!parse_header(d + next, &b) || !same_stream(a, b)) {
        // a Xing / Info frame may differ from the audio frames in nothing that same_stream() compares; anything else is not a sync
        continue;
      }
    } else if (next > n) {
      continue;
    }
    *h = a;
    return off;
  }
  fail(E_DATA, "no MPEG audio frame found");
}

// Xing / Info / VBRI frame?  With a LAME-style tag: encoder delay and padding (FFmpeg's mp3 demuxer reads the same fields)
struct InfoFrame { bool present = false; bool lame = false; int64_t frames = 0; int delay = 0, padding = 0; };
InfoFrame parse_info_frame(const uint8_t* f, const Header& h) {
  InfoFrame r;
  const size_t x = 4 + (size_t)h.side_bytes;         // (the tag frame carries no CRC in practice; FFmpeg looks here as well)
  if ((size_t)h.frame_bytes >= x + 8 && (std::memcmp(f + x, "Xing", 4) == 0 || std::memcmp(f + x, "Info", 4) == 0)) {
    r.present = true;
    const uint32_t flags = ((uint32_t)f[x + 4] << 24) | ((uint32_t)f[x + 5] << 16) | ((uint32_t)f[x + 6] << 8) | f[x + 7];
    size_t p = x + 8;
    if ((flags & 1) && p + 4 <= (size_t)h.frame_bytes) { r.frames = ((int64_t)f[p] << 24) | ((int64_t)f[p + 1] << 16) | ((int64_t)f[p + 2] << 8) | f[p + 3]; p += 4; }
    if (flags & 2) p += 4;
    if (flags & 4) p += 100;
    if (flags & 8) p += 4;
    // 9 bytes encoder version, 1 revision / VBR method, 1 lowpass, 8 replay gain, 1 flags, 1 bit rate, then 3 bytes: delay << 12 | padding
    if (p + 24 <= (size_t)h.frame_bytes && (std::memcmp(f + p, "LAME", 4) == 0 || std::memcmp(f + p, "Lavf", 4) == 0 || std::memcmp(f + p, "Lavc", 4) == 0)) {
      const uint32_t v = ((uint32_t)f[p + 21] << 16) | ((uint32_t)f[p + 22] << 8) | f[p + 23];
      r.lame = true;
      r.delay = (int)(v >> 12);
      r.padding = (int)(v & 4095);
    }
    return r;
  }
  if ((size_t)h.frame_bytes >= 4 + 32 + 4 && std::memcmp(f + 4 + 32, "VBRI", 4) == 0) r.present = true;
  return r;
}

// ---------------------------------------------------------------------------------------------- side information
struct Granule {
  int part2_3_length, big_values, global_gain, scalefac_compress;
  int window_switching, block_type, mixed;
  int table_select[3], subblock_gain[3];
  int region0_count, region1_count;
  int preflag, scalefac_scale, count1table;
  int region1_start, region2_start;
};
struct SideInfo {
  int main_data_begin;
  int scfsi[2][4];
  Granule gr[2][2];
};

void read_side_info(BitReader& br, const Header& h, SideInfo* si) {
  const int nch = h.channels;
  if (!h.lsf()) {
    si->main_data_begin = br.get(9);
    br.get(nch == 1 ? 5 : 3);
    for (int ch = 0; ch < nch; ++ch) for (int b = 0; b < 4; ++b) si->scfsi[ch][b] = br.get(1);
  } else {
    si->main_data_begin = br.get(8);
    br.get(nch == 1 ? 1 : 2);
    for (int ch = 0; ch < 2; ++ch) for (int b = 0; b < 4; ++b) si->scfsi[ch][b] = 0;
  }
  const int ngr = h.lsf() ? 1 : 2;
  for (int g = 0; g < ngr; ++g)
    for (int ch = 0; ch < nch; ++ch) {
      Granule& G = si->gr[g][ch];
      G.part2_3_length = br.get(12);
      G.big_values = br.get(9);
      G.global_gain = br.get(8);
      G.scalefac_compress = br.get(h.lsf() ? 9 : 4);
      G.window_switching = br.get(1);
      G.mixed = 0;
      if (G.window_switching) {
        G.block_type = br.get(2);
        G.mixed = br.get(1);
        G.table_select[0] = br.get(5); G.table_select[1] = br.get(5); G.table_select[2] = 0;
        for (int w = 0; w < 3; ++w) G.subblock_gain[w] = br.get(3);
        if (G.block_type == 0) fail(E_DATA, "window switching with block type 0 (forbidden)");
        G.region0_count = (G.block_type == 2 && !G.mixed) ? 8 : 7;
        G.region1_count = 20 - G.region0_count;
      } else {
        G.block_type = 0;
        for (int r = 0; r < 3; ++r) G.table_select[r] = br.get(5);
        G.region0_count = br.get(4);
        G.region1_count = br.get(3);
        for (int w = 0; w < 3; ++w) G.subblock_gain[w] = 0;
      }
      G.preflag = h.lsf() ? 0 : br.get(1);
      G.scalefac_scale = br.get(1);
      G.count1table = br.get(1);
      if (G.big_values > 288) fail(E_DATA, "big_values > 288");
      for (int r = 0; r < 3; ++r) if (G.table_select[r] == 4 || G.table_select[r] == 14) fail(E_DATA, "Huffman table 4 / 14 selected (do not exist)");
      // region boundaries (in lines).  With window switching the counts are implicit: 36 lines of region 0 for MPEG-1 and for
      // every short block; LSF start / stop blocks take the first 8 long bands (54 lines); MPEG-2.5 counts bands of its own table
      const int* L = SFB_L[h.sr_index];
      if (G.window_switching) {
        if (G.mixed && h.sr_index == 8) fail(E_UNSUPPORTED, "mixed blocks at 8 kHz are not decoded here");
        if (h.version == 3) G.region1_start = L[((G.block_type == 2 && !G.mixed) ? 5 : 7) + 1];
        else G.region1_start = (h.version == 1 || G.block_type == 2) ? 36 : 54;
        G.region2_start = 576;
      } else {
        G.region1_start = L[std::min(G.region0_count + 1, 22)];
        G.region2_start = L[std::min(G.region0_count + G.region1_count + 2, 22)];
      }
    }
}

// ---------------------------------------------------------------------------------------------- scale factors
struct ScaleFac {
  int l[23];          // long bands 0..21 (+ guard)
  int s[3][13];       // short bands 0..12 per window
  int is_max_l[23], is_max_s[3][13];   // LSF intensity: the "illegal" position of each band (= no intensity), 0x7fffffff where any value is legal
};

void scalefactors_mpeg1(BitReader& br, const Granule& G, const int scfsi[4], const ScaleFac* prev, int gr, ScaleFac* sf) {
  const int s1 = SLEN1[G.scalefac_compress], s2 = SLEN2[G.scalefac_compress];
  std::memset(sf, 0, sizeof(*sf));
  if (G.window_switching && G.block_type == 2) {
    if (G.mixed) {
      for (int b = 0; b < 8; ++b) sf->l[b] = br.get(s1);
      for (int b = 3; b < 6; ++b) for (int w = 0; w < 3; ++w) sf->s[w][b] = br.get(s1);
      for (int b = 6; b < 12; ++b) for (int w = 0; w < 3; ++w) sf->s[w][b] = br.get(s2);
    } else {
      for (int b = 0; b < 6; ++b) for (int w = 0; w < 3; ++w) sf->s[w][b] = br.get(s1);
      for (int b = 6; b < 12; ++b) for (int w = 0; w < 3; ++w) sf->s[w][b] = br.get(s2);
    }
  } else {
    static const int B0[5] = {0, 6, 11, 16, 21};
    for (int grp = 0; grp < 4; ++grp)
      for (int b = B0[grp]; b < B0[grp + 1]; ++b) {
        if (gr == 1 && scfsi[grp]) sf->l[b] = prev->l[b];
        else sf->l[b] = br.get(grp < 2 ? s1 : s2);
      }
  }
}

// ISO 13818-3 2.4.3.2; returns the preflag the standard derives from scalefac_compress
int scalefactors_lsf(BitReader& br, const Granule& G, bool intensity_right, ScaleFac* sf) {
  std::memset(sf, 0, sizeof(*sf));
  for (int b = 0; b < 23; ++b) sf->is_max_l[b] = 0x7fffffff;
  for (int w = 0; w < 3; ++w) for (int b = 0; b < 13; ++b) sf->is_max_s[w][b] = 0x7fffffff;
  int slen[4], blocknumber, preflag = 0;
  int sfc = G.scalefac_compress;
  if (!intensity_right) {
    if (sfc < 400) { slen[0] = (sfc >> 4) / 5; slen[1] = (sfc >> 4) % 5; slen[2] = (sfc % 16) >> 2; slen[3] = sfc % 4; blocknumber = 0; }
    else if (sfc < 500) { sfc -= 400; slen[0] = (sfc >> 2) / 5; slen[1] = (sfc >> 2) % 5; slen[2] = sfc % 4; slen[3] = 0; blocknumber = 1; }
    else { sfc -= 500; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; blocknumber = 2; preflag = 1; }
  } else {
    sfc >>= 1;
    if (sfc < 180) { slen[0] = sfc / 36; slen[1] = (sfc % 36) / 6; slen[2] = (sfc % 36) % 6; slen[3] = 0; blocknumber = 3; }
    else if (sfc < 244) { sfc -= 180; slen[0] = (sfc % 64) >> 4; slen[1] = (sfc % 16) >> 2; slen[2] = sfc % 4; slen[3] = 0; blocknumber = 4; }
    else { sfc -= 244; slen[0] = sfc / 3; slen[1] = sfc % 3; slen[2] = 0; slen[3] = 0; blocknumber = 5; }
  }
  const int btn = (G.window_switching && G.block_type == 2) ? (G.mixed ? 2 : 1) : 0;
  int vals[40], maxv[40], k = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < NR_OF_SFB[blocknumber][btn][i]; ++j) {
      vals[k] = slen[i] ? (int)br.get(slen[i]) : 0;
      maxv[k] = intensity_right ? (1 << slen[i]) - 1 : 0x7fffffff;
      ++k;
    }
  while (k < 40) { vals[k] = 0; maxv[k] = 0x7fffffff; ++k; }
  int q = 0;
  if (btn == 0) {
    for (int b = 0; b < 21; ++b) { sf->l[b] = vals[q]; sf->is_max_l[b] = maxv[q]; ++q; }
  } else if (btn == 1) {
    for (int b = 0; b < 12; ++b) for (int w = 0; w < 3; ++w) { sf->s[w][b] = vals[q]; sf->is_max_s[w][b] = maxv[q]; ++q; }
  } else {
    for (int b = 0; b < 6; ++b) { sf->l[b] = vals[q]; sf->is_max_l[b] = maxv[q]; ++q; }
    for (int b = 3; b < 12; ++b) for (int w = 0; w < 3; ++w) { sf->s[w][b] = vals[q]; sf->is_max_s[w][b] = maxv[q]; ++q; }
  }
  return preflag;
}

// ---------------------------------------------------------------------------------------------- Huffman-coded spectrum
inline int huff_symbol(BitReader& br, int t) {
  const HuffDec& d = g_dec[t];
  const unsigned p = br.peek(FAST);
  const uint16_t e = d.fast[p];
  if (e != 0xffff) { br.pos += e >> 12; return e & 0xff; }
  const uint32_t w = (br.peek(24) << 8);
  for (size_t i = 0; i < d.slow_code.size(); ++i) {
    const int len = d.slow_len[i];
    if (((w ^ d.slow_code[i]) >> (32 - len)) == 0) { br.pos += len; return d.slow_sym[i]; }
  }
  fail(E_DATA, "invalid Huffman code word");
}

// -> number of lines decoded up to the last non-zero region (count of spectral lines with defined values); is[576]
void huffman_decode(BitReader& br, const Granule& G, size_t end_bit, int is[576], Stats* st) {
  std::memset(is, 0, 576 * sizeof(int));
  int i = 0;
  const int bv = G.big_values * 2;
  const int bounds[3] = {std::min(G.region1_start, bv), std::min(G.region2_start, bv), bv};
  for (int r = 0; r < 3; ++r) {
    const int t = G.table_select[r];
    const int stop = bounds[r];
    if (t == 0) { i = std::max(i, stop); continue; }
    const int lb = LINBITS[t], ylen = HT[t].ylen;
    while (i < stop) {
      if (br.pos >= end_bit) fail(E_DATA, "Huffman data of a granule runs past part2_3_length inside big_values");
      const int s = huff_symbol(br, t);
      int x = s / ylen, y = s % ylen;
      if (lb && x == 15) x += br.get(lb);
      if (x && br.get(1)) x = -x;
      if (lb && y == 15) y += br.get(lb);
      if (y && br.get(1)) y = -y;
      is[i++] = x; is[i++] = y;
    }
  }
  // count1: quadruples of -1 / 0 / +1 until the granule's bits are used up
  const int t1 = 32 + G.count1table;
  bool over = false;
  while (br.pos < end_bit && i <= 572) {
    const size_t before = br.pos;
    const int s = huff_symbol(br, t1);
    int v[4] = {(s >> 3) & 1, (s >> 2) & 1, (s >> 1) & 1, s & 1};
    for (int k = 0; k < 4; ++k) if (v[k] && br.get(1)) v[k] = -1;
    if (br.pos > end_bit) { br.pos = before; over = true; break; }      // the quadruple started inside the stuffing: not data
    for (int k = 0; k < 4; ++k) is[i++] = v[k];
  }
  if (st) {
    ++st->granules;
    if (over) ++st->huff_overrun;
    else if (br.pos == end_bit) ++st->huff_exact;
    else ++st->huff_short;
  }
  br.pos = end_bit;
}

// ---------------------------------------------------------------------------------------------- requantisation
void requantize(const Header& h, const Granule& G, const ScaleFac& sf, const int is[576], float xr[576]) {
  const Tables& T = tables();
  const int* L = SFB_L[h.sr_index];
  const int* S = SFB_S[h.sr_index];
  const double mult = G.scalefac_scale ? 1.0 : 0.5;
  const double gg = std::pow(2.0, 0.25 * (G.global_gain - 210));
  auto val = [&](int q, double scale) -> float {
    const int a = q < 0 ? -q : q;
    const double m = T.pow43[a] * scale;
    return (float)(q < 0 ? -m : m);
  };
  int i = 0;
  const bool shortb = G.window_switching && G.block_type == 2;
  const int long_end = shortb ? (G.mixed ? (h.lsf() ? L[6] : L[8]) : 0) : 576;      // mixed: 36 lines of long bands (8 bands MPEG-1, 6 LSF)
  for (int b = 0; i < long_end; ++b) {
    const int stop = std::min(L[b + 1], long_end);
    const double scale = gg * std::pow(2.0, -mult * (sf.l[b] + (G.preflag ? PRETAB[b] : 0)));
    for (; i < stop; ++i) xr[i] = val(is[i], scale);
  }
  if (shortb) {
    for (int b = G.mixed ? 3 : 0; b < 13; ++b) {
      const int width = S[b + 1] - S[b];
      for (int w = 0; w < 3; ++w) {
        const double scale = gg * std::pow(2.0, -2.0 * G.subblock_gain[w]) * std::pow(2.0, -mult * (b < 12 ? sf.s[w][b] : 0));
        const int start = 3 * S[b] + w * width;
        for (int k = 0; k < width; ++k) xr[start + k] = val(is[start + k], scale);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- stereo
// Bands of the granule in the order they are stored (before the short-block reordering): {first line, lines, band, window or -1}
struct Band { int start, width, sfb, win; };
int list_bands(const Header& h, const Granule& G, Band out[64]) {
  const int* L = SFB_L[h.sr_index];
  const int* S = SFB_S[h.sr_index];
  int n = 0;
  const bool shortb = G.window_switching && G.block_type == 2;
  if (!shortb) { for (int b = 0; b < 22; ++b) out[n++] = Band{L[b], L[b + 1] - L[b], b, -1}; return n; }
  if (G.mixed) for (int b = 0; b < (h.lsf() ? 6 : 8); ++b) out[n++] = Band{L[b], L[b + 1] - L[b], b, -1};
  for (int b = G.mixed ? 3 : 0; b < 13; ++b) {
    const int width = S[b + 1] - S[b];
    for (int w = 0; w < 3; ++w) out[n++] = Band{3 * S[b] + w * width, width, b, w};
  }
  return n;
}

void stereo(const Header& h, const Granule& G1 /* right channel's granule */, const ScaleFac& sf1, float xr[2][576], Stats* st) {
  const bool ms = (h.mode_ext & 2) != 0, is_on = (h.mode_ext & 1) != 0;
  if (h.mode != 1 || (!ms && !is_on)) return;
  const Tables& T = tables();
  Band bands[64];
  const int nb = list_bands(h, G1, bands);
  // intensity: a band is intensity-coded when it and every higher band (of its window, for short blocks) are zero in the right
  // channel; in a mixed block the long bands only once all three windows are
  std::vector<char> isb(nb, 0);
  if (is_on) {
    bool found_w[3] = {false, false, false}, found_l = false;
    for (int k = nb - 1; k >= 0; --k) {
      const Band& B = bands[k];
      bool nz = false;
      for (int j = 0; j < B.width && !nz; ++j) nz = xr[1][B.start + j] != 0.f;
      if (B.win >= 0) {
        if (nz) found_w[B.win] = true;
        isb[k] = !found_w[B.win];
      } else {
        if (G1.window_switching && G1.block_type == 2 && (found_w[0] || found_w[1] || found_w[2])) found_l = true;
        if (nz) found_l = true;
        isb[k] = !found_l;
      }
    }
    if (st) ++st->intensity_granules;
  }
  if (ms && st) ++st->ms_granules;
  const float r2 = 0.70710678118654752440f;
  for (int k = 0; k < nb; ++k) {
    const Band& B = bands[k];
    bool done = false;
    if (isb[k]) {
      // position: the right channel's scale factor of the band; the last band (21 long / 12 short) takes the one before it
      int pos, illegal;
      if (B.win >= 0) { const int b = std::min(B.sfb, 11); pos = sf1.s[B.win][b]; illegal = h.lsf() ? sf1.is_max_s[B.win][b] : 7; }
      else { const int b = std::min(B.sfb, 20); pos = sf1.l[b]; illegal = h.lsf() ? sf1.is_max_l[b] : 7; }
      if (pos != illegal && (h.lsf() || pos < 7)) {
        float kl, kr;
        if (!h.lsf()) { kl = (float)T.is_ratio[pos][0]; kr = (float)T.is_ratio[pos][1]; }
        else {
          const double io = (G1.scalefac_compress & 1) ? 0.70710678118654752440 : 0.84089641525371454303;     // 2^-1/2, 2^-1/4
          if (pos == 0) { kl = 1.f; kr = 1.f; }
          else if (pos & 1) { kl = (float)std::pow(io, (pos + 1) / 2); kr = 1.f; }
          else { kl = 1.f; kr = (float)std::pow(io, pos / 2); }
        }
        for (int j = 0; j < B.width; ++j) { const float x = xr[0][B.start + j]; xr[0][B.start + j] = x * kl; xr[1][B.start + j] = x * kr; }
        done = true;
      }
    }
    if (!done && ms)
      for (int j = 0; j < B.width; ++j) {
        const float m = xr[0][B.start + j], s = xr[1][B.start + j];
        xr[0][B.start + j] = (m + s) * r2; xr[1][B.start + j] = (m - s) * r2;
      }
  }
}

// ---------------------------------------------------------------------------------------------- reorder, alias reduction, IMDCT
void reorder_short(const Header& h, const Granule& G, float xr[576]) {
  const int* S = SFB_S[h.sr_index];
  float tmp[576];
  std::memcpy(tmp, xr, sizeof(tmp));
  for (int b = G.mixed ? 3 : 0; b < 13; ++b) {
    const int width = S[b + 1] - S[b], base = 3 * S[b];
    for (int w = 0; w < 3; ++w)
      for (int k = 0; k < width; ++k) xr[base + 3 * k + w] = tmp[base + w * width + k];
  }
}

}  // namespace

void hybrid_granule(float xr[576], float overlap[576], int block_type, int mixed, int long_sb_mixed, float out[576]) {
  const Tables& T = tables();
  // alias reduction between neighbouring subbands of long blocks (mixed: the long subbands only)
  const int sb_alias = block_type == 2 ? (mixed ? long_sb_mixed : 0) : 32;
  for (int sb = 1; sb < sb_alias; ++sb)
    for (int i = 0; i < 8; ++i) {
      const float a = xr[18 * sb - 1 - i], b = xr[18 * sb + i];
      xr[18 * sb - 1 - i] = (float)(a * T.cs[i] - b * T.ca[i]);
      xr[18 * sb + i] = (float)(b * T.cs[i] + a * T.ca[i]);
    }
  for (int sb = 0; sb < 32; ++sb) {
    const float* X = xr + 18 * sb;
    float* ov = overlap + 18 * sb;
    double y[36];
    int bt = block_type;
    if (mixed && sb < long_sb_mixed) bt = 0;       // mixed_block_flag: the two lowest subbands take the NORMAL window whatever the block type
                                                   // (a start / stop granule next to a mixed short one carries the flag too: that keeps them TDAC-clean)
    bool zero = true;
    for (int k = 0; k < 18; ++k) zero = zero && X[k] == 0.f;
    if (zero) {
      for (int i = 0; i < 36; ++i) y[i] = 0.0;
    } else if (bt != 2) {
      for (int i = 0; i < 36; ++i) {
        double a = 0.0;
        for (int k = 0; k < 18; ++k) a += X[k] * T.cos36[i][k];
        y[i] = a * T.win[bt][i];
      }
    } else {
      for (int i = 0; i < 36; ++i) y[i] = 0.0;
      for (int w = 0; w < 3; ++w)
        for (int i = 0; i < 12; ++i) {
          double a = 0.0;
          for (int k = 0; k < 6; ++k) a += X[3 * k + w] * T.cos12[i][k];
          y[6 + 6 * w + i] += a * T.win[2][i];
        }
    }
    for (int i = 0; i < 18; ++i) {
      float v = (float)(y[i] + ov[i]);
      ov[i] = (float)y[18 + i];
      if ((sb & 1) && (i & 1)) v = -v;         // frequency inversion of the odd subbands
      out[32 * i + sb] = v;
    }
  }
}

void polyphase_granule(const float sbs[576], float vbuf[1024], int* voff, float pcm[576]) {
  const Tables& T = tables();
  for (int t = 0; t < 18; ++t) {
    const float* S = sbs + 32 * t;
    *voff = (*voff - 64) & 1023;
    float* V = vbuf + *voff;
    // matrixing through the 32 distinct cosine sums: V[i] = sum_k cos((16 + i)(2k + 1) pi / 64) S[k]
    double dct[32];
    for (int m = 0; m < 32; ++m) {
      double a = 0.0;
      for (int k = 0; k < 32; ++k) a += T.dct32[m][k] * S[k];
      dct[m] = a;
    }
    for (int i = 0; i < 64; ++i) {
      const int n = 16 + i;
      double v;
      if (n < 32) v = dct[n];
      else if (n == 32) v = 0.0;
      else if (n < 64) v = -dct[64 - n];
      else v = -dct[n - 64];
      // vbuf is a ring of 1024 (16 blocks of 64): no wrap inside a block, voff is a multiple of 64
      V[i] = (float)v;
    }
    float* o = pcm + 32 * t;
    for (int j = 0; j < 32; ++j) {
      double a = 0.0;
      for (int i = 0; i < 8; ++i) {
        // U[64 i + j] = V[128 i + j], U[64 i + 32 + j] = V[128 i + 96 + j]   (V = the last 1024 matrixed values, newest first)
        a += T.dwin[64 * i + j] * vbuf[(*voff + 128 * i + j) & 1023];
        a += T.dwin[64 * i + 32 + j] * vbuf[(*voff + 128 * i + 96 + j) & 1023];
      }
      o[j] = (float)a;
    }
  }
}

int huffman_table(int t, const uint16_t** codes, const uint8_t** lens, int* linbits) {
  if (linbits) for (int i = 0; i < 32; ++i) linbits[i] = LINBITS[i];
  if (t < 0 || t > 33 || HT[t].n == 0) return t >= 0 && t <= 33 ? 0 : -1;
  *codes = HT[t].code; *lens = HT[t].len;
  return HT[t].n;
}
const float* synthesis_window() { return tables().dwin; }

// ---------------------------------------------------------------------------------------------- stream level
namespace {
struct Stream {
  size_t first;          // offset of the first frame
  Header h0;             // its header
  InfoFrame tag;
};
Stream open_stream(const uint8_t* d, size_t n) {
  Stream s;
  s.first = find_first_frame(d, n, &s.h0);
  if (s.first + (size_t)s.h0.frame_bytes <= n) s.tag = parse_info_frame(d + s.first, s.h0);
  return s;
}
// walk the frames from `off`: f(offset, header); stops at the first position that is not a frame of this stream (an ID3v1 / APE
// tag, or the end of the file); a last frame cut short by the end of the file is dropped
template <typename F>
int64_t walk(const uint8_t* d, size_t n, size_t off, const Header& h0, F&& f) {
  int64_t frames = 0;
  while (off + 4 <= n) {
    Header h;
    if (!parse_header(d + off, &h) || h.layer != 3 || h.frame_bytes == 0 || !same_stream(h, h0)) {
      // resynchronise over a few bytes of garbage (never over a tag: "TAG" / "APETAGEX" end the stream)
      if (n - off >= 3 && std::memcmp(d + off, "TAG", 3) == 0) break;
      if (n - off >= 8 && std::memcmp(d + off, "APETAGEX", 8) == 0) break;
      size_t k = off + 1;
      const size_t lim = std::min(n, off + 4096);
      bool found = false;
      for (; k + 4 <= lim; ++k) {
        Header g;
        if (d[k] == 0xff && parse_header(d + k, &g) && g.layer == 3 && g.frame_bytes && same_stream(g, h0)) {
          const size_t nx = k + (size_t)g.frame_bytes;
          Header g2;
          if (nx + 4 > n || (parse_header(d + nx, &g2) && same_stream(g2, h0))) { found = true; break; }
        }
      }
      if (!found) break;
      off = k;
      continue;
    }
    if (off + (size_t)h.frame_bytes > n) break;
    f(off, h);
    ++frames;
    off += (size_t)h.frame_bytes;
  }
  return frames;
}

uint16_t crc16_mpeg(uint16_t crc, const uint8_t* p, size_t nbits) {
  for (size_t i = 0; i < nbits; ++i) {
    const int bit = (p[i >> 3] >> (7 - (i & 7))) & 1;
    const int top = (crc >> 15) & 1;
    crc = (uint16_t)(crc << 1);
    if (bit ^ top) crc ^= 0x8005;
  }
  return crc;
}
}  // namespace

Info probe(const uint8_t* d, size_t n) {
  const Stream s = open_stream(d, n);
  Info in;
  in.version = s.h0.version; in.channels = s.h0.channels; in.sample_rate = s.h0.sr; in.samples_per_frame = s.h0.spf;
  in.has_info_frame = s.tag.present ? 1 : 0;
  size_t off = s.first + (s.tag.present ? (size_t)s.h0.frame_bytes : 0);
  bool first = true;
  in.audio_frames = walk(d, n, off, s.h0, [&](size_t, const Header& h) { if (first) { in.bitrate_kbps = h.bitrate; first = false; } });
  const int64_t total = in.audio_frames * in.samples_per_frame;
  // Encoder delay / padding of a LAME-style tag, trimmed the way FFmpeg's mp3 demuxer + decoder do (libavformat/mp3dec.c: start skip =
  // delay + 528 + 1 samples -- the encoder's delay plus the decoder's own -- and the declared padding, less those 529, off the end).
  // Without such a tag nothing is trimmed.
  in.start_skip = 0;
  int64_t end_trim = 0;
  if (s.tag.lame) {
    in.start_skip = s.tag.delay + 528 + 1;
    end_trim = std::max(0, s.tag.padding - (528 + 1));
  }
  in.samples = std::max<int64_t>(0, total - in.start_skip - end_trim);
  return in;
}

namespace {
struct FrameRef { size_t off; Header h; };

// One decoding thread's state and the work on one frame.  mode 0 = the frame's main data only joins the reservoir (warm-up far
// ahead of a segment), 1 = decode, keep the state, drop the samples (the two frames right in front of a segment), 2 = decode and emit.
struct Worker {
  const uint8_t* d;
  int nch, c0, c1;                                  // channels of the stream; channels [c0, c1) are synthesised
  std::vector<uint8_t> reservoir;                   // main data not yet consumed: the tail of earlier frames
  std::vector<float> overlap, vbuf;
  int voff[2] = {0, 0};
  ScaleFac sf[2][2];
  Stats st;
  Worker(const uint8_t* data, int channels, int first, int last)
      : d(data), nch(channels), c0(first), c1(last), overlap((size_t)2 * 576, 0.f), vbuf((size_t)2 * 1024, 0.f) { std::memset(sf, 0, sizeof(sf)); }

  void frame(const FrameRef& fr, int mode, float pcm[2][1152]) {
    const Header& h = fr.h;
    const uint8_t* f = d + fr.off;
    size_t p = 4;
    if (h.crc) {
      if (mode == 2) {
        uint16_t c = crc16_mpeg(0xffff, f + 2, 16);
        c = crc16_mpeg(c, f + 6, (size_t)h.side_bytes * 8);
        ++st.crc_checked;
        if (c != (uint16_t)((f[4] << 8) | f[5])) ++st.crc_failed;
      }
      p += 2;
    }
    if (p + (size_t)h.side_bytes > (size_t)h.frame_bytes) fail(E_DATA, "frame shorter than its side information");
    const size_t md0 = p + (size_t)h.side_bytes;
    if (mode == 0) {
      reservoir.insert(reservoir.end(), f + md0, f + h.frame_bytes);
      if (reservoir.size() > 4096) reservoir.erase(reservoir.begin(), reservoir.end() - 2048);
      return;
    }
    Stats* sp = mode == 2 ? &st : nullptr;
    BitReader sbr(f + p, (size_t)h.side_bytes);
    SideInfo si;
    std::memset(&si, 0, sizeof(si));
    read_side_info(sbr, h, &si);
    // bit reservoir: this frame's main data starts main_data_begin bytes before the frame's own main-data bytes
    const bool missing = (size_t)si.main_data_begin > reservoir.size();
    if (sp) sp->max_main_data_begin = std::max<int64_t>(sp->max_main_data_begin, si.main_data_begin);
    std::vector<uint8_t> md;
    if (!missing) md.assign(reservoir.end() - si.main_data_begin, reservoir.end());
    md.insert(md.end(), f + md0, f + h.frame_bytes);
    reservoir.insert(reservoir.end(), f + md0, f + h.frame_bytes);
    if (reservoir.size() > 4096) reservoir.erase(reservoir.begin(), reservoir.end() - 2048);     // main_data_begin <= 511
    const int ngr = h.lsf() ? 1 : 2;
    if (missing) {
      // the stream was cut in front of this frame (or starts in the middle of one): nothing to decode it from -- silence, state kept
      if (sp) ++sp->reservoir_missing;
      for (int g = 0; g < ngr; ++g)
        for (int ch = c0; ch < c1; ++ch) {
          float zero[576], sbs[576];
          std::memset(zero, 0, sizeof(zero));
          hybrid_granule(zero, overlap.data() + 576 * ch, 0, 0, 2, sbs);
          polyphase_granule(sbs, vbuf.data() + 1024 * ch, &voff[ch], pcm[ch] + 576 * g);
        }
      return;
    }
    BitReader br(md.data(), md.size());
    for (int g = 0; g < ngr; ++g) {
      float xr[2][576];
      for (int ch = 0; ch < nch; ++ch) {
        Granule& G = si.gr[g][ch];
        const size_t start = br.pos, end = start + (size_t)G.part2_3_length;
        if (end > br.nbits) fail(E_DATA, "part2_3_length runs past the main data");
        if (!h.lsf()) scalefactors_mpeg1(br, G, si.scfsi[ch], &sf[0][ch], g, &sf[g][ch]);
        else G.preflag = scalefactors_lsf(br, G, ch == 1 && h.mode == 1 && (h.mode_ext & 1), &sf[g][ch]);
        if (br.pos > end) fail(E_DATA, "scale factors run past part2_3_length");
        // a channel that is neither asked for nor needed by the joint-stereo processing of the one that is: its bits are skipped
        const bool joint = nch == 2 && h.mode == 1 && (h.mode_ext & 3);
        if (!joint && (ch < c0 || ch >= c1)) { br.pos = end; continue; }
        int is[576];
        huffman_decode(br, G, end, is, sp);
        requantize(h, G, sf[g][ch], is, xr[ch]);
        if (sp && G.window_switching && G.block_type == 2) { ++sp->short_granules; if (G.mixed) ++sp->mixed_granules; }
      }
      if (nch == 2) stereo(h, si.gr[g][1], sf[g][1], xr, sp);
      for (int ch = c0; ch < c1; ++ch) {
        const Granule& G = si.gr[g][ch];
        if (G.window_switching && G.block_type == 2) reorder_short(h, G, xr[ch]);
        float sbs[576];
        hybrid_granule(xr[ch], overlap.data() + 576 * ch, G.block_type, G.mixed, 2, sbs);
        polyphase_granule(sbs, vbuf.data() + 1024 * ch, &voff[ch], pcm[ch] + 576 * g);
      }
    }
  }
};
}  // namespace

// Frames decode in parallel: a frame's samples depend on the stream before it through (a) the bit reservoir -- at most 511 bytes of
// earlier main data --, (b) the previous granule's IMDCT tail and (c) the last 16 time slots of the synthesis filterbank.  A thread
// that takes frames [a, b) therefore first lets the frames that hold >= 1 KiB of main data in front of frame a - 2 fill its
// reservoir, decodes frames a - 2 and a - 1 for their state only, and is then in exactly the state the sequential decoder would be in:
// the result does not depend on the number of threads (tests/test_mp3.py compares them bit for bit).
int64_t decode(const uint8_t* d, size_t n, int channel, float* out, int64_t capacity, Info* info_out, Stats* st, int threads) {
  Info in = probe(d, n);
  if (info_out) *info_out = in;
  if (!out) return in.samples;
  if (channel >= in.channels) fail(E_ARG, "channel index out of range");
  const int c0 = channel < 0 ? 0 : channel, c1 = channel < 0 ? in.channels : channel + 1;
  if ((int64_t)(c1 - c0) * in.samples > capacity) fail(E_ARG, "output buffer too small");
  const Stream s = open_stream(d, n);
  tables();
  std::vector<FrameRef> frames;
  frames.reserve((size_t)in.audio_frames);
  walk(d, n, s.first + (s.tag.present ? (size_t)s.h0.frame_bytes : 0), s.h0, [&](size_t fo, const Header& h) { frames.push_back(FrameRef{fo, h}); });
  const int64_t nf = (int64_t)frames.size();
  const int spf = in.samples_per_frame;
  const int64_t keep0 = in.start_skip, keep1 = in.start_skip + in.samples;
  int nt = threads > 0 ? threads : (int)std::min<int64_t>(16, std::max<int64_t>(1, nf / 400));      // ~10 s of audio per thread at least
  nt = (int)std::max<int64_t>(1, std::min<int64_t>(nt, std::max<int64_t>(nf / 8, 1)));
  if (info_out) info_out->decode_threads = nt;
  std::vector<Stats> stats((size_t)nt);
  std::vector<Error> errors((size_t)nt, Error{0, ""});
  auto run = [&](int t) {
    try {
      const int64_t a = nf * t / nt, b = nf * (t + 1) / nt;
      Worker w(d, in.channels, c0, c1);
      float pcm[2][1152];
      int64_t first = std::max<int64_t>(0, a - 2), bytes = 0;
      while (first > 0 && bytes < 1024) { --first; bytes += frames[(size_t)first].h.frame_bytes - 4 - frames[(size_t)first].h.side_bytes; }
      for (int64_t k = first; k < b; ++k) {
        const int mode = k >= a ? 2 : (k >= a - 2 ? 1 : 0);
        w.frame(frames[(size_t)k], mode, pcm);
        if (mode != 2) continue;
        for (int i = 0; i < spf; ++i) {
          const int64_t at = k * spf + i;
          if (at < keep0 || at >= keep1) continue;
          for (int c = c0; c < c1; ++c) out[(size_t)(c - c0) * (size_t)in.samples + (size_t)(at - keep0)] = pcm[c][i];
        }
      }
      stats[(size_t)t] = w.st;
    } catch (const Error& e) {
      errors[(size_t)t] = e;
    }
  };
  if (nt == 1) {
    run(0);
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(run, t);
    for (auto& th : pool) th.join();
  }
  for (int t = 0; t < nt; ++t) if (errors[(size_t)t].code != 0) throw errors[(size_t)t];      // the earliest segment's error, as the sequential decoder would report
  if (st) {
    *st = Stats();
    for (const Stats& x : stats) {
      st->granules += x.granules; st->huff_exact += x.huff_exact; st->huff_short += x.huff_short; st->huff_overrun += x.huff_overrun;
      st->crc_checked += x.crc_checked; st->crc_failed += x.crc_failed; st->reservoir_missing += x.reservoir_missing;
      st->short_granules += x.short_granules; st->mixed_granules += x.mixed_granules; st->ms_granules += x.ms_granules;
      st->intensity_granules += x.intensity_granules; st->max_main_data_begin = std::max(st->max_main_data_begin, x.max_main_data_begin);
    }
  }
  return in.samples;
}

}  // namespace mp3
}  // namespace rvb
