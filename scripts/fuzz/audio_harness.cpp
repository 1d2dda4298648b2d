#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include <cstdint>
#include "../../include/rvb.h"
namespace rvb { void set_error(const std::string&) {} }
int main(int argc, char** argv) {
  long ok = 0, bad = 0;
  for (int i = 1; i < argc; ++i) {
    FILE* f = fopen(argv[i], "rb"); if (!f) continue;
    std::vector<unsigned char> d; unsigned char buf[65536]; size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.insert(d.end(), buf, buf + n);
    fclose(f);
    // many cases per file: records of [u32 len][bytes]
    size_t pos = 0;
    while (pos + 4 <= d.size()) {
      uint32_t len = d[pos] | d[pos+1] << 8 | d[pos+2] << 16 | (uint32_t)d[pos+3] << 24; pos += 4;
      if (pos + len > d.size()) break;
      std::vector<unsigned char> one(d.begin() + pos, d.begin() + pos + len);   // exact-size heap block: OOB reads are caught
      pos += len;
      rvb_audio_info info;
      if (rvb_audio_probe(one.data(), (int64_t)one.size(), &info) != 0) { ++bad; continue; }
      if (info.frames < 0 || info.frames > (1 << 24) || info.channels < 1 || info.channels > 64) { ++bad; continue; }
      std::vector<float> out((size_t)info.frames * info.channels + 1);
      for (int th = 0; th < 3; ++th) {
        int64_t r = rvb_audio_decode_f32(one.data(), (int64_t)one.size(), -1, out.data(), (int64_t)out.size(), (th * 3) << 8, &info);
        if (r < 0) { ++bad; break; }
        if (th == 2) ++ok;
      }
    }
  }
  printf("ok %ld rejected %ld\n", ok, bad);
  return 0;
}
