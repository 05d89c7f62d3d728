// C-callable driver around the reference's own saveWav (save-wav.cpp:17-48).
// TEST INFRASTRUCTURE ONLY.  This file is the build's own text; the reference
// sources are compiled where they lie under /root/reference (never copied).
#include <cstdint>
#include <string>
#include <vector>

#include "save-wav.hpp" // resolved via -I/root/reference

extern "C" int ref_save_wav(const char *path, const int16_t *pcm, long m, int sampleRate) {
  const std::vector<int16_t> v(pcm, pcm + m);
  saveWav(path, v, sampleRate);
  return 0;
}
