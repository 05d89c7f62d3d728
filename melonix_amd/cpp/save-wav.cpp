#include "save-wav.hpp"

#include "melonix_amd.h"

auto saveWav(const std::string &fileName, const std::vector<int16_t> &pcm, int sampleRate) -> void {
#ifdef MELONIX_CORRECT_WAV_HEADER
  const int strict = 0;
#else
  const int strict = 1;
#endif
  // like the reference, failures are not signalled to the caller (SURVEY.md §8b "Errors")
  (void)mx_save_wav(fileName.c_str(), pcm.data(), (int64_t)pcm.size(), sampleRate, strict);
}
