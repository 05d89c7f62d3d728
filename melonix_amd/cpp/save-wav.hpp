// save-wav.hpp — drop-in for the reference's save-wav.hpp:5 (same signature).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// Writes a mono PCM16 RIFF file.  By default byte-identical to the reference's writer, including
// the data-size quirk of save-wav.cpp:43 (size field 2M+16, samples 0 and 1 zeroed); define
// MELONIX_CORRECT_WAV_HEADER to write a standards-conforming header instead.
auto saveWav(const std::string &fileName, const std::vector<int16_t> &wav, int sampleRate) -> void;
