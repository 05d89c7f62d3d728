// fake_mx.h — TEST INFRASTRUCTURE: what tests/cpp/facade_stress.cpp knows about the host fake of the C-ABI (fake_mx.cpp).
#pragma once
#include <cstdint>

// the "transform" of the fake: a magnitude that is a function of the column's key and the bin alone
inline float fake_mag(int start, int end, int bin) {
  const std::uint32_t h = static_cast<std::uint32_t>(start) * 2654435761u ^ static_cast<std::uint32_t>(end) * 40503u ^
                          static_cast<std::uint32_t>(bin) * 2246822519u;
  return static_cast<float>((h >> 9) & 0x3FFF) * (1.0f / 4096.0f);  // [0, 4): exact in binary32
}

extern "C" {
// every transform / fetch / colormap call sleeps this long first (the "device" is busy)
void fake_set_latency_us(int us);
// every `keep_nomem_every`-th mx_stft_ranges_keep answers MX_ERR_NOMEM, every `device_every`-th call of any transform
// entry point MX_ERR_DEVICE (0: never)
void fake_set_failures(int keep_nomem_every, int device_every);
void fake_reset_counts(void);
unsigned long long fake_transformed_columns(void);   // columns that went through a transform entry point
int fake_max_transforms_of_one_key(void);            // since the last reset
// live objects (must all be 0 once every Spec is gone)
long fake_live_contexts(void), fake_live_audio(void), fake_live_pinned(void), fake_live_rows(void);
}
