// marker.hpp — drop-in for the reference's Marker (marker.hpp:4-19).  Layout-compatible with
// mx_marker of the C-ABI.  The reference's `ser` property macros are applied when the caller's
// build provides them (MELONIX_WITH_SER), so project files keep loading.
#pragma once
#ifdef MELONIX_WITH_SER
#include <ser/macro.hpp>
#endif

struct Marker {
  int sample;        // source sample the marker is pinned to
  double note;       // display row (set by the UI, app.cpp:923,937); not used by the resynthesis
  double dTime;      // extra warped time inserted before this marker (time map, app.cpp:1035)
  double pitchBend;  // semitones at this marker (time2PitchBend, app.cpp:1089-1122)

#ifdef MELONIX_WITH_SER
#define SER_PROP_LIST \
  SER_PROP(sample);   \
  SER_PROP(note);     \
  SER_PROP(dTime);    \
  SER_PROP(pitchBend);
  SER_DEF_PROPS()
#undef SER_PROP_LIST
#endif
};
