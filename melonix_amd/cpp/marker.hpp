// marker.hpp — drop-in for the reference's Marker (marker.hpp:4-19).  Layout-compatible with
// mx_marker of the C-ABI.  The editor's .melonix load/save (app.cpp:1142-1189) walks the struct
// through mika314's `ser` property macros (marker.hpp:10-17): they are applied whenever the
// including build can see <ser/macro.hpp> — as the editor's own build does — so the project-file
// code keeps compiling with this header in place of the reference's.  -DMELONIX_WITH_SER forces
// them on (a missing header is then an error), -DMELONIX_NO_SER leaves them out.
#pragma once
#if !defined(MELONIX_WITH_SER) && !defined(MELONIX_NO_SER) && defined(__has_include)
#if __has_include(<ser/macro.hpp>)
#define MELONIX_WITH_SER 1
#endif
#endif
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
