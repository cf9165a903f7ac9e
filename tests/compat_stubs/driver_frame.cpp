// tests/compat_stubs: translation unit of the typo guard (g++ -fsyntax-only): Frame.cc as INTEGRATION.md section 2 leaves it --
// the two member bodies come from compat/Frame_stereo.inl, inside namespace ORB_SLAM2
#include <stdexcept>
#include "Frame.h"
#define ORBX_REPLACE_UNDISTORT
namespace ORB_SLAM2 {
#include "Frame_stereo.inl"
}
