// tests/compat_stubs: translation unit of the typo guard (g++ -fsyntax-only): the extractor shim alone, as Frame.h includes it
#include "ORBextractor.h"
int main() { ORB_SLAM2::ORBextractor ex(1000, 1.2f, 8, 20, 7); (void)ex.GetLevels(); return 0; }
