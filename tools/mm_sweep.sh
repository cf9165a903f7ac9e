#!/bin/bash
# k_match timing variants (tools/bin/liborbx_mm_*.so built from a scratch copy; wrong results except the plain library)
python tools/match_rate.py
for v in "$@"; do echo "== $v"; ORBX_LIB=$PWD/tools/bin/liborbx_mm_$v.so python tools/match_rate.py; done
echo "== other shapes, shipped library"
python tools/match_rate.py 32 4000 4000
python tools/match_rate.py 1 1000 1000
python tools/match_rate.py 1 2000 2000
