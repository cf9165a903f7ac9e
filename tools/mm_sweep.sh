#!/bin/bash
# k_match alone on the bench shape and on the other configurations' shapes (tools/match_rate.py); VARIANT libraries from
# tools/build_variant.sh may be named as arguments
python tools/match_rate.py
for v in "$@"; do echo "== $v"; ORBX_LIB=$PWD/tools/bin/liborbx_$v.so python tools/match_rate.py; done
python tools/match_rate.py 32 4000 4000
python tools/match_rate.py 1 1000 1000
python tools/match_rate.py 1 2000 2000
echo "== vector-pipe kernel"
ORBX_MATCH_KERNEL=valu python tools/match_rate.py
