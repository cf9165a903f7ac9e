#!/bin/bash
# build a kernel variant of liborbx.so into tools/bin/: tools/build_variant.sh NAME -DFR_WPS=5 ...
name=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/bin
S=orb_slam2_detailed_comments_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -x hip -ffp-contract=off -fno-fast-math -w "$@" \
  $S/orbx_kernels.hip $S/orbx_api.cpp $S/orbx_geometry.cpp $S/orbx_policies.cpp -o tools/bin/liborbx_$name.so
