#!/bin/bash
# Build librange_hip.so (gfx950 only) in-tree.  Usage: ./build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
exec hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" \
    -o range_amd/librange_hip.so range_amd/csrc/range_hip.hip
