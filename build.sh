#!/bin/bash
# Build librange_hip.so (gfx950 only) in-tree.  Usage: ./build.sh [extra hipcc flags]
# Two translation units (retrieval engine, ridge probe) compiled side by side, then linked.
set -e
cd "$(dirname "$0")"
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
hipcc $flags "$@" -c range_amd/csrc/range_hip.hip -o "$obj/range_hip.o" &
pid=$!
hipcc $flags "$@" -c range_amd/csrc/probe_hip.hip -o "$obj/probe_hip.o"
wait $pid
hipcc --offload-arch=gfx950 -shared -fPIC -o range_amd/librange_hip.so "$obj/range_hip.o" "$obj/probe_hip.o"
