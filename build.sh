#!/bin/bash
# Build librange_hip.so (gfx950 only) in-tree.  Usage: ./build.sh [extra hipcc flags]
# Two translation units (retrieval engine, ridge probe) compiled side by side, then linked.
# RANGE_LIB_OUT=<path> writes the library somewhere else (tuning sweeps build experiment variants
# into a temporary file and load them with RANGE_LIB_PATH, never over the in-tree library).
# The extra flags are recorded in the library (range_build_flags()): range_amd refuses to load a
# build that carries RANGE_EXP_* timing-experiment switches.
set -e
cd "$(dirname "$0")"
out="${RANGE_LIB_OUT:-range_amd/librange_hip.so}"
obj=$(mktemp -d)
trap 'rm -rf "$obj"' EXIT
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
printf '#define RANGE_BUILD_FLAGS "%s"\n' "$*" > "$obj/range_build_flags.h"
# the hash of the sources this library is built from (range_amd/_srchash.py): range_source_sha256()
printf '#define RANGE_SRC_SHA256 "%s"\n' "$(python3 range_amd/_srchash.py)" >> "$obj/range_build_flags.h"
hipcc $flags "$@" -include "$obj/range_build_flags.h" -c range_amd/csrc/range_hip.hip -o "$obj/range_hip.o" &
pid=$!
hipcc $flags "$@" -c range_amd/csrc/probe_hip.hip -o "$obj/probe_hip.o"
wait $pid
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" "$obj/range_hip.o" "$obj/probe_hip.o"
