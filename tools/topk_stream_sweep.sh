#!/bin/bash
# Tuning aid (GPU box): rebuilds the library with other stream-kernel geometries / experiment
# switches and times the small-batch scan.  Leaves the LAST build in place: rebuild afterwards.
set -e
cd "$(dirname "$0")/.."
for cfg in "$@"; do
  ./build.sh $cfg 2>&1 | grep -E "error" || true
  echo "== $cfg"
  timeout -k 10 120 python tools/small_batch_scan.py 2>/dev/null | grep -E "stream kernel"
done
