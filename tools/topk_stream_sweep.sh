#!/bin/bash
# Tuning aid (GPU box): builds the library with other stream-kernel switches into a TEMPORARY file
# (the in-tree library is never replaced) and times the small-batch scan with it.  Experiment
# builds (RANGE_EXP_*) give invalid results: timing only.
# usage: tools/topk_stream_sweep.sh "<hipcc flags>[@ENV=VAL ...]" ...
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
for cfg in "$@"; do
  flags="${cfg%%@*}"; envs=""; [ "$cfg" != "$flags" ] && envs="${cfg#*@}"
  RANGE_LIB_OUT="$tmp/librange_exp.so" ./build.sh $flags 2>&1 | grep -E "error" || true
  echo "== $cfg"
  env $envs RANGE_LIB_PATH="$tmp/librange_exp.so" RANGE_ALLOW_EXPERIMENT_BUILD=1 \
    timeout -k 10 120 python tools/small_batch_scan.py --stream-only 2>/dev/null | grep -E "stream kernel"
done
