#!/bin/bash
# A/B timing of library variants over several bench configurations:
#   tools/ab_multi.sh "B C D W" lib1.so lib2.so ...      (configs as bench.py --config names; C runs at 1e9 bytes)
CONFIGS="$1"; shift
for cfg in $CONFIGS; do
  tools/ab.sh "--config $cfg --steps 10 --warmup 3" "$@" || exit 1
done
