#!/bin/bash
# usage: tools/ab_interleaved.sh <rounds> "<bench args>" tag1 tag2 ...   (GPU box; tag "hip" = the product library)
# Runs the benchmark with each library in turn, <rounds> times over, and prints every ms_per_step and the minimum
# per library: run-to-run noise on one box is a few per cent, so one run per variant decides nothing.
rounds=$1; args=$2; shift 2
declare -A best
for ((r = 0; r < rounds; ++r)); do
  for l in "$@"; do
    lib=$PWD/build/libmercat_$l.so; [ $l = hip ] && lib=$PWD/mercat2_amd/libmercat_hip.so
    ms=$(MERCAT_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu --no-file-leg $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "round $r $l $ms"
    if [ -z "${best[$l]}" ] || awk "BEGIN{exit !($ms < ${best[$l]})}"; then best[$l]=$ms; fi
  done
done
for l in "$@"; do echo "min $l ${best[$l]}"; done
