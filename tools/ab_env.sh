#!/bin/bash
# usage: tools/ab_env.sh <rounds> "<bench args>" "ENV=1" ...   (GPU box) -- like ab_interleaved.sh, but the variants are
# environment settings of the product library ("-" = none), e.g. tools/ab_env.sh 4 "--steps 10" - MK_SCATTER_WALK=1
rounds=$1; args=$2; shift 2
declare -A best
for ((r = 0; r < rounds; ++r)); do
  for e in "$@"; do
    if [ "$e" = - ]; then ms=$(timeout -k 10 200 python bench.py --no-cpu --no-file-leg $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    else ms=$(env "$e" timeout -k 10 200 python bench.py --no-cpu --no-file-leg $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2); fi
    echo "round $r $e $ms"
    if [ -z "${best[$e]}" ] || awk "BEGIN{exit !($ms < ${best[$e]})}"; then best[$e]=$ms; fi
  done
done
for e in "$@"; do echo "min $e ${best[$e]}"; done
