#!/bin/bash
# usage: tools/ab_k63.sh <rounds> "tag[:ENV=1[,ENV2=1]]" ...   (GPU box) -- interleaved A/B on the S3 workload (k = 63, BASELINE config 5)
rounds=$1; shift
declare -A best
for ((r = 0; r < rounds; ++r)); do
  for v in "$@"; do
    tag=${v%%:*}; envs=""; [ "$v" != "$tag" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    lib=$PWD/build/libmercat_$tag.so; [ $tag = hip ] && lib=$PWD/mercat2_amd/libmercat_hip.so
    out=$(env MERCAT_HIP_LIB=$lib $envs timeout -k 10 300 python bench.py --no-cpu --no-file-leg --steps 3 --warmup 2 --k 63 --reads 50000000 --genome 50000000 --genome-seed 6 --read-seed 7 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    km=$(echo "$out" | grep -o '"kernel_ms_per_step": {[^}]*}' | head -1)
    echo "round $r $v step_ms=$ms $km"
    if [ -n "$ms" ] && { [ -z "${best[$v]}" ] || awk "BEGIN{exit !($ms < ${best[$v]})}"; }; then best[$v]=$ms; fi
  done
done
for v in "$@"; do echo "min $v ${best[$v]}"; done
