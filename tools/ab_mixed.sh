#!/bin/bash
# usage: tools/ab_mixed.sh <rounds> "<bench args>" "tag[:ENV=1[,ENV2=1]]" ...   (GPU box) -- interleaved A/B of libraries
# (tag "hip" = the product library, else build/libmercat_<tag>.so) with optional environment switches, minimum per variant.
rounds=$1; args=$2; shift 2
declare -A best
for ((r = 0; r < rounds; ++r)); do
  for v in "$@"; do
    tag=${v%%:*}; envs=""; [ "$v" != "$tag" ] && envs=$(echo "${v#*:}" | tr ',' ' ')
    lib=$PWD/build/libmercat_$tag.so; [ $tag = hip ] && lib=$PWD/mercat2_amd/libmercat_hip.so
    out=$(env MERCAT_HIP_LIB=$lib $envs timeout -k 10 200 python bench.py --no-cpu --no-file-leg $args 2>/dev/null)
    ms=$(echo "$out" | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    cms=$(echo "$out" | grep -o '"ms_per_launch": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "round $r $v step_ms=$ms count_launch_ms=$cms"
    if [ -n "$ms" ] && { [ -z "${best[$v]}" ] || awk "BEGIN{exit !($ms < ${best[$v]})}"; }; then best[$v]=$ms; fi
  done
done
for v in "$@"; do echo "min $v ${best[$v]}"; done
