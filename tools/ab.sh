#!/bin/bash
# usage: tools/ab.sh <outdir> "<bench args>" tag1 tag2 ...   (GPU box; tag "hip" = the product library)
out=$1; args=$2; shift 2
mkdir -p gpurun_out/$out
for l in "$@"; do
  lib=$PWD/build/libmercat_$l.so; [ $l = hip ] && lib=$PWD/mercat2_amd/libmercat_hip.so
  MERCAT_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu --no-file-leg $args > gpurun_out/$out/$l.json 2> gpurun_out/$out/$l.err
  echo "$l: $(grep -o '"value": [0-9.e+]*' gpurun_out/$out/$l.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/$out/$l.json | head -1) $(grep -o '"ms_per_launch": [0-9.]*' gpurun_out/$out/$l.json | head -1) rows $(grep -o '"rows": [0-9]*' gpurun_out/$out/$l.json | head -1) $(grep stamp gpurun_out/$out/$l.err | tail -1)"
done
