#!/bin/bash
# A/B of tuning switches: tools/ab_env.sh <tag> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...  -> gpurun_out/<tag>_<i>.json
tag=$1; args=$2; shift 2
i=0
for envs in "$@"; do
  env QI_TUNE=1 $envs python bench.py $args > gpurun_out/${tag}_$i.json 2> gpurun_out/${tag}_$i.err
  echo "== [$i] $envs rc=$?"
  python tools/show_bench.py gpurun_out/${tag}_$i.json | grep -E "value|stages"
  i=$((i+1))
done
