#!/bin/bash
# A/B of the float64 leg in one job: tools/r4_ab_f64.sh <tag> "ENV=... ENV=..." "ENV=..." ...   (each argument one variant)
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for variant in "$@"; do
  i=$((i+1))
  ( for kv in $variant; do export "$kv"; done
    python bench.py --legs f64 --cpu-seconds 0 --steps 30 --warmup 5 > $out/v$i.json 2> $out/v$i.err
    python - <<PY
import json
d=json.loads([l for l in open('$out/v$i.json') if l.startswith('{')][-1])
print('variant [$variant]: %.0f Mpoints/s, %.3f ms/step' % (d['value'], d['ms_per_step']), d['step_roofline']['stage_ms_per_step'])
PY
  )
done
