#!/bin/bash
# A/B of one bench leg in one job: tools/r4_ab_leg.sh <tag> "<bench args>" "ENV=... ENV=..." ...   (each further argument one variant)
tag=$1; shift
bargs=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
i=0
for variant in "$@"; do
  i=$((i+1))
  ( for kv in $variant; do export "$kv"; done
    python bench.py $bargs --cpu-seconds 0 --two-streams 0 --wrappers 0 > $out/v$i.json 2> $out/v$i.err
    python - <<PY
import json
d=json.loads([l for l in open('$out/v$i.json') if l.startswith('{')][-1])
print('variant [$variant]: %.0f Mpoints/s, %.4f ms/step' % (d['value'], d['ms_per_step']), d['step_roofline']['stage_ms_per_step'])
PY
  )
done
