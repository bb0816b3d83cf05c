#!/bin/bash
# Same-box A/B of one environment switch on the headline bench: ab_env.sh VAR "v0 v1 v0 v1" [bench args]
# prints value / ms per step / stage times / tower roofline fraction per run (run on the GPU box through gpurun)
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  env $VAR=$v python bench.py --steps 8 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | VAL="$VAR=$v" python -c "
import json, os, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(os.environ['VAL'], d['value'], d['ms_per_step'], d['config']['stages_ms_per_step'], d['roofline']['frac'], d['roofline'].get('other_head_launch', {}).get('avg_launch_ms'))"
done
