#!/bin/bash
# Same-box A/B of two BUILDS of libbayesod_hip.so on the headline bench: ab_lib.sh "libA.so libB.so libA.so libB.so" [bench args]
# (BOD_LIB_OVERRIDE selects the library bayes_od_rc_amd._lib loads); prints value / ms per step / stage times / tower roofline
# fraction / fan-out launch ms per run.  Run on the GPU box through gpurun; keep the builds under .ab/ (git-ignored, travels).
LIBS=$1; shift
for l in $LIBS; do
  BOD_LIB_OVERRIDE=$l python bench.py --steps 8 --no-secondary --no-cpu-baseline "$@" 2>/dev/null | VAL="$l" python -c "
import json, os, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(os.environ['VAL'], d['value'], d['ms_per_step'], d['config']['stages_ms_per_step'], d['roofline']['frac'], d['roofline'].get('other_head_launch', {}).get('avg_launch_ms'))"
done
