#!/bin/bash
# N=1 forward (BASELINE config 2) at several batch sizes: frames/s and ms per frame -- does a sub-batch that fits the Infinity Cache
# (256 MB) run the memory-bound backbone layers faster per frame?
for b in "$@"; do
  python bench.py --mc 1 --batch $b --steps 12 --warmup 3 --no-secondary --no-cpu-baseline 2>/dev/null | B=$b python -c "
import json, os, sys
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('batch', os.environ['B'], 'frames/s', d['value'], 'ms/step', d['ms_per_step'], 'us/frame', round(1e3 * d['ms_per_step'] / int(os.environ['B']), 2))"
done
