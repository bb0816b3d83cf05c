#!/bin/bash
# The measurement table of DESIGN.md section 7 (run on the GPU box through gpurun).
run() { echo "== $*"; python bench.py --no-cpu-baseline --no-secondary "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('frames/s %.1f  ms/step %.3f  head TF/s %.1f (frac %.3f, share %.2f)  M %s  post ns/anchor %s' % (d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['share_of_step'], d['config'].get('kept_anchors_M'), d['config'].get('per_anchor_covariance_latency_ns')))"; }
run
run --batch 128
run --batch 64
run --batch 32
run --batch 16
run --batch 8
run --batch 1 --steps 100
run --mc 1
run --mc 30 --batch 64 --steps 15
run --height 384 --width 1248 --mc 30 --batch 16 --steps 10
run --height 384 --width 1248 --mc 10 --batch 64 --steps 20
run --height 384 --width 1248 --mc 30 --batch 1 --steps 50
run --precision fp32 --batch 8 --steps 5 --warmup 1
