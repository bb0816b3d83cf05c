#!/bin/bash
# Same-box A/B of two builds of the library on the tower-kernel micro-benchmark (run through gpurun):
#   .ab/libA.so, .ab/libB.so (git-ignored) are copied over the in-tree library in turn, ABAB.
# usage: ab_head_conv.sh <bench_head_conv.py arguments>
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in A B; do
    cp .ab/lib$v.so bayes-od-rc_amd/lib/libbayesod_hip.so
    echo "== lib$v"
    python3 tests/tools/bench_head_conv.py "$@" 2>&1 | grep "round 1"
  done
done
