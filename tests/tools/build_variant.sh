#!/bin/bash
# Developer aid: a VARIANT build of libbayesod_hip.so for same-box A/B runs (tests/tools/ab_lib.sh, BOD_LIB_OVERRIDE).
# usage: build_variant.sh NAME SOURCE.hip "-DFLAG ..."   -> .ab/libNAME.so (SOURCE recompiled with the flags, every other object from lib/obj;
# no kernel guards: ablation builds may compute wrong results on purpose).  Run bayes_od_rc_amd.build first so that lib/obj is current.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
NAME=$1; SRC=$2; FLAGS=$3
mkdir -p "$ROOT/.ab"
OBJ="$ROOT/.ab/${NAME}_${SRC%.hip}.o"
EXTRA=""
EXTRA="-fno-slp-vectorize"          # (as bayes_od_rc_amd/build.py COMMON; a variant that wants packed fp32 passes -fslp-vectorize in its flags)
case "$SRC" in post_kernels.hip|loss_kernels.hip) EXTRA="$EXTRA -ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-result -Wno-unused-value $EXTRA $FLAGS \
  -c "$ROOT/bayes-od-rc_amd/csrc/$SRC" -o "$OBJ"
OBJS=""
for o in "$ROOT"/bayes-od-rc_amd/lib/obj/*.o; do
  if [ "$(basename "$o")" = "${SRC%.hip}.o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$ROOT/.ab/lib${NAME}.so"
echo "$ROOT/.ab/lib${NAME}.so"
