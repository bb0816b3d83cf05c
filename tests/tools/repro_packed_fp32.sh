#!/bin/bash
# STAND-ALONE reproduction of DESIGN.md 8.4's finding on an MI355X (gfx950, ROCm 7.2) -- two small HIP files, no kernel of this library:
#
#   victim  : tests/tools/row3_victims.hip, mode 17 -- the posterior's prior fusion as plain C++ (two 4x4 Cholesky inverses + two
#             matrix-vector products), compiled with hipcc -O3 defaults, i.e. SLP-vectorised into v_pk_mul_f32 / v_pk_add_f32 /
#             v_pk_fma_f32; every thread evaluates it TWICE on the same input and logs a bitwise difference with HW_REG_HW_ID;
#   company : tests/tools/noise_kernels.hip, mode 7 -- a kernel that interleaves VALU instructions (here: packed fp32; modes 8-12: v_pk_max_i16,
#             v_cvt_pk_bf16_f32, v_mov_b64, v_pk_mul_f32, v_fma_f64 -- any kind will do) with v_mfma_f32_32x32x16_bf16, on another stream
#             from another host thread.
#
# Observed (profiles/round6_selfcheck_probes.txt, section 12): alone 0 differences; beside mode 7: ~4 % of the (wave, iteration) pairs,
# ALWAYS lanes 48-63 of a wave (the last 16-lane row), uniformly over all XCCs and SIMDs; beside MFMAs only (mode 2), packed fp32 only
# (mode 6) or packed int16 (mode 1): 0.  With the victim built -fno-slp-vectorize (no packed fp32 in it) the library's own victim
# kernels show 0 of 56 million waves -- which is how the library is built since round 6.
# The single instruction form (profiles/round6_selfcheck_probes.txt, section 16): v_pk_mul_f32 / v_pk_add_f32 with op_sel:[0,1] -- victim modes
# 20 / 18 / 23 of row3_victims.hip are twenty-line kernels around that one instruction; op_sel:[1,0] (mode 21) and plain operands (mode 12) are immune.
cd "$(dirname "$0")/../.."
ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=none python3 tests/tools/row3_probe.py
ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=synthetic:7 python3 tests/tools/row3_probe.py
ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=synthetic:2 python3 tests/tools/row3_probe.py
ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=synthetic:6 python3 tests/tools/row3_probe.py
ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=synthetic:8 python3 tests/tools/row3_probe.py          # v_pk_max_i16 between MFMAs: the strongest trigger
VICTIM_NOSLP=1 ITERS=${ITERS:-300} LAUNCHES=${LAUNCHES:-3} MODES=17 COMPANY=synthetic:7 python3 tests/tools/row3_probe.py   # the victim without packed fp32: 0
ITERS=${ITERS:-1500} LAUNCHES=${LAUNCHES:-2} MODES=12,20,18,23,21 COMPANY=none python3 tests/tools/row3_probe.py          # explicit instruction forms, alone: 0
ITERS=${ITERS:-1500} LAUNCHES=${LAUNCHES:-2} MODES=12,20,18,23,21 COMPANY=synthetic:8 python3 tests/tools/row3_probe.py   # op_sel:[0,1] forms fail, the others do not
