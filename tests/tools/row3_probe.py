#!/usr/bin/env python3
"""micro-victims (row3_victims.hip: one instruction class each, every chain computed twice and compared) beside
a COMPANY of chosen kernels of the library's forward (BOD_FORWARD_OPS) on another host thread.
  COMPANY=forward | ops:lo:hi | none      MODES=0,1,..,8 (default all)     VICTIM_SLOTS=lo:hi / COMPANY_SLOTS=lo:hi (CU slots per XCD)
prints one JSON line per victim mode: mismatching (thread, iteration) pairs, waves x iterations checked, lane rows, where."""
import ctypes as C, json, os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np
from bayes_od_rc_amd import synthetic
from bayes_od_rc_amd.engine import Engine, make_config

# VICTIM_NOSLP=1: the victims built with -fno-slp-vectorize (no packed fp32 instruction in mode 17's arithmetic)
noslp = os.environ.get("VICTIM_NOSLP") == "1"
so = os.path.join(HERE, "librow3_victims_noslp.so" if noslp else "librow3_victims.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "row3_victims.hip")):
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC"] + (["-fno-slp-vectorize"] if noslp else []) +
                          [os.path.join(HERE, "row3_victims.hip"), "-o", so])
lib = C.CDLL(so)
lib.victim_run.argtypes = [C.c_int] * 5
lib.victim_read.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.c_void_p, C.c_int]
NAMES = {0: "fma chain", 1: "IEEE division chain", 2: "transcendental chain", 3: "integer chain", 4: "fp64 chain", 5: "cross-lane chain",
         6: "fma chain behind a call", 7: "fma chain behind a call, results through scratch", 8: "global loads twice (4 bytes per lane)",
         9: "global_load_dwordx4 twice", 10: "global_load_dwordx2 twice", 11: "scratch array round trip twice",
         12: "packed fp32 chain (v_pk_mul_f32 + v_pk_add_f32)", 13: "packed fp32 chain (v_pk_fma_f32)",
         14: "packed fp32 chain, op_sel_hi broadcast", 15: "packed fp32 chain, neg modifiers", 16: "packed fp32 chain, both modifiers + scalar consumers",
         17: "the posterior's prior fusion (two 4x4 Cholesky inverses + matrix-vector products), SLP-vectorised",
         18: "packed fp32 chain, op_sel:[0,1] op_sel_hi:[1,0]", 19: "packed fp32 chain, op_sel:[1,0]",
         20: "v_pk_mul_f32 op_sel:[0,1] (src1 high half to both lanes)", 21: "v_pk_mul_f32 op_sel:[1,0] op_sel_hi:[0,1] (src0 halves swapped)",
         22: "v_pk_mul_f32 op_sel:[1,1] op_sel_hi:[0,0] (both sources swapped)", 23: "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (src1 halves swapped)"}
company = os.environ.get("COMPANY", "forward")
modes = [int(m) for m in os.environ.get("MODES", "0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23").split(",")]
blocks, iters = int(os.environ.get("BLOCKS", "2048")), int(os.environ.get("ITERS", "40"))
hw, batch = (512, 512), int(os.environ.get("B", "64"))
b = None
lib.company_run.argtypes = [C.c_int] * 3
noise_lib = None
if company.startswith("synthetic:"):      # tests/tools/noise_kernels.hip: one instruction class per mode (7: packed fp32 between MFMAs)
    so2 = os.path.join(HERE, "libnoise_kernels.so")
    if not os.path.exists(so2) or os.path.getmtime(so2) < os.path.getmtime(os.path.join(HERE, "noise_kernels.hip")):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(HERE, "noise_kernels.hip"), "-o", so2])
    noise_lib = C.CDLL(so2)
elif company.startswith("dma"):          # COMPANY=dma:0 (LDS-DMA only) / dma:1 (the same bytes without LDS-DMA): row3_victims.hip's synthetic company
    lib.victim_run(0, 1, 1, 0, 0)      # (allocates the seeds)
elif company != "none":
    if os.environ.get("COMPANY_SLOTS"):
        os.environ["BOD_CU_MASK_SLOTS"] = os.environ["COMPANY_SLOTS"]
    if company.startswith("ops:"):
        os.environ["BOD_FORWARD_OPS"] = company.split(":", 1)[1]
    b = Engine(make_config(hw, batch=batch, mc_samples=2, precision=os.environ.get("COMPANY_PRECISION", "bf16")))
    os.environ.pop("BOD_CU_MASK_SLOTS", None); os.environ.pop("BOD_FORWARD_OPS", None)
    b.load_weights(synthetic.make_weights()); b.upload_images(synthetic.make_frames(batch, hw[0], hw[1], seed=12)); b.forward(None); b.synchronize()
vs = [int(x) for x in os.environ.get("VICTIM_SLOTS", "0:0").split(":")]
stop = False
n_company = [0]


def noise():
    while not stop:
        if noise_lib is not None:
            assert noise_lib.noise_run(int(company.split(":")[1]), 4096, 400) == 0
        elif company.startswith("dma"):
            assert lib.company_run(int(company.split(":")[1]), int(os.environ.get("COMPANY_BLOCKS", "512")), 20000) == 0
        else:
            b.forward(None, seed=1, first_image_id=0)
        n_company[0] += 1


t = threading.Thread(target=noise if (b is not None or company.startswith("dma") or noise_lib is not None) else (lambda: None)); t.start()
cnt, waves = C.c_uint(0), C.c_ulonglong(0)
recs = np.zeros((4096, 8), np.uint32)
try:
    for mode in modes:
        lib.victim_read(C.byref(cnt), C.byref(waves), recs.ctypes.data, 4096)
        c0 = n_company[0]; t0 = time.time()
        for _ in range(int(os.environ.get("LAUNCHES", "6"))):
            rc = lib.victim_run(mode, blocks, iters, vs[0], vs[1])
            assert rc == 0, rc
        dt = time.time() - t0
        lib.victim_read(C.byref(cnt), C.byref(waves), recs.ctypes.data, 4096)
        k = min(int(cnt.value), 4096)
        out = {"victim": NAMES[mode], "company": company, "victim_slots": os.environ.get("VICTIM_SLOTS"), "company_slots": os.environ.get("COMPANY_SLOTS"),
               "seconds": round(dt, 2), "company_forwards": n_company[0] - c0, "wave_iterations": int(waves.value), "mismatches": int(cnt.value)}
        if k:
            r = recs[:k]
            out["lane_rows"] = {str(q): int((r[:, 3] // 16 == q).sum()) for q in range(4)}
            out["by_xcc"] = {str(x): int(((r[:, 1] & 0xF) == x).sum()) for x in range(8)}
            out["by_simd"] = {str(s_): int((((r[:, 0] >> 4) & 3) == s_).sum()) for s_ in range(4)}
            fl = r[:, 6:8].copy().view(np.float32)
            out["examples"] = [[float(fl[i, 0]), float(fl[i, 1])] for i in range(min(k, 3))]
        print(json.dumps(out), flush=True)
finally:
    stop = True; t.join()
if b is not None:
    b.close()
