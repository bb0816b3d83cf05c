#!/usr/bin/env python3
"""WHERE and BESIDE WHAT does post_fuse_kernel miscompute a 16-lane row?  (development probe)

Needs a BOD_POST_SELFCHECK build of the library: post_fuse_kernel computes every slot twice in the same thread through one out-of-line
copy of the code and logs every slot whose two results differ with the wave's HW_REG_HW_ID / HW_REG_XCC_ID.
    tests/tools/build_variant.sh sc2packed post_kernels.hip "-DBOD_POST_SELFCHECK=2 -fslp-vectorize"   # the victim WITH packed fp32 (the fault)
    tests/tools/build_variant.sh sc2       post_kernels.hip "-DBOD_POST_SELFCHECK=2"                    # as the product is built: without (0 events)
    BOD_LIB_OVERRIDE=.ab/libsc2packed.so COMPANY=synthetic:7 python3 tests/tools/selfcheck_probe.py 2000
(-DBOD_POST_SELFCHECK=1: loads and stores inside the twice-run code; =2: inputs in registers; -DBOD_POST_SELFCHECK_PART=1..3 cuts the
twice-run code behind the epistemic / aleatoric / likelihood block.  Records of round 6: profiles/round6_selfcheck_probes.txt.)

The VICTIM handle re-runs the posterior on unchanged MC statistics; a COMPANY handle runs on another host thread:
  COMPANY=forward            the whole bf16 forward (the round-5 trigger)
  COMPANY=ops:lo:hi          ops [lo, hi) of the forward's plan only (BOD_FORWARD_OPS) -- bisection over the real kernels
  COMPANY=none               nothing (control)
  COMPANY=synthetic:m        tests/tools/noise_kernels.hip mode m (one instruction class: 0 cvt_pk_bf16, 1 packed int16, 2 bf16 MFMA, 3 LDS-DMA +
                             ds_read, 4 bitop3 / perm, 5 transcendental, 6 packed fp32, 7 packed fp32 between MFMAs)
  VICTIM_SLOTS=lo:hi / COMPANY_SLOTS=lo:hi   CU slots of every XCD the victim's / the company's main stream may use (disjoint masks:
                             if the fault persists, SIMD co-residency is not what causes it)
usage: selfcheck_probe.py [iterations]   prints one JSON line per configuration"""
import ctypes as C, json, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from conftest import ANCHOR_CFG
from bayes_od_rc_amd import synthetic, _lib
from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
from bayes_od_rc_amd.engine import Engine, make_config

BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"}, "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}
hw, n, batch = (512, 512), 2, int(os.environ.get("B", "64"))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
company = os.environ.get("COMPANY", "forward")
weights = synthetic.make_weights(cls_fg_bias=float(os.environ.get("FG_BIAS", "-1.0")))
anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
frames = synthetic.make_frames(batch, hw[0], hw[1], seed=12)


def make(mask_env, ops=None, prec="bf16"):
    for k in ("BOD_CU_MASK_SLOTS", "BOD_FORWARD_OPS"):
        os.environ.pop(k, None)
    if os.environ.get(mask_env):
        os.environ["BOD_CU_MASK_SLOTS"] = os.environ[mask_env]
    if ops:
        os.environ["BOD_FORWARD_OPS"] = ops
    e = Engine(make_config(hw, batch=batch, mc_samples=n, bayes_od_config=BAYES_CFG, nms_config=NMS_CFG, use_full_covar=True, precision=prec))
    for k in ("BOD_CU_MASK_SLOTS", "BOD_FORWARD_OPS"):
        os.environ.pop(k, None)
    e.load_weights(weights); e.set_anchors(anchors); e.upload_images(frames)
    return e


lib = _lib.load()
read = getattr(lib, "bod_debug_selfcheck_read", None)
if read is None:
    sys.exit("this library is not a BOD_POST_SELFCHECK build")
read.restype = C.c_int
read.argtypes = [C.POINTER(C.c_uint), C.POINTER(C.c_ulonglong), C.c_void_p, C.c_int]

e = make("VICTIM_SLOTS")
e.infer(None, seed=3, first_image_id=0)
e.synchronize()
ref = [e.get_posterior(i) for i in range(0, batch, max(1, batch // 8))]
b = None
noise_lib = None
if company.startswith("synthetic:"):
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so_ = os.path.join(here, "libnoise_kernels.so")
    if not os.path.exists(so_) or os.path.getmtime(so_) < os.path.getmtime(os.path.join(here, "noise_kernels.hip")):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(here, "noise_kernels.hip"), "-o", so_])
    noise_lib = C.CDLL(so_)
elif company != "none":
    b = make("COMPANY_SLOTS", ops=company.split(":", 1)[1] if company.startswith("ops:") else None, prec=os.environ.get("COMPANY_PRECISION", "bf16"))
    b.forward(None)
    b.synchronize()
cnt, waves = C.c_uint(0), C.c_ulonglong(0)
recs = np.zeros((4096, 8), np.uint32)
read(C.byref(cnt), C.byref(waves), recs.ctypes.data, 4096)          # reset (the set-up's own launches)
stop = False
n_company = [0]


def noise():
    while not stop:
        if noise_lib is not None:
            assert noise_lib.noise_run(int(company.split(":")[1]), 4096, 400) == 0
        else:
            b.forward(None, seed=1, first_image_id=0)
        n_company[0] += 1


t = threading.Thread(target=noise if (b is not None or noise_lib is not None) else (lambda: None)); t.start()
t0 = time.time()
try:
    for it in range(iters):
        e.posterior(seed=3, first_image_id=0)
    e.synchronize()
finally:
    stop = True; t.join()
dt = time.time() - t0
read(C.byref(cnt), C.byref(waves), recs.ctypes.data, 4096)
k = min(int(cnt.value), 4096)
r = recs[:k]
out = {"company": company, "victim_slots": os.environ.get("VICTIM_SLOTS"), "company_slots": os.environ.get("COMPANY_SLOTS"), "iterations": iters,
       "seconds": round(dt, 1), "company_forwards": n_company[0], "waves_checked": int(waves.value), "mismatching_slots": int(cnt.value)}
if k:
    hw_id, xcc = r[:, 0], r[:, 1] & 0xF
    simd, cu, sh, se = (hw_id >> 4) & 3, (hw_id >> 8) & 0xF, (hw_id >> 12) & 1, (hw_id >> 13) & 7
    lane = r[:, 4]
    # a wave's mismatching lanes arrive as separate records: group them by (image, slot // 64 * 64 base of the wave's 64 slots is not the
    # wave -- slots are strided -- so group by (hw_id, xcc, image, slot - lane)
    key = {}
    for i in range(k):
        key.setdefault((int(hw_id[i]), int(xcc[i]), int(r[i, 2]), int(r[i, 3]) - int(lane[i])), []).append(int(lane[i]))
    out["faulting_waves"] = len(key)
    out["lane_rows"] = {str(q): int(sum(1 for l in lane if l // 16 == q)) for q in range(4)}
    out["lanes_per_wave"] = sorted({len(v) for v in key.values()})
    where = {}
    for (h, x, _, _), _v in key.items():
        w = "xcc%d se%d sh%d cu%d simd%d" % (x, (h >> 13) & 7, (h >> 12) & 1, (h >> 8) & 0xF, (h >> 4) & 3)
        where[w] = where.get(w, 0) + 1
    out["where"] = dict(sorted(where.items(), key=lambda kv: -kv[1])[:24])
    out["distinct_units"] = len(where)
    out["by_xcc"] = {str(x): int(sum(1 for (h, xx, _, _) in key if xx == x)) for x in range(8)}
    out["by_simd"] = {str(s_): int(sum(1 for (h, _, _, _) in key if ((h >> 4) & 3) == s_)) for s_ in range(4)}
    fl = r[:, 6:8].copy().view(np.float32)
    out["examples"] = [{"elem": int(r[i, 5]), "first": float(fl[i, 0]), "second": float(fl[i, 1])} for i in range(min(k, 4))]
print(json.dumps(out), flush=True)
e.close()
if b is not None:
    b.close()
