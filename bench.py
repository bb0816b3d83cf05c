#!/usr/bin/env python3
"""Headline benchmark: images/sec of the BayesOD inference hot path at N=10 MC samples on
512x512 BDD-shape synthetic frames (BASELINE.json `metric`, config[2]).

A "step" = one pass of the whole hot path (RetinaNet forward -> MC posterior -> soft-NMS ->
cluster-and-fuse -> detections on the host) over one batch of frames that is already resident in
HBM.  One process per GPU; frames are sharded across ranks (weak scaling: fixed per-GPU batch); the
only inter-GPU traffic is one RCCL gather of the final detection records per step.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

The ONE JSON line carries, beside the headline (bf16, BASELINE config 3):
  roofline       the dominant kernel's achieved TFLOP/s from HIP events inside the timed region
  cpu_baseline   the CPU port timed on this host (rank 0, N=1 only)
  parity_mode    the fastest precision mode that meets north_star's 1e-3 END TO END in this run -- f16mx4 / f16mx (round 5: head towers
                 on one f16 product + the two cross terms as one block-scaled e2m1 / e2m3 product) or bf16x3 (rounds 2-4) -- timed
                 in the same run, with its max relative error against the CPU port's fp32 forward of the same frame and Philox masks
                 and the detection-level distance over every frame of the CPU leg; parity_mode_<name>: the other two beside it.
                 Round 6: each carries its own `roofline` (its dominant kernel from the timed steps) and `meets_1e-3` = the gate as
                 clauses with the max beside every p99 and every detection outside 1e-3 counted by CAUSE (a discrete flip of the
                 filter / a cluster member / a draw / a centre, or arithmetic)
  value_at_parity  images/sec of `parity_mode` when a measured mode met every clause (north_star's two clauses together), else null
  secondary      BASELINE configs 2 (N=1 forward), 4's geometry (384x1248, N=30, one GPU) and 5 (ResNet-101 training step)
  value_with_h2d the headline with the uint8 frames crossing PCIe every step (copy stream, overlapped with the convolutions)
"""
import argparse
import datetime
import json
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# HIP runtime: hardware queues the process's streams are multiplexed onto (default 4).  The training step of `secondary` gains from 8
# (docs/DESIGN_HISTORY.md A.1; the inference headline does not move); set here, before anything initialises HIP, and recorded in the
# JSON line (`config.gpu_max_hw_queues`).  An explicit GPU_MAX_HW_QUEUES wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

ANCHOR_CFG = {"layers": [3, 4, 5, 6, 7], "aspect_ratios": [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]],
              "scales": [1.0, 1.26, 1.59], "min_positive_iou": 0.5, "max_negative_iou": 0.4}
BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"},
             "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3       # f32-in MFMA = fp32 vector rate (fp32 precision mode)
# cls foreground bias calibrated so that 500 <= M <= 1500 anchors survive the background filter
# at 512x512 with synthetic.make_weights() (python bench.py --calibrate)
CALIBRATED_FG_BIAS = -3.2
METRIC = "images/sec at N=10 MC samples, 512x512; per-anchor covariance latency"
PARITY_MODES = ("f16mx4", "f16mx", "bf16x3")      # the 1e-3 end-to-end modes, fastest first (f16mx4: least margin)
N_CMP_FRAMES = 16                  # frames whose device detections are kept for the comparison with the CPU leg (it computes ~15 in its budget)
USE_DIST = False                   # process group initialised (N > 1, or BOD_BENCH_FORCE_DIST=1 on one rank)
# conv FLOPs (2 MACs) of backbone + FPN per 512x512 image, linear in the pixel count
BACKBONE_FPN_GFLOP_512 = {50: 49.05, 101: 49.05 + 17 * 2.0 * 1024 * (1024 * 256 + 2304 * 256 + 256 * 1024) / 1e9}


def head_conv_flops(P):
    return 2.0 * P * 256 * 2304          # one 3x3 256->256 head conv over one image's pyramid, one sample


def head_out_flops(P):
    return 2.0 * P * 256 * 9 * (8 + 4 + 10)     # the three 1x1 output convs (cls 9x8, box 9x4, cov 9x10 channels)


def head_flops_per_image(P, N, dedup=True):
    conv = head_conv_flops(P)
    if dedup:
        return 3 * conv + N * 8 * conv
    return N * 11 * conv


def image_gflop(hw, P, n, depth=50, dedup=True):
    """De-duplicated conv GFLOP of one image through the whole network: backbone + FPN + heads."""
    heads = (3 + 8 * n) * head_conv_flops(P) if (dedup and n > 1) else 11 * n * head_conv_flops(P)
    return (BACKBONE_FPN_GFLOP_512[depth] * 1e9 * (hw[0] * hw[1]) / (512.0 * 512.0) + heads + n * head_out_flops(P)) / 1e9


def _rel_errs(got, ref):
    """(max |got-ref| / (|ref| + rms(ref)),  rms(got-ref) / rms(ref),  strict: max |got-ref| / max(|ref|, 1e-5) over |ref| >= 1 % of rms(ref))"""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    rms = float(np.sqrt((ref ** 2).mean()))
    d = np.abs(got - ref)
    big = np.abs(ref) >= 1e-2 * rms
    strict = float((d[big] / np.maximum(np.abs(ref[big]), 1e-5)).max()) if big.any() else 0.0
    return float(np.max(d / (np.abs(ref) + rms))), float(np.sqrt(((got - ref) ** 2).mean()) / rms), strict


def detection_parity(dev, ref, arrays=False, dev_ctx=None, cpu_ctx=None):
    """Final output of the path for one frame -- cluster-fused detections (scores [K,C], means [K,4] or [K,4,1], covs [K,4,4],
    counts [K,C]) -- device vs the CPU leg, both from the same frame, Philox dropout masks and categorical uniforms.
    Every CPU detection is paired with the device detection whose box overlaps it most (IoU > 0.5, each device detection used
    once); `same_order` says whether that pairing is the identity, i.e. both sides selected the same soft-NMS centres in the
    same order (the parity mode: expected; the bf16 headline mode: a few centres differ).  Errors are over
    the matched pairs with the abs floors of tests/conftest.py (means: |ref| + 1 px; covariance entries: |ref| + 1 % of the
    matrix's largest entry; scores: absolute)."""
    if ref is None or dev is None:
        return {"matched": 0, "device_detections": 0 if dev is None else int(len(dev[0])), "cpu_detections": 0 if ref is None else int(len(ref[0]))}
    ds, dm, dc = np.asarray(dev[0], np.float64), np.asarray(dev[1], np.float64).reshape(-1, 4), np.asarray(dev[2], np.float64)
    rs, rm, rc = np.asarray(ref[0], np.float64), np.asarray(ref[1], np.float64).reshape(-1, 4), np.asarray(ref[2], np.float64)
    out = {"matched": 0, "device_detections": int(len(dm)), "cpu_detections": int(len(rm))}
    if len(dm) == 0 or len(rm) == 0:
        return out

    def corners(m):                     # (v, u, h, w) -> (v1, u1, v2, u2)
        return np.stack([m[:, 0] - m[:, 2] / 2, m[:, 1] - m[:, 3] / 2, m[:, 0] + m[:, 2] / 2, m[:, 1] + m[:, 3] / 2], 1)
    a, b = corners(rm), corners(dm)
    iv = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    iu = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iv * iu
    area = lambda c: np.abs((c[:, 2] - c[:, 0]) * (c[:, 3] - c[:, 1]))
    iou = inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-12)
    pairs, used = [], set()
    for i in np.argsort(-iou.max(axis=1)):              # most certain CPU detections first
        for j in np.argsort(-iou[i]):
            if iou[i, j] <= 0.5:
                break
            if j not in used:
                used.add(int(j)); pairs.append((int(i), int(j)))
                break
    out["matched"] = len(pairs)
    if not pairs:
        return out
    ri, di = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
    out["same_order"] = bool(len(pairs) == len(rm) == len(dm) and np.array_equal(ri, di))
    dmu = np.abs(dm[di] - rm[ri])
    floor = np.abs(rc[ri]).reshape(len(ri), -1).max(axis=1)[:, None, None] * 1e-2
    rel_sig = np.abs(dc[di] - rc[ri]) / (np.abs(rc[ri]) + floor)
    dsc = np.abs(ds[di] - rs[ri])
    out.update({"max_abs_dmu_px": float("%.3g" % dmu.max()), "median_abs_dmu_px": float("%.3g" % np.median(dmu.max(axis=1))),
                "max_rel_dmu": float("%.3g" % (dmu / (np.abs(rm[ri]) + 1.0)).max()),
                "max_rel_dSigma": float("%.3g" % rel_sig.max()),
                "median_rel_dSigma": float("%.3g" % np.median(rel_sig.reshape(len(ri), -1).max(axis=1))),
                "max_dscore": float("%.3g" % dsc.max()), "median_dscore": float("%.3g" % np.median(dsc.max(axis=1))),
                "counts_equal": bool(np.array_equal(np.asarray(dev[3], np.float64)[di], np.asarray(ref[3], np.float64)[ri]))})
    if arrays:           # per matched detection: worst coordinate / entry / class (detection_statistics aggregates them over frames)
        out["_dmu_px"] = dmu.max(axis=1)
        out["_rel_dmu"] = (dmu / (np.abs(rm[ri]) + 1.0)).max(axis=1)
        out["_rel_dsigma"] = rel_sig.reshape(len(ri), -1).max(axis=1)
        # the GATE's form for covariance entries: the one the raw head outputs are judged by -- |d| / (|entry| + rms of the matrix's
        # entries).  Off-diagonal entries are signed and cross zero, so an entry-wise relative error needs a floor at the scale of the
        # matrix; the 1 %-of-the-largest-entry floor above (rounds 4-5) stays reported beside it, as does the norm-wise distance below.
        rms_m = np.sqrt((rc[ri] ** 2).reshape(len(ri), -1).mean(axis=1))[:, None, None]
        out["_rms_dsigma"] = (np.abs(dc[di] - rc[ri]) / (np.abs(rc[ri]) + rms_m)).reshape(len(ri), -1).max(axis=1)
        # the same covariance distance norm-wise (no floor to choose): ||dSigma||_F / ||Sigma||_F
        out["_fro_dsigma"] = (np.sqrt(((dc[di] - rc[ri]) ** 2).reshape(len(ri), -1).sum(axis=1)) /
                              np.maximum(np.sqrt((rc[ri] ** 2).reshape(len(ri), -1).sum(axis=1)), 1e-30))
        out["_dscore"] = dsc.max(axis=1)
        out["_cause"] = [explain_detection(int(i), int(j), dev_ctx, cpu_ctx) for i, j in zip(ri, di)]
        out["_unmatched_cause"] = [explain_detection(int(i), None, dev_ctx, cpu_ctx) for i in range(len(rm)) if i not in set(ri.tolist())]
    return out


def _vuhw_corners32(m):
    m = np.asarray(m, np.float32).reshape(-1, 4)
    return np.stack([m[:, 0] - m[:, 2] / 2, m[:, 1] - m[:, 3] / 2, m[:, 0] + m[:, 2] / 2, m[:, 1] + m[:, 3] / 2], 1)


def _iou_column_a15(corners, c):
    """bbox_iou_vuvu of every box against box c (box_utils.py:117-146: + 1 pixel in the intersection, the (min - max + 1) areas)."""
    b = corners[c]
    iv = np.maximum(np.minimum(corners[:, 2], b[2]) - np.maximum(corners[:, 0], b[0]) + 1, 0)
    iu = np.maximum(np.minimum(corners[:, 3], b[3]) - np.maximum(corners[:, 1], b[1]) + 1, 0)
    inter = iv * iu
    area = lambda q: (q[..., 0] - q[..., 2] + 1) * (q[..., 1] - q[..., 3] + 1)
    return inter / (area(corners) + area(b) - inter + 1e-5)


def explain_detection(k_cpu, k_dev, dev_ctx, cpu_ctx):
    """WHY a device detection may differ from the CPU leg's by more than rounding: the pipeline has three discrete decisions between the head
    outputs and a detection, and a 1e-4 perturbation of the outputs can flip each of them for a few anchors per frame --
      centre_differs : soft-NMS selected another anchor as this cluster's centre (ranking order of two candidates);
      filter_flip    : an anchor is kept by one side's categorical filter and not by the other's (a draw at a CDF edge: inference_utils.py:37-51);
      member_flip    : the same anchors are kept, but one sits on the other side of the affinity threshold (IoU > 0.5 of posterior means, :316);
      draw_flip      : same centre, same members, but a member's sampled class counts differ (a draw at a CDF edge that did not change the filter);
      numeric        : none of the above -- the difference is arithmetic.
    Needs the posterior both sides kept (anchor ids, counts, means) and the centres; None when the bench did not collect them."""
    if dev_ctx is None or cpu_ctx is None:
        return None
    ca = cpu_ctx["anchor_index"]
    cpu_centre = int(ca[cpu_ctx["centres"][k_cpu]])
    cpu_members = set(int(x) for x in ca[np.nonzero(cpu_ctx["iou"][:, cpu_ctx["centres"][k_cpu]] > 0.5)[0]])
    da = dev_ctx["anchor_index"]
    dev_kept, cpu_kept = cpu_ctx.setdefault("_dev_kept", set(int(x) for x in da)), cpu_ctx.setdefault("_cpu_kept", set(int(x) for x in ca))
    if k_dev is None:
        return "centre_differs" if cpu_centre in dev_kept else "filter_flip"
    dcen = int(dev_ctx["centres"][k_dev])
    if int(da[dcen]) != cpu_centre:
        return "centre_differs" if cpu_centre in dev_kept else "filter_flip"
    corners = dev_ctx.setdefault("_corners", _vuhw_corners32(dev_ctx["means"]))
    dev_members = set(int(x) for x in da[np.nonzero(_iou_column_a15(corners, dcen) > 0.5)[0]])
    if dev_members != cpu_members:
        diff = dev_members ^ cpu_members
        return "filter_flip" if any((a not in dev_kept) or (a not in cpu_kept) for a in diff) else "member_flip"
    dpos = dev_ctx.setdefault("_pos", {int(a): i for i, a in enumerate(da)})
    cpos = cpu_ctx.setdefault("_pos", {int(a): i for i, a in enumerate(ca)})
    for a in cpu_members:
        if not np.array_equal(np.asarray(dev_ctx["counts"][dpos[a]], np.float32), np.asarray(cpu_ctx["counts"][cpos[a]], np.float32)):
            return "draw_flip"
    return "numeric"


def _posterior_pmc_bytes(B, n, hw):
    """HBM bytes (FETCH_SIZE corrected + WRITE_SIZE) of the unfused posterior's kernels per step from the committed PMC passes of the
    same configuration, or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "round5_posterior_pmc.json")) as fp:
            pmc = json.load(fp)
        c = pmc["config"]
        if (c["height"], c["width"], c["mc_samples"], c["batch"]) == (hw[0], hw[1], n, B):
            return {"bytes_per_step": int(pmc["hbm_bytes_per_step"]), "source": "profiles/round5_posterior_pmc.json"}
    except (OSError, KeyError, ValueError):
        pass
    return None


def detection_statistics(per_frame):
    """Detection-level distance of one precision mode to the CPU leg over EVERY frame the CPU leg computed (round-4 review: one frame is
    an anecdote): match rate, frames whose detections come out in the CPU leg's order, and median / p95 / p99 / max of the per-detection
    worst box coordinate error (pixels and relative to |mu| + 1 px), covariance error (entries relative to |entry| + 1 % of the matrix's
    largest entry; Frobenius-relative) and score error.  Round 6: every detection outside 1e-3 is counted WITH ITS CAUSE (explain_detection:
    a discrete flip of the filter / a cluster member / a draw / the centre, or arithmetic), and the maxima are repeated over the detections
    whose discrete decisions all agree (`numeric_only`): that is the statement the 1e-3 gate is about."""
    per_frame = [p for p in per_frame if p is not None]
    if not per_frame:
        return None
    q = lambda name: np.concatenate([np.asarray(p[name]) for p in per_frame if name in p]) if any(name in p for p in per_frame) else np.zeros(0)
    st = lambda v: ({"median": float("%.3g" % np.median(v)), "p95": float("%.3g" % np.quantile(v, 0.95)), "p99": float("%.3g" % np.quantile(v, 0.99)),
                     "max": float("%.3g" % v.max())} if len(v) else None)
    out = {"frames": len(per_frame),
           "cpu_detections": int(sum(p["cpu_detections"] for p in per_frame)), "device_detections": int(sum(p["device_detections"] for p in per_frame)),
           "matched": int(sum(p["matched"] for p in per_frame)),
           "frames_in_same_order": int(sum(1 for p in per_frame if p.get("same_order"))),
           "frames_counts_equal": int(sum(1 for p in per_frame if p.get("counts_equal"))),
           "abs_dmu_px": st(q("_dmu_px")), "rel_dmu": st(q("_rel_dmu")), "rms_dSigma": st(q("_rms_dsigma")), "rel_dSigma": st(q("_rel_dsigma")),
           "fro_dSigma": st(q("_fro_dsigma")),
           "dscore": st(q("_dscore"))}
    causes = [c for p in per_frame for c in p.get("_cause", [])]
    if causes and len(causes) == len(q("_rel_dmu")):
        rel_mu, rel_sig, dsc, sig_1pct = q("_rel_dmu"), q("_rms_dsigma"), q("_dscore"), q("_rel_dsigma")
        outside = (rel_mu > 1e-3) | (rel_sig > 1e-3) | (dsc > 1e-3)
        known = all(c is not None for c in causes)
        by_cause = {}
        for c, o in zip(causes, outside):
            if o:
                by_cause[c or "unexplained (posterior not collected)"] = by_cause.get(c or "unexplained (posterior not collected)", 0) + 1
        unmatched = [c for p in per_frame for c in p.get("_unmatched_cause", [])]
        for c in unmatched:
            by_cause["unmatched: " + (c or "unexplained")] = by_cause.get("unmatched: " + (c or "unexplained"), 0) + 1
        out["outside_1e-3"] = {"detections": int(outside.sum()) + len(unmatched), "of": out["cpu_detections"], "by_cause": by_cause}
        if known:
            num = np.array([c == "numeric" for c in causes])
            out["discrete_flips"] = {"detections": int((~num).sum()) + len(unmatched),
                                     "by_kind": {k: int(sum(1 for c in causes if c == k)) + int(sum(1 for c in unmatched if c == k))
                                                 for k in ("centre_differs", "filter_flip", "member_flip", "draw_flip")}}
            if num.any():
                out["numeric_only"] = {"detections": int(num.sum()), "max_rel_dmu": float("%.3g" % rel_mu[num].max()),
                                       "max_abs_dmu_px": float("%.3g" % q("_dmu_px")[num].max()),
                                       "max_rms_dSigma": float("%.3g" % rel_sig[num].max()), "max_fro_dSigma": float("%.3g" % q("_fro_dsigma")[num].max()),
                                       "max_dscore": float("%.3g" % dsc[num].max()),
                                       "outside_1e-3": int((outside & num).sum()),
                                       # the rounds-4/5 covariance metric (floor = 1 % of the matrix's largest entry), not gated on
                                       "covariance_entries_1pct_floor": {"max": float("%.3g" % sig_1pct[num].max()), "outside_1e-3": int((sig_1pct[num] > 1e-3).sum())}}
    return out


def cpu_baseline(hw, n, frames, weights, anchors, seconds_budget=25.0, device_raw=None, seed=0, first_image_id=0, device_dets=None):
    """Reference-literal CPU timing with the oracle (kind='port'): PyTorch-CPU fp32 forward
    (oracle/torch_ref.py) + NumPy posterior / soft-NMS / clustering, all host cores.

    ``device_raw``: {mode name: (cls, box, cov)} raw head outputs the GPU produced for frame 0 with (seed, first_image_id).
    The CPU forward of frame 0 then runs with the SAME Philox dropout masks, so the two are directly comparable: the
    returned ``parity`` maps each mode to its max relative error / relative RMS distance against this CPU forward
    (north_star: "outputs match the CPU reference within 1e-3 rel on identical inputs, CPU baseline timed in the same run").
    ``device_dets``: {mode name: [(scores, means, covs, counts) of frame 0, of frame 1, ...]} the device's final detections from the same
    call; ``parity[mode]["detections"]`` compares frame 0's with this leg's own posterior -> soft-NMS -> cluster-and-fuse output, and
    ``parity[mode]["detections_all_frames"]`` is the statistic over EVERY frame this leg computes (each frame's forward then runs with
    that frame's Philox masks, generated outside the clock like frame 0's)."""
    import torch
    from oracle import bayes_od, clustering, geometry, network, nms, philox, torch_ref
    tw = torch_ref.prepare(weights)
    # big hosts lose to thread oversubscription on these small convs: probe a few pool sizes on one
    # head-tower conv (median of 3 after a warm-up) and keep the fastest
    import torch.nn.functional as F
    ncpu = os.cpu_count() or 1
    # (the probe walks ALL FIVE pyramid levels of one tower conv at the MC batch: on 128-core hosts the biggest pool wins the P3 map
    # alone and loses the whole forward to the 32x32 .. 4x4 maps, where its fork/join costs more than the conv -- 0.17 instead of 0.4-0.6
    # frames/s on round 3's boxes with the single-map probe)
    probe_xs = [torch.randn(n, 256, max(1, -(-hw[0] // (1 << l))), max(1, -(-hw[1] // (1 << l)))) for l in range(3, 8)]
    best = (float("inf"), 1)
    for cand in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(cand)
        with torch.no_grad():
            for px in probe_xs:
                F.conv2d(px, tw["pyramid_classification_0"][0], padding=1)
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                for px in probe_xs:
                    F.conv2d(px, tw["pyramid_classification_0"][0], padding=1)
                ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[1]
        if dt < best[0]:
            best = (dt, cand)
    threads = best[1]
    torch.set_num_threads(threads)
    P = anchors.shape[0] // 9
    n_cmp = max([len(v) for v in (device_dets or {}).values()] + [1 if device_raw else 0])      # frames with device outputs to compare

    def masks_of(frame):      # Philox masks of one frame, generated before its clock starts (the reference draws its masks inside the op)
        cache = {}

        def masks(s, lid):
            if (s, lid) not in cache:
                cache[(s, lid)] = philox.dropout_keep_mask(seed, first_image_id + frame, s, lid, P, 256, 0.3)
            return cache[(s, lid)]
        for s in range(n):
            for lid in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11):
                masks(s, lid)
        return masks
    done, t_total = 0, 0.0
    parts = {"forward": 0.0, "posterior": 0.0, "nms": 0.0, "cluster": 0.0}
    parity = {}
    cpu_dets0 = None
    per_frame = {mode: [] for mode in (device_dets or {})}
    while done < len(frames) and (done == 0 or t_total < seconds_budget):
        masks0 = masks_of(done) if done < n_cmp else None
        t0 = time.perf_counter()
        out = torch_ref.retinanet_forward(None, frames[done:done + 1], n, 8, prepared=tw, keep_masks=masks0)
        t1 = time.perf_counter()
        u = philox.categorical_uniforms(seed, first_image_id + done, anchors.shape[0])
        post = bayes_od.bayes_od_posterior(out, anchors, u, BAYES_CFG, use_full_covar=True, dtype=np.float32)
        t2 = time.perf_counter()
        corners = post["corners"].astype(np.float32)
        idx, _ = nms.soft_nms(corners, post["ranking"], 100, 0.5, 0.5)
        t3 = time.perf_counter()
        dets, iou = None, None
        if len(idx):
            iou = geometry.bbox_iou_vuvu(corners, corners)
            dets = clustering.bayes_od_clustering(post["counts"], post["means"], post["covs"], idx, iou, 0.5)
        t4 = time.perf_counter()
        cpu_ctx = ({"anchor_index": np.nonzero(post["keep"])[0], "centres": np.asarray(idx), "iou": iou, "counts": post["counts"]}
                   if iou is not None else None)
        if done == 0:
            cpu_dets0 = dets
        parts["forward"] += t1 - t0; parts["posterior"] += t2 - t1
        parts["nms"] += t3 - t2; parts["cluster"] += t4 - t3
        t_total += t4 - t0
        if done == 0 and device_raw:
            for mode, (cls, box, cov) in device_raw.items():
                errs = [_rel_errs(g, out[k]) for g, k in ((cls, "anchors_class_predictions"), (box, "anchors_box_predictions"),
                                                          (cov, "_covar_params"))]
                parity[mode] = {"max_rel_err": max(e[0] for e in errs), "rel_rms": max(e[1] for e in errs), "max_rel_err_strict": max(e[2] for e in errs)}
            for mode, dev in (device_dets or {}).items():
                parity.setdefault(mode, {})["detections"] = detection_parity(dev[0][:4], cpu_dets0)
        for mode, dev in (device_dets or {}).items():
            if done < len(dev):
                dctx = dict(dev[done][4]) if len(dev[done]) > 4 and dev[done][4] is not None else None
                per_frame[mode].append(detection_parity(dev[done][:4], dets, arrays=True, dev_ctx=dctx, cpu_ctx=dict(cpu_ctx) if cpu_ctx else None))
        done += 1
    for mode, pf in per_frame.items():
        parity.setdefault(mode, {})["detections_all_frames"] = detection_statistics(pf)
    base = {"value": done / t_total, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d frame(s) of %dx%d at N=%d, reference-literal (11*N head convs, no dedup), "
                      "torch %s fp32 CPU forward + NumPy Bayesian stages (Philox dropout masks of the compared frames generated outside the clock); seconds: %s" %
                      (done, hw[0], hw[1], n, torch.__version__,
                       {k: round(v, 2) for k, v in parts.items()})}
    return base, parity


class Telemetry(object):
    """Board power / shader clock / temperature of THIS rank's GPU sampled from the amdgpu hwmon files while the timed region runs
    (a thread reading three sysfs files every 25 ms: no HIP call, no effect on the streams).  The tower kernel runs at the board's
    power management limit -- boxes of the pool differ by a few per cent in the clock they sustain -- so the
    line records what this run's box did: a slow box and a regression are then distinguishable."""

    def __init__(self, device_index, hwmon_dir=None):
        self.dir, self.samples, self._stop, self._thread = hwmon_dir, [], False, None
        if hwmon_dir is not None:                # (tests: a directory with the three files)
            return
        try:
            import glob
            import torch
            props = torch.cuda.get_device_properties(device_index)
            want = None
            if hasattr(props, "pci_bus_id"):
                want = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
            cands = []
            for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
                if os.path.exists(os.path.join(d, "power1_input")) or os.path.exists(os.path.join(d, "power1_average")):
                    cands.append(d)
            for d in cands:
                if want and want in os.path.realpath(os.path.join(d, "..", "..")).lower():
                    self.dir = d
            if self.dir is None and len(cands) == 1:
                self.dir = cands[0]
            self.candidates = cands
        except Exception:                        # telemetry is optional: never fail the bench over it
            self.dir = None

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as fp:
                return float(fp.read().strip())
        except (OSError, ValueError):
            return None

    def _run(self):
        pw = "power1_input" if os.path.exists(os.path.join(self.dir, "power1_input")) else "power1_average"
        while not self._stop:
            self.samples.append((self._read(pw), self._read("freq1_input"), self._read("temp2_input")))
            time.sleep(0.025)

    def start(self):
        if self.dir is None:
            return
        import threading
        self.samples, self._stop = [], False
        self._thread = threading.Thread(target=self._run, daemon=True)
        self._thread.start()

    def stop(self):
        if self._thread is None:
            return None
        self._stop = True
        self._thread.join()
        self._thread = None
        col = lambda k: [s[k] for s in self.samples if s[k] is not None]
        pw, fq, tp = col(0), col(1), col(2)
        if not pw and not fq:
            return None
        cap = self._read("power1_cap")
        out = {"source": self.dir, "samples": len(self.samples), "interval_ms": 25}
        if pw:
            out.update({"board_power_w_mean": round(sum(pw) / len(pw) / 1e6, 1), "board_power_w_max": round(max(pw) / 1e6, 1)})
        if cap:
            out["board_power_cap_w"] = round(cap / 1e6, 1)
        if fq:
            out.update({"shader_clock_mhz_mean": round(sum(fq) / len(fq) / 1e6, 1), "shader_clock_mhz_min": round(min(fq) / 1e6, 1),
                        "shader_clock_mhz_max": round(max(fq) / 1e6, 1)})
        if tp:
            out["hotspot_temp_c_max"] = round(max(tp) / 1e3, 1)
        return out


def make_engine(hw, B, n, device, precision="bf16", weights=None, anchors=None, **kw):
    from bayes_od_rc_amd.engine import Engine, make_config
    eng = Engine(make_config(hw, batch=B, mc_samples=n, device=device, bayes_od_config=BAYES_CFG,
                             nms_config=NMS_CFG, use_full_covar=True, precision=precision, **kw))
    eng.load_weights(weights)
    if anchors is not None:
        eng.set_anchors(anchors)
    return eng


def frame_outputs(eng, i):
    """Detections of frame i of the last synchronous infer + what explain_detection needs from the device: the kept anchors' ids, class
    counts and posterior means, and the soft-NMS centres."""
    det = eng.get_detections(i)
    try:
        post = eng.get_posterior(i)
        ctx = {"anchor_index": post["anchor_index"], "counts": post["counts"], "means": post["means"], "centres": eng.get_nms(i)}
    except Exception:          # (a plan that keeps no posterior readable: the comparison then reports its causes as unexplained)
        ctx = None
    return det + (ctx,)


def raw_of_image0(eng):
    """Raw head outputs of image 0 of the last forward (zero-copy device views, one small D2H)."""
    from bayes_od_rc_amd import distributed as bdist
    v = bdist.raw_views(eng)            # (after an aggregating bod_infer this re-runs the raw flavour of the last tower launches ...
    eng.synchronize()                   #  ... on the engine's stream: wait for it before torch's stream copies)
    return tuple(v[k][0].cpu().numpy() for k in ("cls", "box", "cov"))


def timed_pipeline(eng, steps, warmup, fwd_only, B, first_id=0, image_buffer=None):
    """Single-rank pipelined loop (depth 2) over the device-resident batch: seconds for `steps` steps."""
    pending, host = [], [None, None]

    def step(i):
        if fwd_only:
            eng.forward(None, seed=0, first_image_id=first_id + i * B, image_buffer=image_buffer)
            return
        pending.append(eng.infer_async(None, seed=0, first_image_id=first_id + i * B, image_buffer=image_buffer))
        if len(pending) > 1:
            s = pending.pop(0)
            host[s] = eng.collect(s, host[s])

    def drain():
        while pending:
            s = pending.pop(0)
            host[s] = eng.collect(s, host[s])
        eng.synchronize()
    for i in range(warmup):
        step(i)
    drain()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    drain()
    return time.perf_counter() - t0


def pyramid_pixels(hw):
    """Pixels of the p3..p7 pyramid of an H x W frame (SAME stride-2 chain from the stride-8 map)."""
    lh, lw = -(-hw[0] // 8), -(-hw[1] // 8)
    p = 0
    for _ in range(5):
        p += lh * lw
        lh, lw = -(-lh // 2), -(-lw // 2)
    return p


def train_step_ms(device):
    """BASELINE config 5: (ms per training step, total loss) of ResNet-101 RetinaNet + full-covariance loss at the yaml's minibatch
    of three 512x512 frames: 3 untimed steps, 10 timed."""
    from bayes_od_rc_amd import constants, synthetic
    from bayes_od_rc_amd.engine import Engine, make_config
    from bayes_od_rc_amd.run_training import synthetic_samples
    hw, B = (512, 512), 3
    samples = synthetic_samples(B, hw, ANCHOR_CFG, 7)
    eng = Engine(make_config(hw, batch=B, mc_samples=1, device=device, training=True, backbone_depth=101))
    eng.load_weights(synthetic.make_weights(depth=101))
    eng.set_anchors(np.asarray(samples[0][constants.ANCHORS_KEY], np.float32))
    st = lambda k: np.stack([s[k] for s in samples])
    eng.upload_images(st(constants.IMAGE_NORMALIZED_KEY))
    targets = (st(constants.ANCHORS_CLASS_TARGETS_KEY), st(constants.ANCHORS_BOX_TARGETS_KEY),
               st(constants.POSITIVE_ANCHORS_MASK_KEY), st(constants.NEGATIVE_ANCHOR_MASK_KEY))
    for i in range(3):
        eng.train_step(None, *targets, seed=1, first_image_id=i * B)
    steps = 10
    t0 = time.perf_counter()
    for i in range(steps):
        loss = eng.train_step(None, *targets, seed=1, first_image_id=(3 + i) * B)
    dt = time.perf_counter() - t0
    eng.close()
    return dt / steps * 1e3, float(loss["total_loss"])


def secondary_configs(device, weights, lo):
    """BASELINE.json configs 2, 4 (its geometry on one GPU) and 5, each on its own handle, a few steps each."""
    import torch
    from bayes_od_rc_amd import constants, synthetic
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    from bayes_od_rc_amd.engine import Engine, make_config
    from bayes_od_rc_amd.run_training import synthetic_samples
    out = []

    def entry(workload, ms, value, unit, gflop_per_unit, units_per_step, note=None):
        ach = gflop_per_unit * units_per_step / ms          # GFLOP / ms = TFLOP/s
        e = {"workload": workload, "ms_per_step": round(ms, 3), "value": round(value, 2), "unit": unit,
             "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(ach / PEAK_BF16_TFLOPS, 4), "algorithmic_gflop_per_unit": round(gflop_per_unit, 1)}}
        if note:
            e["note"] = note
        out.append(e)
    gen = FpnAnchorGenerator(ANCHOR_CFG)
    # ---- config 2: ResNet-50 RetinaNet, 512x512, N=1 (no MC): network forward only (raw head outputs)
    hw, B, n = (512, 512), 512, 1          # (512 frames per step like the headline: +1.4 % over 256, launch ramps and tails amortise)
    eng = make_engine(hw, B, n, device, weights=weights)
    eng.upload_images(synthetic.make_frames(B, hw[0], hw[1], seed=lo))
    steps = 8
    dt = timed_pipeline(eng, steps, 2, True, B)
    entry("BASELINE config 2: ResNet-50 RetinaNet, 512x512, N=1 (no MC), forward only (raw head outputs), %d frames/step" % B,
          dt / steps * 1e3, B * steps / dt, "images/sec", image_gflop(hw, eng.P, 1), B)
    eng.close()
    # ---- config 4's geometry on ONE GPU: 384x1248 (KITTI), N=30, full pipeline
    hw, B, n = (384, 1248), 96, 30          # (96 frames per step: +2.7 % over 32; ~85 GB)
    eng = make_engine(hw, B, n, device, weights=weights, anchors=gen.generate_all((hw[0], hw[1], 3)))
    eng.upload_images(synthetic.make_frames(B, hw[0], hw[1], seed=lo))
    steps = 6
    dt = timed_pipeline(eng, steps, 2, False, B)
    entry("BASELINE config 4 geometry on one GPU: ResNet-50 RetinaNet + covar head, N=30 MC-dropout, KITTI 384x1248, full BayesOD "
          "pipeline, %d frames/step" % B, dt / steps * 1e3, B * steps / dt, "images/sec", image_gflop(hw, eng.P, 30), B)
    eng.close()
    # ---- the frame sizes the reference really runs (it never sees 512x512 or 384x1248): native BDD 720x1280
    # (bdd_dataset_handler.py:128-139) and KITTI resized / padded to 512x1696 (kitti_dataset_handler.py:125-132), N=10, full pipeline
    for hw, name in (((720, 1280), "BDD frames at their native 720x1280"), ((512, 1696), "KITTI frames resized to 512x1696")):
        B, n = 128, 10                       # (128 frames per step: +1.2-1.4 % over 64)
        eng = make_engine(hw, B, n, device, weights=weights, anchors=gen.generate_all((hw[0], hw[1], 3)))
        eng.upload_images(synthetic.make_frames(B, hw[0], hw[1], seed=lo))
        steps = 5
        dt = timed_pipeline(eng, steps, 2, False, B)
        entry("the reference's real geometry: %s (SURVEY F7), ResNet-50 RetinaNet + covar head, N=10 MC-dropout, full BayesOD pipeline, "
              "%d frames/step" % (name, B), dt / steps * 1e3, B * steps / dt, "images/sec", image_gflop(hw, eng.P, n), B)
        eng.close()
    # ---- config 5: ResNet-101 RetinaNet + full-covariance loss, one training step (the yaml's minibatch of 3).  The step is a chain of
    # ~1 200 small dependent launches: its time is launch latency, and that depends on what the process did before -- measured on MI355X
    # (.ab/train_probe.py in round 4): 10.9 ms in a fresh process, 12.4 ms once a torch HIP context exists, 13.4 ms after a large handle
    # was created and destroyed.  A training run is its own process: the step is therefore timed in a CHILD process (this file with
    # --train-step-probe), in-process only if that fails.
    hw, B = (512, 512), 3
    ms = loss = None
    how = "timed in a fresh child process"
    try:
        import subprocess
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--train-step-probe", str(device)], capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith("TRAIN_STEP_MS ")]
        if r.returncode == 0 and line:
            ms, loss = float(line[-1].split()[1]), float(line[-1].split()[2])
    except Exception:
        ms = None
    if ms is None:
        ms, loss = train_step_ms(device)
        how = "timed in this process (the child process failed)"
    P101 = pyramid_pixels(hw)
    # forward + input-gradient + weight-gradient GEMMs = 3 x the forward's conv FLOPs (no de-duplication at N=1)
    entry("BASELINE config 5: ResNet-101 RetinaNet + full-covariance loss, one training step (forward, backward, clip, Adam), "
          "bf16, %d frames of 512x512 (the yaml's minibatch)" % B, ms, B / (ms * 1e-3), "images/sec",
          3.0 * image_gflop(hw, P101, 1, depth=101, dedup=False), B,
          note="algorithmic GFLOP = 3 x forward conv FLOPs (forward, dgrad, wgrad GEMMs); total_loss %.3f; %s" % (loss, how))
    torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per GPU per step (512: ~140 GB of the 288 GB; the tower "
                    "launches then hold 436 tiles per CU and the step's fixed costs -- launch ramps and tails -- are amortised: "
                    "+5 %% over 64, +1.1 %% over 256 frames/s on one box, tower roofline 0.495 -> 0.501)")
    ap.add_argument("--mc", type=int, default=10)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip parity_mode / secondary / value_with_h2d (headline only)")
    ap.add_argument("--calibrate", action="store_true", help="print the cls foreground bias for M~1000")
    ap.add_argument("--fg-bias", type=float, default=CALIBRATED_FG_BIAS)
    ap.add_argument("--forward-only", action="store_true",
                    help="time RetinaNetModel.call only (BASELINE config 2: raw head outputs); implied by --mc 1, "
                         "where the Bayesian stages are undefined (sample covariance divides by N-1)")
    ap.add_argument("--precision", choices=("bf16", "fp32", "bf16x3", "f16mx", "f16mx4"), default="bf16",
                    help="bf16 = throughput path (BASELINE.json north_star); bf16x3 = its 1e-3 end-to-end parity mode; "
                         "fp32 = exact-fp32 MFMA")
    ap.add_argument("--train-step-probe", type=int, default=None, metavar="DEVICE",
                    help="(internal) time BASELINE config 5's training step on DEVICE in this fresh process and print TRAIN_STEP_MS")
    args = ap.parse_args()
    if args.train_step_probe is not None:
        ms, loss = train_step_ms(args.train_step_probe)
        print("TRAIN_STEP_MS %.4f %.6f" % (ms, loss))
        return

    import torch
    import torch.distributed as dist
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd import distributed as bdist
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # BOD_BENCH_BACKEND=gloo + BOD_BENCH_SHARE_GPU=1: exercise the N>1 code path on a ONE-GPU box (all ranks on
    # device 0, records gathered through host memory) -- tests/test_gpu_pipeline.py; the driver's runs use RCCL.
    backend = os.environ.get("BOD_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if os.environ.get("BOD_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    if local_rank >= ndev:
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, ndev))
    print("# bench rank %d/%d: local_rank %d of %d visible GPU(s), backend %s" % (rank, world, local_rank, ndev, backend),
          file=sys.stderr, flush=True)
    torch.cuda.set_device(local_rank)
    # BOD_BENCH_FORCE_DIST=1: initialise the process group and run EVERY collective of the N>1 path (gather of the records on
    # device tensors, barrier, max-reduce of the time, all-gather of the per-rank rates) even with one rank -- the only way to run
    # the RCCL branch on a one-GPU box (tests/test_gpu_pipeline.py::test_bench_collectives_run_under_rccl_with_one_rank)
    global USE_DIST
    USE_DIST = world > 1 or os.environ.get("BOD_BENCH_FORCE_DIST") == "1"
    if USE_DIST:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # every collective is bounded: a dead peer turns into an exception on the survivors, not a hang
        timeout = datetime.timedelta(seconds=int(os.environ.get("BOD_BENCH_COLLECTIVE_TIMEOUT_S", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=timeout)
        else:
            dist.init_process_group(backend, timeout=timeout)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    hw, n, B = (args.height, args.width), args.mc, args.batch
    fwd_only = args.forward_only or n < 2
    out = {"metric": METRIC, "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
           "data": "synthetic"}
    if world > 1 and rank == 0:
        # torch.distributed.run answers a dead worker by sending SIGTERM to the survivors -- often before rank 0's own bounded
        # collective has failed.  Rank 0 must still leave its ONE JSON line: report the termination as the error and go.
        import signal

        def on_term(signum, frame):
            out["error"] = "terminated by the launcher (signal %d): a peer rank failed" % signum
            out["value"] = None
            print(json.dumps(out), flush=True)
            os._exit(1)
        signal.signal(signal.SIGTERM, on_term)
    try:
        run(args, out, rank, world, local_rank, backend, hw, n, B, fwd_only)
    except BaseException as exc:            # a peer died / a collective timed out / a HIP error: rank 0 still reports, non-zero exit
        if isinstance(exc, SystemExit) and exc.code in (0, None):
            raise
        traceback.print_exc(file=sys.stderr)
        if rank == 0:
            out["error"] = "%s: %s" % (type(exc).__name__, exc)
            print(json.dumps(out), flush=True)
        # never re-exec, never wait for the dead peer: leave without the collective teardown
        os._exit(1)
    if USE_DIST:
        dist.destroy_process_group()


def run(args, out, rank, world, local_rank, backend, hw, n, B, fwd_only):
    import torch
    import torch.distributed as dist
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd import distributed as bdist
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator

    weights = synthetic.make_weights(cls_fg_bias=args.fg_bias)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
    eng = make_engine(hw, B, n, local_rank, precision=args.precision, weights=weights, anchors=anchors)
    # this rank's shard of a (world*B)-frame synthetic clip, resident in HBM before timing starts
    lo, hi = bdist.shard_range(world * B, world, rank)
    frames = synthetic.make_frames(hi - lo, hw[0], hw[1], seed=lo)
    eng.upload_images(frames)

    if args.calibrate:
        eng.forward(None, seed=0, first_image_id=lo)
        cls = eng.get_raw()[0]
        print("calibrated cls foreground bias:", synthetic.calibrate_fg_bias(cls[0], args.fg_bias))
        return

    # BOD_BENCH_FORCE_GATHER=1: run the N>1 record path (device-side pack, gather, host copy on rank 0) at world size 1 too,
    # to measure what it costs per step on one GPU
    force_gather = os.environ.get("BOD_BENCH_FORCE_GATHER") == "1" or (USE_DIST and world == 1)
    views = [bdist.torch_views(eng, s) for s in (0, 1)] if (world > 1 or force_gather) else None
    host_out = [None, None]
    gathered = None

    def collect(slot):
        """Detections of the batch in `slot` -> host (rank 0 receives every rank's records)."""
        nonlocal gathered
        if world > 1 or force_gather:
            eng.wait_slot(slot)
            v = views[slot]
            rec = bdist.pack_records(v["num"], v["scores"], v["means"], v["covs"], v["counts"])
            allrec = bdist.gather_records(rec if backend == "nccl" else rec.cpu(), dst=0, always=USE_DIST)
            if rank == 0:
                gathered = allrec.cpu()
            # the pack kernels read the engine's slot buffers on torch's stream: they must have finished before the engine's
            # side stream may rewrite this slot (two infer_async calls from now)
            torch.cuda.current_stream().synchronize()
        else:
            host_out[slot] = eng.collect(slot, host_out[slot])

    pending = []

    def step(i):
        if fwd_only:
            eng.forward(None, seed=0, first_image_id=lo + i * world * B)   # asynchronous on the engine stream
            return
        # software pipeline of depth 2: batch i's NMS/cluster-fuse (side stream) and its collection
        # overlap batch i+1's convolutions; every enqueued batch is collected inside the timed region
        pending.append(eng.infer_async(None, seed=0, first_image_id=lo + i * world * B))
        if len(pending) > 1:
            collect(pending.pop(0))

    def drain():
        while pending:
            collect(pending.pop(0))

    def fence():
        eng.synchronize()
        torch.cuda.synchronize()
        if USE_DIST:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    # fault injection for tests/test_gpu_pipeline.py: the named rank dies here, after the warm-up -- the survivors' next collective
    # must turn into an error report (rank 0 prints the JSON line with "error", every rank exits non-zero), never a hang
    if os.environ.get("BOD_BENCH_FAULT_RANK") == str(rank):
        os._exit(17)
    fence()
    # HIP events around every head-tower launch (and every posterior) of the timed steps, recorded on the streams the
    # kernels run on; read back after the closing fence
    eng.profile_begin(which=1)         # 1: the launches of the dominant kernel (row-reuse tower kernel, tower layers 1..3)
    telemetry = Telemetry(local_rank) if rank == 0 else None
    if telemetry:
        telemetry.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    t_local = time.perf_counter() - t0          # this rank's own clock, before the closing barrier
    fence()
    elapsed = time.perf_counter() - t0
    out["telemetry"] = telemetry.stop() if telemetry else None
    per_rank = [B * args.steps / t_local]
    if USE_DIST:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([per_rank[0]], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [float(x.item()) for x in allr]
    total_images = world * B * args.steps
    value = total_images / elapsed
    out["value"] = round(value, 3)
    out["ms_per_step"] = round(elapsed / args.steps * 1e3, 3)
    kept = [] if fwd_only else eng.num_kept()

    # ---- roofline of the dominant kernel (head 3x3 implicit-GEMM): the launches of the timed region itself
    # The dominant kernel is ONE kernel symbol -- conv_igemm_kernel<256,256,2,4,0,true>, the row-reuse loop that runs tower
    # layers 1..3 (4 launches per step -- layer 2 is two, DESIGN.md 5.2 --, 8 of the 11 de-duplicated head convs) -- so that its average launch duration here
    # and in the rocprofv3 --kernel-trace summary under profiles/ are the same quantity.  The first tower layer (N-way
    # dropout fan-out, a different instantiation) is timed in three extra steps below and reported beside it.
    def more_steps(k):
        for i in range(k):
            if fwd_only:
                eng.forward(None, seed=0, first_image_id=lo + i * world * B)
            else:
                eng.infer(None, seed=0, first_image_id=lo + i * world * B)

    conv_flops = head_conv_flops(eng.P)
    out_flops = head_out_flops(eng.P)
    prof_steps = args.steps
    prof = eng.profile_end()
    tower_only = prof["head_conv_launches"] > 0
    if tower_only:
        # SURVEY.md 8d, de-duplicated heads: N * (8 * 6.436 + 0.553) GFLOP per 512x512 image for these launches -- the eight
        # 3x3 convs of tower layers 1..3 and the three 1x1 output convs fused into their epilogues
        # (the plan fuses them whenever the launch uses the full-cout tile: from about 8 frames per step on; below that
        # they are separate small launches and are not counted here)
        algo_flops = n * 8 * conv_flops * B * prof_steps
        layers = "layers 1-3"
        if n == 1 and eng.plan_info()["fan_out_row_reuse"]:
            # N = 1: no fan-out -- the first tower layer is a plain three-head launch of the same kernel symbol (round 4)
            algo_flops += 3 * conv_flops * B * prof_steps
            layers = "layers 0-3 (N = 1: no fan-out launch)"
        kernel_name = "conv_igemm_kernel<256,256,2,4,0,true%s> (head towers, 3x3 256->256, %s; mid-tile-barrier loop)" % (
            ",SPLIT: 3 MFMA products per MAC" if args.precision == "bf16x3" else "", layers)
        if args.precision == "f16mx" and eng.plan_info()["tower_mx"]:
            kernel_name = ("conv_igemm_mx_kernel<1> (head towers, 3x3 256->256, %s; f16mx: one f16 product + half a block-scaled e2m3 product "
                           "per multiplication = 1.5 bf16-product equivalents)" % layers)
        if args.precision == "f16mx4" and eng.plan_info()["tower_mx"]:
            kernel_name = ("conv_igemm_mx_kernel<3> (head towers, 3x3 256->256, %s; f16mx4: one f16 product + the two cross terms as one "
                           "block-scaled e2m1 product of twice the channels: 54 K-tiles and 768 staged bytes per row instead of 72 and 1 024)" % layers)
    else:                                # no row-reuse kernel in the plan (fp32 / bf16x3 mode, BOD_CONV_XREUSE=0): all head 3x3 launches, three extra steps
        prof_steps = max(1, min(3, args.steps))
        eng.profile_begin(which=0)
        more_steps(prof_steps)
        prof = eng.profile_end()
        algo_flops = head_flops_per_image(eng.P, n) * B * prof_steps
        kernel_name = {"fp32": "conv_igemm_f32_kernel", "bf16x3": "conv_igemm_kernel<..., SPLIT> (3 MFMA products per MAC)",
                       "f16mx": "conv_igemm_kernel<..., SPLIT> (3 MFMA products per MAC; BOD_TOWER_MX=0)",
                       "f16mx4": "conv_igemm_kernel<..., SPLIT> (3 MFMA products per MAC; BOD_TOWER_MX=0)",
                       "bf16": "conv_igemm_kernel"}[args.precision] + " (head towers, 3x3 256->256)"
    if abs(prof["head_conv_flops"] - algo_flops) / algo_flops > 1e-6:       # the plan fused the 1x1 output convs into these launches
        algo_flops += n * out_flops * B * prof_steps
    launches = max(1, prof["head_conv_launches"])
    # algorithmic FLOPs: the de-duplicated head convs these launches issue
    assert abs(algo_flops - prof["head_conv_flops"]) / algo_flops < 1e-6
    achieved = algo_flops / (prof["head_conv_ms"] * 1e-3) / 1e12
    fan_out = None
    if tower_only:
        eng.profile_begin(which=2)
        more_steps(3)
        fo = eng.profile_end()
        if fo["head_conv_launches"] > 0:
            fan_out = {"kernel": ("conv_igemm_mx_kernel<%d>" % ({"f16mx": 1, "f16mx4": 3}[args.precision] if os.environ.get("BOD_MX_LAYER0", "1") != "0" else 2) if args.precision in ("f16mx", "f16mx4") else "conv_igemm_kernel<256,256,2,4,5,true>") + " (tower layer 0, %d-way dropout fan-out)" % n,
                       "achieved": round(fo["head_conv_flops"] / (fo["head_conv_ms"] * 1e-3) / 1e12, 2),
                       "avg_launch_ms": round(fo["head_conv_ms"] / fo["head_conv_launches"], 4), "launches_per_step": 1}
        eng.profile_begin(which=0)
        eng.profile_end()
    # algorithmic MFLOP per MAC-FLOP is 1 in every mode; the bf16x3 mode ISSUES three MFMA products per MAC, so its
    # matrix-pipe work is 3 x `achieved` (reported as `mfma_issue_tflops`)
    peak = PEAK_F32_MFMA_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    # MI355X_MICROARCH.md prescribes) when they were taken on this exact configuration, else null
    traffic, traffic_source = None, None
    for pmc_file in ("round6_head_conv_pmc.json", "round5_head_conv_pmc.json", "round4_head_conv_pmc.json", "round3_head_conv_pmc.json", "round2_head_conv_pmc.json", "round1_head_conv_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", pmc_file)) as fp:
                pmc = json.load(fp)
            c = pmc["config"]
            if (c["height"], c["width"], c["mc_samples"], c["batch"]) == (hw[0], hw[1], n, B) and args.precision == "bf16":
                # the per-sample tower launches = the row-reuse kernel's production symbol <256,256,2,4,0,true,...>
                is_tower = lambda l: "4, 0, true" in l["kernel"] or l["kernel"].rstrip().endswith("true>(ConvArgs)") and ", 5, true" not in l["kernel"]
                sel = [l for l in pmc["launches"] if is_tower(l) == tower_only or not tower_only]
                traffic = int(sum(l["hbm_read_bytes_corrected"] + l["hbm_write_bytes"] for l in sel) / len(sel))
                traffic_source = "profiles/" + pmc_file
                break
        except (OSError, KeyError, ValueError, ZeroDivisionError):
            pass
    roofline = {"bound": "mfma", "kernel": kernel_name,
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                # `traffic` is not measured in this run: it is read from the committed rocprofv3 --pmc passes of the same
                # configuration (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes + WRITE_SIZE, per launch)
                "traffic_source": traffic_source,
                "avg_launch_ms": round(prof["head_conv_ms"] / launches, 4), "launches_per_step": launches // prof_steps,
                "share_of_step": round(prof["head_conv_ms"] / prof_steps / (elapsed / args.steps * 1e3), 3)}
    if args.precision == "bf16x3":
        roofline["mfma_issue_tflops"] = round(3 * achieved, 2)
    if args.precision in ("f16mx", "f16mx4"):
        roofline["mfma_issue_tflops_bf16_equivalent"] = round(1.5 * achieved, 2)
    if fan_out:
        roofline["other_head_launch"] = fan_out
        # "the conv heads" as a whole: the de-duplicated 3 + 8 N head convs and the N 1x1 sets of a step = the tower launches of
        # the timed region + the fan-out launch (timed in the three extra steps above)
        heads_ms = prof["head_conv_ms"] / prof_steps + fan_out["avg_launch_ms"]
        heads_flops = (algo_flops / prof_steps + 3 * conv_flops * B) / 1e12
        roofline["heads_total"] = {"achieved": round(heads_flops / (heads_ms * 1e-3), 2), "frac": round(heads_flops / (heads_ms * 1e-3) / peak, 4),
                                   "ms_per_step": round(heads_ms, 3), "tflop_per_step": round(heads_flops, 2)}
    # BASELINE.md section 3, "reported beside it": the whole pipeline's de-duplicated conv FLOPs per image x images/sec
    # (backbone + FPN 49.05 GFLOP at 512x512, linear in the pixel count; heads 3 + 8 N convs + N 1x1 sets)
    img_gflop = image_gflop(hw, eng.P, n)
    roofline["pipeline"] = {"dedup_gflop_per_image": round(img_gflop, 1),
                            "achieved": round(img_gflop * value / world / 1e3, 2), "unit": "TFLOP/s per GPU",
                            "frac": round(img_gflop * value / world / 1e3 / peak, 4)}
    # per-anchor latency of the aggregate / posterior stage (a9-a11): HIP events around the stage's launches of the timed
    # steps (main stream, between one batch's convolutions and the next)
    post_us_per_anchor = prof["posterior_ms"] * 1e3 / max(1, prof["posterior_launches"]) / (B * eng.A)

    # where a step's time goes on the main stream (soft-NMS, cluster-fuse and the D2H of the records run on the side stream
    # underneath the next batch); per-layer detail: BOD_TRACE_OPS=k
    step_ms = elapsed / args.steps * 1e3
    stages = {"head_towers_layers_1_3": round(prof["head_conv_ms"] / prof_steps, 3) if tower_only else None,
              "head_tower_layer_0_fan_out": round(fan_out["avg_launch_ms"], 3) if fan_out else None,
              "posterior": round(prof["posterior_ms"] / max(1, prof["posterior_launches"]), 3) if prof["posterior_launches"] else None}
    known = sum(v for v in stages.values() if v)
    stages["stem_backbone_fpn_and_gaps"] = round(step_ms - known, 3) if tower_only else None

    out["config"] = {"workload": "ResNet-50 RetinaNet + covar head, N=%d MC-dropout, %dx%d, %s"
                                 % (n, hw[0], hw[1], "forward only (raw head outputs, BASELINE config 2)" if fwd_only else
                                    "full BayesOD pipeline (forward+posterior+soft-NMS+cluster-fuse)"),
                     "frames_per_gpu_per_step": B, "global_batch": world * B, "mc_samples": n,
                     "anchors": eng.A, "kept_anchors_M": [int(k) for k in kept[:4]],
                     "parallelism": ("image-sharded x%d, one RCCL gather/step" % world) if world > 1 else "one GPU (no exchange; N > 1: image-sharded, one RCCL gather/step)",
                     "per_rank_images_per_sec": [round(v, 1) for v in per_rank],
                     # valid detections in the records rank 0 received from each rank in the last gathered step (N > 1 path)
                     "gathered_detections_per_rank": ([int(g[:, :, 0].sum().item()) for g in gathered] if gathered is not None else None),
                     "visible_gpus": torch.cuda.device_count(), "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                     # posterior launches only; the default run (extras) replaces both fields with the stage's full cost: posterior +
                     # the A/B-measured share of the tower launches that is the fused MC aggregation (see below)
                     "per_anchor_covariance_latency_ns": round(post_us_per_anchor * 1e3, 4),
                     "per_anchor_covariance_latency": {"posterior_only_ns": round(post_us_per_anchor * 1e3, 4),
                                                       "note": "posterior launches only: the MC aggregation runs inside the tower "
                                                               "epilogues and is not in this figure (A/B skipped in this invocation)"},
                     "stages_ms_per_step": stages}
    out["roofline"] = roofline

    extras = rank == 0 and world == 1 and not args.no_secondary and not fwd_only and args.precision == "bf16"
    device_raw, device_dets = {}, {}
    if extras:
        # ---- value_with_h2d: the same steps with the frames crossing PCIe EVERY step as uint8 (a quarter of the fp32 bytes),
        # copy + device preprocessing on the handle's copy stream, overlapped with the previous step's convolutions
        pinned = torch.empty((2, B, hw[0], hw[1], 3), dtype=torch.uint8).pin_memory()
        rng = np.random.default_rng(lo)
        pinned[0].numpy()[...] = rng.integers(0, 256, size=(B, hw[0], hw[1], 3), dtype=np.uint8)
        pinned[1].copy_(pinned[0].flip(0))
        clips = [pinned[0].numpy(), pinned[1].numpy()]
        h2d_steps, pend, host2 = max(4, min(10, args.steps)), [], [None, None]
        eng.upload_frames_u8_async(clips[0], 0)
        for i in range(2):                                   # warm-up (allocates the second image buffer)
            pend.append(eng.infer_async(None, seed=0, first_image_id=lo + i * B, image_buffer=i & 1))
            eng.upload_frames_u8_async(clips[(i + 1) & 1], (i + 1) & 1)
            if len(pend) > 1:
                s = pend.pop(0); host2[s] = eng.collect(s, host2[s])
        while pend:
            s = pend.pop(0); host2[s] = eng.collect(s, host2[s])
        eng.synchronize()
        t0 = time.perf_counter()
        for i in range(h2d_steps):
            pend.append(eng.infer_async(None, seed=0, first_image_id=lo + i * B, image_buffer=i & 1))
            eng.upload_frames_u8_async(clips[(i + 1) & 1], (i + 1) & 1)     # next step's frames, under this step's convolutions
            if len(pend) > 1:
                s = pend.pop(0); host2[s] = eng.collect(s, host2[s])
        while pend:
            s = pend.pop(0); host2[s] = eng.collect(s, host2[s])
        eng.synchronize()
        dt = time.perf_counter() - t0
        out["value_with_h2d"] = {"value": round(B * h2d_steps / dt, 2), "unit": "images/sec", "steps": h2d_steps,
                                 "ms_per_step": round(dt / h2d_steps * 1e3, 3),
                                 "input": "uint8 RGB frames from pinned host memory every step (%.0f MB/step), bod_upload_frames_u8_async: "
                                          "copy stream + device preprocessing, overlapped with the previous step" % (B * hw[0] * hw[1] * 3 / 1e6)}
        del pinned, clips
        # raw head outputs of frame 0 with (seed 0, image id lo) for the parity figures of the CPU leg
        eng.upload_images(frames)
        eng.infer(None, seed=0, first_image_id=lo)
        device_dets["bf16"] = [frame_outputs(eng, i) for i in range(min(B, N_CMP_FRAMES))]
        device_raw["bf16"] = raw_of_image0(eng)          # (re-runs the raw flavour of the last tower launches: same Philox streams)
        # ---- what the MC aggregation (a9-a10) costs where it now lives, inside the tower epilogues: the same steps on a handle
        # planned WITHOUT it (raw [B,N,A,.] tensors + the posterior's own loops over the samples), same box, same frames
        eng.close()
        del eng
        ab_steps = 6

        def towers_of(fused):
            """tower launches + posterior, ms per step, of `ab_steps` pipelined steps on a fresh handle planned with / without the fused aggregation"""
            if not fused:
                os.environ["BOD_FUSE_AGGREGATION"] = "0"
            try:
                e = make_engine(hw, B, n, local_rank, precision=args.precision, weights=weights, anchors=anchors)
            finally:
                os.environ.pop("BOD_FUSE_AGGREGATION", None)
            e.upload_images(frames)
            timed_pipeline(e, 1, 1, False, B, first_id=lo)
            e.profile_begin(which=1)
            timed_pipeline(e, ab_steps, 0, False, B, first_id=lo)
            p_ = e.profile_end()
            e.close()
            return p_["head_conv_ms"] / ab_steps, p_["posterior_ms"] / max(1, p_["posterior_launches"])
        # off / on / off / on on fresh handles, back to back: the difference of the means is the aggregation's share of the tower launches,
        # the spread between the two runs of one plan is what this box's run-to-run noise allows to be said about it
        t_off1, post_off = towers_of(False)
        t_on1, _ = towers_of(True)
        t_off2, _ = towers_of(False)
        t_on2, _ = towers_of(True)
        towers_on, towers_off = 0.5 * (t_on1 + t_on2), 0.5 * (t_off1 + t_off2)
        noise = max(abs(t_on1 - t_on2), abs(t_off1 - t_off2))
        post_on = prof["posterior_ms"] / max(1, prof["posterior_launches"])
        anchors_per_step = B * out["config"]["anchors"]
        in_tower = towers_on - towers_off                  # signed: ms per step the fused aggregation adds to (or takes off) the tower launches
        bound = max(0.0, in_tower) + noise                 # what the aggregation costs inside the tower launches AT MOST
        algo_bytes = (n * (4 + 10 + 8) * 4 + 16 + (4 + 16 + 8) * 4) * anchors_per_step     # 49.5 MB per 512x512 image at N=10
        out["config"]["per_anchor_covariance_latency_ns"] = round((post_on + bound) * 1e6 / anchors_per_step, 4)
        out["config"]["per_anchor_covariance_latency"] = {
            "definition": "UPPER BOUND of stage a9-a11 (per-anchor mean / 4x4 covariance over the MC samples, aleatoric mix, prior fusion) = "
                          "posterior launches of the timed region + at most what the fused MC aggregation adds to the tower launches "
                          "(mean of two %d-step runs with it - mean of two without it on fresh handles, interleaved, same box, + the larger "
                          "spread between two runs of one plan), / (frames x anchors).  A latency, not an HBM-roofline statement: the "
                          "per-sample tensors the stage's algorithmic bytes describe never reach HBM in this plan" % ab_steps,
            "posterior_only_ns": round(post_on * 1e6 / anchors_per_step, 4),
            "aggregation_in_tower_epilogues_ms_per_step": {"signed_difference": round(in_tower, 3), "run_to_run_spread": round(noise, 3),
                                                           "at_most": round(bound, 3)},
            "towers_ms_per_step": {"fused_aggregation": [round(t_on1, 3), round(t_on2, 3)], "raw_tensors": [round(t_off1, 3), round(t_off2, 3)]},
            # the kernel-quality figure of the stage: the posterior that WALKS the raw [B,N,A,.] tensors (plan without the fusion)
            "unfused_stage_ns": round(post_off * 1e6 / anchors_per_step, 4),
            # SURVEY 8d's bytes over the unfused stage's time: a RATE, not a roofline fraction -- since the compaction kernel (round 4)
            # the per-anchor fusion reads box / covariance samples of the ~2 % kept anchors only, so the stage does not move those bytes
            # (measured FETCH + WRITE of its kernels: profiles/round5_posterior_pmc.json when taken)
            "unfused_stage_algorithmic_TBps": round(algo_bytes / (post_off * 1e-3) / 1e12, 2),
            "unfused_stage_measured_hbm_bytes": _posterior_pmc_bytes(B, n, hw),
            "algorithmic_bytes_per_step": int(algo_bytes),
            # (SURVEY 8d's bytes / this stage's time; above 1 by construction once the bytes are not moved: a ratio, not a fraction of peak)
            "algorithmic_bytes_per_second_over_8TBs": round(algo_bytes / ((post_on + bound) * 1e-3) / 8e12, 3)}
    else:
        eng.close()
        del eng

    if extras:
        # ---- parity_mode: the precision modes in which the pipeline meets north_star's 1e-3 END TO END, timed in the same run on the same
        # frames.  f16mx (round 5) = bf16x3 with the head towers -- 80 % of bf16x3's step -- on one f16 product + half a block-scaled e2m3
        # product per multiplication instead of three bf16 products; bf16x3 (rounds 2-4) beside it as `parity_mode_bf16x3`.
        Bp = min(B, 256)                  # the production data path: fused 1x1 + MC aggregation, no [B,N,A,.] tensors (92 GB of pair planes at 256 frames)
        for mode in PARITY_MODES:
            key = "parity_mode_" + mode
            engp = make_engine(hw, Bp, n, local_rank, precision=mode, weights=weights, anchors=anchors)
            engp.upload_images(frames[:Bp])
            engp.infer(None, seed=0, first_image_id=lo)
            device_dets[mode] = [frame_outputs(engp, i) for i in range(min(Bp, N_CMP_FRAMES))]
            device_raw[mode] = raw_of_image0(engp)
            p_steps = 5
            timed_pipeline(engp, 2, 0, False, Bp, first_id=lo)          # warm-up
            engp.profile_begin(which=1 if mode in ("f16mx", "f16mx4") else 0)      # HIP events on the engine's own stream around every launch of the dominant kernel
            dt = timed_pipeline(engp, p_steps, 0, False, Bp, first_id=lo)
            pprof = engp.profile_end()
            products = 1.5 if (mode in ("f16mx", "f16mx4") and engp.plan_info()["tower_mx"]) else 3.0
            out[key] = {"precision": mode, "images_per_sec": round(Bp * p_steps / dt, 2),
                        "ms_per_step": round(dt / p_steps * 1e3, 3), "frames_per_step": Bp, "steps": p_steps,
                        "fraction_of_headline": round(Bp * p_steps / dt / value, 4),
                        "pipeline_tflops": round(image_gflop(hw, engp.P, n) * Bp * p_steps / dt / 1e3, 2),
                        "tower_bf16_product_equivalents_per_multiplication": products,
                        "max_rel_err": None, "plan": engp.plan_info()}
            if pprof["head_conv_launches"] > 0 and pprof["head_conv_ms"] > 0:
                # the dominant kernel of THIS mode, from the timed steps: algorithmic FLOPs (2 per multiply-add, whatever the mode issues per
                # multiplication) / the launches' summed duration; `frac` against the bf16 dense peak as north_star words it, and
                # `frac_of_own_mfma_floor` against the matrix-pipe time of the mode's own instruction mix (products x the bf16 time)
                ach = pprof["head_conv_flops"] / (pprof["head_conv_ms"] * 1e-3) / 1e12
                mx = mode in ("f16mx", "f16mx4") and engp.plan_info()["tower_mx"]
                out[key]["roofline"] = {
                    "bound": "mfma",
                    "kernel": (("conv_igemm_mx_kernel<1> (box / covariance towers: f16 hi*hi + block-scaled e2m3 cross terms) + <3> (classification tower: e2m1 cross terms), layers 1-3"
                                if (mode == "f16mx" and os.environ.get("BOD_MX_CLS_H4", "1") != "0") else
                                "conv_igemm_mx_kernel<%d> (head towers, layers 1-3: f16 hi*hi + block-scaled %s cross terms)" % ((1, "e2m3") if mode == "f16mx" else (3, "e2m1")))
                               if mx else "conv_igemm_kernel<256,256,2,4,0,%s,true> (every head 3x3 launch, (hi, lo) bf16 pairs: three products per multiplication)" % ("true" if engp.plan_info().get("row_reuse") else "false")),
                    "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    "mfma_issue_tflops_bf16_equivalent": round(products * ach, 2), "frac_of_own_mfma_floor": round(products * ach / PEAK_BF16_TFLOPS, 4),
                    "avg_launch_ms": round(pprof["head_conv_ms"] / pprof["head_conv_launches"], 4),
                    "launches_per_step": int(pprof["head_conv_launches"] // p_steps),
                    "share_of_step": round(pprof["head_conv_ms"] / p_steps / (dt / p_steps * 1e3), 3), "traffic": None}
            engp.close()
            del engp
        out["secondary"] = secondary_configs(local_rank, weights, lo)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline(hw, n, frames, weights, anchors, device_raw=device_raw, seed=0, first_image_id=lo, device_dets=device_dets)
        out["cpu_baseline"] = base
        out["config"]["speedup_vs_cpu_baseline"] = round(value / base["value"], 1)
        for mode in PARITY_MODES:
            key = "parity_mode_" + mode
            if mode not in parity or key not in out:
                continue
            out[key]["max_rel_err"] = float("%.3g" % parity[mode]["max_rel_err"])
            out[key]["rel_rms"] = float("%.3g" % parity[mode]["rel_rms"])
            out[key]["max_rel_err_strict"] = float("%.3g" % parity[mode]["max_rel_err_strict"])
            out[key]["against"] = ("cpu_baseline's fp32 forward of frame 0 with the same Philox dropout masks: raw head outputs "
                                   "(class logits, box deltas, covariance parameters), max |d| / (|ref| + rms(ref)); _strict: SURVEY 8d's "
                                   "max |d| / max(|ref|, 1e-5) over the elements with |ref| >= 1 % of the tensor's rms (a signed output's zero "
                                   "crossings have no relative error)")
            out[key]["speedup_vs_cpu_baseline"] = round(out[key]["images_per_sec"] / base["value"], 1)
            # the path's OUTPUT (boxes, class scores, 4x4 covariances of the cluster-fused detections at full size) against the CPU leg's
            # own posterior -> soft-NMS -> cluster-and-fuse, same Philox streams: frame 0, and the statistic over every frame of the CPU leg
            out[key]["detections"] = parity[mode].get("detections")
            out[key]["detections_all_frames"] = parity[mode].get("detections_all_frames")
        if "bf16" in parity:
            out["config"]["headline_mode_distance_to_cpu_forward"] = {k: (float("%.3g" % v) if not isinstance(v, dict) and v is not None else v)
                                                                      for k, v in parity["bf16"].items()}
    # `parity_mode` = the fastest of the measured modes that meets EVERY clause of north_star's "outputs (boxes, class logits, 4x4
    # covariance) match ... within 1e-3" in this run.  The gate (round 6; SURVEY 8d "Parity gate"), over every frame of the CPU leg:
    #   raw_outputs            max |d| / (|ref| + rms(ref)) of the raw head outputs of frame 0 <= 1e-3 (the strict form is reported beside it);
    #   numeric_max_within     over the detections whose DISCRETE decisions agree with the CPU leg's (same centre, same kept anchors, same
    #                          cluster members, same sampled counts: explain_detection) the MAXIMUM of box means (|d| / (|mu| + 1 px)), scores
    #                          (absolute: they are probabilities) and covariance entries (|d| / (|entry| + rms of the matrix's entries): the raw
    #                          outputs' form -- a signed entry's zero crossings have no relative error; the rounds-4/5 floor, 1 % of the largest
    #                          entry, is reported beside it as `covariance_entries_1pct_floor`, the norm-wise distance as fro_dSigma) <= 1e-3;
    #   covariance_entries_1pct_floor   in the stricter entry-wise form (|d| / (|entry| + 1 % of the largest entry)) at most one such detection
    #                          per frame above 1e-3;
    #   discrete_flips         the others -- a categorical draw at a CDF edge, a cluster member at the affinity threshold, a centre swapped: a
    #                          1e-4 perturbation flips a handful per frame in ANY arithmetic -- counted by kind and bounded at one per frame on
    #                          average; every unmatched CPU detection must be one of them.
    # p99 and max of every metric over ALL matched detections stand beside the gate so that nothing hides behind it.
    measured = [m for m in PARITY_MODES if "parity_mode_" + m in out]
    if measured:
        def clauses(r):
            st = r.get("detections_all_frames") or {}
            if r["max_rel_err"] is None or not st:
                return None
            num, flips = st.get("numeric_only"), st.get("discrete_flips")
            stat = lambda k, f: (st.get(k) or {}).get(f)
            c = {"raw_outputs": bool(r["max_rel_err"] <= 1e-3), "raw_outputs_max_rel_err": r["max_rel_err"], "raw_outputs_strict_max": r.get("max_rel_err_strict"),
                 "p99": {k: stat(k, "p99") for k in ("rel_dmu", "abs_dmu_px", "rms_dSigma", "rel_dSigma", "fro_dSigma", "dscore")},
                 "max": {k: stat(k, "max") for k in ("rel_dmu", "abs_dmu_px", "rms_dSigma", "rel_dSigma", "fro_dSigma", "dscore")},
                 "outside_1e-3": st.get("outside_1e-3")}
            if num is None or flips is None:            # (no posterior collected: the causes are unknown and the max over everything decides)
                c["numeric_max_within_1e-3"] = bool(max(stat("rel_dmu", "max"), stat("rms_dSigma", "max"), stat("dscore", "max")) <= 1e-3)
                c["all_matched"] = st.get("matched") == st.get("cpu_detections") == st.get("device_detections")
                return c
            c["numeric_only"] = num
            c["numeric_max_within_1e-3"] = bool(max(num["max_rel_dmu"], num["max_rms_dSigma"], num["max_dscore"]) <= 1e-3)
            c["discrete_flips"] = flips
            c["discrete_flips_at_most_one_per_frame"] = bool(flips["detections"] <= st["frames"])
            # the STRICTER entry-wise covariance form of rounds 4-5 (floor: 1 % of the matrix's largest entry) stays a clause: at most one
            # detection per frame may exceed 1e-3 in it (f16mx: 1 of 1 600 at 1.2e-3; f16mx4: ~90 of 1 600 up to 4e-3 -- which is why that
            # mode is reported beside the parity mode and never chosen as it, whatever the matrix-scale form says)
            c["covariance_entries_1pct_floor_at_most_one_per_frame"] = bool(num["covariance_entries_1pct_floor"]["outside_1e-3"] <= st["frames"])
            c["unmatched_all_explained"] = bool(st["cpu_detections"] - st["matched"] <=
                                                sum(v for k, v in (st.get("outside_1e-3") or {}).get("by_cause", {}).items()
                                                    if k.startswith("unmatched: ") and "unexplained" not in k))
            return c
        gate_ok = lambda c: bool(c) and all(v for v in c.values() if isinstance(v, bool))
        for m in measured:
            out["parity_mode_" + m]["meets_1e-3"] = clauses(out["parity_mode_" + m])
        ok = [m for m in measured if gate_ok(out["parity_mode_" + m]["meets_1e-3"])]
        chosen = max(ok, key=lambda m: out["parity_mode_" + m]["images_per_sec"]) if ok else ("f16mx" if "f16mx" in measured else measured[0])
        verified = any(out["parity_mode_" + m]["meets_1e-3"] for m in measured)
        out["parity_mode"] = out.pop("parity_mode_" + chosen)
        out["parity_mode"]["chosen"] = ("fastest of %s that meets every clause" % "/".join(measured)) if ok else \
            ("no measured mode meets every clause in this run" if verified else "no mode verified in this run (no CPU leg)")
        # the images/sec that satisfies north_star's accuracy clause and its throughput clause TOGETHER (null when no mode was verified)
        out["value_at_parity"] = out["parity_mode"]["images_per_sec"] if ok else None
        out["value_at_parity_mode"] = chosen if ok else None
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
