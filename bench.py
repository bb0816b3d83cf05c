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
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ANCHOR_CFG = {"layers": [3, 4, 5, 6, 7], "aspect_ratios": [[1.0, 1.0], [1.0, 2.0], [2.0, 1.0]],
              "scales": [1.0, 1.26, 1.59]}
BAYES_CFG = {"ranking_method": "score", "dirichlet_prior": {"type": "non_informative"},
             "gaussian_prior": {"type": "isotropic", "isotropic_variance": 100000.0}}
NMS_CFG = {"max_output_size": 100, "iou_threshold": 0.5, "soft_nms_sigma": 0.5}

PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3       # f32-in MFMA = fp32 vector rate (fp32 precision mode)
# cls foreground bias calibrated so that 500 <= M <= 1500 anchors survive the background filter
# at 512x512 with synthetic.make_weights() (python bench.py --calibrate)
CALIBRATED_FG_BIAS = -3.2


def head_flops_per_image(P, N, dedup=True):
    conv = 2.0 * P * 256 * 2304
    if dedup:
        return 3 * conv + N * 8 * conv
    return N * 11 * conv


def cpu_baseline(hw, n, frames, weights, anchors, seconds_budget=25.0):
    """Reference-literal CPU timing with the oracle (kind='port'): PyTorch-CPU fp32 forward
    (oracle/torch_ref.py) + NumPy posterior / soft-NMS / clustering, all host cores."""
    import torch
    from oracle import bayes_od, clustering, geometry, network, nms, philox, torch_ref
    tw = torch_ref.prepare(weights)
    # big hosts lose to thread oversubscription on these small convs: probe a few pool sizes on one
    # head-tower conv and keep the fastest
    import torch.nn.functional as F
    ncpu = os.cpu_count() or 1
    probe_x = torch.randn(n, 256, hw[0] // 8, hw[1] // 8)
    best = (float("inf"), 1)
    for cand in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)}):
        torch.set_num_threads(cand)
        with torch.no_grad():
            F.conv2d(probe_x, tw["pyramid_classification_0"][0], padding=1)
            t0 = time.perf_counter()
            F.conv2d(probe_x, tw["pyramid_classification_0"][0], padding=1)
            dt = time.perf_counter() - t0
        if dt < best[0]:
            best = (dt, cand)
    threads = best[1]
    torch.set_num_threads(threads)
    done, t_total = 0, 0.0
    parts = {"forward": 0.0, "posterior": 0.0, "nms": 0.0, "cluster": 0.0}
    while done < len(frames) and (done == 0 or t_total < seconds_budget):
        t0 = time.perf_counter()
        out = torch_ref.retinanet_forward(None, frames[done:done + 1], n, 8, prepared=tw)
        t1 = time.perf_counter()
        u = philox.categorical_uniforms(0, done, anchors.shape[0])
        post = bayes_od.bayes_od_posterior(out, anchors, u, BAYES_CFG, use_full_covar=True, dtype=np.float32)
        t2 = time.perf_counter()
        corners = post["corners"].astype(np.float32)
        idx, _ = nms.soft_nms(corners, post["ranking"], 100, 0.5, 0.5)
        t3 = time.perf_counter()
        if len(idx):
            iou = geometry.bbox_iou_vuvu(corners, corners)
            clustering.bayes_od_clustering(post["counts"], post["means"], post["covs"], idx, iou, 0.5)
        t4 = time.perf_counter()
        parts["forward"] += t1 - t0; parts["posterior"] += t2 - t1
        parts["nms"] += t3 - t2; parts["cluster"] += t4 - t3
        t_total += t4 - t0
        done += 1
    return {"value": done / t_total, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "%d frame(s) of %dx%d at N=%d, reference-literal (11*N head convs, no dedup), "
                      "torch %s fp32 CPU forward + NumPy Bayesian stages; seconds: %s" %
                      (done, hw[0], hw[1], n, torch.__version__,
                       {k: round(v, 2) for k, v in parts.items()})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (256: ~60 GB of the 288 GB; the tower "
                    "launches then hold 218 tiles per CU and the step's fixed costs are amortised: +4 %% over 64)")
    ap.add_argument("--mc", type=int, default=10)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--calibrate", action="store_true", help="print the cls foreground bias for M~1000")
    ap.add_argument("--fg-bias", type=float, default=CALIBRATED_FG_BIAS)
    ap.add_argument("--forward-only", action="store_true",
                    help="time RetinaNetModel.call only (BASELINE config 2: raw head outputs); implied by --mc 1, "
                         "where the Bayesian stages are undefined (sample covariance divides by N-1)")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16",
                    help="bf16 = throughput path (BASELINE.json north_star); fp32 = reference-exact arithmetic mode")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bayes_od_rc_amd import synthetic
    from bayes_od_rc_amd import distributed as bdist
    from bayes_od_rc_amd.anchor_generator import FpnAnchorGenerator
    from bayes_od_rc_amd.engine import Engine, make_config

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
    if os.environ.get("BOD_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    hw, n, B = (args.height, args.width), args.mc, args.batch
    fwd_only = args.forward_only or n < 2
    weights = synthetic.make_weights(cls_fg_bias=args.fg_bias)
    anchors = FpnAnchorGenerator(ANCHOR_CFG).generate_all((hw[0], hw[1], 3))
    eng = Engine(make_config(hw, batch=B, mc_samples=n, device=local_rank, bayes_od_config=BAYES_CFG,
                             nms_config=NMS_CFG, use_full_covar=True, precision=args.precision))
    eng.load_weights(weights)
    eng.set_anchors(anchors)
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
    force_gather = os.environ.get("BOD_BENCH_FORCE_GATHER") == "1"
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
            allrec = bdist.gather_records(rec if backend == "nccl" else rec.cpu(), dst=0)
            if rank == 0:
                gathered = allrec.cpu()
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
        if world > 1:
            dist.barrier()
        eng.synchronize()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    drain()
    fence()
    # HIP events around every head-tower launch (and every posterior) of the timed steps, recorded on the streams the
    # kernels run on; read back after the closing fence
    eng.profile_begin(which=1)         # 1: the launches of the dominant kernel (row-reuse tower kernel, tower layers 1..3)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_images = world * B * args.steps
    value = total_images / elapsed
    kept = [] if fwd_only else eng.num_kept()

    # ---- roofline of the dominant kernel (head 3x3 implicit-GEMM): the launches of the timed region itself
    # The dominant kernel is ONE kernel symbol -- conv_igemm_kernel<256,256,2,4,0,true>, the row-reuse loop that runs tower
    # layers 1..3 (3 launches per step, 8 of the 11 de-duplicated head convs) -- so that its average launch duration here
    # and in the rocprofv3 --kernel-trace summary under profiles/ are the same quantity.  The first tower layer (N-way
    # dropout fan-out, a different instantiation) is timed in three extra steps below and reported beside it.
    def more_steps(k):
        for i in range(k):
            if fwd_only:
                eng.forward(None, seed=0, first_image_id=lo + i * world * B)
            else:
                eng.infer(None, seed=0, first_image_id=lo + i * world * B)

    conv_flops = 2.0 * eng.P * 256 * 2304          # one 3x3 256->256 head conv over one image's pyramid, one sample
    out_flops = 2.0 * eng.P * 256 * 9 * (8 + 4 + 10)     # the three 1x1 output convs (cls 9x8, box 9x4, cov 9x10 channels)
    prof_steps = args.steps
    prof = eng.profile_end()
    tower_only = prof["head_conv_launches"] > 0
    if tower_only:
        # SURVEY.md 8d, de-duplicated heads: N * (8 * 6.436 + 0.553) GFLOP per 512x512 image for these launches -- the eight
        # 3x3 convs of tower layers 1..3 and the three 1x1 output convs fused into their epilogues
        # (the plan fuses them whenever the launch uses the full-cout tile: from about 8 frames per step on; below that
        # they are separate small launches and are not counted here)
        algo_flops = n * 8 * conv_flops * B * prof_steps
        kernel_name = "conv_igemm_kernel<256,256,2,4,0,true> (head towers, 3x3 256->256, layers 1-3)"
    else:                                # no row-reuse kernel in the plan (fp32 mode, BOD_CONV_XREUSE=0): all head 3x3 launches, three extra steps
        prof_steps = max(1, min(3, args.steps))
        eng.profile_begin(which=0)
        more_steps(prof_steps)
        prof = eng.profile_end()
        algo_flops = head_flops_per_image(eng.P, n) * B * prof_steps
        kernel_name = ("conv_igemm_f32_kernel" if args.precision == "fp32" else "conv_igemm_kernel") + " (head towers, 3x3 256->256)"
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
            fan_out = {"kernel": "conv_igemm_kernel<..., false> (tower layer 0, %d-way dropout fan-out)" % n,
                       "achieved": round(fo["head_conv_flops"] / (fo["head_conv_ms"] * 1e-3) / 1e12, 2),
                       "avg_launch_ms": round(fo["head_conv_ms"] / fo["head_conv_launches"], 4), "launches_per_step": 1}
        eng.profile_begin(which=0)
        eng.profile_end()
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_MFMA_TFLOPS
    # HBM bytes per launch from the committed PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    # MI355X_MICROARCH.md prescribes) when they were taken on this exact configuration, else null
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "round1_head_conv_pmc.json")) as fp:
            pmc = json.load(fp)
        c = pmc["config"]
        if (c["height"], c["width"], c["mc_samples"], c["batch"]) == (hw[0], hw[1], n, B) and args.precision == "bf16":
            sel = [l for l in pmc["launches"] if ("true>" in l["kernel"]) == tower_only or not tower_only]
            traffic = int(sum(l["hbm_read_bytes_corrected"] + l["hbm_write_bytes"] for l in sel) / len(sel))
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "mfma", "kernel": kernel_name,
                "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "avg_launch_ms": round(prof["head_conv_ms"] / launches, 4), "launches_per_step": launches // prof_steps,
                "share_of_step": round(prof["head_conv_ms"] / prof_steps / (elapsed / args.steps * 1e3), 3)}
    if fan_out:
        roofline["other_head_launch"] = fan_out
    # BASELINE.md section 3, "reported beside it": the whole pipeline's de-duplicated conv FLOPs per image x images/sec
    # (backbone + FPN 49.05 GFLOP at 512x512, linear in the pixel count; heads 3 + 8 N convs + N 1x1 sets)
    image_gflop = (49.05e9 * (hw[0] * hw[1]) / (512.0 * 512.0) + (3 + 8 * n) * conv_flops + n * out_flops) / 1e9
    roofline["pipeline"] = {"dedup_gflop_per_image": round(image_gflop, 1),
                            "achieved": round(image_gflop * value / world / 1e3, 2), "unit": "TFLOP/s per GPU",
                            "frac": round(image_gflop * value / world / 1e3 / peak, 4)}
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

    out = {"metric": "images/sec at N=10 MC samples, 512x512; per-anchor covariance latency",
           "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
           "data": "synthetic",
           "config": {"workload": "ResNet-50 RetinaNet + covar head, N=%d MC-dropout, %dx%d, %s"
                                  % (n, hw[0], hw[1], "forward only (raw head outputs, BASELINE config 2)" if fwd_only else
                                     "full BayesOD pipeline (forward+posterior+soft-NMS+cluster-fuse)"),
                      "frames_per_gpu_per_step": B, "global_batch": world * B, "mc_samples": n,
                      "anchors": eng.A, "kept_anchors_M": [int(k) for k in kept[:4]],
                      "parallelism": "image-sharded x%d, one RCCL gather/step" % world,
                      "per_anchor_covariance_latency_ns": round(post_us_per_anchor * 1e3, 4),
                      "stages_ms_per_step": stages},
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(hw, n, frames, weights, anchors)
        out["config"]["speedup_vs_cpu_baseline"] = round(value / out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
