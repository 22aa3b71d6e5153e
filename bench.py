#!/usr/bin/env python3
"""bench.py -- frames/sec end-to-end (JPEG bytes -> post-NMS boxes), UltraFace-640 @ 640x480.

One "step" = one pass of the hot path over one batch of 32 synthetic 640x480 camera-like JPEGs
(BASELINE.json configs[2]/[3]).  One process per GPU; independent camera
streams shard one-per-GPU ("weak" scaling) and the only collective is the start-up RCCL
broadcast of the weight blob.  Prints ONE JSON line on rank 0.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak

# bench kernel label -> device function name (as rocprofv3 --kernel-trace --stats prints it)
KERNEL_FUNCS = {
    "conv_pw_mfma": "k_pw_mfma",
    "conv_dwpw_mfma": "k_dwpw_mfma",
    "conv_dwpw2_mfma": "k_dwpw2_mfma",
    "conv3x3_mfma": "k_conv3x3_mfma",
    "conv3x3_rows_mfma": "k_conv3x3_rows_mfma",
    "upsample_norm_420": "k_upsample_norm_420",
    "conv_direct_dw": "k_conv_direct<1, true>",
    "conv_direct_full": "k_conv_direct<16|4, false>",
    "idct": "k_idct",
    "upsample_norm": "k_upsample_norm",
    "upsample_rgb": "k_upsample_rgb",
    "resize_norm": "k_resize_norm",
    "head_decode": "k_head_decode",
    "huffman_rst": "k_huffman_rst",
    "huff_unstuff": "k_huff_unstuff", "huff_seed": "k_huff_seed", "huff_extend": "k_huff_extend", "huff_link": "k_huff_link",
    "huff_resolve": "k_huff_resolve", "huff_write": "k_huff_write", "dc_prefix": "k_dc_prefix", "zero_coef": "k_zero_coef",
    "conv_dwpw_coop": "k_dwpw_coop", "stem_planes_mfma": "k_stem_planes_mfma",
    "sort_nms": "k_sort_nms",
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pool", type=int, default=256, help="distinct frames per stream")
    ap.add_argument("--depth", type=int, default=6, help="batches in flight (async submit/wait)")
    ap.add_argument("--input", choices=["hbm", "host"], default="hbm",
                    help="hbm: the JPEG bytes of every batch are staged in HBM before the clock starts "
                         "(ufd_stage_jpeg_batch / ufd_submit_staged; implies --entropy device); "
                         "host: the timed region starts from host buffers (PCIe-inclusive)")
    ap.add_argument("--entropy", choices=["host", "device"], default="device",
                    help="where the Huffman stage runs (device: self-synchronising decoder kernels; "
                         "host: worker threads, coefficient slabs over PCIe)")
    ap.add_argument("--variant", type=int, choices=[640, 320], default=640, help="UltraFace variant (BASELINE C1/C2: 320)")
    ap.add_argument("--src", default=None, help="frame size of the synthetic stream (BASELINE config C5: 1280x720, --batch 16)")
    ap.add_argument("--no-variants", action="store_true", help="skip the PCIe-inclusive comparison runs")
    ap.add_argument("--restart-rows", type=int, default=0,
                    help="JPEG restart interval in MCU rows (0 = none: entropy decoding on host workers; "
                         ">0: restart-interval stream, entropy decoding on the GPU)")
    ap.add_argument("--host-threads", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-every", type=int, default=4, help="record kernel events for every n-th timed step")
    return ap.parse_args()


def cpu_baseline(jpegs, weights, priors, budget_s, W=640, H=480):
    """The CPU oracle (oracle/, a single-threaded plain-C port of the reference path) timed on
    this host on a bounded sample of the same frames."""
    import oracle

    oracle.build()
    oracle.infer_jpeg(jpegs[0], W, H, weights, priors)  # warm
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.infer_jpeg(jpegs[n % len(jpegs)], W, H, weights, priors, 0.5, 0.5)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 4 * len(jpegs):
            break
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the bench's JPEG frames, full path decode->NMS at %dx%d, 1 thread, %.1f s" % (n, W, H, el)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)

    from infercam_onnx_amd import nn, parallel, synth

    W, H, B = (640, 480, args.batch) if args.variant == 640 else (320, 240, args.batch)
    variant = nn.UltrafaceVariant.W640H480 if args.variant == 640 else nn.UltrafaceVariant.W320H240
    # ---- weights: generated on rank 0, broadcast over RCCL (the path's only collective)
    weights = parallel.broadcast_weights(synth.synthetic_weights() if rank == 0 else None, dist,
                                         device=torch.device("cuda", local_rank))
    priors = synth.gen_priors(W, H)

    # ---- this rank's camera stream: pool of distinct synthetic frames (baseline JPEG q90 4:2:0)
    SW, SH = (int(v) for v in args.src.lower().split("x")) if args.src else (W, H)
    jpegs = synth.synth_jpeg_pool(rank, args.pool, SW, SH, quality=90, subsampling="4:2:0", restart_rows=args.restart_rows)
    device_entropy = args.entropy == "device" or args.input == "hbm"
    model = nn.UltrafaceModel(variant, 0.5, 0.5, device_id=local_rank, max_batch=B,
                              weights=weights, priors=priors, max_src=(SW, SH), host_threads=args.host_threads,
                              profile=True, det_cap=256, host_entropy=not device_entropy)
    nb = max(1, args.pool // B)
    if args.input == "hbm":
        # inputs resident in HBM before the clock starts: bytes + parsed headers of every batch
        batches = [model.stage_jpeg_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
    else:
        batches = [model._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
    depth = min(args.depth, nb) if args.input == "hbm" else args.depth  # a staged batch is in flight once at a time

    def run_steps(k, mdl=None, bts=None, staged=None):
        mdl = mdl or model
        bts = bts or batches
        staged = (args.input == "hbm") if staged is None else staged
        inflight = []
        dets = 0
        for s in range(k):
            if len(inflight) >= depth:
                cnt, _ = mdl.wait(inflight.pop(0), collect=False)
                dets += sum(cnt)
            b = bts[s % len(bts)]
            inflight.append(mdl.submit_staged(b) if staged else mdl.submit_jpeg_batch(b))
        for t in inflight:
            cnt, _ = mdl.wait(t, collect=False)
            dets += sum(cnt)
        return dets

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    model.profile_reset()
    model.profile_sampling(args.profile_every)
    barrier()
    t0 = time.perf_counter()
    ndet = run_steps(args.steps)
    barrier()
    el = time.perf_counter() - t0
    el = parallel.max_over_ranks(el, dist, device=torch.device("cuda", local_rank))
    stats = model.profile_read()

    # ---- the same workload over the reference's own boundary (host buffers in, PCIe inside the
    # timed region): reported beside `value`, never as `value`
    variants = {}
    if world == 1 and args.input == "hbm" and not args.no_variants:
        hb = [model._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
        run_steps(args.warmup, bts=hb, staged=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.steps, bts=hb, staged=False)
        torch.cuda.synchronize()
        variants["host_bytes_device_entropy_fps"] = round(B * args.steps / (time.perf_counter() - t1), 1)
        m2 = nn.UltrafaceModel(variant, 0.5, 0.5, device_id=local_rank, max_batch=B, weights=weights,
                               priors=priors, max_src=(SW, SH), host_threads=args.host_threads, det_cap=256,
                               host_entropy=True)
        hb2 = [m2._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
        run_steps(args.warmup, mdl=m2, bts=hb2, staged=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.steps, mdl=m2, bts=hb2, staged=False)
        torch.cuda.synchronize()
        variants["host_bytes_host_entropy_fps"] = round(B * args.steps / (time.perf_counter() - t1), 1)
        m2.close()

    if rank == 0:
        frames = world * B * args.steps
        prof_steps = (args.steps + args.profile_every - 1) // args.profile_every  # steps that carried events
        # ---- roofline of the dominant kernel (device time from HIP events on the library's stream)
        agg = {}
        for st in stats:
            key = st["name"].split(":")[0]
            a = agg.setdefault(key, dict(ms=0.0, launches=0, bytes=0.0, flops=0.0))
            a["ms"] += st["total_ms"]
            a["launches"] += st["launches"]
            a["bytes"] += st["bytes"]
            a["flops"] += st["flops"]
        kern = {k: v for k, v in agg.items() if not k.startswith("h2d_")}
        dom = max(kern, key=lambda k: kern[k]["ms"])
        d = kern[dom]
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] > 0 else 0.0
        tfl = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        # (the roof the kernel sits closer to; on gfx950 fp32 MFMA and fp32 vector work share one datapath)
        if dom in ("conv_pw_mfma", "conv_dwpw_mfma", "conv_dwpw2_mfma", "conv_dwpw_coop", "stem_planes_mfma", "conv3x3_mfma",
                   "conv3x3_rows_mfma") and tfl / MFMA_F32_PEAK_TFLOPS > gbs / HBM_PEAK_GBS:
            roof = {"bound": "mfma", "achieved": round(tfl, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4)}
        else:
            roof = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 4)}
        traffic = None
        try:  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes summarised by tools/pmc_traffic.py
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))["kernels"]
            traffic = round(pmc[KERNEL_FUNCS.get(dom, dom)]["hbm_bytes_per_launch"])
        except Exception:
            pass
        per_launch = d["bytes"] / max(d["launches"], 1)
        roof.update({"traffic": traffic, "algorithmic_bytes_per_launch": round(per_launch),
                     "algorithmic_flops_per_launch": round(d["flops"] / max(d["launches"], 1)),
                     "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "mfma_frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4),
                     "kernel": KERNEL_FUNCS.get(dom, dom),
                     "avg_launch_us": round(d["ms"] * 1e3 / max(d["launches"], 1), 2), "launches": d["launches"]})
        gpu_ms = sum(v["ms"] for v in kern.values())
        out = {
            "metric": "frames/sec end-to-end (decode->NMS), UltraFace-%d @ %dx%d" % (args.variant, W, H),
            "value": round(frames / el, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "UltraFace-" + str(args.variant) + ", one %dx%d synthetic JPEG stream per GPU (q90 4:2:0, %d distinct "
                                   "frames, %s), batch=%d, seeded synthetic weights" % (
                                       SW, SH, args.pool, "DRI = %d MCU row(s)" % args.restart_rows if args.restart_rows
                                       else "no restart markers", B),
                       "global_batch": world * B, "parallelism": "streams x%d (one per GPU), RCCL weight broadcast only" % world,
                       "entropy_decode": "GPU kernels" if device_entropy else "host worker threads",
                       "timed_region": ("JPEG bytes resident in HBM (headers parsed at staging) -> detections in host memory"
                                        if args.input == "hbm" else
                                        "host JPEG bytes -> host detections (%s + PCIe included)" % (
                                            "JPEG bytes H2D" if device_entropy else "host Huffman, coefficient slabs H2D")),
                       "async_depth": depth, "pcie_inclusive": variants or None},
            "roofline": roof,
            "gpu_ms_per_step": round(gpu_ms / prof_steps, 3),
            "kernels_ms_per_step": {k: round(v["ms"] / prof_steps, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])},
            "detections_per_frame": round(ndet / (B * args.steps), 2),
        }
        dump = os.environ.get("UFD_BENCH_DUMP")
        if dump:
            with open(dump, "w") as f:
                json.dump({"steps": prof_steps, "batch": B, "stats": stats}, f, indent=1)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(jpegs[:64], weights, priors, args.cpu_seconds, W, H)
        print(json.dumps(out), flush=True)
    model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
