#!/usr/bin/env python3
"""bench.py -- frames/sec end-to-end (JPEG bytes -> post-NMS boxes), UltraFace-640 @ 640x480.

One "step" = one pass of the hot path over one batch of 32 synthetic 640x480 camera-like JPEGs
(BASELINE.json configs[2]/[3]).  One process per GPU; independent camera
streams shard one-per-GPU ("weak" scaling) and the only collective is the start-up RCCL
broadcast of the weight blob.  Prints ONE JSON line on rank 0.

The timed region is the reference's own boundary (inferer.rs:35-37): JPEG bytes in host memory
-> detections in host memory, PCIe included (`--input host`, the default).  `--input hbm` times
the same path with the JPEG bytes staged in HBM before the clock starts; at N=1 the default run
reports that figure too (`config.hbm_resident_fps`).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA dense peak
ROOF_STEPS = 48             # steps of the fully sampled roofline pass (every launch of every kernel timed)

# bench kernel label -> device function name (as rocprofv3 --kernel-trace --stats prints it)
KERNEL_FUNCS = {
    "conv_pw_mfma": "k_pw_mfma",
    "conv_dwpw_mfma": "k_dwpw_mfma", "conv_dwpw_mfma_sk8": "k_dwpw_mfma_sk8",
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
    "huff_unstuff": "k_huff_unstuff", "huff_seed": "k_huff_seed", "huff_extend": "k_huff_extend", "huff_link": "k_huff_link",
    "huff_resolve": "k_huff_resolve", "huff_write": "k_huff_write", "dc_prefix": "k_dc_prefix", "zero_coef": "k_zero_coef",
    "conv_dwpw_coop": "k_dwpw_coop", "stem_planes_mfma": "k_stem_planes_mfma",
    "sort_nms": "k_sort_nms", "conv_dual_coop": "k_dual_dwpw_coop", "conv_dual_pw": "k_dual_dwpw_pw",
    "nms_matrix": "k_nms_matrix", "nms_scan": "k_nms_scan", "rfb_tail": "k_rfb_tail",
}
MFMA_KERNELS = ("conv_pw_mfma", "conv_dwpw_mfma", "conv_dwpw_mfma_sk8", "conv_dwpw2_mfma", "conv_dwpw_coop", "conv_dual_coop", "conv_dual_pw",
                "stem_planes_mfma", "conv3x3_mfma", "conv3x3_rows_mfma", "rfb_tail")


def base_label(key):
    """'conv_dwpw2_mfma<16, 1, true>' -> 'conv_dwpw2_mfma' (the library labels a launch with its template instance)."""
    return key.split("<")[0]


def device_name(key):
    """Device function name of a profiled label, template arguments included, as rocprofv3's kernel_stats.csv prints it."""
    b = base_label(key)
    return KERNEL_FUNCS.get(b, b) + key[len(b):]


def stage_of(label):
    """Stage of SURVEY 8(d)'s breakdown a profiled kernel label belongs to."""
    label = base_label(label)
    if label.startswith("h2d_"):
        return "h2d"
    if label.startswith("host_"):
        return "host_" + ("plan" if label == "host_plan" else "issue_total")
    if label.startswith("huff_") or label in ("dc_prefix", "zero_coef"):
        return "entropy"
    if label in ("idct", "upsample_norm", "upsample_norm_420", "upsample_rgb", "resize_norm"):
        return "idct_preproc"
    if label in ("head_decode", "sort_nms"):
        return "post"
    if label == "draw_rects" or label.startswith("enc_"):
        return "annotate_encode"
    return "cnn"  # (the fused stem -- upsampling + colour + normalise + conv 0 -- counts as CNN)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pool", type=int, default=256, help="distinct frames per stream")
    ap.add_argument("--depth", type=int, default=6, help="batches in flight (async submit/wait)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="timed regions (each: --warmup untimed steps, then exactly --steps timed ones) run back to back; `value` is the median one")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N>1 script rehearsal on a one-GPU box: all ranks on cuda:0, exchanges over gloo (not a measurement)")
    ap.add_argument("--input", choices=["host", "hbm"], default="host",
                    help="host: the timed region starts from JPEG bytes in host memory (the reference's boundary, PCIe "
                         "included); hbm: the JPEG bytes of every batch are staged in HBM before the clock starts "
                         "(ufd_stage_jpeg_batch / ufd_submit_staged)")
    ap.add_argument("--entropy", choices=["host", "device"], default="device",
                    help="where the Huffman stage runs (device: self-synchronising decoder kernels; "
                         "host: worker threads, coefficient slabs over PCIe)")
    ap.add_argument("--variant", type=int, choices=[640, 320], default=640, help="UltraFace variant (BASELINE C1/C2: 320)")
    ap.add_argument("--src", default=None, help="frame size of the synthetic stream (BASELINE config C5: 1280x720, --batch 16)")
    ap.add_argument("--no-extras", "--no-variants", dest="no_extras", action="store_true",
                    help="skip the side measurements (other boundary, stage breakdown, batch-1 latency, verification)")
    ap.add_argument("--restart-rows", type=int, default=0, help="JPEG restart interval in MCU rows (0 = none)")
    ap.add_argument("--subsampling", choices=["4:2:0", "4:2:2", "4:4:4"], default="4:2:0",
                    help="chroma subsampling of the synthetic stream (SURVEY 8d: 4:2:0 primary, 4:2:2 = UVC-MJPG-like, what the "
                         "reference's sender captures, cam_sender/src/sensors.rs:18-68)")
    ap.add_argument("--no-dht", action="store_true",
                    help="MJPG flavour without DHT segments (cameras omit the Annex-K tables, SURVEY A1): every frame of the pool is stripped")
    ap.add_argument("--host-threads", type=int, default=0, help="host workers of the handle (0: hardware threads / ranks, <= 32)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--annotate", action="store_true",
                    help="time the whole Inferer::run iteration instead (decode -> infer -> rectangles -> JPEG re-encode, "
                         "ufd_submit_annotate_batch): a side workload for profiling N1, not BASELINE.json's metric")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="record kernel events for every n-th step of the timed region (default 0: never -- the timed region and "
                         "the steady-state run carry no event packets; the roofline comes from its own fully sampled pass behind them)")
    ap.add_argument("--cpus", type=int, default=0,
                    help="restrict this process to its first N usable CPUs before anything touches the GPU (host-scaling sweep: "
                         "what 8 ranks sharing one host leave each of them)")
    ap.add_argument("--one-process", action="store_true",
                    help="config C4 in ONE process (the reference server is one process, infer_server.rs:39-68): "
                         "ufd_create_replicas over --gpus devices (RCCL broadcast of the weights) and ONE ufd_sched over the "
                         "replicas, one producer thread per camera stream; no torch.distributed launcher")
    ap.add_argument("--streams", type=int, default=1,
                    help="--one-process: camera streams per GPU (the reference server serves many cameras from one Inferer, "
                         "infer_server.rs:48-50; 8 = VERDICT r4 #6b)")
    ap.add_argument("--no-rfb-tail", action="store_true", help="UFD_FLAG_NO_RFB_TAIL: A/B of k_rfb_tail against the two-launch form")
    ap.add_argument("--no-gate", action="store_true", help="UFD_FLAG_NO_GATE: A/B of the cross-context order of the network's first launches (csrc/pipeline_gate.cpp)")
    ap.add_argument("--spin-wait", action="store_true", help="UFD_FLAG_SPIN_WAIT: ufd_wait always spins in the runtime (A/B of the sleeping wait)")
    ap.add_argument("--host-only", action="store_true",
                    help="timed region + steady state + the `host` object only (no roofline pass, side workloads, verification "
                         "or CPU baseline): one point of tools/host_scaling.py's sweep")
    return ap.parse_args()


def cpu_baseline(jpegs, weights, priors, budget_s, W=640, H=480, annotate=False):
    """The CPU oracle (oracle/, a plain-C port of the reference path; the reference's own tract-onnx
    path cannot be built here) timed on this host on a bounded sample of the bench's frames:
    (a) one thread -- the reference's operating point is a single Inferer task (infer_server.rs:48-50);
    (b) one worker per hardware thread, frames being independent."""
    import oracle

    oracle.build()
    oracle.infer_jpeg(jpegs[0], W, H, weights, priors)  # warm

    def one_frame(j):  # the whole Inferer::run iteration when annotate (inferer.rs:35-40), else decode -> NMS
        d = oracle.infer_jpeg(j, W, H, weights, priors, 0.5, 0.5)
        if annotate:
            oracle.jpeg_encode_rgb(oracle.draw_labels(oracle.jpeg_decode_rgb(j), d, 1280, 720), 95)

    if annotate:  # (one thread only: the reference's operating point; the all-cores leg times decode -> NMS)
        n, t0 = 0, time.perf_counter()
        while True:
            one_frame(jpegs[n % len(jpegs)])
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 4 * len(jpegs):
                break
        return {"value": round(n / el, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                "sample": "%d of the bench's JPEG frames, decode -> infer at %dx%d -> rectangles + labels -> JPEG q95 re-encode "
                          "(the whole Inferer::run iteration), 1 thread, %.1f s" % (n, W, H, el)}
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.infer_jpeg(jpegs[n % len(jpegs)], W, H, weights, priors, 0.5, 0.5)
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 4 * len(jpegs):
            break
    one = n / el
    cores = usable_cpus()
    # all cores: worker threads inside the C library, each running whole frames, for the same time budget
    t1 = time.perf_counter()
    done, _ = oracle.infer_jpeg_many_threads(jpegs, 1 << 30, cores, W, H, weights, priors, 0.5, 0.5, budget_s=budget_s)
    el_all = time.perf_counter() - t1
    try:
        torch_1 = torch_cnn_baseline(min(budget_s, 6.0), W, H)
    except Exception as e:  # (a second opinion: its absence must not take the line down)
        torch_1 = {"error": repr(e)[:200]}
    return {"value": round(one, 3), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d of the bench's JPEG frames, full path decode->NMS at %dx%d, 1 thread, %.1f s" % (n, W, H, el),
            "torch_1thread": torch_1,
            "all_cores": {"value": round(done / el_all, 2), "unit": "frames/s", "cores": cores,
                          "sample": "%d frames, one worker thread per usable hardware thread (affinity mask / cgroup quota; "
                                    "the machine reports %d), %.1f s" % (done, os.cpu_count() or 1, el_all)}}


def torch_cnn_baseline(budget_s, W=640, H=480):
    """SURVEY 8(d)'s second opinion on the CPU side: the UltraFace-RFB forward ALONE (no JPEG decode, resize or NMS)
    through torch.nn.functional.conv2d -- oneDNN's packed-GEMM convolutions, the class of kernel tract-linalg's would be --
    on ONE thread (the reference runs tract's single-threaded SimplePlan inside its single Inferer task), BatchNorm folded
    into the convolutions as the zoo file has it.  The module is the upstream network definition restated in
    tools/ultraface_torch.py (test infrastructure); weights are seeded: a convolution's time does not depend on them."""
    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ultraface_torch as ut

    from torch.nn.utils.fusion import fuse_conv_bn_eval

    def fold(mod):  # Conv2d followed by BatchNorm2d -> one Conv2d (what tract's declutter and the zoo export do)
        for name, child in list(mod.named_children()):
            if isinstance(child, ut.BasicConv) and child.bn is not None:
                child.conv, child.bn = fuse_conv_bn_eval(child.conv, child.bn), None
            elif isinstance(child, torch.nn.Sequential):
                kids = list(child.children())
                for i in range(len(kids) - 1):
                    if isinstance(kids[i], torch.nn.Conv2d) and isinstance(kids[i + 1], torch.nn.BatchNorm2d):
                        child[i] = fuse_conv_bn_eval(kids[i], kids[i + 1])
                        child[i + 1] = torch.nn.Identity()
            fold(child)

    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        model = ut.build_seeded((W, H)).eval()
        fold(model)
        x = torch.randn(1, 3, H, W)
        with torch.no_grad():
            model(x)  # warm (oneDNN primitive creation)
            n, t0 = 0, time.perf_counter()
            while True:
                model(x)
                n += 1
                el = time.perf_counter() - t0
                if el >= budget_s or n >= 400:
                    break
    finally:
        torch.set_num_threads(prev)
    return {"value": round(n / el, 3), "unit": "frames/s", "cores": 1,
            "sample": "%d forwards of the UltraFace-RFB network alone at %dx%d (52 convolutions, BatchNorm folded, softmax + box "
                      "decode), torch %s conv2d on 1 thread, %.1f s -- CNN only: the JPEG decode, resize and NMS of the full path "
                      "are not in this figure" % (n, W, H, torch.__version__.split("+")[0], el)}


def usable_cpus():
    """Hardware threads this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a GPU box reports all 256 threads of the host but gives one GPU's job a share of them)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def kernel_source_sha(root=ROOT):
    """Hash of the kernel sources' CODE (comments and blank space stripped: rewording a comment is not another kernel): a
    PMC traffic figure is only reported for the code it was measured on.  tools/finish_profiles.py stamps with the same
    function."""
    import re

    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "infercam_onnx_amd", "csrc", "*.hip"))):
        text = open(f, "r", encoding="utf-8").read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        h.update("\n".join(" ".join(line.split()) for line in text.splitlines() if line.strip()).encode())
    return h.hexdigest()[:16]


def main_one_process(args):
    """C4 as the reference's process model has it: one server process, N GPUs.  ufd_create_replicas reads the weights once
    and broadcasts the packed image over RCCL; ONE scheduler (ufd_sched over models_640[N]) places stream i on GPU i mod N;
    one producer thread per camera stream pushes its frames (ufd_sched_push_batch, drop-on-full retried by the producer so
    that every step's frames do run); the clock runs from the first push to ufd_sched_flush."""
    import threading

    import torch

    from infercam_onnx_amd import nn, scheduler, synth

    N, B, K, Wm = args.gpus, args.batch, args.steps, args.warmup
    Wd, Hd = (640, 480) if args.variant == 640 else (320, 240)
    variant = nn.UltrafaceVariant.W640H480 if args.variant == 640 else nn.UltrafaceVariant.W320H240
    weights = synth.synthetic_weights()
    priors = synth.gen_priors(Wd, Hd)
    SW, SH = (int(v) for v in args.src.lower().split("x")) if args.src else (Wd, Hd)
    host_threads = args.host_threads or max(2, min(32, usable_cpus() // N))
    kw = dict(max_batch=B, weights=weights, priors=priors, max_src=(SW, SH), host_threads=host_threads, det_cap=256,
              extra_flags=nn.UFD_FLAG_NO_GATE if args.no_gate else 0)
    if args.rehearse_one_gpu:  # N handles on cuda:0: the script path on a one-GPU box, not a measurement
        models = [nn.UltrafaceModel(variant, 0.5, 0.5, device_id=0, **kw) for _ in range(N)]
    else:
        if torch.cuda.device_count() < N:
            raise SystemExit("bench.py --one-process --gpus %d: only %d devices visible" % (N, torch.cuda.device_count()))
        models = nn.UltrafaceModel.create_replicas(variant, 0.5, 0.5, list(range(N)), **kw)
    S = max(1, args.streams)  # camera streams per GPU; stream r lives on replica r mod N (round-robin placement)
    NS = N * S
    pools = [synth.synth_jpeg_pool(r, args.pool, SW, SH, quality=90, subsampling=args.subsampling, restart_rows=args.restart_rows) for r in range(NS)]
    if args.no_dht:
        pools = [[synth.strip_dht(j) for j in pool] for pool in pools]
    nb = max(1, args.pool // B)
    batches = [[models[0]._prep_batch(pools[r][i * B:(i + 1) * B]) for i in range(nb)] for r in range(NS)]
    enc_bytes = models[0]._lib.ufd_encode_bound(SW, SH) if args.annotate else 0
    sch = scheduler.Scheduler(models_640=models if args.variant == 640 else None, models_320=models if args.variant == 320 else None,
                              on_result=False, ring_slots=max(10, 8 * B // S), max_wait_us=2000, max_inflight=args.depth, det_cap=256,
                              jpeg_bytes_per_frame=enc_bytes)
    # --annotate: every stream wants the annotated JPEG back (the /face_stream viewers of endpoints.rs:58-72): rectangles and
    # labels at the router's 1280 x 720, q95 re-encode (inferer.rs:38-40)
    hs = [sch.add_stream(1000 + r, args.variant, annotate=args.annotate) for r in range(NS)]
    where = [sch.stream_replica(h) for h in hs]

    def produce(r, steps):
        for s_ in range(steps):
            b, first = batches[r][s_ % nb], 0
            while first < B:
                first += sch.push_batch(hs[r], b, first)
                if first < B:
                    time.sleep(0.0002)  # ring full: the router would drop (router.rs:65); the bench wants every frame run

    def run(steps):
        th = [threading.Thread(target=produce, args=(r, steps)) for r in range(NS)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        sch.flush()

    def sync_all():
        for d in range(1 if args.rehearse_one_gpu else N):
            torch.cuda.synchronize(d)

    run(Wm)
    sync_all()
    for m in models:
        m.host_stats_reset()
    before = sch.replica_stats(args.variant)
    t0 = time.perf_counter()
    run(K)
    sync_all()
    el = time.perf_counter() - t0
    after = sch.replica_stats(args.variant)
    frames = sum(a["frames"] - b_["frames"] for a, b_ in zip(after, before))
    dets = sum(a["detections"] - b_["detections"] for a, b_ in zip(after, before))
    assert frames == NS * B * K, (frames, NS * B * K)
    hstats = [m.host_stats() for m in models]
    flops_frame = 798315520 if args.variant == 640 else 200837120
    out = {
        "metric": ("frames/sec end-to-end (decode->NMS), UltraFace-%d @ %dx%d" % (args.variant, Wd, Hd)) if not args.annotate else
                  ("frames/sec decode->NMS->rectangles->JPEG q95 re-encode (N1 side workload), UltraFace-%d @ %dx%d" % (args.variant, Wd, Hd)),
        "value": round(frames / el, 1), "unit": "frames/s", "n_gpus": N, "steps": K, "warmup": Wm,
        "ms_per_step": round(el / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "UltraFace-%d, %d %dx%d synthetic JPEG stream(s) per GPU (q90 %s%s, %d distinct frames each), batches of up to %d "
                               "formed by the scheduler across the streams, seeded synthetic weights%s" % (
                                   args.variant, S, SW, SH, args.subsampling, ", no DHT segments (MJPG)" if args.no_dht else "", args.pool, B,
                                   "; every stream gets its frames back annotated (rectangles + labels at 1280x720, JPEG q95)" if args.annotate else ""),
                   "global_batch": N * B, "streams": NS,
                   "parallelism": "ONE process: ufd_create_replicas x%d (RCCL weight broadcast) + one ufd_sched over the replicas, "
                                  "stream i -> GPU i mod %d" % (N, N),
                   "timed_region": "host JPEG bytes pushed by one producer thread per stream -> detections delivered (ufd_sched_flush)",
                   "stream_to_replica": where, "async_depth": args.depth, "host_threads_per_replica": host_threads,
                   "replica_frames": [a["frames"] - b_["frames"] for a, b_ in zip(after, before)]},
        "roofline": None, "cpu_baseline": None,
        "whole_net_mfma_frac": round(frames / el * flops_frame / 1e12 / MFMA_F32_PEAK_TFLOPS / N, 4),
        "detections_per_frame": round(dets / max(frames, 1), 2),
        "host": {"per_replica": hstats, "usable_cpus": usable_cpus()},
    }
    if args.rehearse_one_gpu:
        out["config"]["rehearsal"] = "all %d replicas are handles on cuda:0: a run of the one-process script path, NOT a measurement" % N
    print(json.dumps(out), flush=True)
    sch.close()
    for m in models:
        m.close()


def main():
    args = parse_args()
    if args.one_process:
        if args.cpus > 0:
            cur = sorted(os.sched_getaffinity(0))
            os.sched_setaffinity(0, cur[:max(1, min(args.cpus, len(cur)))])
        return main_one_process(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "RANK" not in os.environ and args.gpus > 1:
            # no launcher: the one-process form of the same configuration (ufd_create_replicas + one scheduler over the
            # replicas; what the reference's single server process would run) instead of refusing
            print("bench.py --gpus %d without torch.distributed.run: running the --one-process form" % args.gpus, file=sys.stderr)
            args.one_process = True
            if args.cpus > 0:
                cur = sorted(os.sched_getaffinity(0))
                os.sched_setaffinity(0, cur[:max(1, min(args.cpus, len(cur)))])
            return main_one_process(args)
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run --nproc-per-node N"
                         % (args.gpus, world))
    if args.cpus > 0:  # before torch / HIP start their threads: they inherit the mask
        cur = sorted(os.sched_getaffinity(0))
        os.sched_setaffinity(0, cur[:max(1, min(args.cpus, len(cur)))])
    # The library's hardware queues FIRST: torch brings its own copy of the HIP runtime, and a handle created behind that
    # runtime's first touch of the device (one 32-byte copy is enough; at N > 1 the weight broadcast is one) runs 21 % slower
    # for the life of the process -- 65.4 k -> 51.5 k frames/s (round 6: tools/ab/r6_torch_queue.py, profiles/r6e/).
    # (Order: torch's runtime INITIALISED first -- is_available / set_device open no queue --, then the library's queues, then
    # whatever torch does on the device.  The other way round torch's runtime no longer finds the device.)
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(0 if args.rehearse_one_gpu else local_rank)
    from infercam_onnx_amd import nn as _nn

    _nn.prime_device(0 if args.rehearse_one_gpu else local_rank)

    dist = None
    if world > 1:
        import torch.distributed as dist

        if args.rehearse_one_gpu:
            # script rehearsal on a one-GPU box: every rank on cuda:0, the exchanges over gloo on host tensors.
            # Not a measurement (the ranks share the GPU); the JSON line says so in config.rehearsal.
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, "RCCL formed %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = None if args.rehearse_one_gpu else dev  # where the tensors of the exchanges live

    from infercam_onnx_amd import nn, parallel, synth

    W, H, B = (640, 480, args.batch) if args.variant == 640 else (320, 240, args.batch)
    variant = nn.UltrafaceVariant.W640H480 if args.variant == 640 else nn.UltrafaceVariant.W320H240
    # ---- weights: generated on rank 0, broadcast over RCCL (the path's only collective)
    weights = parallel.broadcast_weights(synth.synthetic_weights() if rank == 0 else None, dist, device=xdev)
    priors = synth.gen_priors(W, H)

    # ---- this rank's camera stream: pool of distinct synthetic frames (baseline JPEG q90 4:2:0)
    SW, SH = (int(v) for v in args.src.lower().split("x")) if args.src else (W, H)
    jpegs = synth.synth_jpeg_pool(rank, args.pool, SW, SH, quality=90, subsampling=args.subsampling, restart_rows=args.restart_rows)
    if args.no_dht:
        jpegs = [synth.strip_dht(j) for j in jpegs]
    device_entropy = args.entropy == "device"
    if args.input == "hbm" and not device_entropy:
        raise SystemExit("--input hbm needs the device entropy decoder")
    # host workers: this rank's share of the box (8 ranks must not oversubscribe it)
    host_threads = args.host_threads or max(2, min(32, usable_cpus() // world))
    model = nn.UltrafaceModel(variant, 0.5, 0.5, device_id=local_rank, max_batch=B, weights=weights, priors=priors,
                              max_src=(SW, SH), host_threads=host_threads, profile=True, det_cap=256,
                              host_entropy=not device_entropy, extra_flags=(nn.UFD_FLAG_SPIN_WAIT if args.spin_wait else 0) | (nn.UFD_FLAG_NO_RFB_TAIL if args.no_rfb_tail else 0) | (nn.UFD_FLAG_NO_GATE if args.no_gate else 0))
    nb = max(1, args.pool // B)
    host_batches = [model._prep_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
    staged_batches = None

    def get_batches(staged):
        nonlocal staged_batches
        if not staged:
            return host_batches
        if staged_batches is None:  # bytes + parsed headers of every batch resident in HBM
            staged_batches = [model.stage_jpeg_batch(jpegs[i * B:(i + 1) * B]) for i in range(nb)]
        return staged_batches

    annot_batches = []

    def run_steps(k, staged, depth=None, mdl=None, bts=None):
        if args.annotate:  # N1 profiling mode: the same pool through ufd_submit_annotate_batch
            if not annot_batches:
                for i in range(min(nb, args.depth)):
                    annot_batches.append(model.prep_annotate_batch(jpegs[i * B:(i + 1) * B], (1280, 720), out_bytes_per_frame=SW * SH))
            d_ = min(depth or args.depth, len(annot_batches))
            infl, dets_ = [], 0
            for s_ in range(k):
                if len(infl) >= d_:
                    cnt_, _, _ = model.wait(infl.pop(0), collect=False)
                    dets_ += sum(cnt_)
                infl.append(model.submit_annotate_batch(annot_batches[s_ % d_]))
            for t_ in infl:
                cnt_, _, _ = model.wait(t_, collect=False)
                dets_ += sum(cnt_)
            return dets_
        mdl = mdl or model
        bts = bts or get_batches(staged)
        depth = min(depth or args.depth, len(bts))  # a batch object (its output arrays) is in flight once at a time
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

    primary_staged = args.input == "hbm"
    get_batches(primary_staged)
    # (UFD_FLAG_PROFILE only ARMS the per-kernel events; a batch records them when the sampling counter says so, and in
    # the timed region and the steady-state run it never does unless --profile-every asks)
    # The timed region -- W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, max over
    # ranks -- is run --repeats times back to back and `value` is the MEDIAN region (all of them are in the line: `samples`).
    # Why more than one: set-up (frame pool, handle creation) leaves the GPU idle for seconds, after 50 ms of idling its
    # clocks are down and take ~20 ms of work to come back -- the first W + K = 25 steps (15 ms) of a process measure that
    # ramp, whatever the library does: round 6, tools/ab/r6_cold_start.py, profiles/r6b/cold_start.txt (a handle that idled
    # 50 ms: 57 k frames/s in the next 20-step region, 62 k without the pause or with 50 warm-up steps, 65 k steady state).
    regions = []
    for _rep in range(max(1, args.repeats)):
        model.profile_sampling(1 << 30)
        run_steps(args.warmup, primary_staged)
        model.profile_reset()
        model.profile_sampling(args.profile_every if args.profile_every > 0 else 1 << 30)
        barrier()
        t0 = time.perf_counter()
        ndet_r = run_steps(args.steps, primary_staged)
        torch.cuda.synchronize()
        el_local_r = time.perf_counter() - t0
        barrier()
        el_r = time.perf_counter() - t0
        el_r = parallel.max_over_ranks(el_r, dist, device=xdev)
        regions.append((el_r, el_local_r, ndet_r, model.profile_read()))
    el, el_local, ndet, stats = sorted(regions, key=lambda r: r[0])[len(regions) // 2]  # the median region (by the job's time)
    prof_steps = max(1, (args.steps + args.profile_every - 1) // args.profile_every) if args.profile_every > 0 else 0

    # ---- per-rank record: which device each rank ran on and what it did alone
    props = torch.cuda.get_device_properties(local_rank)
    place = model.placement()  # NUMA node of the GPU and the CPUs the handle's host threads are pinned to
    mine = torch.tensor([rank, local_rank, B * args.steps / el_local, getattr(props, "pci_bus_id", -1),
                         getattr(props, "pci_device_id", -1), place["numa_node"], place["pinned_cpus"]],
                        dtype=torch.float64, device=xdev if dist is not None else dev)
    if dist is not None:
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
    else:
        allr = [mine]
    ranks = [{"rank": int(t[0]), "local_rank": int(t[1]), "fps": round(float(t[2]), 1),
              "pci": "%02x:%02x" % (int(t[3]), int(t[4])), "numa_node": int(t[5]), "pinned_cpus": int(t[6])}
             for t in (x.cpu() for x in allr)]
    ranks[rank]["cpu_list"] = place["cpu_list"]
    if dist is not None and not args.rehearse_one_gpu:
        assert len({r["pci"] for r in ranks}) == world or ranks[0]["pci"] == "-1:-1", "two ranks share a GPU: %s" % ranks

    def aggregate(st):
        agg = {}
        for s in st:
            key = s["name"].split(":")[0]
            a = agg.setdefault(key, dict(ms=0.0, launches=0, bytes=0.0, flops=0.0))
            a["ms"] += s["total_ms"]
            a["launches"] += s["launches"]
            a["bytes"] += s["bytes"]
            a["flops"] += s["flops"]
        return agg

    def stage_ms(st, steps):
        out = {}
        for k, v in aggregate(st).items():
            out[stage_of(k)] = out.get(stage_of(k), 0.0) + v["ms"] / steps
        return out

    extras = {}
    host_only = args.host_only
    if not args.no_extras:
        # ---- steady state: the same workload, >= 200 more steps behind the timed region (a 20-step sample holds one
        # pipeline fill and one drain in 15 ms; this figure does not)
        k_ss = max(200, args.steps)
        model.profile_sampling(1 << 30)
        torch.cuda.synchronize()
        model.host_stats_reset()
        t1 = time.perf_counter()
        run_steps(k_ss, primary_staged)
        torch.cuda.synchronize()
        el_ss = time.perf_counter() - t1
        extras["steady_state"] = {"fps": round(B * k_ss / el_ss, 1), "steps": k_ss,
                                  "what": "this rank, same submit/wait loop as the timed region, run right behind it"}
        # ---- the host side of that run (ufd_host_stats: always-on counters of the library, no kernel events recorded)
        hs = model.host_stats()
        hs["usable_cpus"] = usable_cpus()
        hs["host_threads"] = host_threads
        hs["ms_per_batch"] = round(el_ss / k_ss * 1e3, 4)
        hs["what"] = ("steady-state run, kernel profiling off: wall microseconds per batch on the issuing worker (header scan, "
                      "staging memcpy, launch issue) and in the caller's ufd_wait; busy share of each context's issue worker; "
                      "per context the share of wall time its stream had a batch running and the idle gap between the end of one "
                      "batch and the first kernel of the next (HIP events on the stream)")
        extras["host"] = hs
    if not args.no_extras and not host_only:
        # ---- roofline pass: the pipeline loaded exactly as in the timed region (depth batches in flight over the
        # contexts), EVERY launch of every kernel timed with HIP events on the library's own streams
        model.profile_sampling(1)
        run_steps(6, primary_staged)
        model.profile_reset()
        run_steps(ROOF_STEPS, primary_staged)
        extras["kernel_stats_loaded"] = model.profile_read()
        extras["launch_shapes"] = model.profile_shapes()
        model.profile_sampling(1 << 30)
    if world == 1 and not args.no_extras and not host_only:
        # ---- the same workload over the other boundary
        if device_entropy and not args.annotate:
            other = not primary_staged
            run_steps(args.warmup, other)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.steps, other)
            torch.cuda.synchronize()
            extras["hbm_resident_fps" if other else "host_boundary_fps"] = round(B * args.steps / (time.perf_counter() - t1), 1)
        # ---- stage breakdown: device time per batch, every kernel timed (HIP events on the library's
        # streams), with the pipeline loaded (depth batches in flight over the contexts) and alone (one batch in flight)
        sb = {}
        for name, depth, k in (("loaded", args.depth, ROOF_STEPS), ("alone", 1, 12)):
            if name == "alone":
                model.profile_sampling(1)
                run_steps(3, primary_staged, depth=depth)
                model.profile_reset()
                run_steps(k, primary_staged, depth=depth)
                extras["kernel_stats_alone"] = model.profile_read()
            for stg, ms in stage_ms(extras["kernel_stats_" + name], k).items():
                sb.setdefault(stg, {})[name + "_ms"] = round(ms, 4)
        extras["stages"] = sb
        model.profile_sampling(1 << 30)
        # ---- N1: the whole Inferer::run iteration, inferer.rs:35-46 -- decode -> infer -> rectangles -> JPEG q95
        # re-encode on the GPU, annotated streams back in (pinned) host memory; label size = the router's 1280 x 720
        nab = 0 if args.annotate else min(nb, args.depth)  # (--annotate: that IS the timed workload)
        abs_ = [model.prep_annotate_batch(jpegs[i * B:(i + 1) * B], (1280, 720), out_bytes_per_frame=SW * SH) for i in range(nab)]

        def run_annotate(k):
            inflight, out_bytes = [], 0
            for s_ in range(k):
                if len(inflight) >= nab:
                    _, _, lens_ = model.wait(inflight.pop(0), collect=False)
                    out_bytes += sum(lens_)
                inflight.append(model.submit_annotate_batch(abs_[s_ % nab]))
            for t_ in inflight:
                _, _, lens_ = model.wait(t_, collect=False)
                out_bytes += sum(lens_)
            return out_bytes

        if nab:
            run_annotate(max(args.warmup, nab))
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ob = run_annotate(args.steps)
            torch.cuda.synchronize()
            el_a = time.perf_counter() - t1
            extras["annotate"] = {"fps": round(B * args.steps / el_a, 1), "bytes_out_per_frame": round(ob / (B * args.steps)),
                                  "what": "ufd_submit_annotate_batch: host JPEG bytes -> detections + annotated q95 4:2:0 JPEG in host memory"}
        del abs_
        # ---- per-frame latency at batch 1, one frame in flight, host bytes -> host detections
        lat = []
        lat_batches = [model.prep_annotate_batch([jpegs[i % len(jpegs)]], (1280, 720)) for i in range(8)] if args.annotate else None
        for i in range(60):
            j = jpegs[i % len(jpegs)]
            t1 = time.perf_counter()
            if args.annotate:  # (buffers prepared once, as a server's ring slots are: the clock sees submit + wait)
                model.wait(model.submit_annotate_batch(lat_batches[i % 8]), collect=False)
            else:
                model.infer_jpeg(j)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat = sorted(lat[10:])
        extras["latency_ms_batch1"] = {"median": round(lat[len(lat) // 2], 3), "p90": round(lat[int(len(lat) * 0.9)], 3),
                                       "what": ("ufd_annotate_jpeg_batch of one frame (host bytes -> detections + annotated JPEG in host memory)"
                                                if args.annotate else "ufd_infer_jpeg, host bytes -> host detections") +
                                               ", one %dx%d frame at a time" % (SW, SH)}

    verified = None
    if rank == 0 and not args.no_extras and not host_only:
        # ---- after the clock: detections of the batches just timed vs the CPU oracle
        from concurrent.futures import ThreadPoolExecutor

        import oracle

        oracle.build()
        annot_checked = None
        if args.annotate:  # the annotated streams of a few frames, byte for byte against the oracle's (libjpeg-turbo-pinned) encoder
            annot_checked = 0
            for j in jpegs[:min(4, len(jpegs))]:
                adets, stream = model.annotate_jpeg(j, (1280, 720))
                frame = oracle.draw_labels(oracle.jpeg_decode_rgb(j), np.array([list(b_) + [c_] for b_, c_ in adets], np.float32).reshape(-1, 5), 1280, 720)
                if stream != oracle.jpeg_encode_rgb(frame, 95):
                    raise SystemExit("bench.py: annotated JPEG differs from the oracle's")
                annot_checked += 1
        nver = min(2, nb)
        bts = get_batches(primary_staged)[:nver]
        tickets = [(model.submit_staged(b) if primary_staged else model.submit_jpeg_batch(b)) for b in bts]
        got = []
        for t, b in zip(tickets, bts):
            model.wait(t, collect=False)
            arr = np.frombuffer(b.out, np.float32).reshape(b.count, model.det_cap, 5)
            got += [(arr[i, :min(b.cnt[i], model.det_cap)].copy(), int(b.cnt[i])) for i in range(b.count)]
        with ThreadPoolExecutor(min(32, usable_cpus())) as ex:
            refs = list(ex.map(lambda j: oracle.infer_jpeg(j, W, H, weights, priors, 0.5, 0.5), jpegs[:nver * B]))
        from oracle.compare import match_detections

        max_err, bad, excused, unexplained, over_cap = 0.0, 0, 0, 0, 0
        for (g, n), r in zip(got, refs):
            if n > model.det_cap:  # (more detections than the bench's output rows hold: compare what was returned)
                r = r[:len(g)]
            # matched as sets (detections whose confidences differ by less than fp32 rounding may swap places); a detection
            # without a partner is excused only when its decision provably sat on a threshold at fp32 resolution -- the
            # rule of tests/helpers.py:assert_dets_match (oracle/compare.py)
            c = match_detections(g, r, 0.5, 0.5, atol=1e-4)  # (the tests' tolerance; north_star's 1e-3 is the bar on max_err below)
            max_err = max(max_err, c["max_err"])
            bad += len(g) != len(r)
            n_left = len(c["left_got"]) + len(c["left_ref"])
            excused += n_left - len(c["not_borderline"])
            unexplained += len(c["not_borderline"])
            # the cap of tests/helpers.py:assert_dets_match: borderline leftovers stay below 1 % of a frame's list (or 2)
            over_cap += n_left > max(2, 0.01 * max(len(g), len(r)))
        verified = {"frames": len(got), "max_abs_err": max_err, "count_mismatch_frames": bad, "borderline_detections_excused": excused,
                    "annotated_streams_byte_identical": annot_checked,
                    "unexplained_detections": unexplained,
                    "against": "CPU oracle (oracle/) on the same JPEG bytes, every detection of every frame; tolerance 1e-3 (north_star); "
                               "a count mismatch passes only when every unmatched detection sits within 1e-4 of the "
                               "confidence threshold or 1e-3 of the IoU threshold, such detections stay below max(2, 1 %) of the frame's list "
                               "and such frames below max(1, 2 %) of the sample"}
        verified["frames_over_the_excuse_cap"] = over_cap
        if max_err > 1e-3 or unexplained or over_cap or bad > max(1, len(got) // 50):
            raise SystemExit("bench.py: detections differ from the CPU oracle: %s" % verified)

    if rank == 0:
        frames = world * B * args.steps
        # ---- roofline of the dominant kernel INSTANCE: device time from HIP events on the library's streams.  Source:
        # the fully sampled loaded pass (ROOF_STEPS steps behind the timed region, same pipeline depth, every launch
        # timed) -- the timed region itself is only sampled every --profile-every-th step so that the event packets stay
        # out of `value`; its (few) launches are kept beside it as `timed_region_sample`.
        loaded = extras.get("kernel_stats_loaded")
        agg = aggregate(loaded if loaded else stats)
        agg_timed = aggregate(stats)
        kern = {k: v for k, v in agg.items() if not k.startswith(("h2d_", "host_"))}
        # Counter files are per WORKLOAD as well as per build: bytes and busy shares collected on C3 (UltraFace-640, 640x480
        # 4:2:0, batch 32) say nothing about a batch-1 or 1280x720 line, so they are attached only to a line whose own
        # workload is the one they were collected on (the stamp tools/finish_profiles.py writes; files without one are C3's).
        my_workload = {"variant": args.variant, "src": "%dx%d" % (SW, SH), "batch": B, "subsampling": args.subsampling}
        c3_workload = {"variant": 640, "src": "640x480", "batch": 32, "subsampling": "4:2:0"}
        pmc, pmc_ok = {}, False
        try:  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes summarised by tools/pmc_traffic.py
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_latest.json")))
            pmc_ok = pmc.get("kernel_source_sha") == kernel_source_sha() and pmc.get("workload", c3_workload) == my_workload
        except Exception:
            pass

        sq, sq_ok = {}, False
        try:  # rocprofv3 --pmc SQ passes summarised by tools/sq_table.py (tools/pmc_kernel.sh), stamped like the traffic
            sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters_latest.json")))
            sq_ok = sq.get("kernel_source_sha") == kernel_source_sha() and sq.get("workload", c3_workload) == my_workload
        except Exception:
            pass

        ea, ea_ok = {}, False
        try:  # rocprofv3 --pmc TCC_EA0_RDREQ* passes summarised by tools/pmc_ea.sh, stamped like the traffic
            ea = json.load(open(os.path.join(ROOT, "profiles", "ea_reads_latest.json")))
            ea_ok = ea.get("kernel_source_sha") == kernel_source_sha() and ea.get("workload", c3_workload) == my_workload
        except Exception:
            pass

        def roof_of(key, d):
            """Roofline figures of one kernel instance from its aggregated launches."""
            sec = d["ms"] * 1e-3
            n = max(d["launches"], 1)
            gbs = d["bytes"] / sec / 1e9 if sec > 0 else 0.0
            tfl = d["flops"] / sec / 1e12 if sec > 0 else 0.0
            r = {"kernel": device_name(key), "launches": d["launches"], "avg_launch_us": round(d["ms"] * 1e3 / n, 2),
                 "algorithmic_bytes_per_launch": round(d["bytes"] / n), "algorithmic_flops_per_launch": round(d["flops"] / n),
                 "mfma_frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
            # (the roof the kernel sits closer to; on gfx950 fp32 MFMA and fp32 vector work share one datapath)
            if base_label(key) in MFMA_KERNELS and tfl / MFMA_F32_PEAK_TFLOPS > gbs / HBM_PEAK_GBS:
                r.update({"bound": "mfma", "achieved": round(tfl, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": round(tfl / MFMA_F32_PEAK_TFLOPS, 4)})
            else:
                r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": round(gbs / HBM_PEAK_GBS, 4)})
            t = None
            if pmc_ok:
                e = pmc.get("instances", {}).get(device_name(key)) or pmc.get("kernels", {}).get(device_name(base_label(key)))
                if e:
                    t = round(e["hbm_bytes_per_launch"])
            r["traffic"] = t
            r["traffic_ratio"] = round(t / (d["bytes"] / n), 3) if t and d["bytes"] else None
            # matrix-pipe / vector-pipe busy shares and resident waves per SIMD of this instance ALONE on the GPU (SQ counters,
            # one rocprofv3 --pmc pass per counter group at --depth 1; null when not collected for this build of the kernels)
            e = sq.get("instances", {}).get(device_name(key)) if sq_ok else None
            for f in ("mfma_busy", "valu_busy", "waves_per_simd", "wait_share"):
                r[f] = e.get(f) if e else None
            # of `traffic`'s reads, the bytes whose requests were ADDRESSED to local memory (TCC_EA0_RDREQ_DRAM x 128 B).  The
            # Infinity Cache sits behind that address decode and rocprofv3 has no counter for hits in it on this box, so this is
            # an upper bound of the HBM reads, not a measurement of them; beside it the average fabric read latency in L2 clocks
            e = ea.get("instances", {}).get(device_name(key)) if ea_ok else None
            r["dram_bytes"] = e.get("dram_bytes") if e else None
            r["ea_read_latency_clk"] = e.get("ea_read_latency_clk") if e else None
            return r

        roof = None
        if kern:
            # the kernel FUNCTION with the largest share of device time, then its heaviest template instance
            fam_ms = {}
            for k, v in kern.items():
                fam_ms[base_label(k)] = fam_ms.get(base_label(k), 0.0) + v["ms"]
            fam = max(fam_ms, key=fam_ms.get)
            dom = max((k for k in kern if base_label(k) == fam), key=lambda k: kern[k]["ms"])
            roof = roof_of(dom, kern[dom])
            roof["source"] = ("loaded pass: %d steps behind the timed region, %d batches in flight, every launch timed (HIP events on "
                              "the library's streams)" % (ROOF_STEPS, min(args.depth, nb))) if loaded else \
                             "timed region, every %d-th step sampled" % args.profile_every
            roof["sq_source"] = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_VALU / SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY at commit %s, same "
                                 "kernel sources, kernels alone (tools/pmc_kernel.sh, tools/sq_table.py)" % sq.get("commit", "?")) if sq_ok else \
                                "not collected for this build of the kernels on this workload"
            roof["traffic_source"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, KiB units, FETCH x2 on gfx950) at commit %s, "
                                      "same kernel sources, per template instance" % pmc.get("commit", "?")) if pmc_ok else \
                                     "not measured for this build of the kernels on this workload (tools/collect_profiles.sh refreshes it)"
            # every template instance of the same kernel function, each with its own flops / bytes / traffic
            roof["instances"] = [roof_of(k, v) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"]) if base_label(k) == fam]
            alone = aggregate(extras.get("kernel_stats_alone", [])).get(dom)
            if alone and alone["ms"] > 0:  # the same instance with one batch in flight (nothing else on the GPU)
                a = roof_of(dom, alone)
                roof["alone"] = {"avg_launch_us": a["avg_launch_us"], "launches": a["launches"], "mfma_frac": a["mfma_frac"],
                                 "hbm_frac": a["hbm_frac"],
                                 "what": "one batch in flight: the kernel has the GPU to itself; `frac` above is with %d batches in "
                                         "flight over four contexts, whose kernels share the GPU" % min(args.depth, nb)}
            tsamp = agg_timed.get(dom)
            if loaded and tsamp and tsamp["ms"] > 0:
                roof["timed_region_sample"] = {"launches": tsamp["launches"],
                                               "avg_launch_us": round(tsamp["ms"] * 1e3 / max(tsamp["launches"], 1), 2)}
        flops_frame = 798315520 if args.variant == 640 else 200837120  # 2 x MACs of the 52 convs
        out = {
            "metric": ("frames/sec end-to-end (decode->NMS), UltraFace-%d @ %dx%d" % (args.variant, W, H)) if not args.annotate else
                      ("frames/sec decode->NMS->rectangles->JPEG q95 re-encode (N1 side workload), UltraFace-%d @ %dx%d" % (args.variant, W, H)),
            "value": round(frames / el, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(el / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "UltraFace-" + str(args.variant) + ", one %dx%d synthetic JPEG stream per GPU (q90 %s%s, %d distinct "
                                   "frames, %s), batch=%d, seeded synthetic weights" % (
                                       SW, SH, args.subsampling, ", no DHT segments (MJPG)" if args.no_dht else "", args.pool,
                                       "DRI = %d MCU row(s)" % args.restart_rows if args.restart_rows else "no restart markers", B),
                       "global_batch": world * B, "parallelism": "streams x%d (one per GPU), RCCL weight broadcast only" % world,
                       "entropy_decode": "GPU kernels" if device_entropy else "host worker threads",
                       "timed_region": ("JPEG bytes resident in HBM (headers parsed at staging) -> detections in host memory"
                                        if primary_staged else
                                        "host JPEG bytes -> host detections (%s + PCIe included)" % (
                                            "header scan, JPEG bytes H2D" if device_entropy else "host Huffman, coefficient slabs H2D")),
                       "async_depth": min(args.depth, nb), "host_threads_per_rank": host_threads, "ranks": ranks},
            "roofline": roof,
            "whole_net_mfma_frac": round(frames / el * flops_frame / 1e12 / MFMA_F32_PEAK_TFLOPS / world, 4),
            "detections_per_frame": round(ndet / (B * args.steps), 2),
            "samples": {"fps": [round(world * B * args.steps / r[0], 1) for r in regions], "value_is": "the median region",
                        "what": "%d timed regions back to back, each %d untimed warm-up steps then exactly %d timed steps (barrier + synchronize on "
                                "both sides, max over ranks); the first one runs on a GPU whose clocks are still coming up after the idle "
                                "seconds of set-up (profiles/r6b/cold_start.txt)" % (len(regions), args.warmup, args.steps)},
        }
        if args.rehearse_one_gpu and world > 1:
            out["config"]["rehearsal"] = "all %d ranks shared cuda:0 and exchanged over gloo: a run of the N>1 script path, NOT a measurement" % world
        out["config"].update({k: v for k, v in extras.items() if k.endswith("_fps")})
        if "steady_state" in extras:
            out["steady_state_fps"] = extras["steady_state"]["fps"]
            out["steady_state"] = extras["steady_state"]
        if "host" in extras:
            out["host"] = extras["host"]
        if "stages" in extras:
            out["stages"] = extras["stages"]
        if "latency_ms_batch1" in extras:
            out["latency_ms_batch1"] = extras["latency_ms_batch1"]
        if "annotate" in extras:
            out["annotate"] = extras["annotate"]
        if verified is not None:
            out["verified"] = verified
        ksteps = ROOF_STEPS if loaded else prof_steps
        if ksteps:
            gpu_ms = sum(v["ms"] for v in kern.values())
            out["gpu_ms_per_step"] = round(gpu_ms / ksteps, 3)
            out["kernels_ms_per_step"] = {k: round(v["ms"] / ksteps, 4) for k, v in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
        if extras.get("launch_shapes"):
            # how every launch of a batch sits on the GPU: workgroups as issued, workgroups a CU holds at once (registers, LDS:
            # the runtime's occupancy query), rounds over the 256 CUs and how full the last one is (ufd_profile_shapes)
            out["launch_shapes"] = {s.pop("name"): s for s in extras["launch_shapes"]}
        dump = os.environ.get("UFD_BENCH_DUMP")
        if dump:
            with open(dump, "w") as f:
                json.dump({"steps": prof_steps, "batch": B, "stats": stats, "roof_steps": ROOF_STEPS,
                           "stats_loaded": extras.get("kernel_stats_loaded"), "stats_alone": extras.get("kernel_stats_alone")}, f, indent=1)
        if not args.no_cpu_baseline and world == 1 and not host_only:
            out["cpu_baseline"] = cpu_baseline(jpegs[:64], weights, priors, args.cpu_seconds, W, H, annotate=args.annotate)
        print(json.dumps(out), flush=True)
    if staged_batches:
        for b in staged_batches:
            model.free_staged(b)
    model.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
