#!/usr/bin/env python3
"""bench.py — images classified / sec of the diffusion-classifier scoring path on MI355X.

A "step" = one `DiffusionClassifier.classify` call over one batch of synthetic images already
resident in HBM (BASELINE.json metric; default workload = config 2: CIFAR-10 32x32 UNet of
reference experiments/cifar10/inference.py:94-116, 10 classes x 50 trials, bf16).
  python bench.py --gpus N --steps K --warmup W
N>1: one rank per GPU under torch.distributed.run — either the driver launches that itself, or a plain
`python bench.py --gpus N` starts it as a CHILD process (before this process touches the GPU) and relays its output.
The (trial, image) pairs of the step are sharded over the ranks, one RCCL all-gather of the error slab per step
(dist.py).  Default: the global batch grows with N (per-GPU work fixed -> "scaling": "weak"); `--global-batch B`
fixes the total instead ("strong": what north_star quotes for CheXpert-256 1->8 GPUs).
Prints ONE JSON line (rank 0).  `roofline` = the dominant kernel family (MFMA implicit GEMM)
timed with HIP events on the launch stream inside the last timed step; `cpu_baseline` = the
oracle's reference-structured loop on the host cores (N=1 only, bounded sample); `parity` = a small sub-grid of
the same workload scored by the HIP path and by the CPU oracle (N=1 only, outside the timed region).
The batch x sits in PINNED host memory and is copied to HBM inside every timed step (SURVEY §8d counts the H2D of x in the metric:
196 KB per cfg2 step, 3 MB per CheXpert step — enqueued on the launch stream in front of the step's first kernel).
`other_workloads` (default flags only, after the headline's timed region): a few timed steps each of BASELINE configs 3, 5 and 4
(CheXpert-DWT UNet, DiT-B/4 f16, IPMSA-5 UNet) with value / ms_per_step / dominant-kernel roofline fraction, so that the driver's
record covers them; at N > 1 the CheXpert-DWT UNet runs there under STRONG scaling (fixed global batch: what north_star quotes).
`--simulate-rank r/N` (one GPU, no process group): run exactly rank r's share of an N-rank grid-sharded step of `--global-batch B`
images and report its time — a PROJECTION of the N-GPU step time (the all-gather of <= 100 KB per stage is not in it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    # name: (arch kwargs fn, encoder, classes, trials, default images per GPU per step, flop/forward)
    "cifar10-unet-10x50": ("cifar10_unet_kwargs", "nn", 10, 50, 16, 10.454e9),
    "small-unet-2x8": ("small_unet_kwargs", "nn", 2, 8, 8, None),
    "chexpert256-dwt-unet-2x100": ("chexpert_dwt_unet_kwargs", "nn", 2, 100, 2, 176.47e9),
    "ipmsa5-unet-5x200": ("ipmsa5_unet_kwargs", "nn", 5, 200, 1, 634.96e9),
    "chexpert256-dwt-dit-b4-2x250": ("chexpert_dit_b4_kwargs", "DiT", 2, 250, 2, 213.31e9),
    # BASELINE config 5' (models/chexpert-256-dit-b4.py:7-13 with wavelet_transform=False): the raw 3x256x256 image, 4096 tokens
    "chexpert256-dit-b4-raw-2x250": ("chexpert_dit_b4_raw_kwargs", "DiT", 2, 250, 1, 1314.97e9),
}
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA, MI355X_MICROARCH.md


def usable_cores():
    """Host threads this process may really use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole machine on a shared GPU box and oversubscribes badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("BENCH_CPU_THREADS", "64"))))


def relaunch_under_torchrun(n, argv):
    """`python bench.py --gpus N` typed by hand: start the N ranks as a child process group and relay rank 0's line.
    Runs before anything in this process has initialised the GPU (a process that has must never exec/replace itself)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cifar10-unet-10x50", choices=list(WORKLOADS))
    ap.add_argument("--images-per-gpu", type=int, default=None)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--units-per-launch", type=int, default=None)
    ap.add_argument("--no-share-trunk", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="fix the TOTAL images per step (strong scaling) instead of images per GPU (weak, default)")
    ap.add_argument("--stages", default=None,
                    help="multi-stage pruning variant, e.g. '10:5,50:1' = trials-so-far:classes-kept per stage (not the BASELINE metric)")
    ap.add_argument("--breakdown", default=None, help="write the per-op event timings of the last step to this JSON file")
    ap.add_argument("--simulate-rank", default=None, metavar="r/N",
                    help="one GPU, no process group: run rank r's share of an N-rank grid-sharded step (use with --global-batch); a projection")
    ap.add_argument("--other-workloads", dest="other_workloads", action="store_true", default=None,
                    help="after the headline: a few timed steps of BASELINE configs 3 / 5 / 4 (default: on with default flags)")
    ap.add_argument("--no-other-workloads", dest="other_workloads", action="store_false")
    ap.add_argument("--no-haar", action="store_true", help="skip the Haar DWT micro-measurement (2 x 201 MB of HBM, rank 0)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(relaunch_under_torchrun(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # DCAMD_BENCH_REHEARSE=1: rehearsal of the N-rank harness on a box with fewer GPUs than ranks — ranks share the devices
    # round-robin and talk over gloo (RCCL refuses two ranks on one device).  Never a measurement: the line says so in `data`.
    rehearse = os.environ.get("DCAMD_BENCH_REHEARSE") is not None and world > 1
    if rehearse:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import diffusion_classifier_amd as dca
    sim = None
    if args.simulate_rank:
        r_, n_ = (int(v) for v in args.simulate_rank.split("/"))
        assert world == 1 and 0 <= r_ < n_, "--simulate-rank r/N runs on one GPU without a process group"
        sim = (r_, n_)

    def build(workload, dtype, B_, stages=None, share_trunk=True):
        """Backbone + classifier of a workload with random-init weights and its synthetic batch, resident in HBM."""
        arch_fn, enc, classes, T, _, flop_fwd = WORKLOADS[workload]
        kw = dca.chexpert_dit_b4_kwargs(False) if arch_fn == "chexpert_dit_b4_raw_kwargs" else getattr(dca, arch_fn)()
        torch.manual_seed(0)
        backbone = dca.UNetCondition2D(**kw) if enc == "nn" else dca.DiT(**kw)
        size, cin = kw["sample_size"], kw["in_channels"]
        ev, keep = [T], [1]
        if stages:
            ev, keep = zip(*[tuple(int(v) for v in st.split(":")) for st in stages.split(",")])
            ev, keep, T = list(ev), list(keep), ev[-1]
        cfg = dict(pred_param="eps", schedule="cosine", noise_d=size, image_size=size, cfg_w=0.0, ema_beta=0.999, ema_warmup=0,
                   ema_update_freq=1, encoder_type=enc, classes=classes, n_stages=len(ev), evaluation_per_stage=ev,
                   n_keep_per_stage=keep, n_fast_classes=2, fast_classification=False, compute_dtype=dtype,
                   units_per_launch=args.units_per_launch, shard_grid=world > 1, simulate_rank=sim)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):      # the constructor prints the parameter count (as the reference does): stdout carries the JSON line only
            dc_ = dca.DiffusionClassifier(backbone, dca.Config(**cfg))
        dc_.ema.ema_model.share_trunk = share_trunk
        dc_ = dc_.to(dev)
        g = torch.Generator().manual_seed(0)
        dwt_ = "dwt" in workload
        if dwt_:
            # the DWT configs score x = haar_dwt2(x0)/2 of a [-1,1] image (dataset/chexpert.py:146-147), through the HIP kernel
            x0 = (torch.rand(B_, cin // 4, 2 * size, 2 * size, generator=g) * 2 - 1).to(dev)
            x_ = dca.wavelet_dec_2(x0, scale=0.5)
        else:
            x_ = (torch.rand(B_, cin, size, size, generator=g) * 2 - 1).to(dev)   # SURVEY §8d synthetic inputs, resident in HBM
        # the step's input as a dataloader hands it over: pinned host memory; `x_` is the device buffer every timed step copies it into
        xh_ = torch.empty(x_.shape, dtype=x_.dtype, pin_memory=True)
        xh_.copy_(x_)
        return dc_, (x_, xh_), dict(kw=kw, enc=enc, classes=classes, T=T, flop_fwd=flop_fwd, cfg=cfg, size=size, cin=cin, dwt=dwt_)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(dc_, xs_, steps, warmup, tag=""):
        """W untimed steps, then exactly K steps between barrier + synchronize fences; MAX over ranks.  Every step starts with the
        H2D copy of its batch (pinned host memory -> the resident device buffer, on the launch stream).  The last timed step
        carries a HIP-event pair around every op (on the launch stream)."""
        x_, xh_ = xs_
        torch.manual_seed(1234)
        for i in range(warmup):
            x_.copy_(xh_, non_blocking=True)
            dc_.classify(x_, rng="philox", seed=1234 + i)
            if rank == 0:
                print(f"[bench] {tag}warmup {i + 1}/{warmup}", file=sys.stderr, flush=True)
        fence()
        t0 = time.perf_counter()
        for i in range(steps):
            if i == steps - 1:
                dc_._timed_sink = []           # HIP-event pair around every op of this step, on the launch stream
            x_.copy_(xh_, non_blocking=True)   # the H2D of x is part of the step
            dc_.classify(x_, rng="philox", seed=1234 + warmup + i)
        fence()
        dt_ = time.perf_counter() - t0
        sink_, dc_._timed_sink = dc_._timed_sink, None
        tt = torch.tensor([dt_], dtype=torch.float64, device="cpu" if rehearse else dev)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item(), sink_

    def families(sink_):
        fam_ = {}
        for plan, ms in sink_:
            for meta, m in zip(plan.pb.meta, ms):
                f = fam_.setdefault(meta["family"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                f["ms"] += m; f["flops"] += meta["flops"]; f["bytes"] += meta["bytes"]; f["launches"] += 1
        return fam_

    arch_fn, enc, classes, T, ipg, flop_fwd = WORKLOADS[args.workload]
    ipg = args.images_per_gpu or ipg
    # CheXpert-256 workloads under N > 1: north_star quotes STRONG scaling (1 -> 8 GPUs on one batch).  Default global batch 8:
    # every rank of 8 keeps 8 x 100 x 2 / 8 = 200 units = one full launch (the floor of 192 units per launch, _units_per_launch)
    strong_default = world > 1 and args.workload.startswith("chexpert256") and not args.images_per_gpu and not args.global_batch
    if strong_default:
        args.global_batch = 8
    B = args.global_batch if args.global_batch else ipg * world
    dc, xs, info = build(args.workload, args.dtype, B, stages=args.stages, share_trunk=not args.no_share_trunk)
    x = xs[0]
    kw, cfg, size, cin, dwt = info["kw"], info["cfg"], info["size"], info["cin"], info["dwt"]
    T = info["T"]
    dt, sink = timed(dc, xs, args.steps, args.warmup)
    bd_first = (sink[0][0].pb.meta, sink[0][1]) if (args.breakdown and sink) else None

    # ---- roofline of the dominant kernel family, from the event timings of the last timed step ----
    def roofline_of(fam_, dtype, workload):
        dom_ = max(fam_, key=lambda k: fam_[k]["ms"])
        d = fam_[dom_]
        peak = PEAK_TFLOPS[dtype if "f32" not in dom_ else "f32"]
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
        return dom_, d, peak, achieved

    fam = families(sink)
    del sink
    dom, d, peak, achieved = roofline_of(fam, args.dtype, args.workload)
    # HBM traffic of that kernel from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled for the
    # gfx950 half-count of wide streaming reads + WRITE_SIZE, per launch), recorded under profiles/
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    headline_cfg = args.workload == "cifar10-unet-10x50" and args.dtype == "bf16"
    TRAFFIC_UNIT = ("HBM bytes/launch: the STORED rocprofv3 PMC average of this same command on the committed tree (profiles/pmc_traffic.json: "
                    "2 x FETCH_SIZE + WRITE_SIZE, separate passes) - not a counter of this run")

    def stored_traffic(workload, dtype, kernel):
        """Per-launch HBM bytes of `kernel` from the committed PMC passes of `bench.py --workload workload` (None: not recorded)."""
        if not os.path.exists(tf):
            return None
        allt = json.load(open(tf))
        key = "cifar10-unet-10x50" if workload == "cifar10-unet-10x50" and dtype == "bf16" else workload
        rec_t = (allt.get("workloads", {}).get(key) or (allt if key == "cifar10-unet-10x50" else {})).get(kernel)
        return round((2.0 * rec_t["fetch_kb_per_launch"] + rec_t["write_kb_per_launch"]) * 1024.0) if rec_t else None

    traffic = stored_traffic(args.workload, args.dtype, dom)
    roofline = dict(bound="mfma", kernel=dom, achieved=round(achieved, 2), peak=peak, unit="TFLOP/s",
                    frac=round(achieved / peak, 4), traffic=traffic, traffic_unit=TRAFFIC_UNIT,
                    alg_bytes_per_launch=round(d["bytes"] / d["launches"]), launches=d["launches"],
                    avg_launch_ms=round(d["ms"] / d["launches"], 5),
                    alg_gflop_per_launch=round(d["flops"] / d["launches"] / 1e9, 4),
                    note="peak = dense spec at 2.4 GHz (MI355X_MICROARCH.md); the clock this kernel actually holds is not sampled in this run — "
                         "see profiles/README.md (in-kernel s_memtime / s_memrealtime stamps of a diagnostic build)")
    total_ms = sum(v["ms"] for v in fam.values())
    kernels = {k: dict(ms=round(v["ms"], 3), share=round(v["ms"] / total_ms, 4), launches=v["launches"],
                       tflops=round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1) if v["flops"] else None,
                       gbps=round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)) for k, v in sorted(fam.items())}
    if rank == 0 and world == 1 and not args.no_haar and (dwt or headline_cfg):
        # the Haar lifting kernel (utils/wavelet.py:4-35 replacement) on a CheXpert-shaped batch: HBM-bound, 8 bytes per input value
        # (rank 0 of single-GPU runs of the DWT workloads and of the headline only: 2 x 201 MB of HBM)
        hx = torch.rand(256, 3, 256, 256, device=dev) * 2 - 1
        dca.wavelet_dec_2(hx)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            hy = dca.wavelet_dec_2(hx, scale=0.5)
        ev1.record()
        torch.cuda.synchronize()
        h_ms = ev0.elapsed_time(ev1) / 20
        kernels["haar_dwt2"] = dict(ms=round(h_ms, 4), share=0.0, launches=1, tflops=None, gbps=round(2 * hx.numel() * 4 / h_ms / 1e6, 1),
                                    note="256x3x256x256 f32 -> 256x12x128x128, outside the timed step; peak 8000 GB/s")
        del hx, hy
    if args.breakdown and rank == 0:
        rows = [dict(meta, ms=m) for meta, m in zip(*bd_first)] if bd_first else []
        with open(args.breakdown, "w") as fh:
            json.dump(dict(families=kernels, ops=rows), fh, indent=1)

    value = B * args.steps / dt
    if sim:
        par_mode = f"PROJECTION: rank {sim[0]} of {sim[1]} of a grid-sharded step, run alone on one GPU (no process group, no all-gather)"
    else:
        par_mode = f"grid-shard x{world} ({'strong: fixed global batch' if args.global_batch else 'weak: fixed images per GPU'})" if world > 1 else "single"
    rec = dict(metric="images classified/sec (node), CIFAR-10 10-class x 50-step ELBO scoring"
               if args.workload == "cifar10-unet-10x50" else f"images classified/sec (node), {args.workload}",
               value=round(value, 3), unit="images/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(dt / args.steps * 1e3, 3), higher_is_better=True,
               scaling="strong" if args.global_batch else "weak", vs_baseline=None,
               dtype=args.dtype, data="synthetic" + (" (REHEARSAL: ranks share devices, gloo — not a measurement)" if rehearse else "")
               + (f" (PROJECTION of a {sim[1]}-GPU step from rank {sim[0]}'s share on one GPU — not a measurement of {sim[1]} GPUs)" if sim else ""),
               config=dict(workload=args.workload, images_per_step=B, classes=classes, trials=T,
                           forwards_per_image=classes * T, share_trunk=not args.no_share_trunk,
                           stages=args.stages, input="haar_dwt2(x0)/2 (HIP kernel)" if dwt else "uniform [-1,1]",
                           h2d=f"x ({x.numel() * x.element_size()} bytes) copied from pinned host memory to HBM inside every timed step (in value)",
                           parallelism=par_mode),
               roofline=roofline, kernels=kernels)
    if flop_fwd:
        rec["ref_equiv_tflops"] = round(value * classes * T * flop_fwd / 1e12, 1)     # reference-equivalent FLOPs

    # ---- the other BASELINE configurations, a few timed steps each (after the headline's timed region) ----
    default_flags = (headline_cfg and not args.stages and not sim and not args.no_share_trunk and args.units_per_launch is None
                     and not args.images_per_gpu and (world > 1 or not args.global_batch))
    do_other = default_flags if args.other_workloads is None else args.other_workloads
    if do_other and not sim:
        dc._score_plans.clear()
        torch.cuda.empty_cache()
        # (workload, dtype, images per step, timed steps, warmup, scaling).  The CheXpert-DWT UNet runs on ONE global batch of 8 at
        # every N (strong scaling, what north_star quotes: 8 x 100 x 2 / 8 = 200 units per rank at N = 8); the DiT and the IPMSA
        # UNet are single-GPU records (the N > 1 run stays short)
        others = [("chexpert256-dwt-unet-2x100", "bf16", 8, 3, 1, "strong")]
        if world == 1:
            others += [("chexpert256-dwt-dit-b4-2x250", "f16", 2, 3, 1, "weak"), ("ipmsa5-unet-5x200", "bf16", 1, 3, 1, "weak"),
                       ("chexpert256-dit-b4-raw-2x250", "f16", 1, 2, 1, "weak")]
        ow = {}
        for wl, dt_o, B_o, st_o, wu_o, sc_o in others:
            if rank == 0:
                print(f"[bench] other workload {wl} ({dt_o}, {B_o} images per step) ...", file=sys.stderr, flush=True)
            dco, xo, io = build(wl, dt_o, B_o)
            dto, sko = timed(dco, xo, st_o, wu_o, tag=wl + " ")
            fo = families(sko)
            domo, do_, pko, aco = roofline_of(fo, dt_o, wl)
            val_o = B_o * st_o / dto
            ow[wl] = dict(value=round(val_o, 3), unit="images/s", ms_per_step=round(dto / st_o * 1e3, 2), steps=st_o, warmup=wu_o, dtype=dt_o,
                          n_gpus=world, images_per_step=B_o, scaling=sc_o, classes=io["classes"], trials=io["T"],
                          ref_equiv_tflops=round(val_o * io["classes"] * io["T"] * io["flop_fwd"] / 1e12, 1),
                          roofline=dict(bound="mfma", kernel=domo, achieved=round(aco, 2), peak=pko, unit="TFLOP/s", frac=round(aco / pko, 4),
                                        traffic=stored_traffic(wl, dt_o, domo), traffic_unit=TRAFFIC_UNIT,
                                        alg_bytes_per_launch=round(do_["bytes"] / do_["launches"]),
                                        launches=do_["launches"], avg_launch_ms=round(do_["ms"] / do_["launches"], 5),
                                        share_of_step=round(do_["ms"] / sum(v["ms"] for v in fo.values()), 4)))
            del dco, xo, sko, fo
            torch.cuda.empty_cache()
        rec["other_workloads"] = ow
    # producer-side GroupNorm launches (csrc/epi_pn.h) wait for their sample's other workgroups with a bounded poll: a timed-out wait
    # poisons its outputs and is counted; any count invalidates the run
    rec["pn_timeouts"] = int(dca._lib.lib().dc_pn_timeouts())
    assert rec["pn_timeouts"] == 0, rec["pn_timeouts"]
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- CPU oracle legs (N=1 only, after the timed region): parity sub-grid, then the timed CPU baseline ----
    if world == 1 and not (args.no_cpu_baseline and args.no_parity):
        import oracle
        cores = usable_cores()
        torch.set_num_threads(cores)
        lowp_kw = dict(lowp=True) if enc == "nn" else dict(lowp=True, lowp_dtype=torch.float16 if args.dtype == "f16" else torch.bfloat16)

        def make_oracle(lowp, T_):
            okw = dict(kw, **(lowp_kw if lowp else {}))
            ob = (oracle.OracleUNetCondition2D(**okw) if enc == "nn" else oracle.OracleDiT(**okw))
            if lowp and enc == "nn" and args.dtype == "f16":
                ob._q = lambda t_: t_.to(torch.float16).float()
            ob.load_state_dict(dc.model.state_dict())
            oc_ = oracle.OracleDiffusionClassifier(ob, oracle.AttrBag(**dict(cfg, n_stages=1, evaluation_per_stage=[T_], n_keep_per_stage=[1])))
            if oc_.encoder is not None:
                oc_.encoder.load_state_dict(dc.encoder.state_dict())
            return oc_

    if world == 1 and not args.no_parity:
        # SURVEY §8d "parity gates reported with every number": the same weights and inputs, a sub-grid of the workload
        # (pb images x pt trials x ALL classes, draws injected), HIP f32 vs the fp32 oracle (bar 1e-4) and the HIP
        # compute dtype vs the oracle with the kernels' storage rounding / under torch autocast / in fp32.  Two quantities:
        # the per-cell eps-MSE (what north_star bounds; dominated by ||eps||^2, so it barely sees a slightly wrong prediction) and
        # `pred_rel_l2` — the relative L2 error of the backbone outputs of the SCORED plan themselves, the worst (trial, image,
        # class) sample — plus label agreement over pb x pt x classes cells.
        pb_, pt_ = (8, 4) if flop_fwd is None or flop_fwd < 50e9 else (1, 1)
        pb_ = min(pb_, x.shape[0])
        print(f"[bench] parity sub-grid: {pb_} images x {pt_} trials x {classes} classes ...", file=sys.stderr, flush=True)
        gp = torch.Generator().manual_seed(4321)
        xp = x[:pb_].cpu()
        tp_ = torch.rand(pt_, pb_, generator=gp)
        ep_ = torch.randn(pt_, pb_, *xp.shape[1:], generator=gp)
        saved = (dc.config.compute_dtype, dc.config.evaluation_per_stage, dc.config.n_stages, dc.config.n_keep_per_stage)
        dc.config.evaluation_per_stage, dc.config.n_stages, dc.config.n_keep_per_stage = [pt_], 1, [1]
        rel = lambda a, b: float(((a.float() - b.float()).abs() / b.float()).max())

        def hip_preds():
            """Backbone outputs of the last classify call, [pt, pb, classes, C, H, W]: the score plan's prediction buffer (the
            sub-grid fits one micro-batch; pairs trial-major, units = pair x class)."""
            (sp,) = list(dc._score_plans.values())
            assert sp["n_bj"] == pt_ * pb_
            pv = sp["plan"].pred_view().float().cpu()
            bbc = dc.ema.ema_model.config
            pch = int(getattr(bbc, "patch_size", 0) or 0)
            if pch > 1:
                U, gg, _, _ = pv.shape
                img = pv[..., :pch * pch * bbc.out_channels].reshape(U, gg, gg, pch, pch, bbc.out_channels).permute(0, 5, 1, 3, 2, 4)
                img = img.reshape(U, bbc.out_channels, gg * pch, gg * pch)
            else:
                img = pv[..., :bbc.out_channels].permute(0, 3, 1, 2)
            return img.reshape(pt_, pb_, sp["k"], *img.shape[1:])

        def prel(got, ref):       # worst sample's relative L2
            g_, r_ = got.double().flatten(3), ref.double().flatten(3)
            return float(((g_ - r_).norm(dim=3) / r_.norm(dim=3).clamp_min(1e-30)).max())

        par = dict(subgrid=f"{pb_} images x {pt_} trials x {classes} classes, injected (t, eps)")
        o32 = make_oracle(False, pt_)
        l32, e32, p32 = o32.classify(xp, t=tp_, eps=ep_, return_errors=True, return_preds=True)
        dc.config.compute_dtype = "f32"
        dc._score_plans.clear()
        lg, eg = dc.classify(xp.to(dev), t=tp_, eps=ep_.to(dev), return_errors=True)
        par.update(f32_max_rel_eps_mse=rel(eg, e32), f32_bar=1e-4, f32_pred_rel_l2=prel(hip_preds(), p32),
                   f32_labels_equal=bool((lg.cpu() == l32).all()), f32_labels_compared=int(pb_))
        if args.dtype != "f32":
            dc.config.compute_dtype = args.dtype
            dc._score_plans.clear()
            lg, eg = dc.classify(xp.to(dev), t=tp_, eps=ep_.to(dev), return_errors=True)
            pg = hip_preds()
            olp = make_oracle(True, pt_)
            llp, elp, plp = olp.classify(xp, t=tp_, eps=ep_, return_errors=True, return_preds=True)
            par[f"{args.dtype}_max_rel_vs_storage_rounded_oracle"] = rel(eg, elp)
            par[f"{args.dtype}_max_rel_vs_fp32_oracle"] = rel(eg, e32)
            par[f"{args.dtype}_pred_rel_l2_vs_storage_rounded_oracle"] = prel(pg, plp)
            par[f"{args.dtype}_pred_rel_l2_vs_fp32_oracle"] = prel(pg, p32)
            if args.dtype == "bf16" and enc == "nn":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    lac, eac, pac = o32.classify(xp, t=tp_, eps=ep_, return_errors=True, return_preds=True)
                par["bf16_max_rel_vs_autocast_oracle"] = rel(eg, eac)
                par["autocast_oracle_max_rel_vs_fp32_oracle"] = rel(eac, e32)
                par["autocast_oracle_pred_rel_l2_vs_fp32_oracle"] = prel(pac, p32)
            # labels: agreement over all sub-grid images, and over the DECIDED ones — images whose best and second-best class means
            # (fp32 oracle) are further apart than twice the compute dtype's per-cell error; random-init weights leave the rest
            # within that error of a tie, where a flip says nothing
            srt_ = e32.mean(2).sort(1).values
            dec_ = ((srt_[:, 1] - srt_[:, 0]) / srt_[:, 0]) > 2.0 * rel(eg, e32)
            par["labels_decided"] = int(dec_.sum())
            par[f"{args.dtype}_label_agreement_on_decided_images"] = float((lg.cpu()[dec_] == l32[dec_]).float().mean()) if bool(dec_.any()) else None
            par[f"{args.dtype}_label_agreement_with_fp32_oracle"] = float((lg.cpu() == l32).float().mean())
            par[f"{args.dtype}_label_agreement_with_storage_rounded_oracle"] = float((lg.cpu() == llp).float().mean())
            par["labels_compared"] = int(pb_)
            del olp
        dc.config.compute_dtype, dc.config.evaluation_per_stage, dc.config.n_stages, dc.config.n_keep_per_stage = saved
        dc._score_plans.clear()
        rec["parity"] = {k: (round(v, 7) if isinstance(v, float) else v) for k, v in par.items()}
        del o32

    # ---- CPU baseline: the oracle's reference-structured loop on the host cores (N=1 only) ----
    if world == 1 and not args.no_cpu_baseline:
        oc = make_oracle(False, T)
        print(f"[bench] cpu baseline on {cores} host threads ...", file=sys.stderr, flush=True)
        bs_c = 2 if size <= 32 else 1
        xc = x[:bs_c].cpu()
        oc.config.evaluation_per_stage = [1]
        el, tr_c = 0.0, 0
        while tr_c < T and (tr_c == 0 or (el < 10.0 and el * (tr_c + 1) / tr_c < 30.0)):
            tc = time.perf_counter()
            oc.classify(xc)                                 # one trial x all classes, sequential forwards at batch bs_c
            el += time.perf_counter() - tc
            tr_c += 1
            print(f"[bench] cpu baseline: {tr_c} trial(s) x {classes} classes in {el:.1f} s", file=sys.stderr, flush=True)
        cpu_val = bs_c / (el * T / tr_c)
        rec["cpu_baseline"] = dict(value=round(cpu_val, 5), unit="images/s", cores=torch.get_num_threads(), kind="port",
                                   sample=f"{bs_c} images x {tr_c} of {T} trials x {classes} classes "
                                          f"({bs_c * tr_c * classes} unit-forwards, {el:.1f} s), scaled linearly to {T} trials")
    print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
