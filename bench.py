#!/usr/bin/env python3
"""Headline benchmark: images/s of one ×4 SR 64×64→256×256 TRAINING step of the full VmambaIR UNet
(BASELINE.json ``metric``; workload = ``configs[1]``: MambaSISR6 dim 48, blocks [15,1,1,1] + 15
refinement, bf16 autocast, batch 8 per MI355X, fwd + L1 loss + bwd + Adam + EMA, exactly the
reference step: SRGAN/options/MambaSISR15_x4.yml:55-90, SRGAN/VmambaIR/models/MambaSISR_model.py:120-147).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by ``python -m torch.distributed.run --nproc-per-node N ...``: one process per
  GPU, DDP over RCCL, the image batch sharded by rank (per-GPU batch fixed => weak scaling).
Prints ONE JSON line on rank 0 with the metric, ``roofline`` (dominant scan kernel: algorithmic
bytes / HIP-event kernel time, measured inside the timed region by the library's own events) and
``cpu_baseline`` (the same training step on the host cores with the CPU oracle as the scan, N = 1
only, one batch-1 step).  Synthetic data, random-init weights (no network on the box).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# kernel arguments in device memory: shortens every launch / graph node on MI300-class GPUs
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NET = dict(type="MambaSISR6", inp_channels=3, out_channels=3, dim=48, num_blocks=[15, 1, 1, 1],
           num_refinement_blocks=15, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
           LayerNorm_type="WithBias")
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
_T0 = time.time()


def log(msg):
    print(f"[bench +{time.time() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def usable_cores() -> int:
    """cores this process may really use: affinity mask, cgroup quota, capped at 64"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def make_step(net, ema_params, opt, autocast_dtype, device_type):
    params = [p for p in net.parameters()]

    def step(lq, gt):
        opt.zero_grad(set_to_none=True)
        with torch.autocast(device_type, dtype=autocast_dtype, enabled=autocast_dtype is not None):
            out = net(lq)
        loss = F.l1_loss(out.float(), gt)
        loss.backward()
        opt.step()
        with torch.no_grad():  # model_ema(decay=0.999), MambaSISR_model.py:146-147
            torch._foreach_mul_(ema_params, 0.999)
            torch._foreach_add_(ema_params, [p.detach() for p in params], alpha=0.001)
        return loss

    return step


def collect_prof(lib):
    """all non-empty profiler buckets -> list of dicts"""
    recs = []
    for which in (0, 1):
        for variant in range(12):
            for io, name in ((0, "f32"), (1, "f16"), (2, "bf16")):
                ms, n, by = C.c_double(), C.c_longlong(), C.c_double()
                if lib.oss_prof_collect(which, variant, io, C.byref(ms), C.byref(n), C.byref(by)) != 0:
                    continue
                if n.value:
                    recs.append(dict(kernel="oss_scan_fwd_kernel" if which == 0 else "oss_scan_bwd_kernel",
                                     variant=variant, io=name, launches=n.value, total_ms=ms.value, alg_bytes=by.value))
    return recs


def cpu_baseline(seed=0):
    """One batch-1 training step of the same net on the host cores, scans routed to the CPU oracle
    (oracle/, the restatement of the reference's sequential selective_scan).  kind = "port"."""
    from oracle import oss_oracle
    from vmambair_amd.archs import build_network
    import vmambair_amd.ops  # noqa: F401

    cores = usable_cores()
    torch.set_num_threads(cores)
    oss_oracle.set_threads(cores)
    from oracle import cpu_twins
    cpu_twins.install()  # CPU dispatch of torch.ops.vmambair = oracle / plain torch references
    torch.manual_seed(seed)
    net = build_network(NET)
    for m in net.modules():  # the host baseline runs the reference's literal data flow (four flattenings)
        if hasattr(m, "omni"):
            m.omni = False
    ema = [p.detach().clone() for p in net.parameters()]
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99))
    step = make_step(net, ema, opt, None, "cpu")
    lq, gt = torch.rand(1, 3, 64, 64), torch.rand(1, 3, 256, 256)
    step(lq, gt)   # untimed: thread pools, allocator, first-call set-up
    n, t0 = 0, time.perf_counter()
    while True:    # a bounded sample: whole steps until >= 10 s of host work
        step(lq, gt)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= 10.0 or n >= 8:
            break
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} training steps (fwd+bwd+Adam+EMA) after one untimed, batch 1, 64x64 LQ, fp32, whole MambaSISR6 net; "
                      "scan = oracle/oss_scan_oracle.c (OpenMP), rest = torch CPU", "seconds": round(dt, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=8)
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--micro-streams", type=int, default=int(os.environ.get("VMAMBAIR_MICRO_STREAMS", "1")),
                    help="micro-batches of the per-GPU batch run as parallel branches of the step's graph (train_graph.py)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VMAMBAIR_BENCH_GRAPH", "1")),
                    help="1: replay the training step as one hipGraph (single GPU, or manual flat-gradient "
                         "all-reduce outside the graph for N > 1)")
    ap.add_argument("--skip-roofline", action="store_true",
                    help="graph mode: do not run the trailing eager steps that time the scan kernels (profiling runs)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP scan has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from vmambair_amd import _capi
    from vmambair_amd.archs import build_network
    lib = _capi.load()

    torch.manual_seed(0)
    net = build_network(NET).to(dev)
    acdt = torch.bfloat16 if args.dtype == "bf16" else None
    B = args.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(1000 + rank)  # per-rank shard of the synthetic batch
    lq = torch.rand(B, 3, 64, 64, device=dev, generator=g)
    gt = torch.rand(B, 3, 256, 256, device=dev, generator=g)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph:
        # whole step replayed as a hipGraph; N > 1: one flat-gradient all-reduce between two graphs
        from vmambair_amd.train_graph import GraphedTrainStep
        if world > 1:  # same initial weights on every rank (DDP's constructor broadcast)
            for p_ in net.parameters():
                dist.broadcast(p_.data, 0)
        step = GraphedTrainStep(net, lr=2e-4, betas=(0.9, 0.99), ema_decay=0.999, autocast_dtype=acdt,
                                micro_streams=args.micro_streams)
        log("capturing the training step")
        step.capture(lq, gt)
        log("captured")
    else:
        ema = [p.detach().clone() for p in net.parameters()]
        model = net
        if world > 1:
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], bucket_cap_mb=50,
                                                              gradient_as_bucket_view=True)
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
        step = make_step(model, ema, opt, acdt, "cuda")

    log(f"model on {dev}, warmup {args.warmup} steps")
    for i in range(args.warmup):
        step(lq, gt)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")

    lib.oss_prof_reset()
    lib.oss_prof_enable(0 if args.graph else 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(lq, gt)
    fence()
    dt = time.perf_counter() - t0
    lib.oss_prof_enable(0)
    log(f"timed {args.steps} steps in {dt:.3f}s")
    loss_val = float(loss.item())
    prof_note = "HIP events around every scan kernel launch inside the timed region"
    if args.graph and args.skip_roofline:
        prof_steps = 1
    elif args.graph:
        # kernels inside a replayed graph cannot be bracketed by host-recorded events: time the very
        # same kernels on the same tensors with eager steps right after the timed region
        ema2 = [p.detach().clone() for p in net.parameters()]
        for p_ in net.parameters():
            p_.grad = None
        opt2 = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
        eager = make_step(net, ema2, opt2, acdt, "cuda")
        eager(lq, gt)
        torch.cuda.synchronize()
        lib.oss_prof_reset()
        lib.oss_prof_enable(1)
        for _ in range(min(3, args.steps)):
            eager(lq, gt)
        torch.cuda.synchronize()
        lib.oss_prof_enable(0)
        prof_note = ("HIP events around every scan kernel launch in %d eager steps run right after the timed "
                     "region (the timed region replays a hipGraph)" % min(3, args.steps))
        prof_steps = min(3, args.steps)
    else:
        prof_steps = args.steps
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        recs = collect_prof(lib)
        roof = None
        if recs:
            dom = max(recs, key=lambda r: r["total_ms"])
            avg_ms = dom["total_ms"] / dom["launches"]
            achieved = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            kkey = f"{dom['kernel']} variant {dom['variant']} io {dom['io']}"
            traffic, traffic_note = None, "no PMC record for this kernel build"
            valu_busy = None
            try:  # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, tools/pmc_traffic.sh)
                rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_traffic.json")))
                if kkey in rec:
                    traffic = int(rec[kkey]["fetch_bytes"] + rec[kkey]["write_bytes"])
                    valu_busy = rec[kkey].get("valu_busy")
                    traffic_note = ("FETCH_SIZE (x2, gfx950) + WRITE_SIZE per dispatch of this kernel at u:(8,384,4096), "
                                    "profiles/r01_pmc_scan_traffic.txt; the excess over alg_bytes is the per-row-tile dB/dC "
                                    "partials (written here, re-read by the finishing kernel) and B/C re-read per row tile")
            except (OSError, ValueError, KeyError):
                pass
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": kkey,
                    # the scan is a recurrence with ~29 vector-ALU instructions per (element, state): its roof is the VALU
                    # issue rate, not HBM -- share of the kernel with a SIMD's VALU issuing, from the SQ counters
                    # (profiles/r01_pmc_sq_scan.txt, tools/pmc_sq.sh); DESIGN.md section 5
                    "valu_busy": valu_busy,
                    "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                    "alg_bytes_per_launch": round(dom["alg_bytes"] / dom["launches"]),
                    "all_scan_kernels": [
                        {"kernel": r["kernel"], "variant": r["variant"], "io": r["io"], "launches": r["launches"],
                         "avg_ms": round(r["total_ms"] / r["launches"], 4),
                         "alg_GBps": round(r["alg_bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)} for r in recs],
                    "scan_ms_per_step": round(sum(r["total_ms"] for r in recs) / prof_steps, 3),
                    "measured": prof_note}
        # achievable HBM bandwidth, same run (copy kernel, 1 GiB)
        n = 1 << 30
        src = torch.empty(n, dtype=torch.uint8, device=dev).fill_(1)
        dst = torch.empty_like(src)
        st = torch.cuda.current_stream().cuda_stream
        lib.oss_hbm_copy(src.data_ptr(), dst.data_ptr(), n, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.oss_hbm_copy(src.data_ptr(), dst.data_ptr(), n, st)
        e1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9
        if roof is not None:
            roof["copy_kernel_GBps"] = round(copy_gbps, 1)
        del src, dst

        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            log("cpu baseline (subprocess, 300 s limit)")
            import subprocess
            try:  # own process: no GPU context, own thread pools, bounded time
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True,
                                   text=True, timeout=300, env={**os.environ, "HIP_VISIBLE_DEVICES": ""})
                cpu = json.loads(r.stdout.strip().splitlines()[-1])
            except Exception as e:  # the headline number must survive a broken baseline leg
                cpu = {"error": str(e)[:200]}

        images = world * B * args.steps
        line = {
            "metric": "images/sec, x4 SR 64->256 training step (fwd+bwd+Adam+EMA), full VmambaIR UNet",
            "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: x4 SR 64x64 LQ, MambaSISR6 dim48 [15,1,1,1]+15, "
                                   f"{args.dtype} autocast (scan arithmetic f32), batch {B} per GPU",
                       "global_batch": world * B, "per_gpu_batch": B, "lq": [64, 64], "gt": [256, 256],
                       "parallelism": f"dp{world}", "step_launch": "hipGraph replay" if args.graph else "eager", "optimizer": "Adam 2e-4 (0.9,0.99) + EMA 0.999", "loss": "L1"},
            "final_loss": round(loss_val, 5), "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
