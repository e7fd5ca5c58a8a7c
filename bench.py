#!/usr/bin/env python3
"""Headline benchmark: images/s of one ×4 SR 64×64→256×256 TRAINING step of the full VmambaIR UNet
(BASELINE.json ``metric``; workload = ``configs[1]``: MambaSISR6 dim 48, blocks [15,1,1,1] + 15
refinement, bf16 autocast, batch 8 per MI355X, fwd + L1 loss + bwd + Adam + EMA, exactly the
reference step: SRGAN/options/MambaSISR15_x4.yml:55-90, SRGAN/VmambaIR/models/MambaSISR_model.py:120-147).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1: one process per GPU, DDP over RCCL, the image batch sharded by rank (per-GPU batch fixed => weak scaling).
  Either the caller starts the ranks (``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
  LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment), or -- when WORLD_SIZE is not set -- this script
  starts them itself (``launch_ranks``: re-executes itself under torch.distributed.run on 127.0.0.1 and a free port,
  passes the ranks' output through and returns their exit code; the reference's counterpart is
  ``python -m torch.distributed.launch --nproc_per_node=N``, SRGAN/train_S1.sh:1-8).
Prints ONE JSON line on rank 0 with the metric, ``roofline`` (dominant scan kernel: algorithmic
bytes / HIP-event kernel time, measured inside the timed region by the library's own events) and
``cpu_baseline`` (the same training step on the host cores with the CPU oracle as the scan, N = 1
only, one batch-1 step; plus BASELINE.json configs[0] -- ONE OSS block on (2,48,48,48), forward and
forward+backward, median of 3 -- as SURVEY.md section 8d defines it).  Synthetic data, random-init weights.

Other workloads of BASELINE.json (not the driver's default line):
  --global-batch 32      configs[2]: fixed global batch split over the ranks (32/16/8/4 per GPU), "scaling": "strong"
  --config deraining     configs[3]: Mamber32 [3,5,7,9]+2, (4,3,128,128) per GPU, AdamW 3e-4 / decay 1e-4 + clip_grad_norm 0.01
                         (Deraining/Deraining/Options/Deraining_mamber33.yml:52-103, image_restoration_model.py:144-173)
  --config realsr-tiled  configs[4]: MambaRealSR11 [6,2,2,1]+6, fp16, 512x512 -> 2048x2048 by the RealESRGANer tile rule,
                         one hipGraph per padded-tile shape (a "step" = one image), tile by tile and with the 4 tiles of a shape
                         stacked on the batch axis; tiles/s in ``config``
  --config srgan-split64 validation-time inference of the SRGAN tree (MambaSISRModel2.test: 64x64 cells, no overlap) on a
                         256x256 LQ image: eager cells / ONE hipGraph replayed per cell / the same graph on 16 stacked cells
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
NET_DERAIN = dict(type="Mamber32", inp_channels=3, out_channels=3, dim=48, num_blocks=[3, 5, 7, 9],
                  num_refinement_blocks=2, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
                  LayerNorm_type="WithBias", dual_pixel_task=False)
NET_REALSR = dict(type="MambaRealSR11", inp_channels=3, out_channels=3, scale=4, dim=48, num_blocks=[6, 2, 2, 1],
                  num_refinement_blocks=6, heads=[1, 2, 4, 8], ffn_expansion_factor=2.66, bias=False,
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


def make_eager(model, net, ema, derain, acdt):
    """the same step op by op (``--graph 0`` and the roofline leg): Adam + EMA (SR) or AdamW + clip 0.01 (Deraining)"""
    if not derain:
        opt = torch.optim.Adam(net.parameters(), lr=2e-4, betas=(0.9, 0.99), fused=True)
        return make_step(model, ema, opt, acdt, "cuda")
    opt = torch.optim.AdamW(net.parameters(), lr=3e-4, betas=(0.9, 0.999), weight_decay=1e-4, fused=True)

    def step(lq, gt):
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=acdt, enabled=acdt is not None):
            out = model(lq)
        loss = F.l1_loss(out.float(), gt)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(net.parameters(), 0.01)
        opt.step()
        return loss

    return step


def bench_realsr_tiled(args):
    """BASELINE.json configs[4]: RealSR inference 512x512 -> 2048x2048, tiled (RealESRGANer rule: tile 128 + halo 16), fp16, one
    hipGraph per padded-tile shape.  One "step" = one image.  Single GPU.  Also timed: tile 256 + halo 16 and the reference's
    default tile = 0 (untiled), each with its own scan roofline.  pre_pad 0: the 3-level UNet needs every window to
    be a multiple of 8 pixels (PixelUnshuffle), and 512 + the script's default pre-pad of 10 leaves a 10-pixel last column
    of cells -- the reference net raises on it just the same."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = torch.device("cuda", 0)
    from vmambair_amd import _capi
    from vmambair_amd.archs import build_network
    from vmambair_amd.infer import RealSREnhancer
    lib = _capi.load()
    # A-B timing only (test-suite setters): VMAMBAIR_SCAN_SEGMENTS="fwd,bwd" time segments per row, VMAMBAIR_SCAN_VARIANT="fwd,bwd" kernel variants
    if os.environ.get("VMAMBAIR_SCAN_SEGMENTS"):
        fs, bs = (int(v) for v in os.environ["VMAMBAIR_SCAN_SEGMENTS"].split(","))
        lib.oss_scan_set_segments(fs, bs)
    if os.environ.get("VMAMBAIR_SCAN_VARIANT"):
        fv, bv = (int(v) for v in os.environ["VMAMBAIR_SCAN_VARIANT"].split(","))
        lib.oss_scan_set_variant(fv, bv)
    # profiling only: VMAMBAIR_REALSR_LEGS="eager,graph_4_tiles_stacked_shapes_side_by_side" times just these legs (an eager leg of the
    # tiling first: the graph legs are compared with it) and prints them alone -- the kernel trace then ends on the leg wanted
    only_legs = [v for v in os.environ.get("VMAMBAIR_REALSR_LEGS", "").split(",") if v]
    torch.manual_seed(0)
    net = build_network(NET_REALSR).to(dev)
    img = torch.rand(1, 3, 512, 512, device=dev)
    res = {}
    ref = None
    # (round 5) tile 256: the reference's tile size is a free argument (RealSR/VmambaIR/utils.py:33, default 0 = the whole image in one
    # forward); 512 x 512 in tiles of 256 + halo 16 is FOUR tiles of ONE padded shape (272 x 272) = one stacked forward per image
    # instead of four (13 % halo pixels instead of 41 %), and closer to the untiled result the reference computes by default
    # (round 6) the reference's DEFAULT is tile = 0 (RealSR/VmambaIR/utils.py:32): the whole 512 x 512 image in ONE forward, L = 262 144
    # per scan row -- two more legs; and the headline VALUE stays on the baseline tiling of BASELINE.json configs[4] / rounds 2-4
    # (tile 128 + halo 16) so that it is comparable round over round; the other tilings are labelled lines of their own
    for name, graph, bt, conc, tile in (("eager", False, 1, False, 128), ("graph", True, 1, False, 128), ("graph_4_tiles_stacked", True, 4, False, 128),
                                        ("graph_4_tiles_stacked_shapes_side_by_side", True, 4, True, 128),
                                        ("tile256_eager", False, 1, False, 256), ("tile256_graph_4_tiles_stacked", True, 4, False, 256),
                                        ("untiled_eager", False, 1, False, 0), ("untiled_graph", True, 1, False, 0)):
        if only_legs and name not in only_legs:
            continue
        drv = RealSREnhancer(net, 4, tile=tile, tile_pad=16, pre_pad=0, half=True, use_graph=graph, batch_tiles=bt, concurrent_shapes=conc)
        runs = (lambda d=drv: d.tiled.tiles_run) if tile else (lambda d=drv: d.whole.calls)
        out = drv.enhance_tensor(img)   # capture / warm-up
        for _ in range(max(0, args.warmup - 1)):
            drv.enhance_tensor(img)
        torch.cuda.synchronize()
        n0 = runs()
        if graph:
            lib.oss_prof_reset()
        mark = graph and (bt == 1 or tile == 256 or bool(only_legs))   # kernel-trace markers (tools/prof_summary.py reads the LAST marked leg)
        if mark:
            lib.oss_prof_marker(1, torch.cuda.current_stream().cuda_stream)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = drv.enhance_tensor(img)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        if mark:
            lib.oss_prof_marker(2, torch.cuda.current_stream().cuda_stream)
        tiles = (runs() - n0) // args.steps
        assert tuple(out.shape) == (1, 3, 2048, 2048) and torch.isfinite(out.float()).all()
        if not graph:
            ref = out.float().clone()   # the eager run of the same tiling (the tilings are different approximations of the image)
        res[name] = {"s_per_image": round(dt, 4), "images_per_s": round(1.0 / dt, 3), "tiles_per_s": round(tiles / dt, 2),
                     "tiles_per_image": tiles, "graphs": drv.tiled.n_graphs if tile else drv.whole.n_graphs, "tiles_per_forward": bt,
                     "shapes_side_by_side": conc, "tile": tile, "tile_pad": 16 if tile else 0,
                     "max_abs_diff_vs_eager": float((out.float() - ref).abs().max())}
        del drv
        torch.cuda.empty_cache()

    if only_legs:
        print(json.dumps({"legs": res}), flush=True)
        return

    def scan_roofline(tile, shape):
        """roofline of the dominant scan kernel of one tiling: the same tiles once more, eager, one at a time, with the library's events on"""
        drv = RealSREnhancer(net, 4, tile=tile, tile_pad=16, pre_pad=0, half=True, use_graph=False)
        lib.oss_prof_reset()
        lib.oss_prof_enable(1)
        drv.enhance_tensor(img)
        torch.cuda.synchronize()
        lib.oss_prof_enable(0)
        recs = collect_prof(lib)
        if not recs:
            return None
        dom = max(recs, key=lambda r: r["total_ms"])
        ach = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
        kkey = f"{dom['kernel']} variant {dom['variant']} io {dom['io']}" + (" segmented" if dom["segmented"] else "")
        traffic, traffic_note, _ = pmc_lookup(lib, kkey, shape)
        per = round(dom["alg_bytes"] / dom["launches"])
        return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "traffic": traffic, "traffic_note": traffic_note, "traffic_over_algorithmic": None if not traffic else round(traffic / per, 2),
                "kernel": kkey, "call_shape": shape + " x 4 directions f16, omni form", "tile": tile,
                "avg_launch_ms": round(dom["total_ms"] / dom["launches"], 4), "launches": dom["launches"],
                "alg_bytes_per_launch": per,
                "segments": {"fwd_last_call": int(lib.oss_scan_last_segments(0))},
                "all_scan_kernels": [{"kernel": r["kernel"], "variant": r["variant"], "segmented": r["segmented"], "io": r["io"],
                                      "launches": r["launches"], "avg_ms": round(r["total_ms"] / r["launches"], 4),
                                      "alg_GBps": round(r["alg_bytes"] / (r["total_ms"] * 1e-3) / 1e9, 1)} for r in recs],
                "scan_ms_per_image": round(sum(r["total_ms"] for r in recs), 3),
                "measured": "HIP events around every scan launch of one eager pass over the same tiles, one tile at a time"}

    # padded tile of the baseline tiling: 128 + 2 x 16 = 160 x 160 interior tiles; tile 256: 272 x 272; untiled: 512 x 512
    roof = scan_roofline(128, "u:(1,96,25600)")
    roof256 = scan_roofline(256, "u:(1,96,73984)")
    roof0 = scan_roofline(0, "u:(1,96,262144)")
    base = {k: v for k, v in res.items() if v["tile"] == 128}
    g = max(base.values(), key=lambda r: r["images_per_s"])
    best256 = max((v for v in res.values() if v["tile"] == 256), key=lambda r: r["images_per_s"])
    best0 = max((v for v in res.values() if v["tile"] == 0), key=lambda r: r["images_per_s"])
    other = {"tile_256_plus_halo_16": {"images_per_s": best256["images_per_s"], "s_per_image": best256["s_per_image"],
                                       "note": "four tiles of ONE padded shape (272 x 272) = one stacked forward per image; a different approximation "
                                               "of the image than tile 128 (the channel branch pools per tile)", "roofline": roof256},
             "untiled_tile_0_reference_default": {"images_per_s": best0["images_per_s"], "s_per_image": best0["s_per_image"],
                                                   "note": "RealESRGANer's default tile = 0 (utils.py:32): the whole image in one forward, L = 262144",
                                                   "roofline": roof0}}
    print(json.dumps({
        "metric": "images/sec, x4 real-world SR inference 512x512 -> 2048x2048, tiled, fp16", "value": g["images_per_s"],
        "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(g["s_per_image"] * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[4]: MambaRealSR11 [6,2,2,1]+6 dim48, fp16 autocast (scan arithmetic f32), "
                               "no_grad, 512x512 LQ, RealESRGANer rule tile 128 + halo 16, pre_pad 0 (`value` = the fastest leg of THIS tiling, "
                               "the one rounds 2-4 and BASELINE.md quote; tile 256 and the untiled default are in `other_tilings`)",
                   "tiles_per_image": g["tiles_per_image"], "tiles_per_s": g["tiles_per_s"], "hipgraphs": g["graphs"],
                   "tiles_per_forward": g["tiles_per_forward"], "other_tilings": other, **res},
        "roofline": roof, "cpu_baseline": None}), flush=True)


def bench_srgan_split64(args):
    """``MambaSISRModel2.test`` (SRGAN/VmambaIR/models/MambaSISR2_model.py:99-193): the LQ image in 64x64 cells, one forward per
    cell.  Every cell has the same shape, so the forward is one hipGraph; cells are independent, so they can also share ONE
    forward on the batch axis.  MambaSISR6 [15,1,1,1]+15, 256x256 LQ -> 1024x1024 (16 cells), bf16 autocast."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU")
    dev = torch.device("cuda", 0)
    from vmambair_amd.archs import build_network
    from vmambair_amd.infer import Split64SR
    torch.manual_seed(0)
    net = build_network(NET).to(dev)
    lq = torch.rand(1, 3, 256, 256, device=dev)
    res, ref = {}, None
    for name, graph, bt in (("eager_cell_by_cell", False, 1), ("graph_cell_by_cell", True, 1), ("graph_16_cells_stacked", True, 16)):
        drv = Split64SR(net, 4, autocast_dtype=torch.bfloat16, use_graph=graph, batch_tiles=bt)
        out = drv(lq)
        for _ in range(max(0, args.warmup - 1)):
            drv(lq)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = drv(lq)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        assert tuple(out.shape) == (1, 3, 1024, 1024) and torch.isfinite(out).all()
        if ref is None:
            ref = out.clone()
        res[name] = {"s_per_image": round(dt, 4), "images_per_s": round(1.0 / dt, 3), "cells_per_s": round(16 / dt, 1),
                     "forwards_per_image": 16 // bt, "max_abs_diff_vs_eager": float((out - ref).abs().max())}
    best = max(res.values(), key=lambda r: r["images_per_s"])
    print(json.dumps({
        "metric": "images/sec, x4 SR validation inference in 64x64 cells (MambaSISRModel2.test), 256x256 LQ", "value": best["images_per_s"],
        "unit": "images/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(best["s_per_image"] * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "SRGAN-tree validation path: MambaSISR6 dim48 [15,1,1,1]+15, bf16 autocast, no_grad, 256x256 LQ in 16 "
                               "cells of 64x64 (SURVEY.md 8f row 3)", **res},
        "roofline": None, "cpu_baseline": None}), flush=True)


def pmc_lookup(lib, kkey, shape=None):
    """HBM bytes per launch of kernel ``kkey`` from the PMC record under profiles/ (separate rocprofv3 --pmc passes,
    tools/pmc_traffic.sh + tools/pmc_record.py) -- only when the record was measured on THIS build of the scan kernels
    (``oss_scan_build_id()``); a record of another build is reported as stale, never silently.
    Time-segmented calls are two launches (the profiler bucket times both): the local / carry pass is added to the main one.
    ``shape``: "u:(B,D,L)" of the call the line is about -- the record holds one entry per measured shape (`<key> @ <shape>`);
    with a shape given, an entry measured at ANOTHER shape is not used (round 4's batch-4 line quoted the Deraining record)."""
    prof_dir = os.path.join(ROOT, "profiles")
    build = lib.oss_scan_build_id().decode()
    extra = []
    if kkey.endswith(" segmented"):
        io = kkey.split(" io ")[1].split()[0]
        if "fwd" in kkey:
            extra = [kkey.replace(" segmented", " local pass")]
        else:
            extra = [f"oss_scan_bwd_carry_kernel rows 12 io {io}"]
    try:
        names = sorted((f for f in os.listdir(prof_dir) if f.endswith("pmc_traffic.json")), reverse=True)
        for f in names:
            rec = json.load(open(os.path.join(prof_dir, f)))
            if kkey not in rec:
                continue
            if shape is not None:
                if f"{kkey} @ {shape}" not in rec:
                    return None, f"no PMC record of this kernel at {shape} in profiles/{f}", None
                suffix = f" @ {shape}"
                kkey_s, extra = kkey + suffix, [k + suffix for k in extra]
            else:
                kkey_s = kkey
            if rec.get("_build_id") != build:
                return None, (f"stale: profiles/{f} was measured on scan-kernel build {rec.get('_build_id', '(unrecorded)')}, "
                              f"this library is {build}"), None
            e = rec[kkey_s]
            total = int(e["fetch_bytes"] + e["write_bytes"]) + sum(int(rec[k]["fetch_bytes"] + rec[k]["write_bytes"]) for k in extra if k in rec)
            note = f"FETCH_SIZE (x2, gfx950) + WRITE_SIZE per dispatch, separate rocprofv3 --pmc passes, build {build} [profiles/{f}]"
            if extra:
                note += "; main + " + ("local" if "fwd" in kkey else "carry") + " pass, at " + str(e.get("shape", ""))
            return total, note, e.get("valu_busy", e.get("valu_active_share_of_wave_cycles"))
    except (OSError, ValueError, KeyError):
        pass
    return None, "no PMC record for this kernel", None


def collect_prof(lib):
    """all non-empty profiler buckets -> list of dicts"""
    recs = []
    names = {0: "oss_scan_fwd_kernel", 1: "oss_scan_bwd_kernel", 2: "oss_scan_bwd_finish"}
    for which in (0, 1, 2):
        for variant in range(32):   # 16 + v: time-segmented launches of variant v (their own bucket)
            for io, name in ((0, "f32"), (1, "f16"), (2, "bf16")):
                ms, n, by, own = C.c_double(), C.c_longlong(), C.c_double(), C.c_double()
                if lib.oss_prof_collect2(which, variant, io, C.byref(ms), C.byref(n), C.byref(by), C.byref(own)) != 0:
                    continue
                if n.value:
                    recs.append(dict(kernel=names[which], variant=variant % 16, segmented=variant >= 16, io=name, launches=n.value,
                                     total_ms=ms.value, alg_bytes=by.value, own_bytes=own.value))
    return recs


def family_counts(lib, passes):
    """algorithmic bytes / entry-point calls per kernel family since ``oss_prof_family_enable(1)`` (include/vmambair_oss.h), per pass"""
    out = []
    for f in range(int(lib.oss_prof_family_count())):
        name, pat, by, calls = C.c_char_p(), C.c_char_p(), C.c_double(), C.c_longlong()
        if lib.oss_prof_family(f, C.byref(name), C.byref(pat), C.byref(by), C.byref(calls)) != 0:
            continue
        out.append({"family": name.value.decode(), "patterns": pat.value.decode().split("|"), "alg_bytes_per_step": by.value / passes,
                    "entry_calls_per_step": calls.value / passes})
    return out


def steady_state_kernel_trace(argv, timeout=300):
    """``rocprofv3 --kernel-trace`` of THIS script (a subprocess replaying the same captured step) -> per kernel name
    (launches per step, ms per step, avg us) inside the marker-bracketed timed region; None + reason when the tracer is not
    there or fails.  The rows are what tools/prof_summary.py prints as the STEADY STATE table of profiles/*rocprof*.txt."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    d = tempfile.mkdtemp(prefix="bench-trace-", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), *argv, "--steps", "3", "--warmup", "1",
               "--no-cpu-baseline", "--no-secondary", "--skip-roofline", "--no-non-scan"]
        r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, capture_output=True, text=True, timeout=timeout)
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        if r.returncode != 0 or not dbs:
            return None, f"rocprofv3 rc {r.returncode}, {len(dbs)} result databases: {(r.stderr or '')[-160:]}"
        cur = sqlite3.connect(dbs[0]).cursor()
        mb = cur.execute("select max(end) from kernels where name like '%oss_prof_marker_begin%'").fetchone()[0]
        me = cur.execute("select min(start) from kernels where name like '%oss_prof_marker_end%' and start > ?", (mb or 0,)).fetchone()[0]
        if not (mb and me):
            return None, "marker kernels not found in the trace"
        rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels where start>=? and end<=? "
                                "group by name order by 3 desc", (mb, me)))
        steps = sum(r_[1] for r_ in rows if "oss_adam_tick_kernel" in r_[0]) or 3
        return {"steps": steps, "wall_ms_per_step": (me - mb) / 1e6 / steps,
                "kernels": [{"name": n, "launches_per_step": c / steps, "ms_per_step": ms / steps, "avg_us": us} for n, c, ms, us in rows]}, None
    except Exception as e:   # noqa: BLE001  the headline must survive a broken tracer
        return None, f"{type(e).__name__}: {str(e)[:160]}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def non_scan_roofline(fams, trace):
    """the part of the step that is NOT a scan kernel under a roofline too: per kernel family, algorithmic
    bytes per step (the library's entry-point accounting, counted while the step was captured) over the family's kernel time per
    step in the steady-state trace; plus the ten largest non-scan kernels by time."""
    def short(n):
        n = n.replace("void oss::", "").replace("oss::", "")
        return n if len(n) <= 110 else n[:107] + "..."
    ks = trace["kernels"]
    fam_of = {}
    for k in ks:
        nm = k["name"]
        if "oss_scan_" in nm:
            fam_of[nm] = "scan"
            continue
        fam_of[nm] = next((f["family"] for f in fams if any(pt and pt in nm for pt in f["patterns"])), "other (vendor 3x3 convolutions / transposes, aten)")
    out = []
    for f in fams + [{"family": "other (vendor 3x3 convolutions / transposes, aten)", "alg_bytes_per_step": None, "entry_calls_per_step": None}]:
        mine = [k for k in ks if fam_of[k["name"]] == f["family"]]
        if not mine:
            continue
        ms = sum(k["ms_per_step"] for k in mine)
        n = sum(k["launches_per_step"] for k in mine)
        by = f["alg_bytes_per_step"]
        gbps = None if not by else by / (ms * 1e-3) / 1e9
        out.append({"family": f["family"], "launches_per_step": round(n, 1), "ms_per_step": round(ms, 4), "avg_us": round(1e3 * ms / max(n, 1e-9), 2),
                    "alg_MB_per_step": None if by is None else round(by / 1e6, 1), "alg_GBps": None if gbps is None else round(gbps, 1),
                    "frac_of_hbm_peak": None if gbps is None else round(gbps / HBM_PEAK_GBPS, 4)})
    out.sort(key=lambda r: -r["ms_per_step"])
    scan = [k for k in ks if fam_of[k["name"]] == "scan"]
    rest = [k for k in ks if fam_of[k["name"]] != "scan" and "oss_prof_marker" not in k["name"]]
    top_scan = max(scan, key=lambda k: k["ms_per_step"]) if scan else None
    fam_gbps = {r["family"]: r["alg_GBps"] for r in out}
    return {"launches_per_step": round(sum(k["launches_per_step"] for k in ks)), "kernel_ms_per_step": round(sum(k["ms_per_step"] for k in ks), 3),
            "wall_ms_per_step_under_the_tracer": round(trace["wall_ms_per_step"], 3),
            "scan": {"launches_per_step": round(sum(k["launches_per_step"] for k in scan), 1), "ms_per_step": round(sum(k["ms_per_step"] for k in scan), 3),
                     # the dominant scan kernel as the kernel trace saw it INSIDE the replayed graph (roofline.avg_launch_ms is taken from
                     # eager steps after the timed region, VERDICT r5 weak #7: the two must agree)
                     "dominant_kernel_in_graph": None if top_scan is None else {
                         "kernel": short(top_scan["name"]), "launches_per_step": round(top_scan["launches_per_step"], 1),
                         "avg_launch_ms": round(top_scan["avg_us"] * 1e-3, 4)}},
            "non_scan": {"launches_per_step": round(sum(k["launches_per_step"] for k in rest), 1), "ms_per_step": round(sum(k["ms_per_step"] for k in rest), 3),
                         "alg_MB_per_step": round(sum(r["alg_MB_per_step"] or 0 for r in out), 1)},
            "families": out,
            "top_kernels": [{"kernel": short(k["name"]), "family": fam_of[k["name"]], "launches_per_step": round(k["launches_per_step"], 1),
                             "avg_us": round(k["avg_us"], 2), "ms_per_step": round(k["ms_per_step"], 4),
                             "family_alg_GBps": fam_gbps.get(fam_of[k["name"]])} for k in rest[:10]],
            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "measured": "kernel time: rocprofv3 --kernel-trace of this script replaying the same captured step (3 steps between the marker "
                        "kernels, no vendor solver search in that run); bytes: algorithmic (each operand once, scratch partials excluded), "
                        "counted by the library's entry points while THIS run captured its step"}



def measure_valu_busy(shape, dtype, timeout=180):
    """share of a SIMD's time with a vector-ALU instruction issuing, for the scan kernels of the call ``shape`` = "B,D,L": ONE
    ``rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace`` pass over tools/scan_one.py in a subprocess (counters are
    never combined with other trace domains), averaged per kernel.  busy = ACTIVE_INST_VALU / (WAVE_CYCLES / waves per SIMD): the
    wide variants keep one workgroup per CU, so 12-wave workgroups put 3 waves on every SIMD (8-wave: 2).  None when the tool is missing."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="bench-pmc-", dir="/tmp")
    try:
        r = subprocess.run([exe, "--pmc", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                            sys.executable, os.path.join(ROOT, "tools", "scan_one.py")], cwd="/tmp", capture_output=True, text=True, timeout=timeout,
                           env={**os.environ, "TMPDIR": "/tmp", "SHAPE": shape, "DTYPE": dtype, "REPS": "3"})
        acc = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection*.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    a = acc.setdefault(row.get("Kernel_Name", "?"), {}).setdefault(row.get("Counter_Name"), [0.0, 0])
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
        out = {}
        for name, c in acc.items():
            m = re.search(r"oss_scan_(bwd2|fwd)_kernel<[^,]+, (\d+), (\d+), (\d+)", name)
            if not m or "SQ_ACTIVE_INST_VALU" not in c or "SQ_WAVE_CYCLES" not in c:
                continue
            waves = int(m.group(2)) if m.group(1) == "bwd2" else int(m.group(4))
            if waves % 4:
                continue
            act, cyc = c["SQ_ACTIVE_INST_VALU"][0] / c["SQ_ACTIVE_INST_VALU"][1], c["SQ_WAVE_CYCLES"][0] / c["SQ_WAVE_CYCLES"][1]
            out[("bwd" if m.group(1) == "bwd2" else "fwd") + f"_{waves}_waves"] = round(act * (waves // 4) / cyc, 4)
        return out or None
    except Exception:   # noqa: BLE001
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def cpu_config1(cores):
    """BASELINE.json configs[0] / SURVEY.md 8d "Config 1": ``MamberBlock(dim=48)`` (x2 SR, 48x48 LQ, d_state 16, ONE OSS block),
    ``torch.manual_seed(0); x = randn(2,48,48,48)``, fp32, forward and forward+backward, median of 3, on the host cores.
    The reference's literal data flow (four flattenings, per-direction scans) with the sequential CPU scan = oracle/."""
    import statistics
    from vmambair_amd.oss_block import MamberBlock
    torch.manual_seed(0)
    blk = MamberBlock(48, variant="srgan")
    for m in blk.modules():
        if hasattr(m, "omni"):
            m.omni = False
    x = torch.randn(2, 48, 48, 48)
    with torch.no_grad():
        blk(x)                                   # untimed first call
    fwd, both = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        with torch.no_grad():
            blk(x)
        fwd.append(time.perf_counter() - t0)
    for _ in range(3):
        xi = x.clone().requires_grad_()
        blk.zero_grad()
        t0 = time.perf_counter()
        blk(xi).square().mean().backward()
        both.append(time.perf_counter() - t0)
    return {"workload": "BASELINE.json configs[0]: MamberBlock(48) on (2,48,48,48), fp32, d_state 16", "cores": cores,
            "fwd_s": round(statistics.median(fwd), 4), "fwd_bwd_s": round(statistics.median(both), 4),
            "images_per_s_fwd_bwd": round(2.0 / statistics.median(both), 3), "median_of": 3,
            "reference_anchor_s": {"fwd": 2.3, "bwd": 46.4, "where": "reference MamberBlock through selective_scan_ref, 8 vCPU build "
                                                                      "container (BASELINE.md section 3); unpublished"}}


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
    cfg1 = cpu_config1(cores)
    return {"value": round(n / dt, 4), "unit": "images/s", "cores": cores, "kind": "port", "config1_block": cfg1,
            "sample": f"{n} training steps (fwd+bwd+Adam+EMA) after one untimed, batch 1, 64x64 LQ, fp32, whole MambaSISR6 net; "
                      "scan = oracle/oss_scan_oracle.c (OpenMP), rest = torch CPU", "seconds": round(dt, 2)}


def secondary_workloads():
    """BASELINE.json configs[3] and configs[4] in front of the driver: a few Deraining training steps and one
    RealSR tiled image, each in its own process after the headline's timed region, each with its dominant scan kernel's
    roofline fraction.  Bounded: 5 + 2 steps; a failure of this leg never touches the headline line."""
    import subprocess
    me = os.path.abspath(__file__)
    out = {}
    # (round 6, VERDICT r5 next #3a / #6) also: the headline net at the REFERENCE's own precision (it never autocasts, SURVEY.md
    # App. C) and two later stages of the Deraining tree's progressive schedule (patch 256 x batch 2, patch 384 x batch 1:
    # Deraining_mamber33.yml:27-30) with their time-segmented scan rooflines.  No vendor solver search on the added legs (start-up).
    for name, extra in (("fp32", ["--config", "sr", "--dtype", "fp32", "--steps", "5", "--warmup", "2", "--miopen-find", "0"]),
                        ("deraining", ["--config", "deraining", "--steps", "5", "--warmup", "2"]),
                        ("deraining_256", ["--config", "deraining", "--patch", "256", "--batch-per-gpu", "2", "--steps", "3", "--warmup", "1",
                                           "--miopen-find", "0"]),
                        ("deraining_384", ["--config", "deraining", "--patch", "384", "--batch-per-gpu", "1", "--steps", "3", "--warmup", "1",
                                           "--miopen-find", "0"]),
                        ("realsr_tiled", ["--config", "realsr-tiled", "--steps", "2", "--warmup", "1"])):
        t0 = time.time()
        try:
            r = subprocess.run([sys.executable, me, *extra, "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True,
                               timeout=240)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            roof = j.get("roofline") or {}
            out[name] = {"metric": j["metric"], "value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"],
                         "steps": j["steps"], "warmup": j["warmup"], "dtype": j["dtype"], "workload": j["config"]["workload"],
                         "roofline": {k: roof.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_note",
                                                                "avg_launch_ms", "launches", "alg_bytes_per_launch", "frac_with_finish",
                                                                "segments")},
                         "seconds": round(time.time() - t0, 1)}
            if name == "realsr_tiled":
                out[name]["tiles_per_s"] = j["config"].get("tiles_per_s")
                out[name]["tiles_per_forward"] = j["config"].get("tiles_per_forward")
                out[name]["other_tilings"] = j["config"].get("other_tilings")
        except Exception as e:   # noqa: BLE001
            out[name] = {"error": str(e)[:200]}
    return out


def launch_ranks(n, argv):
    """``--gpus N`` without a launcher around us: start the N ranks ourselves (one process per GPU, torch.distributed.run on
    127.0.0.1 -- the container hostname may not resolve -- and a free port), stream their output (rank 0's JSON line is the
    last stdout line) and return their exit code.  Under torch.distributed.run already (WORLD_SIZE set) this is never called."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    log(f"--gpus {n}: no WORLD_SIZE in the environment, starting the ranks: {' '.join(cmd)}")
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    return subprocess.call(cmd, env=env)


def rccl_preflight(dev, nbytes, world, timeout_s=120.0):
    """The first RCCL collective of the process, before anything is captured: an all-reduce of the size of the gradient exchange
    (one warm call creates the communicators, a second one is timed), run in a watchdog thread so that a hang becomes a reason
    instead of a dead job.  -> (ok, reason or None, ms or None)."""
    import threading
    box = {}

    def run():
        try:
            t = torch.ones(max(1, nbytes // 4), dtype=torch.float32, device=dev)
            dist.all_reduce(t)
            torch.cuda.current_stream().synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(t)
            e1.record()
            e1.synchronize()
            box["ms"] = e0.elapsed_time(e1)
            box["sum"] = float(t[0].item())
        except Exception as e:   # noqa: BLE001
            box["err"] = f"{type(e).__name__}: {str(e)[:200]}"
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return False, f"the {nbytes >> 20} MB RCCL all-reduce did not return within {timeout_s:.0f} s", None
    if "err" in box:
        return False, box["err"], None
    if abs(box["sum"] - float(world) ** 2) > 1e-3:   # ones -> world after the first call -> world^2 after the second
        return False, f"RCCL all-reduce returned {box['sum']} instead of {float(world) ** 2}", None
    return True, None, box["ms"]


def host_barrier():
    """a barrier over the CPU backend (gloo): works whatever state RCCL is in"""
    dist.all_reduce(torch.zeros(1))


def rendezvous_only(world, rank):
    """``--rendezvous-only`` (tests/test_ddp_gloo.py): the launcher path without a GPU -- the ranks meet over gloo, agree on
    the world size with one all-reduce, rank 0 prints a line.  Everything up to the first device call of a real run."""
    dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"rendezvous": "ok", "backend": "gloo", "n_ranks": dist.get_world_size(), "sum_of_ranks_plus_1": float(t.item()),
                          "master": f"{os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}"}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="default 8 (sr) / 4 (deraining)")
    ap.add_argument("--patch", type=int, default=128,
                    help="--config deraining: patch size of the progressive schedule (Deraining_mamber33.yml:27-30: gt_sizes "
                         "[128,160,192,256,320,384] with mini_batch_sizes [8,5,3,2,1,1] per GPU); default 128 = BASELINE.json configs[3]")
    ap.add_argument("--global-batch", type=int, default=0,
                    help="fixed GLOBAL batch split over the ranks (BASELINE.json configs[2]: 32 -> 32/16/8/4 per GPU); scaling = strong")
    ap.add_argument("--config", choices=["sr", "deraining", "realsr-tiled", "srgan-split64"], default="sr")
    ap.add_argument("--dtype", choices=["bf16", "fp32"], default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the trailing Deraining / RealSR legs of the default single-GPU run (the `secondary` block)")
    ap.add_argument("--micro-streams", type=int, default=int(os.environ.get("VMAMBAIR_MICRO_STREAMS", "1")),
                    help="micro-batches of the per-GPU batch run as parallel branches of the step's graph (train_graph.py)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VMAMBAIR_BENCH_GRAPH", "1")),
                    help="1: replay the training step as one hipGraph (single GPU, or manual flat-gradient "
                         "all-reduce outside the graph for N > 1)")
    ap.add_argument("--skip-roofline", action="store_true",
                    help="graph mode: do not run the trailing eager steps that time the scan kernels (profiling runs)")
    ap.add_argument("--miopen-find", type=int, default=None,
                    help="1: torch.backends.cudnn.benchmark = True, as the reference's training pipelines set it "
                         "(SRGAN/VmambaIR/train_pipeline.py:97, Deraining/basicsr/train.py:135) -- the vendor library searches its solvers "
                         "for the UNet skeleton's GEMM-shaped 3x3 convolutions during the warm-up (before the capture) instead of taking "
                         "its heuristic's pick: 228.1 -> 232.0 images/s, 37 s more start-up (profiles/r04_multirank_flow_check_and_"
                         "miopen_find.txt).  Default: 1 for the 16-bit training workloads, 0 for --dtype fp32 (3 minutes of search for "
                         "+0.3 %) and for the inference configs; VMAMBAIR_MIOPEN_FIND overrides the default")
    ap.add_argument("--no-non-scan", action="store_true",
                    help="skip `roofline.non_scan` (a second, kernel-traced run of this script: rocprofv3 --kernel-trace, about 40 s)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rendezvous-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 smoke run on a box with FEWER GPUs than ranks: every rank uses GPU (rank %% device_count) and the ranks talk "
                         "over gloo instead of RCCL (RCCL refuses two ranks on one device).  Exercises the whole multi-rank flow of this "
                         "script -- broadcast, two graphs, flat gradient all-reduce, max-over-ranks timing -- but NOT RCCL; its line is "
                         "marked and is not a measurement")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline()), flush=True)
        return
    if args.config == "realsr-tiled":
        return bench_realsr_tiled(args)
    if args.config == "srgan-split64":
        return bench_srgan_split64(args)

    if args.miopen_find is None:
        env = os.environ.get("VMAMBAIR_MIOPEN_FIND")
        args.miopen_find = int(env) if env is not None else (1 if (args.dtype == "bf16" and args.config in ("sr", "deraining")) else 0)
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks: use --nproc-per-node {args.gpus}")
    if args.rendezvous_only:
        return rendezvous_only(world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP scan has no CPU path)")
    if args.share_gpu:
        local_rank = local_rank % max(1, torch.cuda.device_count())
    if torch.cuda.device_count() < (1 if args.share_gpu else max(world, local_rank + 1)):
        raise SystemExit(f"--gpus {world}: rank {rank} needs GPU {local_rank}, but only {torch.cuda.device_count()} GPU(s) are visible "
                         "on this node (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl = {"ok": None, "fallback": None, "preflight_ms": None}
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # VMAMBAIR_SHARE_GPU_RCCL=1 (test of the fall-back on a 1-GPU box): the ranks share a GPU but take the RCCL path -- RCCL refuses
        # two ranks on one device, so the preflight fails and the host-staged exchange takes over
        if args.share_gpu and os.environ.get("VMAMBAIR_SHARE_GPU_RCCL", "0") != "1":
            dist.init_process_group("gloo")
            rccl.update(ok=False, fallback="--share-gpu: gloo by request")
        else:
            # CPU tensors over gloo, device tensors over RCCL: the host side (agreement between ranks, the timing reduction, barriers in
            # the fall-back) never depends on the state of RCCL
            dist.init_process_group("cpu:gloo,cuda:nccl")
            ok, why, ms = rccl_preflight(dev, 48 << 20, world)
            flag = torch.tensor([1.0 if ok else 0.0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank takes the same path
            agreed = bool(flag.item() >= 1.0)
            rccl.update(ok=agreed, preflight_ms=None if ms is None else round(ms, 3),
                        fallback=None if agreed else ("RCCL preflight failed" + (f" on this rank: {why}" if why else " on another rank") +
                                                      "; gradients are exchanged through pinned host memory over gloo"))
            log(f"RCCL preflight: {'ok, %.3f ms for 48 MB' % ms if agreed and ms is not None else rccl['fallback']}")
    via_host = world > 1 and not rccl["ok"]

    from vmambair_amd import _capi
    from vmambair_amd.archs import build_network
    lib = _capi.load()
    if os.environ.get("VMAMBAIR_SCAN_SEGMENTS"):   # A-B timing: "fwd,bwd" time segments per row (-1 heuristic, 1 off, n)
        fs, bs = (int(v) for v in os.environ["VMAMBAIR_SCAN_SEGMENTS"].split(","))
        lib.oss_scan_set_segments(fs, bs)

    torch.manual_seed(0)
    derain = args.config == "deraining"
    net = build_network(NET_DERAIN if derain else NET).to(dev)
    acdt = torch.bfloat16 if args.dtype == "bf16" else None
    B = args.batch_per_gpu or (4 if derain else 8)
    scaling = "weak"
    if args.global_batch:
        from vmambair_amd.ddp import split_global_batch
        try:
            B, scaling = split_global_batch(args.global_batch, world), "strong"
        except ValueError as e:
            raise SystemExit(f"--global-batch: {e}")
    g = torch.Generator(device=dev).manual_seed(1000 + rank)  # per-rank shard of the synthetic batch
    if derain:
        lq = torch.rand(B, 3, args.patch, args.patch, device=dev, generator=g)
        gt = torch.rand(B, 3, args.patch, args.patch, device=dev, generator=g)
    else:
        lq = torch.rand(B, 3, 64, 64, device=dev, generator=g)
        gt = torch.rand(B, 3, 256, 256, device=dev, generator=g)
    opt_kw = dict(lr=3e-4, betas=(0.9, 0.999), ema_decay=0.0, weight_decay=1e-4, clip_grad_norm=0.01) if derain else \
        dict(lr=2e-4, betas=(0.9, 0.99), ema_decay=0.999)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            host_barrier()
        torch.cuda.synchronize()

    if world > 1 and args.miopen_find:
        # the vendor library's solver search writes its picks to a per-user database: let rank 0 search first (one untimed eager
        # forward + backward, no collectives), then the other ranks, which find the entries -- eight searches at once would only
        # contend for that file
        def prewarm():
            with torch.autocast("cuda", dtype=acdt, enabled=acdt is not None):
                out = net(lq)
            F.l1_loss(out.float(), gt).backward()
            for p_ in net.parameters():
                p_.grad = None
            torch.cuda.synchronize()
        for first in (True, False):
            if (rank == 0) == first:
                prewarm()
            host_barrier()
        log("vendor solver search done (rank 0 first)")

    bucket_note = None
    if args.graph:
        # whole step replayed as a hipGraph; N > 1: the flat-gradient all-reduce (in buckets) between the backward and optimizer graphs
        from vmambair_amd.train_graph import GraphedTrainStep
        if world > 1:  # same initial weights on every rank (DDP's constructor broadcast)
            for p_ in net.parameters():
                if via_host:
                    c_ = p_.data.cpu()
                    dist.broadcast(c_, 0)
                    p_.data.copy_(c_)
                else:
                    dist.broadcast(p_.data, 0)
        # N > 1: the gradient exchange in 3 reverse-order buckets, each all-reduced on a side stream while the next one is flushed
        # (train_graph.py: grad_buckets); VMAMBAIR_GRAD_BUCKETS=1 = ONE all-reduce between two graphs (rounds 3-5); also settable on one
        # rank (A-B of what the bucketed flow costs without anything to overlap)
        buckets = int(os.environ.get("VMAMBAIR_GRAD_BUCKETS", "3" if world > 1 else "1"))
        if args.micro_streams > 1 or os.environ.get("VMAMBAIR_OVERLAP_WGRADS", "0") == "1":
            buckets = 1

        def make_step(nb):
            return GraphedTrainStep(net, autocast_dtype=acdt, micro_streams=args.micro_streams,
                                    split_graphs=os.environ.get("VMAMBAIR_BENCH_SPLIT", "0") == "1",   # A-B: the two-graph form of N > 1 on one rank
                                    overlap_wgrads=os.environ.get("VMAMBAIR_OVERLAP_WGRADS", "0") == "1", grad_buckets=nb,
                                    allreduce_via_host=via_host, **opt_kw)
        log("capturing the training step")
        lib.oss_prof_family_enable(1)     # algorithmic bytes per non-scan kernel family, per pass (warm-up passes + the captured one)
        try:
            step = make_step(buckets)
            step.capture(lq, gt)
        except Exception as e:   # noqa: BLE001
            if buckets <= 1:
                raise
            # the bucketed flow has never met a real multi-GPU box before the driver's run: fall back to the single exchange
            bucket_note = f"bucketed capture failed ({type(e).__name__}: {str(e)[:160]}); ONE all-reduce between two graphs instead"
            log(bucket_note)
            lib.oss_set_defer_wgrad(0)
            lib.oss_set_defer_finish(0)
            from vmambair_amd.ops import _common as _oc
            _oc._DEFER_KEEP = _oc._DEFER_OUTS = _oc._WGRAD_OPERANDS = _oc._WGRAD_STORAGES = _oc._WGRAD_FLUSHER = None
            torch.cuda.synchronize()
            lib.oss_prof_family_enable(1)
            step = make_step(1)
            step.capture(lq, gt)
        lib.oss_prof_family_enable(0)
        fam_counts = family_counts(lib, step.warmup + 1)
        log("captured")
    else:
        ema = [p.detach().clone() for p in net.parameters()]
        model = net
        if world > 1:
            model = torch.nn.parallel.DistributedDataParallel(net, device_ids=[local_rank], bucket_cap_mb=50,
                                                              gradient_as_bucket_view=True)
        step = make_eager(model, net, ema, derain, acdt)

    log(f"model on {dev}, warmup {args.warmup} steps")
    for i in range(args.warmup):
        step(lq, gt)
        torch.cuda.synchronize()
        log(f"warmup step {i} done")

    if args.graph and world > 1:
        step.time_allreduce()
    lib.oss_prof_reset()
    lib.oss_prof_enable(0 if args.graph else 1)
    st_ = torch.cuda.current_stream().cuda_stream
    lib.oss_prof_marker(1, st_)   # kernel-trace markers around the timed region (tools/prof_summary.py), outside the clock
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(lq, gt)
    fence()
    dt = time.perf_counter() - t0
    lib.oss_prof_marker(2, st_)
    lib.oss_prof_enable(0)
    log(f"timed {args.steps} steps in {dt:.3f}s")
    loss_val = float(loss.item())
    # peak device memory of the captured step (graph pool included) and what the deferred weight gradients held (ADVICE r3:
    # recorded products keep both operands alive until the grouped launch; bounded by VMAMBAIR_WGRAD_KEEP_MB, ops/_common.py)
    peak_mem_gb = round(torch.cuda.max_memory_allocated(dev) / 1e9, 3)
    wgrad_stats = getattr(step, "wgrad_stats", None) if args.graph else None
    allreduce_ms = step.collect_allreduce_ms() if (args.graph and world > 1) else None
    prof_note = "HIP events around every scan kernel launch inside the timed region"
    if args.graph and args.skip_roofline:
        prof_steps = 1
    elif args.graph:
        # kernels inside a replayed graph cannot be bracketed by host-recorded events: time the very
        # same kernels on the same tensors with eager steps right after the timed region
        ema2 = [p.detach().clone() for p in net.parameters()]
        for p_ in net.parameters():
            p_.grad = None
        eager = make_eager(net, net, ema2, derain, acdt)
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
    tmax = torch.tensor([dt], dtype=torch.float64)   # a CPU tensor: over gloo, whatever state RCCL is in
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        recs = collect_prof(lib)
        roof = None
        fin = {(r["variant"], r["io"], r["segmented"]): r for r in recs if r["kernel"] == "oss_scan_bwd_finish"}
        recs = [r for r in recs if r["kernel"] != "oss_scan_bwd_finish"]
        if recs:
            dom = max(recs, key=lambda r: r["total_ms"])
            avg_ms = dom["total_ms"] / dom["launches"]
            achieved = dom["alg_bytes"] / (dom["total_ms"] * 1e-3) / 1e9
            kkey = f"{dom['kernel']} variant {dom['variant']} io {dom['io']}" + (" segmented" if dom["segmented"] else "")
            fdom = fin.get((dom["variant"], dom["io"], dom["segmented"])) if dom["kernel"] == "oss_scan_bwd_kernel" else None
            with_fin_ms = dom["total_ms"] + (fdom["total_ms"] if fdom else 0.0)
            # the dominant call of the workload: SS2D_1 of the widest full-resolution level, D = d_inner rows per direction
            call_shape = f"u:({B},48,{args.patch * args.patch})" if derain else f"u:({B},96,4096)"
            traffic, traffic_note, valu_busy = pmc_lookup(lib, kkey, call_shape)
            roof = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_note": traffic_note,
                    "kernel": kkey,
                    # the scan is a recurrence with ~29 vector-ALU instructions per (element, state): its roof is the VALU
                    # issue rate, not HBM -- share of the kernel with a SIMD's VALU issuing, from the SQ counters
                    # (profiles/r01_pmc_sq_scan.txt, tools/pmc_sq.sh); DESIGN.md section 5
                    "valu_busy": valu_busy,
                    #  not a counter read in THIS run: the SQ record of this build id under profiles/, or null
                    "valu_busy_source": (None if valu_busy is None else
                                         "looked up in the SQ counter record of this scan build id under profiles/ (the file `traffic_note` "
                                         "names), not measured in this run"),
                    "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                    "alg_bytes_per_launch": round(dom["alg_bytes"] / dom["launches"]),
                    # time segments per row of the LAST scan call (1 = one workgroup walks the whole row, as the reference does)
                    "segments": {"fwd": int(lib.oss_scan_last_segments(0)), "bwd": int(lib.oss_scan_last_segments(1))},
                    # SURVEY.md 8d asks for both: `achieved` prices the unfused-equivalent bytes (every direction its own
                    # u / dout rows); the omni kernel's own algorithmic bytes share them between directions k, k + 2
                    "own_alg_bytes_per_launch": round(dom["own_bytes"] / dom["launches"]),
                    "achieved_own_GBps": round(dom["own_bytes"] / (dom["total_ms"] * 1e-3) / 1e9, 1),
                    # the backward's finishing kernel (adds the row-tile dB/dC partials, casts; one per call) priced in
                    "finish_avg_ms": round(fdom["total_ms"] / fdom["launches"], 4) if fdom else None,
                    "frac_with_finish": round(dom["alg_bytes"] / (with_fin_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                    "all_scan_kernels": [
                        {"kernel": r["kernel"], "variant": r["variant"], "segmented": r["segmented"], "io": r["io"],
                         "launches": r["launches"], "avg_ms": round(r["total_ms"] / r["launches"], 4),
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
        if roof is not None and args.graph and world == 1 and not args.no_non_scan and not args.skip_roofline:
            log("non-scan roofline: kernel trace of the same step (rocprofv3 subprocess)")
            torch.cuda.empty_cache()
            targv = ["--config", args.config, "--dtype", args.dtype, "--batch-per-gpu", str(B), "--patch", str(args.patch), "--miopen-find", "0",
                     "--micro-streams", str(args.micro_streams)]
            trace, why = steady_state_kernel_trace(targv)
            roof["non_scan"] = non_scan_roofline(fam_counts, trace) if trace else {"error": why}
            if not derain:   # the SQ counters of the dominant call, read in THIS run (next to the looked-up record above)
                roof["valu_busy_measured"] = measure_valu_busy(f"{B},96,4096", "bf16" if args.dtype == "bf16" else "f32")

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

        second = None
        if world == 1 and args.config == "sr" and not args.global_batch and not args.no_secondary:
            log("secondary workloads (configs[3], configs[4]; subprocesses, 240 s limit each)")
            second = secondary_workloads()

        images = world * B * args.steps
        line = {
            "metric": (f"images/sec, deraining {args.patch}x{args.patch} training step (fwd+bwd+clip+AdamW), Mamber32 [3,5,7,9]+2" if derain else
                       "images/sec, x4 SR 64->256 training step (fwd+bwd+Adam+EMA), full VmambaIR UNet"),
            "value": round(images / dt, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": ((("BASELINE.json configs[3]: Deraining 128x128 patches" if args.patch == 128 else
                                      f"Deraining progressive schedule stage (Deraining_mamber33.yml:27-30): {args.patch}x{args.patch} patches") +
                                     f", Mamber32 dim48 [3,5,7,9]+2, {args.dtype} autocast (scan arithmetic f32), batch {B} per GPU") if derain else
                                    (f"BASELINE.json configs[2]: x4 SR 64x64 LQ, global batch {args.global_batch} over {world} GPU(s), "
                                     if args.global_batch else "BASELINE.json configs[1]: x4 SR 64x64 LQ, ") +
                                    f"MambaSISR6 dim48 [15,1,1,1]+15, {args.dtype} autocast (scan arithmetic f32), batch {B} per GPU"),
                       "global_batch": world * B, "per_gpu_batch": B, "lq": list(lq.shape[-2:]), "gt": list(gt.shape[-2:]),
                       "parallelism": f"dp{world}", "step_launch": "hipGraph replay" if args.graph else "eager",
                       "vendor_conv_solver_search": bool(args.miopen_find),
                       "optimizer": ("AdamW 3e-4 (0.9,0.999) decay 1e-4 + clip_grad_norm 0.01, no EMA" if derain else
                                     "Adam 2e-4 (0.9,0.99) + EMA 0.999"), "loss": "L1",
                       # multi-GPU exchange: ONE flat fp32 all-reduce between the forward+backward graph and the optimizer graph
                       "rccl_ranks": (dist.get_world_size() if world > 1 else 1),
                       **({"smoke_only": "--share-gpu: ranks share GPUs and talk over gloo -- flow check of the multi-rank path, not a measurement"}
                          if args.share_gpu else {}),
                       "allreduce_ms_per_step": None if allreduce_ms is None else round(allreduce_ms, 3),
                       # bucketed: every bucket but the last is exchanged on a side stream while the main stream flushes the next
                       # bucket's weight gradients (train_graph.py); False = ONE all-reduce between two graphs
                       "allreduce_overlapped": bool(args.graph and getattr(step, "nbuckets", 1) > 1 and world > 1),
                       "allreduce_buckets": getattr(step, "nbuckets", 1) if args.graph else None,
                       "rccl_preflight_ms_48MB": rccl["preflight_ms"],
                       **({"fallback": "; ".join(x for x in (rccl["fallback"], bucket_note) if x)}
                          if (world > 1 and (rccl["fallback"] or bucket_note)) else {}),
                       "allreduce_bytes": 4 * sum(p.numel() for p in net.parameters()) if world > 1 else 0,
                       "peak_memory_GB": peak_mem_gb,
                       "deferred_weight_gradients": None if not wgrad_stats else {
                           "grouped_launches": wgrad_stats.get("grouped_launches"), "budget_flushes": wgrad_stats.get("budget_flushes"),
                           "operands_held_GB_max": round(wgrad_stats.get("held_bytes_max", 0) / 1e9, 3),
                           "budget_GB": round(float(os.environ.get("VMAMBAIR_WGRAD_KEEP_MB", "8192")) / 1024, 2)}},
            "final_loss": round(loss_val, 5), "roofline": roof, "cpu_baseline": cpu, "secondary": second,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        host_barrier()
        if via_host and rccl["ok"] is False and "by request" not in (rccl["fallback"] or ""):
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)   # RCCL is in an unknown state (a preflight thread may still sit in it): leave without its teardown
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
