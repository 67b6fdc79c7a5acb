#!/usr/bin/env python3
"""SoftRas fwd+bwd benchmark on MI355X — the contract the driver runs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: jr_softras_forward (tile binning + raster) followed by jr_softras_backward
with a device-resident upstream gradient.  Workload = BASELINE.json's metric configuration
(configs[2]): UV-sphere with 39 000 faces seen from 8 turntable cameras, 1024x1024, batch 8 per
GPU, jrender Renderer defaults (sigma 1e-5, gamma 1e-4, euclidean / softmax / prod, K=16).
Every rank renders its own 8 views (batch sharding, no data-path collective) => weak scaling.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (HIP-event brackets on the
context stream around each kernel phase, collected over the timed region) against the 8 TB/s
HBM peak using ALGORITHMIC bytes; `cpu_baseline` times the CPU oracle
(reference kernels compiled for the host when oracle/_ref exists, else the C port) on a bounded
sample — the only place the oracle is touched, never the measured path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(B, NF, T, IS, K):
    """compulsory HBM traffic of one fwd+bwd, split per kernel phase."""
    P, F = B * IS * IS, B * NF
    fwd = 4 * (F * (9 + 27 + 3 * T) + P * (4 + 2 + K))           # faces+info+tex in, rgba+aggr+ids out
    bwd = 4 * (P * (4 + 4 + 2 + K) + F * (9 + 27 + 3 * T) + F * (9 + 3 * T))
    setup = 4 * (F * 9 + F * 27)                                  # faces in, faces_info out
    return dict(fwd_raster=fwd, bwd_raster=bwd, setup=setup,
                step=4 * (P * (16 + 2 * K) + F * (81 + 9 * T)))


def cpu_baseline(NF, K, budget_s=20.0):
    """Oracle fwd+bwd on a bounded sample of the same workload, all host cores; images/s at 1024^2."""
    from oracle import Oracle, have_ref
    from jrender_amd import synthetic as syn
    kind = "reference" if have_ref() else "port"
    orc = Oracle(kind, nthreads=0)
    cores = orc.num_procs()
    fv, tex = syn.sphere_views(NF, 1)

    def run(IS):
        t0 = time.perf_counter()
        a = orc.forward(fv, tex, image_size=IS, max_faces_per_pixel_for_grad=K)
        g = np.ones_like(a["soft_colors"])
        orc.backward(a, g, nthreads=cores)
        return time.perf_counter() - t0

    run(32)                                         # thread-pool / page-cache warm-up, not timed
    IS, t = 128, run(128)                           # cost is O(pixels x faces): grow while 4x still fits
    while IS < 1024 and 4.0 * t <= budget_s:
        IS *= 2
        t = run(IS)
    ips = 1.0 / (t * (1024.0 / IS) ** 2)
    return {"value": ips, "unit": "images/s", "cores": int(cores), "kind": kind,
            "sample": "1 view of the %d-face sphere at %dx%d fwd+bwd in %.1f s on %d threads, "
                      "scaled by pixel count to 1024x1024" % (NF, IS, IS, t, cores)}


def bench_n3mr(args):
    """Secondary workload (BASELINE.json configs[4]): NMR hard raster + approximate gradients,
    39k-face sphere with fill_back (78k faces), 1024x1024, B=1, fwd+bwd on one GPU.  Not the headline
    metric; prints its own JSON line when called with --workload n3mr."""
    from jrender_amd import _ffi
    import jrender_amd as jr
    from jrender_amd.renderer.dr.n3mr import RasterizeFunction
    ctx = _ffi.Context(int(os.environ.get("LOCAL_RANK", "0")))
    IS, B, ts = args.image_size, 1, 2
    v, f = jr.synthetic.sphere_mesh(args.faces)
    eye = np.asarray(jr.get_points_from_angles(2.732, 30., 0.), np.float32)
    ndc = jr.perspective(jr.look_at(v[None], eye), 30.)
    ff = np.concatenate([f, f[:, ::-1]])
    faces = ctx.array(np.ascontiguousarray(ndc[:, ff]))
    tex = ctx.array(np.random.default_rng(0).uniform(0, 1, (B, ff.shape[0], ts, ts, ts, 3)).astype(np.float32))
    rng = np.random.default_rng(1)
    g_rgb = ctx.array(rng.uniform(-1, 1, (B, IS, IS, 3)).astype(np.float32))
    g_a = ctx.array(rng.uniform(-1, 1, (B, IS, IS)).astype(np.float32))
    g_d = ctx.array(rng.uniform(-1, 1, (B, IS, IS)).astype(np.float32))
    fn = RasterizeFunction(IS, 0.1, 100, 1e-3, (0, 0, 0), True, True, True, ctx=ctx)

    def step():
        fn.execute(faces, tex)
        fn.grad(g_rgb, g_a, g_d)
    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    print(json.dumps({"metric": "NMR fwd+bwd ms @%dx%d, %d faces (fill_back x2)" % (IS, IS, args.faces),
                      "value": ms, "unit": "ms", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                      "ms_per_step": ms, "higher_is_better": False, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "n3mr rgb+alpha+depth, texture_size 2, batch 1"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="softras", choices=["softras", "n3mr"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scene", default="sphere", choices=["sphere", "soup"],
                    help="sphere: 39k-face UV sphere turntable (BASELINE configs[2], the headline); "
                         "soup: random triangles over the whole screen (north_star's 'random-triangle batches')")
    ap.add_argument("--faces", type=int, default=39000)
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8, help="views per GPU")
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.workload == "n3mr":
        return bench_n3mr(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        # torch is plumbing here: process-group rendezvous, barrier and the max-over-ranks reduction
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if ndev >= world:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:       # fewer GPUs than ranks (plumbing test on a 1-GPU box): ranks share GPUs, gloo carries the control traffic
            dist.init_process_group("gloo")
            local_rank = local_rank % max(ndev, 1)

    from jrender_amd import _ffi, synthetic as syn
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction

    ctx = _ffi.Context(local_rank)
    B, NF, IS, K, T = args.batch, args.faces, args.image_size, args.K, 1
    # every rank gets its own 8 cameras of the turntable (different azimuth offset per rank)
    if args.scene == "sphere":
        fv_h, tex_h = syn.sphere_views(NF, B, azimuth0=360.0 * rank / max(world, 1) / B)
    else:
        fv_h, tex_h = syn.triangle_soup(NF, B, seed=100 + rank)
    fv, tex = ctx.array(fv_h), ctx.array(tex_h)
    grad = ctx.array(np.random.default_rng(7 + rank).uniform(-1, 1, (B, 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, max_faces_per_pixel_for_grad=K, ctx=ctx)

    def step():
        fn.execute(fv, tex)
        fn.grad(grad)

    def barrier():
        if dist is not None:
            dist.barrier()
        ctx.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_collect()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    phases = ctx.profile_collect()
    ctx.profile_enable(False)
    if dist is not None:
        import torch
        tt = torch.tensor([elapsed], dtype=torch.float64,
                          device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        ab = algorithmic_bytes(B, NF, T, IS, K)
        per_launch = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()}   # ms per bracket
        dom = max(("fwd_raster", "bwd_raster"), key=lambda k: per_launch[k])
        achieved = ab[dom] / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        profiled = (args.scene, NF, IS, B, K) == ("sphere", 39000, 1024, 8, 16)    # the configuration the PMC passes ran
        if profiled and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "SoftRas fwd+bwd images/s @1024x1024, 39k faces",
            "value": world * B / (elapsed / args.steps),
            "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_image_fwd_bwd": ms_step / B,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s %d faces x %d views/GPU, %dx%d, SoftRas fwd+bwd, K=%d, "
                                   "sigma=1e-5 gamma=1e-4 euclidean/softmax/prod"
                                   % ("UV-sphere" if args.scene == "sphere" else "random-triangle soup", NF, B, IS, IS, K),
                       "faces": NF, "image_size": IS, "batch_per_gpu": B, "global_batch": B * world,
                       "parallelism": "batch-sharded x%d, no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": "k_softras_%s" % ("forward" if dom == "fwd_raster" else "backward"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": ab[dom],
                         "avg_launch_ms": per_launch[dom],
                         "step_frac": ab["step"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "phase_ms_per_step": {k: v[0] / args.steps for k, v in phases.items()},
            "tile_stats": ctx.last_stats(),
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(NF, K)
            except Exception as e:                      # the baseline must never break the bench line
                out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
