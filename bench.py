#!/usr/bin/env python3
"""SoftRas fwd+bwd benchmark on MI355X — the contract the driver runs.

    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks when N > 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
                                                           # (only RANK / LOCAL_RANK / WORLD_SIZE /
                                                           #  MASTER_PORT are READ; torch is never imported)

One "step" = one pass of the hot path over one batch of synthetic input that is already
resident in HBM: jr_softras_forward (tile binning + raster) followed by jr_softras_backward
with a device-resident upstream gradient.  Workload = BASELINE.json's metric configuration
(configs[2]): UV-sphere with 39 000 faces seen from 8 turntable cameras, 1024x1024, batch 8 per
GPU, jrender Renderer defaults (sigma 1e-5, gamma 1e-4, euclidean / softmax / prod, K=16).
Every rank renders its own 8 views (batch sharding, no collective inside the op) => weak scaling.
With N > 1 the step ends with the exchange a data-parallel mesh optimisation needs
(--exchange allreduce_vertex_grads, the default for N > 1: face->vertex scatter of the rank's
gradients + ONE ncclAllReduce of the shared-vertex gradient, demo2-deform.py:45); --exchange
allgather_images puts the RCCL all-gather of the rendered image shards into the step instead,
--exchange none times independent replicas.

Rank 0 prints ONE JSON line.  `value` comes from the wall clock around exactly K steps between
barriers (max over ranks); per-step HIP events give median / p10 / p90.  `roofline` prices the
dominant kernel with ALGORITHMIC bytes over its average launch time, measured with
HIP-event brackets on the context stream in a SECOND pass of K steps (so that the brackets are not
inside the headline timing) against the 8 TB/s HBM peak; the line also says what actually binds
(VALU issue, from the committed PMC pass in profiles/).  `cpu_baseline` times the CPU oracle
(reference kernels compiled for the host when oracle/_ref exists, else the C port) on a bounded
sample — the only place the oracle is touched, never the measured path.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
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
    read = 4 * (P * (4 + 4 + 2 + K) + 2 * F * (9 + 27 + 3 * T))   # what fwd+bwd must READ (north_star's wording)
    return dict(fwd_raster=fwd, bwd_raster=bwd, setup=setup, read=read,
                step=4 * (P * (16 + 2 * K) + F * (81 + 9 * T)))


def n3mr_algorithmic_bytes(B, NF, TS, IS):
    """per pixel the forward writes face_index 4 + weight 12 + depth 4 + face_inv 36 + rgb 12 +
    alpha 4 + sampling index/weight 64 = 136 B; the backward reads those (minus weight for rgb/alpha-only
    pixels it still needs: all of them with return_depth) plus three upstream gradients 20 B = 156 B; per
    face: 36 B in, 36 B faces_inv out, textures ts^3*12 B in (forward), 36 B + ts^3*12 B gradients out."""
    P, F, t = B * IS * IS, B * NF, TS ** 3 * 12
    fwd = P * 136 + F * (36 + 36 + t)
    bwd = P * 156 + F * (36 + 36 + t)
    return dict(fwd=fwd, bwd=bwd, step=fwd + bwd)


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


def n3mr_cpu_baseline(faces_h, tex_h, IS_full, budget_s=25.0):
    """The reference's NMR kernels compiled for the host (serial, like their one-thread-per-face CUDA form
    run on one core), fwd+bwd at a reduced image size; ms scaled by pixel count to the full size."""
    from oracle import N3mrOracle
    orc = N3mrOracle()
    IS, t = 64, None
    rng = np.random.default_rng(1)
    while True:
        t0 = time.perf_counter()
        s = orc.forward(faces_h, tex_h, image_size=IS, near=0.1, far=100, eps=1e-3)
        orc.backward(s, rng.uniform(-1, 1, s["rgb_map"].shape).astype(np.float32),
                     rng.uniform(-1, 1, s["alpha_map"].shape).astype(np.float32),
                     rng.uniform(-1, 1, s["depth_map"].shape).astype(np.float32))
        t = time.perf_counter() - t0
        if IS * 2 > min(IS_full, 512) or 4.0 * t > budget_s:
            break
        IS *= 2
    ms = t * 1e3 * (float(IS_full) / IS) ** 2
    return {"value": ms, "unit": "ms", "cores": 1, "kind": orc.kind,
            "sample": "fwd+bwd at %dx%d in %.1f s on 1 thread, scaled by pixel count to %dx%d"
                      % (IS, IS, t, IS_full, IS_full)}


def percentiles(ms):
    a = np.sort(np.asarray(ms, np.float64))
    return {"median": float(np.median(a)), "p10": float(np.percentile(a, 10)), "p90": float(np.percentile(a, 90)),
            "min": float(a[0]), "max": float(a[-1])}


def timed_steps(ctx, comm, step, steps, warmup):
    """W untimed steps, then exactly K steps between barriers -> (wall seconds max over ranks, per-step ms)."""
    for _ in range(warmup):
        step()
    events = [ctx.event() for _ in range(steps + 1)]
    ctx.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    ctx.record(events[0])
    for i in range(steps):
        step()
        ctx.record(events[i + 1])
    ctx.synchronize()
    comm.barrier()
    elapsed = comm.all_reduce_max(time.perf_counter() - t0)
    per_step = [ctx.elapsed_ms(events[i], events[i + 1]) for i in range(steps)]
    return elapsed, per_step


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def bench_n3mr(args, ctx, comm, rank, world):
    """Secondary workload (BASELINE.json configs[4]): NMR hard raster + approximate gradients,
    39k-face sphere with fill_back (78k faces), 1024x1024, B=1, fwd+bwd on one GPU."""
    import jrender_amd as jr
    from jrender_amd.renderer.dr.n3mr import RasterizeFunction
    IS, B, ts = args.image_size, 1, 2
    v, f = jr.synthetic.sphere_mesh(args.faces)
    eye = np.asarray(jr.get_points_from_angles(2.732, 30., 0.), np.float32)
    ndc = jr.perspective(jr.look_at(v[None], eye), 30.)
    ff = np.concatenate([f, f[:, ::-1]])
    faces_h = np.ascontiguousarray(ndc[:, ff])
    tex_h = np.random.default_rng(0).uniform(0, 1, (B, ff.shape[0], ts, ts, ts, 3)).astype(np.float32)
    faces, tex = ctx.array(faces_h), ctx.array(tex_h)
    rng = np.random.default_rng(1)
    g_rgb = ctx.array(rng.uniform(-1, 1, (B, IS, IS, 3)).astype(np.float32))
    g_a = ctx.array(rng.uniform(-1, 1, (B, IS, IS)).astype(np.float32))
    g_d = ctx.array(rng.uniform(-1, 1, (B, IS, IS)).astype(np.float32))
    fn = RasterizeFunction(IS, 0.1, 100, 1e-3, (0, 0, 0), True, True, True, ctx=ctx)

    def step():
        fn.execute(faces, tex)
        fn.grad(g_rgb, g_a, g_d)
    elapsed, per_step = timed_steps(ctx, comm, step, args.steps, args.warmup)
    # forward / backward halves with events (second pass)
    e = [ctx.event() for _ in range(3)]
    fwd_ms, bwd_ms = [], []
    for _ in range(min(args.steps, 20)):
        ctx.record(e[0]); fn.execute(faces, tex); ctx.record(e[1]); fn.grad(g_rgb, g_a, g_d); ctx.record(e[2])
        fwd_ms.append(ctx.elapsed_ms(e[0], e[1])); bwd_ms.append(ctx.elapsed_ms(e[1], e[2]))
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    NF2 = ff.shape[0]
    ab = n3mr_algorithmic_bytes(B, NF2, ts, IS)
    dom, dom_ms = ("bwd", float(np.mean(bwd_ms))) if np.mean(bwd_ms) >= np.mean(fwd_ms) else ("fwd", float(np.mean(fwd_ms)))
    achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
    traffic = (load_json("traffic_n3mr_latest.json") or {}).get(dom) if (NF2, IS) == (78000, 1024) else None
    out = {"metric": "NMR fwd+bwd ms @%dx%d, %d faces (fill_back x2)" % (IS, IS, args.faces),
           "value": ms, "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms, "step_ms": percentiles(per_step), "higher_is_better": False, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "n3mr rgb+alpha+depth, %d faces, %dx%d, texture_size %d, batch 1" % (NF2, IS, IS, ts)},
           "roofline": {"bound": "hbm", "kernel": "n3mr %s kernels (events around jr_n3mr_%s)" % (dom, "backward" if dom == "bwd" else "forward"),
                        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": traffic, "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": dom_ms,
                        "step_frac": ab["step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "phase_ms_per_step": {"forward": float(np.mean(fwd_ms)), "backward": float(np.mean(bwd_ms))}}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = n3mr_cpu_baseline(faces_h, tex_h, IS)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(out), flush=True)


def bench_softras(args, ctx, comm, rank, world):
    from jrender_amd import synthetic as syn
    from jrender_amd.parallel import shared_vertex_gradient
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    B, NF, IS, K, T = args.batch, args.faces, args.image_size, args.K, 1
    # every rank gets its own 8 cameras of the turntable (different azimuth offset per rank)
    if args.scene == "sphere":
        fv_h, tex_h = syn.sphere_views(NF, B, azimuth0=360.0 * rank / max(world, 1) / B)
        verts, mesh_faces = syn.sphere_mesh(NF)
        NV = verts.shape[0]
    else:
        fv_h, tex_h = syn.triangle_soup(NF, B, seed=100 + rank)
        mesh_faces, NV = np.arange(3 * NF, dtype=np.int32).reshape(NF, 3), 3 * NF
    fv, tex = ctx.array(fv_h), ctx.array(tex_h)
    faces_d = ctx.array(np.ascontiguousarray(mesh_faces, np.int32))
    grad = ctx.array(np.random.default_rng(7 + rank).uniform(-1, 1, (B, 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, max_faces_per_pixel_for_grad=K, ctx=ctx)
    exchange = args.exchange or ("allreduce_vertex_grads" if world > 1 else "none")

    def render():
        img = fn.execute(fv, tex)
        gf, _ = fn.grad(grad)
        return img, gf

    def do_exchange(img, gf):
        if exchange == "allreduce_vertex_grads":
            shared_vertex_gradient(gf, faces_d, NV, comm)
        elif exchange == "allgather_images":
            comm.all_gather(img, B * world)

    def step():
        do_exchange(*render())

    elapsed, per_step = timed_steps(ctx, comm, step, args.steps, args.warmup)

    # second pass: per-phase brackets (and the exchange on its own) — not part of the headline timing
    ctx.profile_enable(True)
    ctx.profile_collect()
    e0, e1 = ctx.event(), ctx.event()
    ex_ms = []
    t1 = time.perf_counter()
    for _ in range(args.steps):
        img, gf = render()
        ctx.record(e0)
        do_exchange(img, gf)
        ctx.record(e1)
        ex_ms.append(ctx.elapsed_ms(e0, e1))
    ctx.synchronize()
    bracketed_ms = (time.perf_counter() - t1) / args.steps * 1e3
    phases = ctx.profile_collect()
    ctx.profile_enable(False)
    comm.barrier()
    if rank != 0:
        return

    ms_step = elapsed / args.steps * 1e3
    ab = algorithmic_bytes(B, NF, T, IS, K)
    per_launch = {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()}   # ms per bracket
    dom = max(("fwd_raster", "bwd_raster"), key=lambda k: per_launch[k])
    achieved = ab[dom] / (per_launch[dom] * 1e-3) / 1e9 if per_launch[dom] > 0 else 0.0
    # HBM bytes and VALU counters of the dominant kernel come from the committed PMC passes of THIS
    # configuration (profiles/traffic_latest.json, profiles/valu_latest.json; tools/collect_profiles.sh)
    profiled = (args.scene, NF, IS, B, K) == ("sphere", 39000, 1024, 8, 16)
    traffic = (load_json("traffic_latest.json") or {}).get(dom) if profiled else None
    valu = (load_json("valu_latest.json") or {}).get(dom) if profiled else None
    out = {
        "metric": "SoftRas fwd+bwd images/s @1024x1024, 39k faces",
        "value": world * B / (elapsed / args.steps),
        "unit": "images/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step,
        "step_ms": percentiles(per_step),
        "ms_per_image_fwd_bwd": ms_step / B,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %d faces x %d views/GPU, %dx%d, SoftRas fwd+bwd, K=%d, "
                               "sigma=1e-5 gamma=1e-4 euclidean/softmax/prod"
                               % ("UV-sphere" if args.scene == "sphere" else "random-triangle soup", NF, B, IS, IS, K),
                   "faces": NF, "image_size": IS, "batch_per_gpu": B, "global_batch": B * world,
                   "parallelism": "batch-sharded x%d, one process per GPU, exchange=%s over %s"
                                  % (world, exchange, comm.backend)},
        "roofline": {"bound": "hbm", "binding_resource": "valu_issue",
                     "kernel": "k_softras_%s" % ("forward" if dom == "fwd_raster" else "backward"),
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ab[dom],
                     "avg_launch_ms": per_launch[dom],
                     "timing": "HIP-event brackets on the context stream, second pass of %d steps (%.4f ms/step with brackets)"
                               % (args.steps, bracketed_ms),
                     "step_frac": ab["step"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "hbm_read_frac": ab["read"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "valu": valu},
        "phase_ms_per_step": {k: v[0] / args.steps for k, v in phases.items()},
        "exchange": {"kind": exchange, "backend": comm.backend, "ms_per_step": percentiles(ex_ms)["median"] if ex_ms else 0.0},
        "tile_stats": ctx.last_stats(),
    }
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(NF, K)
        except Exception as e:                      # the baseline must never break the bench line
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    print(json.dumps(out), flush=True)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside any launcher: start N ranks of this script (one process per
    GPU), rendezvous through a private directory, relay rank 0's JSON line."""
    rdzv = tempfile.mkdtemp(prefix="jrender_bench_")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   JRENDER_RDZV=os.path.join(rdzv, "rdzv"), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    try:
        for f in os.listdir(rdzv):
            os.unlink(os.path.join(rdzv, f))
        os.rmdir(rdzv)
    except OSError:
        pass
    if any(rcs):
        sys.exit("bench.py: rank exit codes %s" % rcs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="softras", choices=["softras", "n3mr"])
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scene", default="sphere", choices=["sphere", "soup"],
                    help="sphere: 39k-face UV sphere turntable (BASELINE configs[2], the headline); "
                         "soup: random triangles over the whole screen (north_star's 'random-triangle batches')")
    ap.add_argument("--faces", type=int, default=39000)
    ap.add_argument("--image-size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8, help="views per GPU")
    ap.add_argument("--K", type=int, default=16)
    ap.add_argument("--exchange", default=None, choices=["none", "allreduce_vertex_grads", "allgather_images"],
                    help="exchange step at the end of every step (default: allreduce_vertex_grads when N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from jrender_amd import _ffi, comm as jcomm
    ndev = _ffi.device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible (there is no CPU fallback)")
    ctx = _ffi.Context(local_rank % ndev)       # fewer GPUs than ranks: plumbing run, ranks share GPUs (host communicator)
    comm = jcomm.init_from_env(ctx)
    try:
        if args.workload == "n3mr":
            bench_n3mr(args, ctx, comm, rank, world)
        else:
            bench_softras(args, ctx, comm, rank, world)
    finally:
        comm.close()


if __name__ == "__main__":
    main()
