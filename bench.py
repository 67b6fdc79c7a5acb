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
With N > 1 the step ends with the exchange north_star names ("RCCL all-gather of images/gradients"):
--exchange both, the default for N > 1 = face->vertex scatter of the rank's gradients + ONE
ncclAllReduce of the shared-vertex gradient (what a data-parallel mesh optimisation needs,
demo2-deform.py:45) AND the ncclAllGather of the rendered image shards; --exchange
allreduce_vertex_grads / allgather_images time one of the two, --exchange none independent replicas.

Rank 0 prints ONE JSON line.  `value` comes from the wall clock around exactly K steps between
barriers (max over ranks); per-step HIP events give median / p10 / p90.  `roofline` prices the
dominant kernel with ALGORITHMIC bytes over its average launch time, measured with
HIP-event brackets on the context stream in a SECOND pass of K steps (so that the brackets are not
inside the headline timing) against the 8 TB/s HBM peak; the line also says what actually binds
(VALU issue, from the committed PMC pass in profiles/).  `cpu_baseline` times the CPU oracle
(reference kernels compiled for the host when oracle/_ref exists, else the C port) on a bounded
sample — the only place the oracle is touched, never the measured path.  The oracle view rendered for that baseline is
ALSO the parity check of the line: the GPU renders the same view at the same size with the same upstream gradient and
`parity` reports the index-buffer match fraction (must be 1.0) and the RGBA / grad_faces / vertex-gradient errors,
SURVEY.md §8(d)'s element-wise formula next to the max-normalised one.  `latency_ms_b1` is fwd+bwd of ONE 39k-face
image (BASELINE's "ms @1024^2, 39k faces"); `secondary` carries the demo2 loop (configs[3]), the NMR workload (configs[4]) and the random-triangle
scene, all measured AFTER the timed region.
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


def csrc_hash(root=None):
    """Content hash of the kernel sources the running library was built from (jrender_amd/csrc/*.hip|*.h|*.cpp).  The
    PMC figures attached to `roofline` (profiles/traffic_latest.json, valu_latest.json) carry the hash of the sources they
    were collected on (tools/pmc_to_json.py): a mismatch prints `profile_stale: true` instead of passing old counters off
    as the shipped kernels'.  (A hash of file contents, not a git object: the GPU box has no .git.)"""
    import hashlib
    d = os.path.join(root or ROOT, "jrender_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".cpp")):
            h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read() + b"\0")
    return h.hexdigest()[:16]


def profile_staleness(traffic_json, valu_json, isa_json, here):
    """(profile_stale, profiled_kernels_device_code_unchanged) of the bench line.  The counters' stamp covers ALL kernel sources:
    any edit makes them stale.  When only another translation unit changed since they were taken, tools/isa_same.py (CPU,
    reproducible) has compiled every unit of both trees to gfx950 assembly and compared; its record counts only if it is about
    exactly these two source states.  The second value is None when there is nothing to say (fresh counters, or no such record)."""
    stale = not (traffic_json.get("csrc_hash") == here and valu_json.get("csrc_hash") == here)
    same = None
    if (stale and isa_json and isa_json.get("new_csrc_hash") == here
            and isa_json.get("old_csrc_hash") == traffic_json.get("csrc_hash") == valu_json.get("csrc_hash")):
        same = all((isa_json.get("units") or {}).get(u, {}).get("same") is True for u in ("softras_forward.hip", "softras_backward.hip"))
    return stale, same


def algorithmic_bytes(B, NF, T, IS, K):
    """compulsory HBM traffic of one fwd+bwd, split per kernel phase."""
    P, F = B * IS * IS, B * NF
    fwd = 4 * (F * (9 + 27 + 3 * T) + P * (4 + 2 + K))           # faces+info+tex in, rgba+aggr+ids out
    bwd = 4 * (P * (4 + 4 + 2 + K) + F * (9 + 27 + 3 * T) + F * (9 + 3 * T))
    setup = 4 * (F * 9 + F * 27)                                  # faces in, faces_info out
    read = 4 * (P * (10 + K) + F * (45 + 6 * T))                  # what fwd+bwd must READ (north_star's wording; SURVEY 8(d): faces_info is produced on chip for the forward)
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
    """Oracle fwd+bwd on a bounded sample of the same workload, all host cores; images/s at 1024^2.
    -> (baseline dict, sample) with sample = the largest oracle view rendered (inputs, outputs, upstream gradient,
    gradients) for the parity check of the line."""
    from oracle import Oracle, have_ref
    from jrender_amd import synthetic as syn
    kind = "reference" if have_ref() else "port"
    orc = Oracle(kind, nthreads=0)
    cores = orc.num_procs()
    fv, tex = syn.sphere_views(NF, 1)
    last = {}

    def run(IS):
        g = np.random.default_rng(11).uniform(-1, 1, (1, 4, IS, IS)).astype(np.float32)
        t0 = time.perf_counter()
        a = orc.forward(fv, tex, image_size=IS, max_faces_per_pixel_for_grad=K)
        gf, gt = orc.backward(a, g, nthreads=cores)
        t = time.perf_counter() - t0
        last.update(IS=IS, fv=fv, tex=tex, saved=a, g=g, grad_faces=gf, grad_textures=gt)
        return t

    run(32)                                         # thread-pool / page-cache warm-up, not timed
    IS, t = 128, run(128)                           # cost is O(pixels x faces): grow while 4x still fits
    while IS < 1024 and 4.0 * t <= budget_s:
        IS *= 2
        t = run(IS)
    ips = 1.0 / (t * (1024.0 / IS) ** 2)
    if kind == "reference":
        try:
            gradient_references(orc, last, cores)
        except Exception as e:                      # never break the baseline
            last["gradient_references_error"] = repr(e)
    return {"value": ips, "unit": "images/s", "cores": int(cores), "kind": kind,
            "sample": "1 view of the %d-face sphere at %dx%d fwd+bwd in %.1f s on %d threads, "
                      "scaled by pixel count to 1024x1024" % (NF, IS, IS, t, cores)}, last


def err_metrics(ours, ref):
    """max of |a-b| / (|b| + 1e-6 max|b|) — and, because float atomics make the gradient sums order
    dependent (in the reference too), the same with a 1e-3 floor and the max-normalised error the tests gate on."""
    a, b = np.asarray(ours, np.float64).ravel(), np.asarray(ref, np.float64).ravel()
    fin = np.isfinite(b)
    if not fin.any():
        return {"max_norm": None, "rel_floor_1e-6": None, "rel_floor_1e-3": None}
    m = float(np.abs(b[fin]).max()) or 1.0
    d = np.abs(a - b)[fin]
    bb = np.abs(b[fin])
    return {"max_norm": float(d.max() / m), "rel_floor_1e-6": float((d / (bb + 1e-6 * m)).max()),
            "rel_floor_1e-3": float((d / (bb + 1e-3 * m)).max()),
            "nonfinite_mismatch": int((np.isfinite(np.asarray(ours, np.float64).ravel()) != fin).sum())}


def gradient_references(orc, sample, cores):
    """What the reference's OWN gradient is good to, measured on the sample view (oracle/_ref only): the float backward on
    one thread against the same on all cores (order noise of its float atomics, cuda/soft_rasterize.py:1349-1358), both
    against the exact sum of the same float terms (float atomics shadowed in double), and against the reference's backward
    kernel instantiated for double (backward_soft_rasterize_cuda_kernel<scalar_t>, cuda/soft_rasterize.py:1177)."""
    a, g = sample["saved"], sample["g"]
    gf1, gt1 = orc.backward(a, g, nthreads=1)
    gfS, gtS = orc.backward_exactsum(a, g, nthreads=cores)
    gf64, gt64 = orc.backward_f64(a, g, nthreads=cores)
    sample.update(gf_serial=gf1, gt_serial=gt1, gf_sum=gfS, gt_sum=gtS, gf_f64=gf64, gt_f64=gt64)


def reference_platform_envelope(NF, K, IS=256):
    """How far the reference moves from ITSELF under a legitimate recompile: its own kernel sources compiled with
    floating-point contraction on (-ffp-contract=fast -mfma, oracle/_ref/libsoftras_ref_fma.so - what nvcc does by
    default, --fmad=true) against the -ffp-contract=off build that parity is pinned to, same view, reduced size.  The
    HIP path reproduces the latter bit for bit in the index buffer; the CUDA build of the reference would not."""
    from oracle import Oracle
    from jrender_amd import synthetic as syn
    a, b = Oracle("reference", nthreads=0), Oracle("reference_fma", nthreads=0)
    fv, tex = syn.sphere_views(NF, 1)
    g = np.random.default_rng(11).uniform(-1, 1, (1, 4, IS, IS)).astype(np.float32)
    ra = a.forward(fv, tex, image_size=IS, max_faces_per_pixel_for_grad=K)
    rb = b.forward(fv, tex, image_size=IS, max_faces_per_pixel_for_grad=K)
    ga, gb = a.backward_exactsum(ra, g)[0], b.backward_exactsum(rb, g)[0]
    bad = (ra["faces_id_buffer"] != rb["faces_id_buffer"]).any(1)
    return {"what": "reference kernels built with -ffp-contract=fast -mfma (nvcc's default contraction) vs -ffp-contract=off, "
                    "one %d-face view at %dx%d" % (NF, IS, IS),
            "ids_mismatching_pixels": int(bad.sum()), "ids_mismatch_frac": float(bad.mean()),
            "faces_info_words_differing": int((ra["faces_info"].view(np.uint32) != rb["faces_info"].view(np.uint32)).sum()),
            "rgba": err_metrics(rb["soft_colors"], ra["soft_colors"]), "grad_faces": err_metrics(gb, ga)}


def parity_vs_sample(ctx, sample, K, mesh_faces, NV, precise_colour=False):
    """The GPU path on the oracle's view: same inputs, same size, same upstream gradient.  precise_colour: through the
    conformant kernel set (the forward's colour path in the reference's own arithmetic, jr_softras_set_precise_colour)."""
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    from jrender_amd.structures.mesh import face_vertices_backward
    IS, a = sample["IS"], sample["saved"]
    fn = SoftRasterizeFunction(image_size=IS, max_faces_per_pixel_for_grad=K, ctx=ctx, precise_colour=precise_colour)
    img = fn.execute(ctx.array(sample["fv"]), ctx.array(sample["tex"]))
    gf, gt = fn.grad(ctx.array(sample["g"]))
    ids = fn.save_vars[5].numpy()
    fb = np.asarray(mesh_faces, np.int64).reshape(1, -1, 3)
    gv = lambda g: face_vertices_backward(np.asarray(g, np.float32).reshape(1, -1, 3, 3), fb, NV)
    covered = a["faces_id_buffer"][:, 0] >= 0
    noise = None
    if "gf_sum" in sample:
        # gradients against references that carry NO order noise: `exact_sum` = the reference's float per-pair terms summed in
        # double; `f64` = the reference's kernel in double (differs from its float self wherever a float decision - inside /
        # outside, clipping - falls the other way: conditioning of the scene, not noise)
        gfh, gth = gf.numpy(), gt.numpy()
        well = np.abs(sample["gf_sum"] - sample["gf_f64"]).reshape(-1, 9).max(1) <= 1e-4 * np.abs(sample["gf_f64"]).max()
        sel = lambda x: np.asarray(x).reshape(-1, 9)[well]
        noise = {
            "ref_self_noise": {"grad_faces_serial_vs_all_cores": err_metrics(sample["gf_serial"], sample["grad_faces"]),
                               "grad_faces_all_cores_vs_exact_sum": err_metrics(sample["grad_faces"], sample["gf_sum"]),
                               "grad_textures_all_cores_vs_exact_sum": err_metrics(sample["grad_textures"], sample["gt_sum"]),
                               "vertex_grad_all_cores_vs_exact_sum": err_metrics(gv(sample["grad_faces"]), gv(sample["gf_sum"]))},
            "vs_exact_sum": {"grad_faces": err_metrics(gfh, sample["gf_sum"]), "grad_textures": err_metrics(gth, sample["gt_sum"]),
                             "vertex_grad": err_metrics(gv(gfh), gv(sample["gf_sum"]))},
            "vs_f64": {"reference_f32_grad_faces": err_metrics(sample["gf_sum"], sample["gf_f64"]),
                       "hip_grad_faces": err_metrics(gfh, sample["gf_f64"]),
                       "faces_where_f32_and_f64_agree_to_1e-4_of_max": float(well.mean()),
                       "reference_f32_grad_faces_on_those": err_metrics(sel(sample["gf_sum"]), sel(sample["gf_f64"])),
                       "hip_grad_faces_on_those": err_metrics(sel(gfh), sel(sample["gf_f64"]))},
            "what": "exact_sum = the reference's float per-pair gradient terms summed in double (its result without the order noise "
                    "of the float atomics); f64 = the reference's backward kernel instantiated for double on the same saved tensors"}
    return {"gradient_references": noise, "view": "oracle view 0 at %dx%d (%s), the whole image" % (IS, IS, "same size as the timed batch" if IS == 1024 else "reduced: CPU budget"),
            "ids_match_frac": float((ids == a["faces_id_buffer"]).all(axis=1).mean()),
            "ids_mismatching_pixels": int((~(ids == a["faces_id_buffer"]).all(axis=1)).sum()),
            "covered_pixel_frac": float(covered.mean()),
            "faces_info_bit_exact": bool((fn.save_vars[3].numpy().view(np.uint32) == a["faces_info"].view(np.uint32)).all()),
            "rgba_err": err_metrics(img.numpy(), a["soft_colors"]),
            "aggrs_err": err_metrics(fn.save_vars[4].numpy(), a["aggrs_info"]),
            "grad_faces_err": err_metrics(gf.numpy(), sample["grad_faces"]),
            "grad_textures_err": err_metrics(gt.numpy(), sample["grad_textures"]),
            "vertex_grad_err": err_metrics(gv(gf.numpy()), gv(sample["grad_faces"])),
            "formula": "max |a-b| / (|b| + floor * max|b|): rel_floor_1e-6 is SURVEY 8(d)'s metric, max_norm = max|a-b| / max|b| "
                       "is what tests/ gate at 1e-4; gradients are float-atomic sums in the reference too"}


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


def measure_n3mr(ctx, comm, nfaces, IS, steps, warmup):
    """NMR hard raster + approximate gradients (BASELINE.json configs[4]): nfaces-sphere with fill_back (x2 faces),
    IS x IS, B=1, rgb+alpha+depth, fwd+bwd on one GPU -> dict of timings and the host copies the CPU baseline needs."""
    import jrender_amd as jr
    from jrender_amd.renderer.dr.n3mr import RasterizeFunction
    B, ts = 1, 2
    v, f = jr.synthetic.sphere_mesh(nfaces)
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
    elapsed, per_step = timed_steps(ctx, comm, step, steps, warmup)
    # forward / backward halves with events (second pass)
    e = [ctx.event() for _ in range(3)]
    fwd_ms, bwd_ms = [], []
    for _ in range(min(steps, 20)):
        ctx.record(e[0]); fn.execute(faces, tex); ctx.record(e[1]); fn.grad(g_rgb, g_a, g_d); ctx.record(e[2])
        fwd_ms.append(ctx.elapsed_ms(e[0], e[1])); bwd_ms.append(ctx.elapsed_ms(e[1], e[2]))
    ms = elapsed / steps * 1e3
    NF2 = ff.shape[0]
    ab = n3mr_algorithmic_bytes(B, NF2, ts, IS)
    dom, dom_ms = ("bwd", float(np.mean(bwd_ms))) if np.mean(bwd_ms) >= np.mean(fwd_ms) else ("fwd", float(np.mean(fwd_ms)))
    achieved = ab[dom] / (dom_ms * 1e-3) / 1e9
    traffic = (load_json("traffic_n3mr_latest.json") or {}).get(dom) if (NF2, IS) == (78000, 1024) else None
    return {"ms": ms, "per_step": per_step, "NF2": NF2, "ts": ts, "faces_h": faces_h, "tex_h": tex_h,
            "roofline": {"bound": "hbm", "kernel": "n3mr %s kernels (events around jr_n3mr_%s)" % (dom, "backward" if dom == "bwd" else "forward"),
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_launch": ab[dom], "avg_launch_ms": dom_ms,
                         "step_frac": ab["step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "phase_ms_per_step": {"forward": float(np.mean(fwd_ms)), "backward": float(np.mean(bwd_ms))}}


def bench_n3mr(args, ctx, comm, rank, world):
    """`--workload n3mr`: the NMR line on its own (the headline line carries it under `secondary`)."""
    IS = args.image_size
    m = measure_n3mr(ctx, comm, args.faces, IS, args.steps, args.warmup)
    if rank != 0:
        return
    out = {"metric": "NMR fwd+bwd ms @%dx%d, %d faces (fill_back x2)" % (IS, IS, args.faces),
           "value": m["ms"], "unit": "ms", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": m["ms"], "step_ms": percentiles(m["per_step"]), "higher_is_better": False, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "n3mr rgb+alpha+depth, %d faces, %dx%d, texture_size %d, batch 1" % (m["NF2"], IS, IS, m["ts"])},
           "roofline": m["roofline"], "phase_ms_per_step": m["phase_ms_per_step"]}
    if world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = n3mr_cpu_baseline(m["faces_h"], m["tex_h"], IS)
        except Exception as e:
            out["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    print(json.dumps(out), flush=True)


def fwd_bwd_ms(ctx, comm, fv_h, tex_h, IS, K, steps=30, warmup=5, seed=3, **op_kwargs):
    """fwd+bwd of one device-resident batch: median ms per step and the per-phase brackets."""
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    fv, tex = ctx.array(fv_h), ctx.array(tex_h)
    grad = ctx.array(np.random.default_rng(seed).uniform(-1, 1, (fv_h.shape[0], 4, IS, IS)).astype(np.float32))
    fn = SoftRasterizeFunction(image_size=IS, max_faces_per_pixel_for_grad=K, ctx=ctx, **op_kwargs)

    def step():
        fn.execute(fv, tex)
        fn.grad(grad)
    _elapsed, per_step = timed_steps(ctx, comm, step, steps, warmup)
    ctx.profile_enable(True)
    ctx.profile_collect()
    for _ in range(10):
        step()
    ctx.synchronize()
    phases = ctx.profile_collect()
    ctx.profile_enable(False)
    return percentiles(per_step), {k: (v[0] / v[1] if v[1] else 0.0) for k, v in phases.items()}


def demo2_loop_ms():
    """examples/demo2_deform.py (the reference's demo2-deform.py on this library) timed by its own clock around the
    optimisation loop: camera -> gather -> SoftRas -> IoU -> SoftRas backward -> scatter + camera VJP -> regularisers ->
    Adam per iteration, with the chain around the rasteriser on the device and, for comparison, through the NumPy mirrors."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("demo2_deform", os.path.join(ROOT, "examples", "demo2_deform.py"))
    demo2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo2)
    res = {"workload": "demo2 silhouette fitting: 64 views at 64x64, 1 352-vertex sphere template, sigma 1e-4, one rank",
           "unit": "ms per iteration"}
    for fe, iters in (("device", 200), ("host", 40)):
        demo2.main(["--iters", "5", "--quiet", "--front-end", fe])
        hist = demo2.main(["--iters", str(iters), "--quiet", "--front-end", fe])
        res[fe] = demo2.main.loop_seconds / iters * 1e3
        res["loss_%s" % fe] = [float(hist[0]), float(hist[-1])]
    return res


def secondary_lines(args, ctx, comm):
    """What the driver's plain `bench.py` run should also put on the record (measured after the timed region):
    the single-image latency, larger K, the random-triangle scene and the NMR path."""
    from jrender_amd import synthetic as syn
    NF, IS, K = args.faces, args.image_size, args.K
    out = {}
    fv1, tex1 = syn.sphere_views(NF, 1)
    st, ph = fwd_bwd_ms(ctx, comm, fv1, tex1, IS, K)
    out["b1"] = {"workload": "ONE %d-face sphere view %dx%d fwd+bwd, K=%d" % (NF, IS, IS, K), "ms": st, "phase_ms": ph, "bin_size": ctx.bin_size()}
    fv8, tex8 = syn.sphere_views(NF, args.batch)
    for k2 in (32, 64):
        st, ph = fwd_bwd_ms(ctx, comm, fv8, tex8, IS, k2, steps=10, warmup=2)
        ab = algorithmic_bytes(args.batch, NF, 1, IS, k2)
        out["K%d" % k2] = {"workload": "the headline batch with max_faces_per_pixel_for_grad=%d" % k2, "ms": st, "phase_ms": ph,
                           "step_frac": ab["step"] / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    fvs, texs = syn.triangle_soup(NF, args.batch, seed=100)
    st, ph = fwd_bwd_ms(ctx, comm, fvs, texs, IS, K, steps=20, warmup=3)
    ab = algorithmic_bytes(args.batch, NF, 1, IS, K)
    out["soup"] = {"workload": "random-triangle soup %d faces x %d views, %dx%d (north_star's random-triangle batches)" % (NF, args.batch, IS, IS),
                   "ms": st, "images_per_s": args.batch / (st["median"] * 1e-3), "phase_ms": ph,
                   "step_frac": ab["step"] / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
    m = measure_n3mr(ctx, comm, NF, IS, 20, 3)
    out["n3mr"] = {"workload": "NMR rgb+alpha+depth fwd+bwd, %d faces (fill_back x2), %dx%d, batch 1 (BASELINE configs[4])" % (m["NF2"], IS, IS),
                   "ms": percentiles(m["per_step"]), "phase_ms": m["phase_ms_per_step"], "roofline": m["roofline"]}
    # the other NAMED BASELINE configurations as operator calls (fwd + bwd, device-resident inputs): the spot
    # cow of configs[0..1] on its own data (tests/golden/g1_spot.npz: 5 856 faces, texture_res 5), config 4's operator shape
    try:
        spot = np.load(os.path.join(ROOT, "tests", "golden", "g1_spot.npz"))
        for name, size in (("c1_spot_256", 256), ("c2_spot_1024", 1024)):
            st, ph = fwd_bwd_ms(ctx, comm, spot["fv"], spot["tex"], size, K)
            ab = algorithmic_bytes(1, spot["fv"].shape[1], spot["tex"].shape[2], size, K)
            out[name] = {"workload": "spot cow %d faces, T=%d, %dx%d, one view, fwd+bwd (BASELINE configs[%d])" % (
                spot["fv"].shape[1], spot["tex"].shape[2], size, size, 0 if size == 256 else 1),
                "ms": st, "phase_ms": ph, "bin_size": ctx.bin_size(), "step_frac": ab["step"] / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        fv4, tex4 = syn.sphere_views(3300, 64)
        st, ph = fwd_bwd_ms(ctx, comm, fv4, tex4, 64, K, sigma_val=1e-4, aggr_func_rgb="hard")
        out["c4_operator"] = {"workload": "config 4's operator call: 3 300-face sphere, 64 views at 64x64, sigma 1e-4, hard rgb, fwd+bwd",
                              "ms": st, "phase_ms": ph, "bin_size": ctx.bin_size()}
    except Exception as e:
        out["named_configs_error"] = repr(e)
    try:        # BASELINE configs[3]: the demo2 silhouette-fitting loop end to end (64 views at 64^2, 1 352-vertex template)
        out["c4_demo2"] = demo2_loop_ms()
    except Exception as e:
        out["c4_demo2"] = {"error": repr(e)}
    if not args.no_cpu_baseline:
        try:                                        # the reference's NMR kernels on one host thread (their CUDA form is one thread per face), bounded sample
            out["n3mr"]["cpu_baseline"] = n3mr_cpu_baseline(m["faces_h"], m["tex_h"], IS, budget_s=8.0)
        except Exception as e:
            out["n3mr"]["cpu_baseline"] = {"value": None, "unit": "ms", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
    return out


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
    # north_star: "RCCL all-gather of images/gradients": with more than one rank a step ends with BOTH collectives - the
    # all-reduce of the shared-vertex gradient and the all-gather of the rendered images (--exchange both), timed in the line
    exchange = args.exchange or ("both" if world > 1 else "none")

    def render():
        img = fn.execute(fv, tex)
        gf, _ = fn.grad(grad)
        return img, gf

    def do_exchange(img, gf):
        if exchange in ("allreduce_vertex_grads", "both"):
            shared_vertex_gradient(gf, faces_d, NV, comm)
        if exchange in ("allgather_images", "both"):
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
    # The communicator's work is done: every rank reads what RCCL says about it and tears it down HERE, together, while all
    # ranks are alive - rank 0 then goes on alone (latency, CPU baseline, parity) for about a minute, and a communicator whose
    # peers have exited is not something to find out about in an N-GPU run that cannot be rehearsed on this pool.
    rccl_ranks = comm.size if comm.backend == "rccl" else None            # ncclCommCount of the live communicator (jr_comm_size)
    backend = comm.backend
    comm.close()
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
    tj, vj = load_json("traffic_latest.json") or {}, load_json("valu_latest.json") or {}
    traffic = tj.get(dom) if profiled else None
    valu = dict(vj.get(dom) or {}) if profiled and vj.get(dom) else None
    here = csrc_hash()
    profile_stale = None
    kernels_same = None
    if profiled:
        profile_stale, kernels_same = profile_staleness(tj, vj, load_json("isa_same_latest.json"), here)
    if valu and per_launch[dom] > 0:
        # issue-slot occupancy against THIS run's launch time (the counters are per launch; the duration is measured live):
        # VALU wavefront-instructions x the mean issue cost of the kernel's opcode mix / (SIMDs x cycles), see tools/pmc_to_json.py
        cyc = per_launch[dom] * 1e-3 * valu.get("clock_ghz", 2.4) * 1e9 * valu.get("simds", 1024)
        raw = valu["valu_insts_per_launch"] * valu.get("mean_issue_cycles", 4.0) / cyc
        valu["busy_raw"] = raw
        valu["busy"] = min(1.0, raw)
        valu["useful_lane_frac"] = valu["busy"] * valu["lane_util"]
        # how close is the forward to what ITS arithmetic costs with every lane busy?  tools/sim/min_valu.py
        # prices the shipped kernel's ISA, region by region, with the trip / lane counts of an instrumented GPU run
        # (profiles/r05_path_counts.json) - it reproduces the PMC's VALU count - and the same at 64 lanes per trip:
        mv = load_json("min_valu_latest.json") if dom == "fwd_raster" else None
        if mv:
            valu["attainable_ms"] = mv["attainable_ms"]
            valu["frac_of_attainable"] = mv["attainable_ms"] / per_launch[dom]
            valu["model_over_measured_valu"] = mv["model_over_measured"]
            valu["split_ms_of_the_profiled_launch"] = mv["split_ms"]
            valu["gap_owner"] = mv["gap_owner"]
    # ... and the same for the OTHER raster kernel (VERDICT r5 next #1a / #3: the line carries both kernels' floors): PMC counters of
    # the backward + tools/sim/min_valu_bwd.py's model (ISA by loop nest x counts of the instrumented build)
    valu_bwd = None
    if profiled and vj.get("bwd_raster") and per_launch["bwd_raster"] > 0:
        valu_bwd = dict(vj["bwd_raster"])
        cyc = per_launch["bwd_raster"] * 1e-3 * valu_bwd.get("clock_ghz", 2.4) * 1e9 * valu_bwd.get("simds", 1024)
        valu_bwd["busy_raw"] = valu_bwd["valu_insts_per_launch"] * valu_bwd.get("mean_issue_cycles", 4.0) / cyc
        valu_bwd["busy"] = min(1.0, valu_bwd["busy_raw"])
        mb = load_json("min_valu_bwd_latest.json")
        if mb:
            valu_bwd.update(attainable_ms=mb["attainable_ms"], frac_of_attainable=mb["attainable_ms"] / per_launch["bwd_raster"],
                            pair_arithmetic_only_ms=mb["pair_arithmetic_only_ms"], model_over_measured_valu=mb["model_over_measured"],
                            split_ms_of_the_profiled_launch=mb["split_ms"], lanes_per_trip=mb["lanes_per_trip"])
        valu_bwd["avg_launch_ms"] = per_launch["bwd_raster"]
        valu_bwd["hbm_frac"] = ab["bwd_raster"] / (per_launch["bwd_raster"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    valu_fwd = None
    if profiled and vj.get("fwd_raster") and per_launch["fwd_raster"] > 0:
        mv = load_json("min_valu_latest.json")
        valu_fwd = {"avg_launch_ms": per_launch["fwd_raster"], "hbm_frac": ab["fwd_raster"] / (per_launch["fwd_raster"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if mv:
            valu_fwd.update(attainable_ms=mv["attainable_ms"], frac_of_attainable=mv["attainable_ms"] / per_launch["fwd_raster"])
    if dom == "bwd_raster" and valu_bwd:
        valu = valu_bwd
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
                                  % (world, exchange, backend)},
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
                     "valu": valu, "valu_bwd": valu_bwd, "valu_fwd_floor": valu_fwd, "profile_stale": profile_stale, "profiled_kernels_device_code_unchanged": kernels_same, "csrc_hash": here,
                     "profile_csrc_hash": tj.get("csrc_hash") if profiled else None},
        "phase_ms_per_step": {k: v[0] / args.steps for k, v in phases.items()},
        "exchange": {"kind": exchange, "backend": backend, "ms_per_step": percentiles(ex_ms)["median"] if ex_ms else 0.0},
        "tile_stats": ctx.last_stats(),
    }
    out["rccl_ranks"] = rccl_ranks
    out["comm_init"] = {"attempts": 2 if os.environ.get("JRENDER_IPC_RETRY") else 1,
                        "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    if backend == "rccl" and rccl_ranks != world:
        sys.exit("bench.py: RCCL reports %s ranks for a launch of %d" % (rccl_ranks, world))
    # Everything below is measured by rank 0 on its own GPU AFTER the timed region, at any world size (north_star: latency
    # and vertex-gradient error "reported at 1/2/4/8 GPUs"); the other ranks have left, so nothing here talks to them.
    from jrender_amd.comm import SingleCommunicator
    solo = SingleCommunicator()
    if not args.no_secondary:
        try:
            if world == 1:
                sec = secondary_lines(args, ctx, solo)
                out["latency_ms_b1"] = sec["b1"]["ms"]["median"]
                out["secondary"] = sec
            else:                                   # N > 1: the single-image latency only (K = 32 / 64, soup, NMR are N = 1 records)
                fv1, tex1 = syn.sphere_views(NF, 1) if args.scene == "sphere" else syn.triangle_soup(NF, 1, seed=100)
                st, ph = fwd_bwd_ms(ctx, solo, fv1, tex1, IS, K)
                out["latency_ms_b1"] = st["median"]
                out["secondary"] = {"b1": {"workload": "ONE %d-face view %dx%d fwd+bwd, K=%d, on rank 0's GPU" % (NF, IS, IS, K), "ms": st, "phase_ms": ph}}
        except Exception as e:                      # never break the headline line
            out["secondary"] = {"error": repr(e)}
    sample = None
    if not args.no_cpu_baseline:
        try:
            out["cpu_baseline"], sample = cpu_baseline(NF, K)
            if args.scene == "sphere":
                out["parity"] = parity_vs_sample(ctx, sample, K, mesh_faces, NV)
                if out["cpu_baseline"]["kind"] == "reference":
                    try:
                        out["parity"]["reference_platform_envelope"] = reference_platform_envelope(NF, K)
                    except Exception as e:
                        out["parity"]["reference_platform_envelope"] = {"error": repr(e)}
        except Exception as e:                      # the baseline must never break the bench line
            out["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": 0, "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    # a CONFORMANT number beside the fast one - the same batch through the precise-colour kernel set
    # (SoftRasterizeFunction(precise_colour=True): libm expf, IEEE quotients, the double-precision sigmoid of SRK:344, :401-411 in
    # the forward; the backward is the same kernel), with its own parity block on the same oracle view
    if not args.no_secondary and world == 1:
        try:
            fvp, texp = syn.sphere_views(NF, B) if args.scene == "sphere" else syn.triangle_soup(NF, B, seed=100)
            st, ph = fwd_bwd_ms(ctx, solo, fvp, texp, IS, K, steps=20, warmup=3, precise_colour=True)
            hp = {"workload": "the headline batch through the precise-colour kernel set (jr_softras_set_precise_colour)",
                  "ms": st, "phase_ms": ph, "images_per_s": B / (st["median"] * 1e-3),
                  "step_frac": ab["step"] / (st["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "vs_default_step": st["median"] / out["step_ms"]["median"]}
            if sample is not None and args.scene == "sphere":
                pp = parity_vs_sample(ctx, sample, K, mesh_faces, NV, precise_colour=True)
                hp["parity"] = {k: pp[k] for k in ("ids_match_frac", "rgba_err", "aggrs_err", "grad_faces_err", "vertex_grad_err", "view")}
                if pp.get("gradient_references"):
                    hp["parity"]["vs_exact_sum"] = pp["gradient_references"]["vs_exact_sum"]
            out.setdefault("secondary", {})["headline_precise"] = hp
        except Exception as e:
            out.setdefault("secondary", {})["headline_precise"] = {"error": repr(e)}
    print(json.dumps(out), flush=True)


def launch_ranks(n, argv, timeout_s=1500.0):
    """`python bench.py --gpus N` outside any launcher: start N ranks of this script (one process per GPU, each its
    own process group), rendezvous through a private 0700 directory, relay rank 0's JSON line.  ALL children are polled
    (jrender_amd/parallel.py: launch_ranks): a rank that exits non-zero - e.g. before ncclCommInitRank completes - ends
    the launch at once, the others are killed, and the launcher exits non-zero with the dead rank's stderr tail."""
    from jrender_amd.parallel import launch_ranks as _launch
    if _launch(os.path.abspath(__file__), n, argv, timeout_s=timeout_s, name="bench.py"):
        sys.exit(1)


def dry_run(args, ctx, rank, world):
    """`bench.py --dry-run-ranks N`: the N-rank start-up up to (not including) ncclCommInitRank, on however many GPUs the box
    has (VERDICT r4 next #7a).  Every rank validates its environment, takes part in the unique-id rendezvous, maps to its
    device and allocates the buffers the bench's exchange step would move (the all-gathered image batch, the all-reduced
    shared-vertex gradient); rank 0 prints ONE JSON line with every rank's report.  Exit code 1 when a check fails."""
    from jrender_amd import comm as jcomm, synthetic as syn
    B, NF, IS = args.batch, args.faces, args.image_size
    nv = syn.sphere_mesh(NF)[0].shape[0] if args.scene == "sphere" and NF in syn.SPHERE_SHAPES else 3 * NF
    payload = {"allgather_images_recv": 4 * 4 * IS * IS * B * world, "allgather_images_send": 4 * 4 * IS * IS * B,
               "allreduce_vertex_grads": 4 * 3 * nv}
    rep = jcomm.rccl_dry_run(ctx, rank, world, payload)
    if rank == 0:
        line = {"dry_run": True, "n_ranks": world, "ok": rep["ok"], "problems": rep["problems"],
                "one_gpu_per_rank": rep["one_gpu_per_rank"], "stopped_before": rep["stopped_before"],
                "unique_id_sha256": rep["id_sha256"], "payload_bytes": payload, "ranks": rep["ranks"],
                "note": "no communicator was created: ncclCommInitRank with more than one rank needs one GPU per rank"}
        print(json.dumps(line), flush=True)
        if not rep["ok"]:
            sys.exit(1)


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
    ap.add_argument("--exchange", default=None, choices=["none", "allreduce_vertex_grads", "allgather_images", "both"],
                    help="exchange step at the end of every step (default when N > 1: both - the all-reduce of the shared-vertex "
                         "gradient AND the all-gather of the images, as north_star words it)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (cpu_baseline + parity)")
    ap.add_argument("--no-secondary", action="store_true", help="skip latency_ms_b1 / secondary (K=32, K=64, soup, NMR)")
    ap.add_argument("--dry-run-ranks", type=int, default=0, metavar="N",
                    help="spawn N ranks that do everything an N-GPU start-up does up to ncclCommInitRank - environment, rendezvous, "
                         "unique id, device mapping, exchange buffers - without creating the communicator; one JSON line (runs on a 1-GPU box)")
    ap.add_argument("--allow-shared-gpus", action="store_true",
                    help="plumbing runs only: let ranks share GPUs over the host communicator when fewer GPUs than ranks are "
                         "visible (without it such a launch exits non-zero: an N-rank line must mean N GPUs)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.dry_run_ranks > 1:
        return launch_ranks(args.dry_run_ranks, sys.argv[1:], timeout_s=300.0)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("JRENDER_BENCH_FAIL_RANK") is not None:        # tests/test_parallel.py: the launcher must notice a dead rank
        if os.environ["JRENDER_BENCH_FAIL_RANK"] == str(rank):
            sys.exit("bench.py: injected failure of rank %d (launcher test)" % rank)
        time.sleep(5.0)       # the injected failure must be the FIRST exit the launcher sees (on a box without a GPU the
                              # healthy ranks die too, at the device query below, and used to win that race now and then)
    from jrender_amd import _ffi, comm as jcomm
    ndev = _ffi.device_count()
    if ndev < 1:
        sys.exit("bench.py: no HIP device visible (there is no CPU fallback)")
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if args.dry_run_ranks:
        return dry_run(args, _ffi.Context(local_rank % ndev), rank, world)
    if ndev < local_world and not args.allow_shared_gpus:
        sys.exit("bench.py: %d ranks on this node but only %d GPU(s) visible - an N-rank line must mean N GPUs "
                 "(--allow-shared-gpus runs the plumbing over the host communicator instead)" % (local_world, ndev))
    if world > 1 and rank != 0:
        # under an external launcher every rank shares ONE stdout: only rank 0 may write there (the JSON line); what the others or
        # their libraries print (RCCL's start-up banner) goes to stderr
        sys.stdout.flush()
        os.dup2(2, 1)
    ctx = _ffi.Context(local_rank % ndev)       # --allow-shared-gpus: plumbing run, ranks share GPUs (host communicator)
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
