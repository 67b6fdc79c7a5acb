#!/usr/bin/env python3
"""Mesh deformation from silhouettes — the workload of the reference's demo2-deform.py (BASELINE.json
configs[3]) on the HIP SoftRas path, with the backward chain written out (no autograd framework):

    displace, center --(sigmoid/tanh parametrisation, demo2-deform.py:35-41)--> vertices
      --look_at + perspective--> face_vertices --SoftRas (HIP)--> silhouettes --neg-IoU loss
      + 0.03 Laplacian + 0.0003 flatten regularisers;   Adam(0.01, betas=(0.5, 0.99))

    python examples/demo2_deform.py [--iters 200] [--batch-size 64] [-i source.npy -c camera.npy]
    python examples/demo2_deform.py --gpus 8                 # views sharded over 8 GPUs (one process each, RCCL)

Without -i/-c (the reference's data files are not shipped here) the target is synthetic: silhouettes of a
squashed, shifted ellipsoid seen from a ring of cameras.  With several ranks every rank renders its slice of
the views; the vertex gradient (the mesh is shared by all views) is summed with one all-reduce.

--front-end device (default): NOTHING of an iteration runs on the host (round 5).  Template, displacement map, centre and
the Adam moments live on the GPU; an iteration is a chain of launches on one stream - parametrisation kernel, camera kernel
(one vertex set broadcast over the views), face gather, SoftRas forward, IoU loss + its gradient, SoftRas backward, scatter to
the vertices + camera VJP summed over the views, RCCL all-reduce, Laplacian and flatten regularisers with their gradients, the
parametrisation's VJP (which also combines the three gradients), one Adam launch per parameter - and the three loss terms of
the iteration land in a history array ON the device, downloaded when a line is printed (every 20 iterations) and at the end.
--front-end host runs the same chain through the NumPy mirrors (the 6 MB face arrays cross PCIe twice per iteration); both
produce the same loss curve (tests/test_gpu_named_configs.py, tests/test_gpu_device_chain.py).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jrender_amd as jr                                                     # noqa: E402
from jrender_amd import comm as jcomm                                        # noqa: E402
from jrender_amd.parallel import shard_bounds                                # noqa: E402


Model = jr.DeformModel      # demo2-deform.py:17-47 with an explicit backward; on the GPU when given a context (jrender_amd/deform.py)


def synthetic_target(renderer, template_v, faces, n_views):
    """Silhouettes of an ellipsoid (semi-axes 0.45, 0.2, 0.3, shifted) from a ring of cameras."""
    dist = np.full(n_views, 2.732, np.float32)
    elev = np.where(np.arange(n_views) % 2 == 0, 30.0, -20.0).astype(np.float32)
    azim = (np.arange(n_views, dtype=np.float32) * 360.0 / n_views)
    cameras = np.stack([dist, elev, azim], 1)
    v = template_v * np.asarray([0.45, 0.2, 0.3], np.float32) + np.asarray([0.1, 0.05, 0.0], np.float32)
    renderer.transform.set_eyes_from_angles(dist, elev, azim)
    mesh = jr.Mesh(np.broadcast_to(v[None], (n_views,) + v.shape).copy(), np.broadcast_to(faces[None], (n_views,) + faces.shape).copy())
    sil = renderer.render_mesh(mesh, mode='silhouettes').numpy()
    return sil.reshape(n_views, sil.shape[-2], sil.shape[-1]), cameras


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-i', '--filename-input', default=None, help="reference-format source.npy [N,4,64,64] uint8")
    ap.add_argument('-c', '--camera-input', default=None, help="reference-format camera.npy [N,3] (distance, elevation, azimuth)")
    ap.add_argument('-b', '--batch-size', type=int, default=64)
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--image-size', type=int, default=64)
    ap.add_argument('-o', '--output', default=None, help="write the optimised mesh to this .obj")
    ap.add_argument('--template-vertices', default=None, help=".obj, or .npz with 'vertices' / 'faces' (default: a 1 352-vertex UV sphere)")
    ap.add_argument('--quiet', action='store_true')
    ap.add_argument('--gpus', type=int, default=1, help="spawn this many ranks (one process per GPU)")
    ap.add_argument('--history-out', default=None, help="rank 0 writes the loss of every iteration to this .npy")
    ap.add_argument('--graph', action='store_true', help="device front end, one rank: run the iteration as ONE HIP graph (recorded at the third iteration, replayed afterwards)")
    ap.add_argument('--front-end', choices=['device', 'host'], default='device',
                    help="where the camera / gather / loss steps around the rasteriser run (see the module docstring)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.batch_size < world:
        raise SystemExit("demo2_deform: %d views cannot be sharded over %d ranks" % (args.batch_size, world))
    # one process per GPU: the context follows LOCAL_RANK; RCCL when every rank owns a GPU, else host sockets
    comm = jcomm.init_from_env(jr.Context.default()) if world > 1 else None

    if args.template_vertices and args.template_vertices.endswith(".npz"):
        z = np.load(args.template_vertices)
        tv, tf = z["vertices"], z["faces"]
    elif args.template_vertices:
        tv, tf = jr.load_obj(args.template_vertices)                         # demo2-deform.py:61 (sphere_1352.obj)
    else:
        tv, tf = jr.synthetic.uv_sphere(52, 27)                               # 1 352-vertex class template
    device = args.front_end == 'device'
    graph_mode = bool(args.graph)
    if graph_mode and (not device or world > 1):
        raise SystemExit("--graph needs the device front end on one rank (a collective cannot be recorded)")
    ctx = jr.Context.default()
    model = Model(tv, tf, ctx=ctx if device else None)
    # (demo2-deform.py:65 also passes bin_size=16, max_elems_per_bin=2700 - tuning of the reference's own binned kernels.  Here
    #  the bin size only selects how finely the lists follow the image, never the result, and the library's own choice for
    #  64^2 images - 8-pixel bins - is the fastest: 0.35 ms per operator call against 0.41 with 16; pass bin_size=16 to compare)
    renderer = jr.Renderer(image_size=args.image_size, sigma_val=1e-4, aggr_func_rgb='hard', camera_mode='look_at',
                           viewing_angle=15, dr_type='softras', max_faces_per_pixel_for_grad=16)
    if args.filename_input and args.camera_input:
        images = np.load(args.filename_input).astype(np.float32) / 255.
        cameras = np.load(args.camera_input).astype(np.float32)
        target = images[:args.batch_size, 3]
        cameras = cameras[:args.batch_size]
    else:
        target, cameras = synthetic_target(renderer, tv, tf, args.batch_size)
    B = target.shape[0]
    lo, hi = (0, B) if comm is None else shard_bounds(B, world)[rank]
    nb = hi - lo
    if nb < 1:                           # (cannot happen after the check above; every rank takes the SAME branch below whatever its share)
        raise SystemExit("demo2_deform: rank %d got no view" % rank)
    renderer.transform.set_eyes_from_angles(cameras[lo:hi, 0], cameras[lo:hi, 1], cameras[lo:hi, 2])
    optimizer = jr.Adam(model.parameters(), 0.01, betas=(0.5, 0.99))

    t0 = time.time()
    # per iteration: [sum of this rank's IoUs, Laplacian loss, flatten loss]
    hist = ctx.zeros((max(args.iters, 1), 3)) if device else np.zeros((max(args.iters, 1), 3), np.float32)
    target_d = ctx.array(np.ascontiguousarray(target[lo:hi], np.float32)) if device else None

    def losses_upto(n):
        """[n] total losses of iterations 0 .. n-1 (device front end: ONE download; the IoU sums of the ranks are added up)"""
        h = hist.view(0, n).numpy() if device else hist[:n]
        iou_sum = h[:, 0].astype(np.float64)
        if comm is not None:
            iou_sum = np.asarray(comm.all_reduce_sum_host(np.ascontiguousarray(iou_sum, np.float32)), np.float64)
        return (1.0 - iou_sum / B) + 0.03 * h[:, 1] + 0.0003 * h[:, 2], h

    # --graph: the iteration number lives on the device (Adam's step, the history row), two iterations run as usual, the third is
    # RECORDED into a HIP graph (ctx.capture) and every further one is a replay of it: one launch per iteration instead of ~30
    it_dev = ctx.array(np.zeros(1, np.int32)) if graph_mode else None

    def device_iteration(it):
        vertices = model.forward()                                            # [1,nv,3] DeviceArray
        # ONE vertex set on the device; the camera step broadcasts it over this rank's eyes (demo2-deform.py:45
        # materialises the copies with repeat())
        mesh = jr.Mesh(vertices, model.faces)
        pred = renderer.render_mesh(mesh, mode='silhouettes')             # DeviceArray [nb,IS,IS]
        iou, g_sil = jr.neg_iou_loss_and_grad(pred, target_d, total_views=B)
        g_v = renderer.grad_vertices(grad_silhouettes=g_sil)              # DeviceArray [1,nv,3]: summed over the views
        if comm is not None:
            g_v = comm.all_reduce_sum(g_v)           # RCCL on the device buffer
        # the regularisers: value and gradient on the same device vertices; nothing has waited for the GPU
        lap, g_lap = model.laplacian_loss.value_and_grad(vertices)
        flat, g_flat = model.flatten_loss.value_and_grad(vertices)
        # parametrisation VJP of g_v + 0.03 g_lap + 0.0003 g_flat (demo2-deform.py:85-88), Adam on the device
        optimizer.step(model.backward(g_v, (0.03, g_lap), (0.0003, g_flat)), iteration=it_dev)
        if it_dev is None:
            ctx.scalar_accumulate(hist, 3 * it + 0, iou)
            ctx.scalar_accumulate(hist, 3 * it + 1, lap.values)
            ctx.scalar_accumulate(hist, 3 * it + 2, flat.values)
        else:
            ctx.scalar_accumulate(hist, 0, iou, iteration=it_dev, stride=3)
            ctx.scalar_accumulate(hist, 1, lap.values, iteration=it_dev, stride=3)
            ctx.scalar_accumulate(hist, 2, flat.values, iteration=it_dev, stride=3)
            ctx.counter_add(it_dev, 1)

    graph = None
    for it in range(args.iters):
        if device:
            if graph is not None:
                graph.launch()
            elif graph_mode and it == 2:
                with ctx.capture() as g:
                    device_iteration(it)
                graph = g
                graph.launch()                       # (the capture itself executed nothing)
            else:
                device_iteration(it)
        else:
            vertices = model.forward()
            mesh = jr.Mesh(np.repeat(vertices, nb, 0), np.repeat(model.faces, nb, 0))
            pred = renderer.render_mesh(mesh, mode='silhouettes').numpy().reshape(nb, args.image_size, args.image_size)
            # neg-IoU over ALL views: per-view IoUs are independent, so the local part is (1/B) * sum over local views
            inter = (pred * target[lo:hi]).sum((1, 2))
            union = (pred + target[lo:hi] - pred * target[lo:hi]).sum((1, 2)) + 1e-6
            g_sil = jr.neg_iou_loss_backward(pred, target[lo:hi]) * (nb / B)      # that helper averages over its own batch
            g_v = renderer.grad_vertices(grad_silhouettes=g_sil.reshape(nb, 1, args.image_size, args.image_size)).sum(0, keepdims=True)
            if comm is not None:
                g_v = comm.all_reduce_sum_host(g_v)          # [1,nv,3]: the mesh is shared by all views
            reg = (model.laplacian_loss(vertices), model.laplacian_loss.backward(vertices)) + model.flatten_loss.value_and_grad(vertices)
            hist[it] = (float((inter / union).sum()), float(np.mean(reg[0])), float(np.mean(reg[2])))
            optimizer.step(model.backward(g_v, (0.03, reg[1]), (0.0003, reg[3])))
        if graph is not None and (it % 20 == 0 or it == args.iters - 1):
            graph.check()                            # (waits; fails when a replayed forward outgrew the captured pool)
        if not args.quiet and (it % 20 == 0 or it == args.iters - 1):          # (every rank: the IoU sums are all-reduced)
            loss, h = losses_upto(it + 1)
            iou_all = B * (1.0 - (loss[it] - 0.03 * h[it, 1] - 0.0003 * h[it, 2]))
            if rank == 0:
                print("iter %4d  loss %.4f  (1-IoU %.4f, laplacian %.4f, flatten %.4f)" % (it, loss[it], 1.0 - iou_all / B, h[it, 1], h[it, 2]), flush=True)
    if device:
        ctx.synchronize()
    if graph is not None:
        graph.check()
        graph.close()
    history = [float(x) for x in losses_upto(args.iters)[0]] if args.iters else []
    main.loop_seconds = time.time() - t0           # the optimisation loop alone (bench.py's secondary.c4_demo2 reads it)
    if rank == 0 and not args.quiet:
        print("%d iterations, %d views on %d rank(s): %.2f s" % (args.iters, B, world, main.loop_seconds))
    if rank == 0 and args.history_out:
        np.save(args.history_out, np.asarray(history, np.float64))
    if rank == 0 and args.output:
        jr.save_obj(args.output, np.asarray(model.forward().numpy() if device else model.forward())[0], model.faces[0])
    if comm is not None:
        comm.close()
    return history


def launch_ranks(n, argv):
    """Start n ranks of this script (one process per GPU) with a private rendezvous; every rank is polled, a rank that dies
    (e.g. before ncclCommInitRank) ends the launch with its stderr instead of leaving the others blocked
    (jrender_amd/parallel.py: launch_ranks, the launcher bench.py --gpus N uses)."""
    from jrender_amd.parallel import launch_ranks as _launch
    if _launch(os.path.abspath(__file__), n, argv, name="demo2_deform.py"):
        raise SystemExit(1)


if __name__ == '__main__':
    main()
