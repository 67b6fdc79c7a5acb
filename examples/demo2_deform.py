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

--front-end device (default): the 16 KB vertex set goes to the GPU once per iteration and everything between it and
the vertex gradient stays there — camera kernel (one vertex set broadcast over the views), face gather, SoftRas
forward, IoU loss + its gradient, SoftRas backward, scatter to the vertices + camera VJP summed over the views,
RCCL all-reduce, Laplacian and flatten regularisers with their gradients — and three 16 KB gradients come back.  --front-end host runs the same chain through the NumPy mirrors (the
6 MB face arrays cross PCIe twice per iteration); both produce the same loss curve (tests/test_gpu_named_configs.py).
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


class Model:
    """demo2-deform.py:17-47 with an explicit backward."""

    def __init__(self, vertices, faces):
        self.vertices = (np.asarray(vertices, np.float32) * 0.5)[None]      # [1,nv,3], |v| < 1
        self.faces = np.asarray(faces, np.int32)[None]
        self.displace = np.zeros_like(self.vertices)
        self.center = np.zeros((1, 1, 3), np.float32)
        self.laplacian_loss = jr.LaplacianLoss(self.vertices[0], self.faces[0])
        self.flatten_loss = jr.FlattenLoss(self.faces[0])

    def parameters(self):
        return [self.displace, self.center]

    def forward(self):
        a = np.abs(self.vertices)
        with np.errstate(divide="ignore"):
            base = np.log(a / (1 - a))
        self._c = np.tanh(self.center)
        self._s = 1.0 / (1.0 + np.exp(-(base + self.displace)))
        self._sign = np.sign(self.vertices)
        u = self._s * self._sign
        self._u = u
        v = np.maximum(u, 0) * (1 - self._c) - np.maximum(-u, 0) * (self._c + 1) + self._c
        return v.astype(np.float32)

    def backward(self, g):
        """g = d(loss)/d(vertices) [1,nv,3] -> (d/d displace, d/d center)."""
        u, c = self._u, self._c
        g_c = (g * (1 - np.maximum(u, 0) - np.maximum(-u, 0))).sum(1, keepdims=True)
        g_u = g * ((u > 0) * (1 - c) + (u < 0) * (c + 1))
        g_disp = g_u * self._sign * self._s * (1 - self._s)
        return g_disp.astype(np.float32), (g_c * (1 - c * c)).astype(np.float32)


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
    model = Model(tv, tf)
    renderer = jr.Renderer(image_size=args.image_size, sigma_val=1e-4, aggr_func_rgb='hard', camera_mode='look_at',
                           viewing_angle=15, dr_type='softras', bin_size=16, max_elems_per_bin=2700,
                           max_faces_per_pixel_for_grad=16)
    if args.filename_input and args.camera_input:
        images = np.load(args.filename_input).astype(np.float32) / 255.
        cameras = np.load(args.camera_input).astype(np.float32)
        target = images[:args.batch_size, 3]
        cameras = cameras[:args.batch_size]
    else:
        target, cameras = synthetic_target(renderer, tv, tf, args.batch_size)
    B = target.shape[0]
    lo, hi = (0, B) if comm is None else shard_bounds(B, world)[rank]
    renderer.transform.set_eyes_from_angles(cameras[lo:hi, 0], cameras[lo:hi, 1], cameras[lo:hi, 2])
    optimizer = jr.Adam(model.parameters(), 0.01, betas=(0.5, 0.99))

    t0 = time.time()
    history = []
    nb = hi - lo
    ctx = jr.Context.default()
    target_d = ctx.array(np.ascontiguousarray(target[lo:hi], np.float32)) if args.front_end == 'device' and nb else None
    for it in range(args.iters):
        vertices = model.forward()                                            # [1,nv,3]
        if args.front_end == 'device' and nb:
            # ONE vertex set on the device; the camera step broadcasts it over this rank's eyes (demo2-deform.py:45
            # materialises the copies with repeat())
            vertices_d = ctx.array(vertices)
            mesh = jr.Mesh(vertices_d, model.faces)
            pred = renderer.render_mesh(mesh, mode='silhouettes')             # DeviceArray [nb,IS,IS]
            iou, g_sil = jr.neg_iou_loss_and_grad(pred, target_d, total_views=B)
            g_v = renderer.grad_vertices(grad_silhouettes=g_sil)              # DeviceArray [1,nv,3]: summed over the views
            if comm is not None:
                g_v = comm.all_reduce_sum(g_v)           # RCCL on the device buffer
            # the regularisers: one launch each on the same device vertices; nothing so far has waited for the GPU
            reg = model.laplacian_loss.value_and_grad(vertices_d) + model.flatten_loss.value_and_grad(vertices_d)
            reg = tuple(r.numpy() for r in reg)
            iou_sum = float(iou.numpy().sum())
            if comm is not None:
                iou_sum = comm.all_reduce_scalar(iou_sum, "sum")
            g_v = g_v.numpy()
        else:
            mesh = jr.Mesh(np.repeat(vertices, nb, 0), np.repeat(model.faces, nb, 0))
            pred = renderer.render_mesh(mesh, mode='silhouettes').numpy().reshape(nb, args.image_size, args.image_size)
            # neg-IoU over ALL views: per-view IoUs are independent, so the local part is (1/B) * sum over local views
            inter = (pred * target[lo:hi]).sum((1, 2))
            union = (pred + target[lo:hi] - pred * target[lo:hi]).sum((1, 2)) + 1e-6
            iou_sum = float((inter / union).sum())
            g_sil = jr.neg_iou_loss_backward(pred, target[lo:hi]) * (nb / B)      # that helper averages over its own batch
            g_v = renderer.grad_vertices(grad_silhouettes=g_sil.reshape(nb, 1, args.image_size, args.image_size)).sum(0, keepdims=True)
            if comm is not None:
                g_v = comm.all_reduce_sum_host(g_v)          # [1,nv,3]: the mesh is shared by all views
                iou_sum = comm.all_reduce_scalar(iou_sum, "sum")
            reg = (model.laplacian_loss(vertices), model.laplacian_loss.backward(vertices)) + model.flatten_loss.value_and_grad(vertices)
        lap, flat = float(np.mean(reg[0])), float(np.mean(reg[2]))
        loss = (1.0 - iou_sum / B) + 0.03 * lap + 0.0003 * flat
        g_v = g_v + 0.03 * reg[1] + 0.0003 * reg[3]
        optimizer.step(model.backward(g_v))
        history.append(loss)
        if rank == 0 and not args.quiet and (it % 20 == 0 or it == args.iters - 1):
            print("iter %4d  loss %.4f  (1-IoU %.4f, laplacian %.4f, flatten %.4f)" % (it, loss, 1.0 - iou_sum / B, lap, flat), flush=True)
    main.loop_seconds = time.time() - t0           # the optimisation loop alone (bench.py's secondary.c4_demo2 reads it)
    if rank == 0 and not args.quiet:
        print("%d iterations, %d views on %d rank(s): %.2f s" % (args.iters, B, world, main.loop_seconds))
    if rank == 0 and args.history_out:
        np.save(args.history_out, np.asarray(history, np.float64))
    if rank == 0 and args.output:
        jr.save_obj(args.output, model.forward()[0], model.faces[0])
    if comm is not None:
        comm.close()
    return history


def launch_ranks(n, argv):
    """Start n ranks of this script (one process per GPU) with a private file rendezvous."""
    import subprocess
    import tempfile
    rdzv = os.path.join(tempfile.mkdtemp(prefix="jrender_demo2_"), "rdzv")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv,
                              env=dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n),
                                       LOCAL_WORLD_SIZE=str(n), JRENDER_RDZV=rdzv))
             for r in range(n)]
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit("demo2_deform: rank exit codes %s" % rcs)


if __name__ == '__main__':
    main()
