#!/usr/bin/env python3
"""Turntable + blur sweep of a lit, textured mesh — the workload of the reference's demo1-render.py:26-59
(BASELINE.json configs[0] / [1]) on the HIP SoftRas path:

    python examples/demo1_render.py -i data/obj/spot/spot_triangulated.obj -o out/      # the reference's input
    python examples/demo1_render.py -o out/                                              # no OBJ: a synthetic sphere

1. 90 views on a turntable (`renderer.transform.set_eyes_from_angles`, `render_mesh(mesh, mode='rgb')`,
   `mesh.reset_()` between renders because Transform and Lighting mutate the mesh like the reference's);
2. ten renders with sigma = 10^(g-1), gamma = 10^g for g = -4 ... -2.2 (`set_sigma` / `set_gamma`).  The reference
   bakes both scalars into its JIT-compiled CUDA source (SRK:485-516), so every step of this sweep recompiles the
   kernel there; here they are kernel ARGUMENTS (jr_softras_forward) and the sweep runs at the turntable's pace —
   the per-frame times are printed so that this is visible;
3. the mesh is written back with `save_obj`.

Frames go to rotation.gif / bluring.gif when Pillow is importable, else to .npy stacks.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import jrender_amd as jr                                                     # noqa: E402


def save_frames(frames, path_gif):
    frames = [np.clip(255 * f, 0, 255).astype(np.uint8) for f in frames]
    try:
        from PIL import Image
        ims = [Image.fromarray(f) for f in frames]
        ims[0].save(path_gif, save_all=True, append_images=ims[1:], duration=40, loop=0)
        return path_gif
    except ImportError:
        out = os.path.splitext(path_gif)[0] + ".npy"
        np.save(out, np.stack(frames))
        return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-i', '--filename-input', type=str, default=None, help="Wavefront OBJ (with its MTL / texture); default: a synthetic sphere")
    ap.add_argument('-o', '--output-dir', type=str, default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'output_render'))
    ap.add_argument('--image-size', type=int, default=256)
    ap.add_argument('--step', type=int, default=4, help="azimuth step of the turntable in degrees")
    args = ap.parse_args(argv)
    camera_distance, elevation = 2.732, 30

    if args.filename_input:
        mesh = jr.Mesh.from_obj(args.filename_input, load_texture=True, texture_res=5, texture_type='surface', dr_type='softras')
    else:
        v, f = jr.synthetic.sphere_mesh(3300)
        mesh = jr.Mesh(v * 0.8, f, textures=jr.synthetic.face_colors(f.shape[0], 25)[None], texture_res=5,
                       texture_type='surface', dr_type='softras')
    renderer = jr.Renderer(image_size=args.image_size, dr_type='softras')
    os.makedirs(args.output_dir, exist_ok=True)

    frames, times = [], []
    for azimuth in range(0, 360, args.step):                                 # demo1-render.py:36-44
        mesh.reset_()
        renderer.transform.set_eyes_from_angles(camera_distance, elevation, azimuth)
        t0 = time.perf_counter()
        rgb = renderer.render_mesh(mesh, mode='rgb')
        frames.append(rgb.numpy()[0].transpose((1, 2, 0)))
        times.append(time.perf_counter() - t0)
    print("turntable: %d frames, median %.2f ms per frame (render + download) -> %s"
          % (len(frames), 1e3 * float(np.median(times)), save_frames(frames, os.path.join(args.output_dir, 'rotation.gif'))))

    frames, times = [], []
    renderer.transform.set_eyes_from_angles(camera_distance, elevation, 45)
    for gamma_pow in np.arange(-4, -2, 0.2):                                 # demo1-render.py:47-58
        mesh.reset_()
        renderer.set_gamma(10 ** gamma_pow)
        renderer.set_sigma(10 ** (gamma_pow - 1))
        t0 = time.perf_counter()
        images = renderer.render_mesh(mesh, mode='rgb')
        frames.append(images.numpy()[0].transpose((1, 2, 0)))
        times.append(time.perf_counter() - t0)
    print("blur sweep: %d (sigma, gamma) pairs, median %.2f ms per frame, slowest %.2f ms - no recompilation between them -> %s"
          % (len(frames), 1e3 * float(np.median(times)), 1e3 * max(times), save_frames(frames, os.path.join(args.output_dir, 'bluring.gif'))))

    mesh.reset_()
    mesh.save_obj(os.path.join(args.output_dir, 'saved_mesh.obj'))
    return frames


if __name__ == '__main__':
    main()
