"""CPU oracle for the SoftRas hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``jrender_amd`` never does.

Two interchangeable back-ends behind one NumPy interface:

* ``kind="port"`` — ``oracle/softras_oracle.c``: our plain-C restatement of the
  reference arithmetic (every function cites SRK/SRW file:line).
* ``kind="reference"`` — ``oracle/_ref/libsoftras_ref.so``: the reference's own
  kernel source compiled for the host by ``oracle/build_ref.py`` (only
  buildable where ``/root/reference`` is mounted; the built ``.so`` travels).

Argument handling mirrors the reference wrapper (SRW:10-42): ``dist_eps`` is
mapped to ``log(1/dist_eps - 1)`` (SRW:25), the enum strings to the integer
ids of SRW:39-42, every scalar is passed to the kernels as ``float``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_LIB = os.path.join(HERE, "libsoftras_oracle.so")
REF_LIB = os.path.join(HERE, "_ref", "libsoftras_ref.so")

DIST = {"hard": 0, "barycentric": 1, "euclidean": 2}          # SRW:39
RGB = {"hard": 0, "softmax": 1, "none": 2}                    # SRW:40
ALPHA = {"hard": 0, "sum": 1, "prod": 2}                      # SRW:41
TEX = {"surface": 0, "vertex": 1}                             # SRW:42

DEFAULTS = dict(image_size=256, near=1, far=100, fill_back=True, eps=1e-3, sigma_val=1e-5,
                dist_func="euclidean", dist_eps=1e-4, gamma_val=1e-4, aggr_func_rgb="softmax",
                aggr_func_alpha="prod", texture_type="surface", max_faces_per_pixel_for_grad=16)


def make_port(force=False):
    src = os.path.join(HERE, "softras_oracle.c")
    if (not force and os.path.exists(PORT_LIB)
            and os.path.getmtime(PORT_LIB) >= os.path.getmtime(src)):
        return PORT_LIB
    tmp = PORT_LIB + ".tmp%d" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math",
                           "-fopenmp", "-fPIC", "-shared", src, "-o", tmp, "-lm"])
    os.replace(tmp, PORT_LIB)
    return PORT_LIB


N3MR_PORT_LIB = os.path.join(HERE, "libn3mr_oracle.so")


def make_n3mr_port(force=False):
    """gcc build of oracle/n3mr_oracle.c (the plain-C restatement of the NMR kernels)."""
    src = os.path.join(HERE, "n3mr_oracle.c")
    if (not force and os.path.exists(N3MR_PORT_LIB)
            and os.path.getmtime(N3MR_PORT_LIB) >= os.path.getmtime(src)):
        return N3MR_PORT_LIB
    tmp = N3MR_PORT_LIB + ".tmp%d" % os.getpid()
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-shared",
                           src, "-o", tmp, "-lm"])
    os.replace(tmp, N3MR_PORT_LIB)
    return N3MR_PORT_LIB


def make_ref(force=False):
    import importlib
    return importlib.import_module(__name__ + ".build_ref").build(force=force)


def have_ref():
    return os.path.exists(REF_LIB) or make_ref() is not None


def _scalars(kw):
    p = dict(DEFAULTS)
    # bin_size / max_elems_per_bin: arguments of the reference operator (SRW:15) that select ITS coarse-to-fine kernels; the bin_size = 0
    # kernels these oracles restate ignore them (tests/test_oracle_c2f.py: with ascending lists the binned forward gives the same bits)
    unknown = set(kw) - set(p) - {"background_color", "bin_size", "max_elems_per_bin"}
    if unknown:
        raise TypeError("unknown parameters: %s" % sorted(unknown))
    p.update(kw)
    f32 = lambda v: C.c_float(float(np.float32(v)))
    dist_eps_log = np.log(1.0 / p["dist_eps"] - 1.0)             # SRW:25
    return p, dict(near=f32(p["near"]), far=f32(p["far"]), eps=f32(p["eps"]),
                   sigma=f32(p["sigma_val"]), dist=DIST[p["dist_func"]],
                   dist_eps=f32(dist_eps_log), gamma=f32(p["gamma_val"]),
                   rgb=RGB[p["aggr_func_rgb"]], alpha=ALPHA[p["aggr_func_alpha"]],
                   tex=TEX[p["texture_type"]], ds=int(bool(p["fill_back"])))


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Oracle:
    """forward()/backward() on NumPy arrays; layouts are the reference's."""

    def __init__(self, kind="port", nthreads=0):
        self.kind = kind
        self.nthreads = int(nthreads)
        if kind == "port":
            self.lib = C.CDLL(make_port())
            self.lib.orc_ub_events.restype = C.c_long
        elif kind in ("reference", "reference_fma"):
            # "reference_fma": the same sources with floating-point contraction on (what nvcc does by default) - a
            # measuring stick for the reference's own platform sensitivity, never an oracle (build_ref.build_fma)
            if kind == "reference":
                path = make_ref()
            else:
                import importlib
                path = importlib.import_module(__name__ + ".build_ref").build_fma()
            if path is None or not os.path.exists(path):
                raise FileNotFoundError("oracle/_ref not built and /root/reference not mounted")
            self.lib = C.CDLL(path)
            self.kind = "reference"
        else:
            raise ValueError(kind)

    def num_procs(self):
        return (self.lib.orc_num_procs if self.kind == "port" else self.lib.ref_num_procs)()

    def ub_events(self):
        return int(self.lib.orc_ub_events()) if self.kind == "port" else 0

    def forward(self, face_vertices, textures, **kw):
        p, s = _scalars(kw)
        fv = np.ascontiguousarray(face_vertices, np.float32)
        B, NF = fv.shape[:2]
        fv = fv.reshape(B, NF, 9)
        tex = np.ascontiguousarray(textures, np.float32).reshape(B, NF, -1, 3)
        T, IS, K = tex.shape[2], int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        info = np.empty((B, NF, 27), np.float32)
        aggr = np.empty((B, 2, IS, IS), np.float32)
        rgba = np.empty((B, 4, IS, IS), np.float32)
        ids = np.empty((B, K, IS, IS), np.int32)
        common = (B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"],
                  s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"])
        if self.kind == "port":
            bg = kw.get("background_color")
            bgp = None if bg is None else _fp(np.ascontiguousarray(bg, np.float32))
            self.lib.orc_ub_reset()
            rc = self.lib.orc_softras_forward(_fp(fv), _fp(tex), _fp(info), _fp(aggr), _fp(rgba),
                                              _ip(ids), *common, bgp, self.nthreads)
        else:
            rc = self.lib.ref_softras_forward(_fp(fv), _fp(tex), _fp(info), _fp(aggr), _fp(rgba),
                                              _ip(ids), *common, self.nthreads)
        if rc:
            raise RuntimeError("oracle forward failed rc=%d" % rc)
        return dict(face_vertices=fv, textures=tex, soft_colors=rgba, faces_info=info,
                    aggrs_info=aggr, faces_id_buffer=ids, params=p)

    def backward(self, saved, grad_soft_colors, nthreads=1):
        p, s = _scalars({k: v for k, v in saved["params"].items()})
        fv, tex = saved["face_vertices"], saved["textures"]
        B, NF, T = fv.shape[0], fv.shape[1], tex.shape[2]
        IS, K = int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        g = np.ascontiguousarray(grad_soft_colors, np.float32).reshape(B, 4, IS, IS)
        gf = np.empty((B, NF, 9), np.float32)
        gt = np.empty((B, NF, T, 3), np.float32)
        common = (B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"],
                  s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"], int(nthreads))
        ids = saved["faces_id_buffer"]
        if self.kind == "port":
            fn = self.lib.orc_softras_backward
        else:
            fn = self.lib.ref_softras_backward
            ids = np.ascontiguousarray(ids.transpose(0, 2, 3, 1))            # SRW:108
        rc = fn(_fp(fv), _fp(tex), _fp(saved["soft_colors"]), _fp(saved["faces_info"]),
                _fp(saved["aggrs_info"]), _ip(ids), _fp(g), _fp(gf), _fp(gt), *common)
        if rc:
            raise RuntimeError("oracle backward failed rc=%d" % rc)
        return gf.reshape(B, NF, 3, 3), gt

    def backward_exactsum(self, saved, grad_soft_colors, nthreads=0):
        """The reference's FLOAT backward with its float atomics also summed in double (kind='reference' only) ->
        float64 (grad_faces, grad_textures) = the exact sum of the reference's float per-pair terms: its gradient
        without the order noise of the float atomics."""
        if self.kind != "reference":
            raise ValueError("the shadowed run exists for kind='reference' only")
        p, s = _scalars({k: v for k, v in saved["params"].items()})
        fv, tex = saved["face_vertices"], saved["textures"]
        B, NF, T = fv.shape[0], fv.shape[1], tex.shape[2]
        IS, K = int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        g = np.ascontiguousarray(grad_soft_colors, np.float32).reshape(B, 4, IS, IS)
        gf, gt = np.empty((B, NF, 9), np.float32), np.empty((B, NF, T, 3), np.float32)
        gf64, gt64 = np.empty((B, NF, 9), np.float64), np.empty((B, NF, T, 3), np.float64)
        ids = np.ascontiguousarray(saved["faces_id_buffer"].transpose(0, 2, 3, 1))            # SRW:108
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        rc = self.lib.ref_softras_backward_exactsum(
            _fp(fv), _fp(tex), _fp(saved["soft_colors"]), _fp(saved["faces_info"]), _fp(saved["aggrs_info"]), _ip(ids),
            _fp(g), _fp(gf), _fp(gt), dp(gf64), dp(gt64), B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"],
            s["dist"], s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"], int(nthreads))
        if rc:
            raise RuntimeError("oracle backward_exactsum failed rc=%d" % rc)
        return gf64.reshape(B, NF, 3, 3), gt64

    def backward_f64(self, saved, grad_soft_colors, nthreads=0):
        """The reference's backward kernel instantiated for DOUBLE on the same saved tensors (oracle/ref_driver.cpp:
        ref_softras_backward_f64; kind='reference' only) -> float64 (grad_faces [B,NF,3,3], grad_textures): the truth
        that the float instantiation and the HIP kernels are both measured against."""
        if self.kind != "reference":
            raise ValueError("the double instantiation exists for kind='reference' only")
        p, s = _scalars({k: v for k, v in saved["params"].items()})
        fv, tex = saved["face_vertices"], saved["textures"]
        B, NF, T = fv.shape[0], fv.shape[1], tex.shape[2]
        IS, K = int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        g = np.ascontiguousarray(grad_soft_colors, np.float32).reshape(B, 4, IS, IS)
        gf = np.empty((B, NF, 9), np.float64)
        gt = np.empty((B, NF, T, 3), np.float64)
        ids = np.ascontiguousarray(saved["faces_id_buffer"].transpose(0, 2, 3, 1))            # SRW:108
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        rc = self.lib.ref_softras_backward_f64(
            _fp(fv), _fp(tex), _fp(saved["soft_colors"]), _fp(saved["faces_info"]), _fp(saved["aggrs_info"]), _ip(ids),
            _fp(g), dp(gf), dp(gt), B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"],
            s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"], int(nthreads))
        if rc:
            raise RuntimeError("oracle backward_f64 failed rc=%d" % rc)
        return gf.reshape(B, NF, 3, 3), gt

    # ---- pixel-subset variants (port only): full-size parity checks in seconds ----
    def forward_subset(self, face_vertices, textures, pixels, **kw):
        """Oracle outputs for the listed global pixel indices (b*IS*IS + row*IS + col) only.
        -> dict(rgba [n,4], aggr [n,2], ids [n,K], faces_info [B,NF,27])."""
        if self.kind != "port":
            raise ValueError("subset evaluation exists for kind='port' only")
        p, s = _scalars(kw)
        fv = np.ascontiguousarray(face_vertices, np.float32)
        B, NF = fv.shape[:2]
        fv = fv.reshape(B, NF, 9)
        tex = np.ascontiguousarray(textures, np.float32).reshape(B, NF, -1, 3)
        T, IS, K = tex.shape[2], int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        pix = np.ascontiguousarray(pixels, np.int64)
        n = pix.size
        info = np.empty((B, NF, 27), np.float32)
        self.lib.orc_faces_info(_fp(fv), _fp(info), C.c_long(B * NF))
        rgba = np.empty((n, 4), np.float32)
        aggr = np.empty((n, 2), np.float32)
        ids = np.empty((n, K), np.int32)
        bg = kw.get("background_color")
        bgp = None if bg is None else _fp(np.ascontiguousarray(bg, np.float32))
        self.lib.orc_ub_reset()
        rc = self.lib.orc_softras_forward_subset(
            _fp(fv), _fp(tex), _fp(info), pix.ctypes.data_as(C.POINTER(C.c_int64)), C.c_long(n),
            _fp(rgba), _fp(aggr), _ip(ids), B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"],
            s["dist"], s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"], bgp,
            self.nthreads)
        if rc:
            raise RuntimeError("oracle forward_subset failed rc=%d" % rc)
        return dict(rgba=rgba, aggr=aggr, ids=ids, faces_info=info, params=p)

    def backward_subset(self, saved, grad_soft_colors, pixels):
        """Backward over the listed pixels only (== full backward when the gradient is zero elsewhere).
        ``saved`` holds full-layout arrays (e.g. downloaded from the GPU forward)."""
        if self.kind != "port":
            raise ValueError("subset evaluation exists for kind='port' only")
        p, s = _scalars({k: v for k, v in saved["params"].items()})
        fv, tex = saved["face_vertices"], saved["textures"]
        B, NF, T = fv.shape[0], fv.shape[1], tex.shape[2]
        IS, K = int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        g = np.ascontiguousarray(grad_soft_colors, np.float32).reshape(B, 4, IS, IS)
        pix = np.ascontiguousarray(pixels, np.int64)
        gf = np.empty((B, NF, 9), np.float32)
        gt = np.empty((B, NF, T, 3), np.float32)
        rc = self.lib.orc_softras_backward_subset(
            _fp(np.ascontiguousarray(fv, np.float32)), _fp(np.ascontiguousarray(tex, np.float32)),
            _fp(saved["soft_colors"]), _fp(saved["faces_info"]), _fp(saved["aggrs_info"]),
            _ip(saved["faces_id_buffer"]), _fp(g), pix.ctypes.data_as(C.POINTER(C.c_int64)),
            C.c_long(pix.size), _fp(gf), _fp(gt), B, NF, T, IS, K, s["near"], s["far"], s["eps"],
            s["sigma"], s["dist"], s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"])
        if rc:
            raise RuntimeError("oracle backward_subset failed rc=%d" % rc)
        return gf.reshape(B, NF, 3, 3), gt


class C2fOracle:
    """The reference's COARSE-TO-FINE forward (soft_rasterize_coarse_to_fine.py) compiled for the host, its coarse kernel launched
    as ONE thread so that every bin list is ascending by face id (oracle/ref_c2f_driver.cpp says why that launch is legal and why it
    is the only comparable one).  Only exists where oracle/_ref was built.  forward() takes the reference operator's keyword
    arguments plus bin_size / max_elems_per_bin (SRW:85-90: 0 -> num_faces / 5 as the reference's wrapper does)."""

    def __init__(self, nthreads=0):
        import importlib
        path = importlib.import_module(__name__ + ".build_ref").build_c2f()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("oracle/_ref/libsoftras_c2f_ref.so not built and /root/reference not mounted")
        self.lib = C.CDLL(path)
        self.nthreads = int(nthreads)

    def forward(self, face_vertices, textures, bin_size=16, max_elems_per_bin=0, **kw):
        p, s = _scalars(kw)
        fv = np.ascontiguousarray(face_vertices, np.float32)
        B, NF = fv.shape[:2]
        fv = fv.reshape(B, NF, 9)
        tex = np.ascontiguousarray(textures, np.float32).reshape(B, NF, -1, 3)
        T, IS, K = tex.shape[2], int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        M = int(max_elems_per_bin) if max_elems_per_bin else int(NF / 5)            # SRW:85-90
        bins = 1 + (IS - 1) // int(bin_size)
        info = np.empty((B, NF, 27), np.float32)
        aggr = np.empty((B, 2, IS, IS), np.float32)
        rgba = np.empty((B, 4, IS, IS), np.float32)
        ids = np.empty((B, K, IS, IS), np.int32)
        per_bin = np.empty((B, bins, bins), np.int32)
        elems = np.empty((B, bins, bins, max(M, 1)), np.int32)
        rc = self.lib.ref_softras_forward_c2f(_fp(fv), _fp(tex), _fp(info), _fp(aggr), _fp(rgba), _ip(ids),
                                              B, NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"],
                                              s["dist_eps"], s["gamma"], s["rgb"], s["alpha"], s["tex"], s["ds"],
                                              int(bin_size), M, _ip(per_bin), _ip(elems), self.nthreads)
        if rc:
            raise ValueError("reference coarse-to-fine forward refused the geometry (more than 27 bins per edge, C2F:16-18) rc=%d" % rc)
        return dict(face_vertices=fv, textures=tex, soft_colors=rgba, faces_info=info, aggrs_info=aggr, faces_id_buffer=ids,
                    elems_per_bin=per_bin, bin_elems=elems, max_elems_per_bin=M, params=p)


class N3mrOracle:
    """The reference's NMR kernels compiled for the host (oracle/_ref/libn3mr_ref.so, serial) or their
    plain-C restatement (oracle/n3mr_oracle.c) behind the same entry points.
    Layouts are the reference's op-level ones (NHWC, bottom-up rows); background compositing and
    alpha are the host ops of n3mr.py:135-148, restated in NumPy."""

    def __init__(self, kind=None):
        """kind='reference': the reference's own kernels compiled for the host (oracle/_ref/libn3mr_ref.so; a
        fixed list of image sizes is instantiated); kind='port': oracle/n3mr_oracle.c (any size, builds with
        gcc alone); None: the reference build when it exists or can be built, else the port."""
        import importlib
        path = None
        if kind in (None, "reference"):
            path = importlib.import_module(__name__ + ".build_ref").build_n3mr()
            if (path is None or not os.path.exists(path)) and kind == "reference":
                raise FileNotFoundError("oracle/_ref/libn3mr_ref.so not built and /root/reference not mounted")
        self.kind = "reference"
        if path is None or not os.path.exists(path):
            path = make_n3mr_port()
            self.kind = "port"
        self.lib = C.CDLL(path)

    def forward(self, faces, textures=None, image_size=256, near=0.1, far=100, eps=1e-3,
                background_color=(0, 0, 0), return_rgb=True, return_alpha=True, return_depth=True):
        f = np.ascontiguousarray(faces, np.float32)
        B, NF = f.shape[:2]
        f = f.reshape(B, NF, 9)
        IS = int(image_size)
        tex = np.ascontiguousarray(textures, np.float32) if return_rgb else np.zeros(1, np.float32)
        TS = tex.shape[2] if return_rgb else 0
        faces_inv = np.zeros((B, NF, 9), np.float32)
        fim = np.empty((B, IS, IS), np.int32)
        wm = np.empty((B, IS, IS, 3), np.float32)
        dm = np.empty((B, IS, IS), np.float32)
        fivm = np.zeros((B, IS, IS, 9), np.float32)
        rgb = np.zeros((B, IS, IS, 3), np.float32)
        sidx = np.zeros((B, IS, IS, 8), np.int32)
        swt = np.zeros((B, IS, IS, 8), np.float32)
        f32 = lambda v: C.c_float(float(np.float32(v)))
        rc = self.lib.ref_n3mr_forward(_fp(f), _fp(tex), _fp(faces_inv), _ip(fim), _fp(wm), _fp(dm), _fp(fivm),
                                       _fp(rgb), _ip(sidx), _fp(swt), B, NF, TS, IS, f32(near), f32(far), f32(eps),
                                       int(return_rgb), int(return_depth))
        if rc:
            raise RuntimeError("n3mr reference forward failed rc=%d (image size not instantiated?)" % rc)
        mask = (fim >= 0).astype(np.float32)
        if return_rgb:                                                                       # n3mr.py:135-143
            bg = np.asarray(background_color, np.float32)
            rgb = rgb * mask[..., None] + (1 - mask[..., None]) * bg[None, None, None]
        return dict(faces=f, textures=tex, face_index_map=fim, weight_map=wm, depth_map=dm, face_inv_map=fivm,
                    rgb_map=rgb.astype(np.float32), alpha_map=mask, sampling_index_map=sidx,
                    sampling_weight_map=swt, faces_inv=faces_inv,
                    params=dict(image_size=IS, eps=eps, return_rgb=return_rgb, return_alpha=return_alpha,
                                return_depth=return_depth, TS=TS))

    def backward(self, s, grad_rgb=None, grad_alpha=None, grad_depth=None):
        p = s["params"]
        f = s["faces"]
        B, NF = f.shape[:2]
        IS, TS = p["image_size"], p["TS"]
        z = lambda like: np.zeros(like.shape, np.float32)
        g_rgb = np.ascontiguousarray(grad_rgb, np.float32) if grad_rgb is not None else z(s["rgb_map"])
        g_a = np.ascontiguousarray(grad_alpha, np.float32) if grad_alpha is not None else z(s["alpha_map"])
        g_d = np.ascontiguousarray(grad_depth, np.float32) if grad_depth is not None else z(s["depth_map"])
        gf = np.empty((B, NF, 9), np.float32)
        gt = np.zeros((B, NF, max(TS, 1), max(TS, 1), max(TS, 1), 3), np.float32)
        f32 = lambda v: C.c_float(float(np.float32(v)))
        rc = self.lib.ref_n3mr_backward(_fp(f), _ip(s["face_index_map"]), _fp(s["weight_map"]), _fp(s["depth_map"]),
                                        _fp(s["face_inv_map"]), _fp(np.ascontiguousarray(s["rgb_map"])),
                                        _fp(s["alpha_map"]), _fp(s["sampling_weight_map"]),
                                        _ip(s["sampling_index_map"]), _fp(g_rgb), _fp(g_a), _fp(g_d), _fp(gf),
                                        _fp(gt), B, NF, TS, IS, f32(p["eps"]), int(p["return_rgb"]),
                                        int(p["return_alpha"]), int(p["return_depth"]))
        if rc:
            raise RuntimeError("n3mr reference backward failed rc=%d" % rc)
        return gf.reshape(B, NF, 3, 3), gt


class TexturesOracle:
    """The reference's texel-sampler kernels compiled for the host (oracle/_ref/libtextures_ref.so;
    jrender/io/utils/load_textures.py:11-69 and :103-219).  Only exists where oracle/_ref was built."""

    def __init__(self):
        import importlib
        path = importlib.import_module(__name__ + ".build_ref").build_textures()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("oracle/_ref/libtextures_ref.so not built and /root/reference not mounted")
        self.lib = C.CDLL(path)

    def softras(self, image, faces, textures, is_update):
        img = np.ascontiguousarray(image, np.float32)
        f = np.ascontiguousarray(faces, np.float32)
        out = np.array(textures, np.float32, copy=True)
        upd = np.ascontiguousarray(is_update, np.int32)
        NF, RR = out.shape[:2]
        self.lib.ref_load_textures_softras(_fp(img), _fp(f), _ip(upd), _fp(out), NF, int(round(np.sqrt(RR))),
                                           img.shape[0], img.shape[1])
        return out

    def n3mr(self, image, faces, textures, is_update, texture_wrapping=0, use_bilinear=True):
        img = np.ascontiguousarray(image, np.float32)
        f = np.ascontiguousarray(faces, np.float32)
        out = np.array(textures, np.float32, copy=True)
        upd = np.ascontiguousarray(is_update, np.int32)
        rc = self.lib.ref_load_textures_n3mr(_fp(img), _fp(f), _ip(upd), _fp(out), out.shape[0], out.shape[1],
                                             img.shape[0], img.shape[1], int(texture_wrapping), int(bool(use_bilinear)))
        if rc:
            raise RuntimeError("unknown (texture_wrapping, use_bilinear) pair")
        return out
