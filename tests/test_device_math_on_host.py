"""The HIP kernels' per-pair arithmetic, compiled for the HOST and compared with the oracle pair by pair (CPU only).

jrender_amd/csrc/softras_device.h holds what the raster kernels compute per (pixel, face): face set-up, the packed record, barycentrics,
the distance tree, clipping, depth, texel choice, coverage.  tests/host_math/ compiles that very header with g++ under a 30-line shim
(no code of it is restated) and tests/host_math/harness.cpp runs every face against every pixel centre of its widened box, next to the
oracle's own functions: the bit-critical quantities - everything the face-index buffer depends on - must be THE SAME BITS, in the plain
IEEE instantiation and in the reciprocal-refinement one the kernels run on well-conditioned faces.  The GPU suite shows this through
whole kernels; here it holds without a GPU, on millions of pairs per second, and under AddressSanitizer + UBSan (this pool has no GPU
sanitizer).  Not comparable on the host and not compared: paths built on the device's approximate v_rcp_f32 / v_exp_f32 (the default
colour path; the inside pairs' second and third projection outside 'hard' alpha)."""
import ctypes as C
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

from jrender_amd import synthetic as syn

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_math")
NAMES = ["pairs", "ub_skipped", "unsafe_faces", "faces_info", "border", "w", "sign", "dis_ieee", "dis_refined", "bary_dist", "w_clip",
         "zp_ieee", "zp_refined", "coverage_ref_form", "texel", "split_vs_joint", "bwd_vs_fwd_form", "cull_decision"]
MISMATCH = NAMES[3:]
FLAGS = ["-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-w"]            # the device build's floating-point flags (jrender_amd/_build.py)
DIST_EPS_LOG = float(np.float32(np.log(1.0 / 1e-4 - 1.0)))                    # SRW:25

pytestmark = pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")


def build(out_dir, extra=(), exe=False):
    objs = []
    for src, cc, std in (("oracle_wrap.c", "gcc", "-std=gnu11"), ("harness.cpp", "g++", "-std=c++17")) + ((("main.cpp", "g++", "-std=c++17"),) if exe else ()):
        o = os.path.join(out_dir, src + ".o")
        subprocess.check_call([cc, std, *FLAGS, *extra, "-I", os.path.join(HERE, "shim"), "-c", os.path.join(HERE, src), "-o", o])
        objs.append(o)
    out = os.path.join(out_dir, "hm_main" if exe else "libhm.so")
    subprocess.check_call(["g++", *(() if exe else ("-shared",)), *extra, *objs, "-o", out, "-lm"])
    return out


def build_hip_for_the_host(name, out_dir, extra=(), load=True):
    """tests/host_math/<name>.hip -> shared object with the HOST side only.  The file includes a kernel source of the product with its
    device functions also built for the host; compiling the device side too would cost minutes (the forward: 2 min 16 s) for code this
    test never launches.  --cuda-host-only leaves ONE undefined symbol, the embedded device binary the object registers when it is
    loaded: an empty one (a HIP file without kernels, hipcc --genco) is supplied under that name."""
    hipcc = "/opt/rocm/bin/hipcc"
    root = os.path.join(os.path.dirname(HERE), "..")
    obj = os.path.join(out_dir, name + ".o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "--cuda-host-only", "-cuid=jr_" + name, "-O2", "-std=c++17", "-ffp-contract=off",
                           "-fno-fast-math", "-fPIC", "-w", *extra, "-I", os.path.join(root, "include"), "-I", os.path.join(root, "jrender_amd", "csrc"),
                           "-c", os.path.join(HERE, name + ".hip"), "-o", obj])
    syms = [l.split()[-1] for l in subprocess.check_output(["nm", obj], text=True).splitlines() if " U __hip_fatbin_" in l]
    assert len(syms) == 1, syms
    empty = os.path.join(out_dir, "empty.hip")
    with open(empty, "w") as f:
        f.write("// no kernels\n")
    fb = os.path.join(out_dir, "empty.hipfb")
    subprocess.check_call([hipcc, "--genco", "--offload-arch=gfx950", empty, "-o", fb])
    asm = os.path.join(out_dir, name + "_fatbin.S")
    with open(asm, "w") as f:
        f.write('.section .hip_fatbin,"a",@progbits\n.globl %s\n.p2align 12\n%s:\n.incbin "%s"\n' % (syms[0], syms[0], fb))
    fbo = os.path.join(out_dir, name + "_fatbin.o")
    subprocess.check_call(["gcc", "-c", asm, "-o", fbo])
    out = os.path.join(out_dir, "lib" + name + ".so")
    subprocess.check_call([hipcc, "-shared", obj, fbo, "-o", out])
    if not load:
        return out
    try:
        return C.CDLL(out)
    except OSError as e:                                   # (the HIP runtime library the object links against does not load on this host)
        pytest.skip(str(e))


@pytest.fixture(scope="module")
def workdir():
    d = tempfile.mkdtemp(prefix="jr_hm_")
    yield d
    shutil.rmtree(d, ignore_errors=True)


@pytest.fixture(scope="module")
def lib(workdir):
    return C.CDLL(build(workdir))


def compare(lib, faces, IS, sigma=1e-5, max_px=48):
    f = np.ascontiguousarray(np.asarray(faces, np.float32).reshape(-1, 9))
    cnt = (C.c_long * 24)()
    lib.hm_compare(f.ctypes.data_as(C.POINTER(C.c_float)), C.c_long(f.shape[0]), int(IS), C.c_float(sigma), C.c_float(DIST_EPS_LOG), int(max_px), cnt)
    return dict(zip(NAMES, list(cnt)[:len(NAMES)]))


def hostile_faces(n, seed):
    """slivers, zero-area faces, repeated vertices, coordinates far outside the screen (records that are not FLAG_SAFE), depths from
    1e-3 to 1e3, a few NaN / inf coordinates - the inputs the kernels route through their plain IEEE instantiation"""
    rng = np.random.default_rng(seed)
    f = np.empty((n, 3, 3), np.float32)
    c = rng.uniform(-0.9, 0.9, (n, 1, 2))
    f[:, :, :2] = c + rng.uniform(-0.05, 0.05, (n, 3, 2))
    f[:, :, 2] = 10.0 ** rng.uniform(-3, 3, (n, 3))
    k = n // 8
    f[:k, 2, :2] = 0.5 * (f[:k, 0, :2] + f[:k, 1, :2]) + rng.uniform(-1e-6, 1e-6, (k, 2))      # slivers
    f[k:2 * k, 2, :2] = f[k:2 * k, 1, :2]                                                       # repeated vertex
    f[2 * k:3 * k, :, :2] *= 0.0                                                               # a point at the origin
    f[3 * k:4 * k, :, :2] = rng.uniform(-3e3, 3e3, (k, 3, 2))                                    # beyond the fast-division range of the record
    f[4 * k:5 * k, 0, :2] += rng.uniform(-2, 2, (k, 2))                                         # long thin faces across the screen
    f[5 * k, 0, 0] = np.nan; f[5 * k + 1, 1, 1] = np.inf; f[5 * k + 2, 2, 2] = 0.0; f[5 * k + 3, :, 2] = np.nan
    return f


SCENES = {
    "sphere3300_two_views_256": lambda: (syn.sphere_views(3300, 2)[0], 256, 1e-5),
    "sphere39000_part_1024": lambda: (syn.sphere_views(39000, 1)[0][:, ::13], 1024, 1e-5),
    "soup2000_128_sigma1e-4": lambda: (syn.triangle_soup(2000, 1, seed=1, scale=3.0)[0], 128, 1e-4),
    "soup500_tiny_17": lambda: (syn.triangle_soup(500, 1, seed=2, scale=4.0)[0], 17, 1e-6),
    "hostile_96": lambda: (hostile_faces(4000, 7), 96, 3e-5),
}


@pytest.mark.parametrize("scene", sorted(SCENES))
def test_device_header_matches_the_oracle_pair_by_pair(lib, scene):
    faces, IS, sigma = SCENES[scene]()
    r = compare(lib, faces, IS, sigma)
    assert r["pairs"] > 10000, r
    assert all(r[k] == 0 for k in MISMATCH), {k: v for k, v in r.items() if v}
    if scene.startswith("hostile"):
        assert r["unsafe_faces"] > 400                      # the plain-division instantiation was really exercised on records of its own


def test_the_refinement_quotients_lean_on_the_hardware_reciprocal(workdir):
    """recip_exact = v_rcp_f32 + ONE Newton step is the IEEE reciprocal because of what v_rcp_f32 returns on gfx950 (checked for every
    float on the device: jr_selftest_reciprocal, tests/test_gpu_parity.py::test_fast_division_identity) - it is NOT a property of any
    reciprocal that is 1 ulp off.  With an adversarial neighbour as the seed a handful of depths per million differ by an ulp; nothing
    else moves (the distance tree and the decisions never touch the approximate reciprocal)."""
    d = os.path.join(workdir, "ulp")
    os.makedirs(d, exist_ok=True)
    lib1 = C.CDLL(build(d, extra=("-DHM_RCP_ULP_OFF=1",)))
    faces, IS, sigma = SCENES["sphere3300_two_views_256"]()
    r = compare(lib1, faces, IS, sigma)
    assert all(r[k] == 0 for k in MISMATCH if k != "zp_refined"), {k: v for k, v in r.items() if v}
    assert r["zp_refined"] <= 2e-5 * r["pairs"]


def test_under_address_and_undefined_behaviour_sanitizers(workdir):
    """The same comparison as a stand-alone executable built with -fsanitize=address,undefined -fno-sanitize-recover: out-of-bounds
    indexing (edge / vertex tables of the record), signed overflow, invalid float -> int conversions and shifts in the per-pair
    arithmetic would abort it.  (GPU AddressSanitizer is not available on this pool; this covers the arithmetic header, not the kernels'
    memory traffic.)"""
    d = os.path.join(workdir, "san")
    os.makedirs(d, exist_ok=True)
    try:
        exe = build(d, extra=("-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"), exe=True)
    except subprocess.CalledProcessError:
        pytest.skip("the sanitizer runtimes are not installed")
    for name in ("hostile_96", "soup500_tiny_17"):
        faces, IS, sigma = SCENES[name]()
        if name == "hostile_96":
            faces = faces[np.isfinite(faces).all((1, 2))]       # (float -> int of NaN in the texel choice is the reference's own undefined corner: SRK:159-166)
        path = os.path.join(d, name + ".bin")
        np.ascontiguousarray(np.asarray(faces, np.float32).reshape(-1, 9)).tofile(path)
        out = subprocess.run([exe, path, str(IS), repr(sigma), repr(DIST_EPS_LOG), "48"], capture_output=True, text=True, timeout=600,
                             env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert out.returncode == 0, out.stderr[-3000:]
        r = dict(zip(NAMES, [int(x) for x in out.stdout.split()]))
        assert r["pairs"] > 10000 and all(r[k] == 0 for k in MISMATCH), r


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_pixel_range_of_the_list_building_is_exact(workdir):
    """jr::pixel_range (binning.hip) on the host - tests/host_math/binning_on_host.hip builds that file with its device functions compiled
    for the host as well - against its definition: the pixel centres that pass the reference's border test on one axis.  Exactness is what
    lets the forward skip its own box test and keeps faces that touch no pixel centre out of the lists.  Bounds are pixel centres moved by
    0 - 3 ulps, uniform values, values far outside the image, infinities and NaN; every image size from 1 to 80 and the usual larger ones."""
    lib = build_hip_for_the_host("binning_on_host", workdir)
    lib.hm_pixel_range_check.restype = C.c_long
    bad = (C.c_float * 4)()
    total = 0
    for IS in list(range(1, 81)) + [96, 100, 127, 128, 255, 256, 511, 512, 1000, 1024, 2048, 4096]:
        n = 20000 if IS <= 256 else 4000
        wrong = lib.hm_pixel_range_check(IS, C.c_long(n), C.c_ulonglong(2024), bad)
        assert wrong == 0, (IS, wrong, list(bad))
        total += n
    assert total > 1.5e6


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_backward_pair_arithmetic_on_the_host_against_the_reference_s_exact_sums(workdir):
    """jr::backward_pair (softras_backward.hip: the gradient of one (pixel, face) pair - all the backward kernel computes between fetching
    a pair and reducing its components) built for the host by tests/host_math/backward_on_host.hip and run over whole images in the order
    of the saved index buffer, its float terms summed in double.  Held to the exact (double) sum of the REFERENCE's float per-pair terms
    (oracle/_ref: ref_softras_backward_exactsum) in all 18 distance x colour x alpha modes with one texel, 2 x 2 texels and vertex colours:
    what remains is per-pair arithmetic only - no atomic order, no kernel organisation.  Measured 7e-7 of the largest component,
    1.1e-4 element-wise (floor 1e-3): the gradient-only reciprocal multiplies."""
    import itertools
    from oracle import Oracle, _scalars, have_ref
    if not have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = build_hip_for_the_host("backward_on_host", workdir)
    ref = Oracle("reference", nthreads=0)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))

    def host_backward(saved, g):
        p, s = _scalars(dict(saved["params"]))
        fv, tex = saved["face_vertices"], saved["textures"]
        NF, T = fv.shape[1], tex.shape[2]
        IS, K = int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        gf, gt, st = np.empty((NF, 9), np.float64), np.empty((NF, T, 3), np.float64), (C.c_long * 4)()
        rc = lib.hm_backward_image(fp(fv), fp(tex), fp(saved["soft_colors"]), fp(saved["aggrs_info"]),
                                   saved["faces_id_buffer"].ctypes.data_as(C.POINTER(C.c_int32)), fp(np.ascontiguousarray(g, np.float32)),
                                   NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"], s["dist_eps"], s["gamma"], s["rgb"], s["alpha"],
                                   s["tex"], s["ds"], dp(gf), dp(gt), st)
        assert rc == 0
        return gf.reshape(1, NF, 3, 3), gt.reshape(1, NF, T, 3)

    def errors(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        assert (np.isfinite(a) == np.isfinite(b)).all()
        m = np.isfinite(b)
        if not m.any() or np.abs(b[m]).max() == 0:
            return 0.0, 0.0
        top = np.abs(b[m]).max()
        d = np.abs(a[m] - b[m])
        return float(d.max() / top), float((d / (np.abs(b[m]) + 1e-3 * top)).max())

    soup = syn.triangle_soup(300, 1, seed=3, texels=4, scale=5.0)
    soup[0][..., :2] *= 0.55
    scenes = {"surface4": soup, "surface1": syn.sphere_views(280, 1), "vertex": syn.sphere_views(280, 1, texels=3)}
    worst = [0.0, 0.0]
    for dist, rgb, alpha, tt in itertools.product(["hard", "barycentric", "euclidean"], ["hard", "softmax"], ["hard", "sum", "prod"], sorted(scenes)):
        f, t = scenes[tt]
        kw = dict(image_size=40, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha, texture_type="vertex" if tt == "vertex" else "surface",
                  sigma_val=1e-4, max_faces_per_pixel_for_grad=6)
        s = ref.forward(f, t, **kw)
        g = np.random.default_rng(1).uniform(-1, 1, s["soft_colors"].shape).astype(np.float32)
        rf, rt = ref.backward_exactsum(s, g)
        hf, ht = host_backward(s, g)
        for name, (mx, el) in (("grad_faces", errors(hf, rf)), ("grad_textures", errors(ht, rt))):
            assert mx <= 5e-6 and el <= 1e-3, (kw, tt, name, mx, el)
            worst = [max(worst[0], mx), max(worst[1], el)]
    assert worst[0] > 0                                      # (float terms against their exact sum: not literally the same computation)
    # randomised cases of the GPU sweep's generator (tests/fuzz_parity.py), one view each; non-finite gradients (overflowed weights poison
    # a face's texels in the reference, SRK:1317-1320) must appear in the same places
    from tests.fuzz_parity import draw_case
    rng = np.random.default_rng(7)
    done = 0
    while done < 80:
        kind, f, t, kw = draw_case(rng)
        if f.shape[1] * kw["image_size"] ** 2 > 1e7:
            continue
        s = ref.forward(f[:1], t[:1], **kw)
        g = rng.uniform(-1, 1, s["soft_colors"].shape).astype(np.float32)
        rf, rt = ref.backward_exactsum(s, g)
        hf, ht = host_backward(s, g)
        for name, (mx, el) in (("grad_faces", errors(hf, rf)), ("grad_textures", errors(ht, rt))):
            assert mx <= 2e-5 and el <= 1e-3, (kind, kw, name, mx, el)
        done += 1


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_forward_state_machine_on_the_host_gives_the_oracle_s_index_buffer(workdir):
    """jr::forward_pair + init_colour_state + KBuffer + final_colour (softras_forward.hip: the complete per-pixel state machine of the forward
    kernels - distance, cull, coverage, alpha, depth cull, K-nearest insert with its id store, online softmax / hard colour) built for the
    host by tests/host_math/forward_on_host.hip and run for every pixel of an image over the faces in ascending order, with the kernels' own
    choice of instantiation.  Against the oracle in all 108 distance x colour x alpha x culling x texture combinations: faces_info and the
    face-index buffer BIT FOR BIT, colours and aggregates within the colour tolerance (measured 2 % of it: the host's exp2f and 1/x are
    exact where the device's are approximations).  What the GPU suite shows through whole kernels holds here without a GPU for the arithmetic
    and the slot semantics; the wavefront organisation around it (lists, staging, ballots) is the GPU suite's."""
    import itertools
    from oracle import Oracle, _scalars
    from tests.util import RGBA_ATOL, bits_equal, rel_err
    lib = build_hip_for_the_host("forward_on_host", workdir)
    port = Oracle("port", nthreads=0)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))

    def host_forward(fv, tex, **kw):
        p, s = _scalars(kw)
        fv = np.ascontiguousarray(fv, np.float32)
        NF = fv.shape[1]
        fv = fv.reshape(1, NF, 9)
        tex = np.ascontiguousarray(tex, np.float32).reshape(1, NF, -1, 3)
        T, IS, K = tex.shape[2], int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
        info, aggr = np.empty((1, NF, 27), np.float32), np.empty((1, 2, IS, IS), np.float32)
        rgba, ids = np.empty((1, 4, IS, IS), np.float32), np.empty((1, K, IS, IS), np.int32)
        rc = lib.hm_forward_image(fp(fv), fp(tex), NF, T, IS, K, s["near"], s["far"], s["eps"], s["sigma"], s["dist"], s["dist_eps"], s["gamma"],
                                  s["rgb"], s["alpha"], s["tex"], s["ds"], fp(info), fp(aggr), fp(rgba), ids.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0
        return dict(faces_info=info, aggrs_info=aggr, soft_colors=rgba, faces_id_buffer=ids)

    soup = syn.triangle_soup(300, 1, seed=3, texels=4, scale=5.0)
    soup[0][..., :2] *= 0.55
    scenes = {"surface4": soup, "surface1": syn.sphere_views(280, 1), "vertex": syn.sphere_views(280, 1, texels=3)}
    # randomised cases of the GPU sweep's own generator (tests/fuzz_parity.py: scene kind, modes, K up to 64, sigma down to 1e-6, gamma,
    # near / far, texture layout), one view each, capped in size for the CPU
    from tests.fuzz_parity import draw_case
    rng = np.random.default_rng(20261001)
    fuzzed = 0
    while fuzzed < 250:
        kind, f, t, kw = draw_case(rng)
        if f.shape[1] * kw["image_size"] ** 2 > 2e7:
            continue
        f, t = f[:1], t[:1]
        a = port.forward(f, t, **kw)
        if port.ub_events():
            continue
        b = host_forward(f, t, **kw)
        assert bits_equal(a["faces_info"], b["faces_info"]), (kind, kw)
        assert bits_equal(a["faces_id_buffer"], b["faces_id_buffer"]), (kind, kw, int((a["faces_id_buffer"] != b["faces_id_buffer"]).any(1).sum()))
        assert rel_err(b["soft_colors"], a["soft_colors"], RGBA_ATOL) <= 1.0 and rel_err(b["aggrs_info"], a["aggrs_info"], RGBA_ATOL) <= 1.0, (kind, kw)
        fuzzed += 1
    compared, worst = 0, 0.0
    for dist, rgb, alpha, tt, fb in itertools.product(["hard", "barycentric", "euclidean"], ["hard", "softmax"], ["hard", "sum", "prod"], sorted(scenes), [True, False]):
        f, t = scenes[tt]
        kw = dict(image_size=40, dist_func=dist, aggr_func_rgb=rgb, aggr_func_alpha=alpha, texture_type="vertex" if tt == "vertex" else "surface",
                  sigma_val=1e-4, max_faces_per_pixel_for_grad=6, fill_back=fb)
        a = port.forward(f, t, **kw)
        if port.ub_events():
            continue                                        # the reference's undefined corner (SRK:107-121)
        b = host_forward(f, t, **kw)
        assert bits_equal(a["faces_info"], b["faces_info"]), kw
        assert bits_equal(a["faces_id_buffer"], b["faces_id_buffer"]), (kw, tt, int((a["faces_id_buffer"] != b["faces_id_buffer"]).any(1).sum()))
        e = max(rel_err(b["soft_colors"], a["soft_colors"], RGBA_ATOL), rel_err(b["aggrs_info"], a["aggrs_info"], RGBA_ATOL))
        assert e <= 1.0, (kw, tt, e)
        compared, worst = compared + 1, max(worst, e)
    assert compared >= 100
    # the committed vectors: written by the reference's kernels (bin_size = 0 and coarse-to-fine), and the inputs of the pinned last-bit corners
    import glob
    import json
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    seen = 0
    for path in sorted(glob.glob(os.path.join(gold, "*.npz"))):
        z = np.load(path)
        if "params" in z and "face_vertices" in z:
            kw = {k: v for k, v in json.loads(str(z["params"])).items() if k not in ("bin_size", "max_elems_per_bin")}
            for v in range(z["face_vertices"].shape[0]):
                b = host_forward(z["face_vertices"][v:v + 1], z["textures"][v:v + 1], **kw)
                assert bits_equal(b["faces_id_buffer"], z["faces_id_buffer"][v:v + 1]) and bits_equal(b["faces_info"], z["faces_info"][v:v + 1]), path
                assert rel_err(b["soft_colors"], z["soft_colors"][v:v + 1], RGBA_ATOL) <= 1.0, path
            seen += 1
        elif os.path.basename(path).startswith("regress_") and "kw" in z:
            import ast
            kw = ast.literal_eval(str(z["kw"]))             # (written as a Python dict literal by the sweep)
            for v in range(z["fv"].shape[0]):
                a = port.forward(z["fv"][v:v + 1], z["tex"][v:v + 1], **kw)
                b = host_forward(z["fv"][v:v + 1], z["tex"][v:v + 1], **kw)
                assert bits_equal(a["faces_id_buffer"], b["faces_id_buffer"]) and rel_err(b["soft_colors"], a["soft_colors"], RGBA_ATOL) <= 1.0, path
            seen += 1
    assert seen >= 12
    # a crowded scene at the operator's defaults: sigma 1e-5, K = 16 with replacements, two bigger images
    for (f, t), IS in ((syn.sphere_views(3300, 1), 128), (soup, 96)):
        a, b = port.forward(f, t, image_size=IS), host_forward(f, t, image_size=IS)
        if not port.ub_events():
            assert bits_equal(a["faces_id_buffer"], b["faces_id_buffer"]) and rel_err(b["soft_colors"], a["soft_colors"], RGBA_ATOL) <= 1.0


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_nmr_forward_face_and_pixel_functions_on_the_host(workdir):
    """The second renderer path (dr_type='n3mr'): n3_face_inv and n3_pixel of n3mr_kernels.hip - face set-up, coverage test,
    clamped weights and depth of one (face, pixel) pair (N3K:63-134) - built for the host by tests/host_math/n3mr_on_host.hip and driven by
    a serial form of k_n3mr_zbuffer (depth test = minimum of the packed (depth bits, face index) key).  Against the reference's own NMR
    kernels compiled for the host: faces_inv, the face-index map, the depth map and the winner's weights, bit for bit."""
    from oracle import N3mrOracle
    from tests.util import bits_equal
    try:
        orc = N3mrOracle("reference")
    except FileNotFoundError:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    lib = build_hip_for_the_host("n3mr_on_host", workdir)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    back = syn.sphere_views(280, 1)[0][:, :, ::-1].copy()                                        # reversed winding: every face of the near side is a back side
    scenes = {"sphere280_64": (syn.sphere_views(280, 1)[0], 64), "soup800_96": (syn.triangle_soup(800, 1, seed=5, scale=3.0)[0], 96),
              "sphere3300_128": (syn.sphere_views(3300, 1)[0], 128), "sphere280_reversed_48": (back, 48),
              "hostile_64": (hostile_faces(1500, 11)[np.isfinite(hostile_faces(1500, 11)).all((1, 2))][None], 64)}
    for name, (fv, IS) in scenes.items():
        f = np.ascontiguousarray(np.asarray(fv, np.float32).reshape(1, -1, 9))
        NF = f.shape[1]
        a = orc.forward(f.reshape(1, NF, 3, 3), None, image_size=IS, near=0.1, far=100, return_rgb=False, return_alpha=True, return_depth=True)
        finv, fim = np.empty((1, NF, 9), np.float32), np.empty((1, IS, IS), np.int32)
        dm, wm = np.empty((1, IS, IS), np.float32), np.empty((1, IS, IS, 3), np.float32)
        assert lib.hm_n3mr_zbuffer(fp(f), NF, IS, C.c_float(0.1), C.c_float(100.0), fp(finv), fim.ctypes.data_as(C.POINTER(C.c_int32)), fp(dm), fp(wm)) == 0
        for key, mine in (("faces_inv", finv), ("face_index_map", fim), ("depth_map", dm), ("weight_map", wm)):
            assert bits_equal(mine, a[key]), (name, key)
        if "reversed" not in name and "hostile" not in name:
            assert (fim >= 0).sum() > 1000


UBSAN_DRIVER = r"""
import ctypes as C, sys
import numpy as np
sys.path.insert(0, sys.argv[3])
from oracle import Oracle, _scalars
from tests.fuzz_parity import draw_case
fwd, bwd = C.CDLL(sys.argv[1]), C.CDLL(sys.argv[2])
port = Oracle("port", nthreads=0)
fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
rng = np.random.default_rng(99)
done = 0
while done < 60:
    kind, f, t, kw = draw_case(rng)
    if f.shape[1] * kw["image_size"] ** 2 > 6e6:
        continue
    s = port.forward(f[:1], t[:1], **kw)
    p, sc = _scalars(kw)
    fv, tex = s["face_vertices"], s["textures"]
    NF, T, IS, K = fv.shape[1], tex.shape[2], int(p["image_size"]), int(p["max_faces_per_pixel_for_grad"])
    common = (sc["near"], sc["far"], sc["eps"], sc["sigma"], sc["dist"], sc["dist_eps"], sc["gamma"], sc["rgb"], sc["alpha"], sc["tex"], sc["ds"])
    info, aggr = np.empty((1, NF, 27), np.float32), np.empty((1, 2, IS, IS), np.float32)
    rgba, ids = np.empty((1, 4, IS, IS), np.float32), np.empty((1, K, IS, IS), np.int32)
    assert fwd.hm_forward_image(fp(fv), fp(tex), NF, T, IS, K, *common, fp(info), fp(aggr), fp(rgba), ids.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    assert (ids == s["faces_id_buffer"]).all()
    g = rng.uniform(-1, 1, rgba.shape).astype(np.float32)
    gf, gt, st = np.empty((NF, 9), np.float64), np.empty((NF, T, 3), np.float64), (C.c_long * 4)()
    assert bwd.hm_backward_image(fp(fv), fp(tex), fp(rgba), fp(aggr), ids.ctypes.data_as(C.POINTER(C.c_int32)), fp(g), NF, T, IS, K, *common, dp(gf), dp(gt), st) == 0
    done += 1
print("UBSAN_OK", done)
"""


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_forward_and_backward_pair_code_under_trapping_ubsan(workdir):
    """The host builds of the forward state machine and the backward pair function once more with -fsanitize=undefined -fsanitize-trap=all
    (undefined behaviour executes a trap instruction: no sanitizer runtime to preload into python) over 60 randomised cases in a child
    process: signed overflow, out-of-range shifts, misaligned or null accesses and out-of-bounds indexing of the K-buffer / record tables
    in the kernels' per-pair code would kill the child."""
    import sys
    d = os.path.join(workdir, "ubsan")
    os.makedirs(d, exist_ok=True)
    flags = ("-g", "-fsanitize=undefined", "-fsanitize-trap=all", "-fno-sanitize=float-divide-by-zero")
    try:
        fwd = build_hip_for_the_host("forward_on_host", d, extra=flags, load=False)
        bwd = build_hip_for_the_host("backward_on_host", d, extra=flags, load=False)
    except subprocess.CalledProcessError:
        pytest.skip("this hipcc does not build with -fsanitize=undefined")
    root = os.path.abspath(os.path.join(os.path.dirname(HERE), ".."))
    out = subprocess.run([sys.executable, "-c", UBSAN_DRIVER, fwd, bwd, root], capture_output=True, text=True, timeout=900)
    if out.returncode != 0 and "cannot open shared object" in out.stderr:
        pytest.skip(out.stderr[-300:])
    assert out.returncode == 0 and "UBSAN_OK 60" in out.stdout, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
