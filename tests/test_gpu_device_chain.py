"""The callers either side of the op, device-resident (BASELINE configs[3]).

The reference runs camera transform, face gather, loss and their backward as Jittor tensor ops on the GPU
(transform/look_at.py:3-39, look.py:3-54, perspective.py:4-17, orthogonal.py:3-16, structures/utils/faces_vertices.py:4-19,
loss/iou_loss.py:1-9).  The NumPy mirrors of those files are pinned to the reference's own Python
(tests/test_host_reference.py); the HIP kernels behind `DeviceArray` vertices are held to the mirrors here:
element-wise float32 steps to a few ulp, sums to 1e-6 of their scale, and the whole demo2 chain
(camera -> gather -> SoftRas -> IoU -> SoftRas backward -> scatter -> camera VJP) to the host chain's loss curve.
"""
import importlib.util
import os

import numpy as np
import pytest

import jrender_amd as jr
from jrender_amd import _ffi
from jrender_amd.renderer import transform as T
from jrender_amd.structures.mesh import face_vertices, face_vertices_backward

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    return _ffi.Context.default()


def close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return bool(np.all(np.abs(a - b) <= atol + rtol * np.abs(b)))


def ring(n, seed=0):
    r = np.random.default_rng(seed)
    return (r.uniform(2.0, 3.5, n).astype(np.float32), r.uniform(-60, 60, n).astype(np.float32),
            r.uniform(0, 360, n).astype(np.float32))


@pytest.mark.parametrize("perspective", [True, False])
@pytest.mark.parametrize("shared", [True, False])
def test_look_at_camera_forward_and_vjp(ctx, perspective, shared):
    B, NV = 7, 1000
    r = np.random.default_rng(3)
    v = r.uniform(-0.6, 0.6, (1 if shared else B, NV, 3)).astype(np.float32)
    cam = T.LookAt(perspective=perspective, viewing_angle=15, viewing_scale=0.8)
    cam._eye = T.get_points_from_angles(*ring(B))
    host_in = np.repeat(v, B, 0) if shared else v
    want = cam(host_in)
    got = cam(ctx.array(v))
    assert isinstance(got, _ffi.DeviceArray) and got.shape == (B, NV, 3)
    assert close(got.numpy(), want, 3e-6, 1e-6)
    g = r.uniform(-1, 1, (B, NV, 3)).astype(np.float32)
    want_g = cam.backward(g, host_in)
    if shared:
        want_g = want_g.astype(np.float64).sum(0, keepdims=True)
    got_g = cam.backward(ctx.array(g), ctx.array(v))
    assert got_g.shape == v.shape
    assert close(got_g.numpy(), want_g, 1e-5, 1e-5 * np.abs(want_g).max())


def test_single_eye_follows_the_vertex_batch(ctx):
    v = np.random.default_rng(5).uniform(-0.5, 0.5, (3, 50, 3)).astype(np.float32)
    cam = T.LookAt(viewing_angle=30)                       # default eye: one 3-vector (transform.py:44-45)
    assert close(cam(ctx.array(v)).numpy(), cam(v), 3e-6, 1e-6)
    with pytest.raises(ValueError):
        cam._eye = T.get_points_from_angles(*ring(2))
        cam(ctx.array(v))                                  # 3 vertex sets, 2 eyes


@pytest.mark.parametrize("coordinate", ["right", "left"])
def test_look_camera_forward(ctx, coordinate):
    v = np.random.default_rng(6).uniform(-0.5, 0.5, (4, 300, 3)).astype(np.float32)
    cam = T.Look(camera_direction=(0.2, -0.1, 1.0), viewing_angle=20, eye=[0.1, 0.2, -2.5], coordinate=coordinate)
    assert close(cam(ctx.array(v)).numpy(), cam(v), 3e-6, 1e-6)


def test_face_gather_and_scatter(ctx):
    v, f = jr.synthetic.uv_sphere(20, 11)
    B = 5
    vb = (v[None] * np.linspace(0.5, 1.5, B, dtype=np.float32)[:, None, None]).astype(np.float32)
    fb = np.repeat(f[None], B, 0)
    got = face_vertices(ctx.array(vb), fb)
    assert np.array_equal(got.numpy(), face_vertices(vb, fb))          # a gather: bit for bit
    # views with DIFFERENT face arrays take one launch per view
    fb2 = fb.copy()
    fb2[1] = fb2[1][:, [2, 1, 0]]
    assert np.array_equal(face_vertices(ctx.array(vb), fb2).numpy(), face_vertices(vb, fb2))
    g = np.random.default_rng(7).uniform(-1, 1, (B, f.shape[0], 3, 3)).astype(np.float32)
    for faces in (fb, fb2):
        want = face_vertices_backward(g, faces, v.shape[0])
        got = face_vertices_backward(ctx.array(g), faces, v.shape[0]).numpy()
        assert close(got, want, 1e-5, 1e-6 * np.abs(want).max())


@pytest.mark.parametrize("perspective", [True, False])
def test_fused_scatter_and_camera_vjp_for_a_shared_vertex_set(ctx, perspective):
    """LookAt.backward_from_faces == face_vertices_backward followed by LookAt.backward (the VJP is linear)."""
    from jrender_amd.structures.mesh import device_faces
    v, f = jr.synthetic.uv_sphere(30, 16)
    B = 9
    cam = T.LookAt(perspective=perspective, viewing_angle=15, viewing_scale=0.7)
    cam._eye = T.get_points_from_angles(*ring(B, seed=4))
    vd = ctx.array((v * 0.5)[None])
    g = ctx.array(np.random.default_rng(8).uniform(-1, 1, (B, f.shape[0], 3, 3)).astype(np.float32))
    two_step = cam.backward(face_vertices_backward(g, f[None], v.shape[0]), vd).numpy()
    fused = cam.backward_from_faces(g, device_faces(ctx, f), vd).numpy()
    assert fused.shape == (1, v.shape[0], 3)
    assert close(fused, two_step, 1e-5, 2e-6 * np.abs(two_step).max())
    # and against the host mirrors end to end
    want = cam.backward(face_vertices_backward(g.numpy(), np.repeat(f[None], B, 0), v.shape[0]), np.repeat((v * 0.5)[None], B, 0))
    want = want.astype(np.float64).sum(0, keepdims=True)
    assert close(fused, want, 1e-5, 1e-5 * np.abs(want).max())


@pytest.mark.parametrize("shape", [(6, 64, 64), (3, 1, 37, 41), (1, 5)])
def test_neg_iou_loss_and_gradient(ctx, shape):
    r = np.random.default_rng(9)
    p = r.uniform(0, 1, shape).astype(np.float32)
    t = (r.uniform(0, 1, shape) > 0.6).astype(np.float32)
    iou, g = jr.neg_iou_loss_and_grad(ctx.array(p), ctx.array(t))
    iou_h, g_h = jr.neg_iou_loss_and_grad(p, t)
    assert isinstance(iou, _ffi.DeviceArray) and close(iou.numpy(), iou_h, 2e-6, 0) and g.shape == p.shape
    assert close(g.numpy(), g_h, 1e-5, 1e-6 * np.abs(g_h).max())
    assert close(jr.neg_iou_loss(ctx.array(p), t), jr.neg_iou_loss(p, t), 2e-6, 1e-7)
    assert close(jr.neg_iou_loss_backward(ctx.array(p), t).numpy(), jr.neg_iou_loss_backward(p, t), 1e-5, 1e-6 * np.abs(g_h).max())
    # a shard of a larger batch: the mean runs over total_views
    _, g2 = jr.neg_iou_loss_and_grad(ctx.array(p), ctx.array(t), total_views=4 * shape[0])
    assert close(g2.numpy(), g_h / 4, 1e-5, 1e-6 * np.abs(g_h).max())


@pytest.mark.parametrize("average", [False, True])
def test_mesh_regularisers_value_and_gradient(ctx, average):
    """LaplacianLoss / FlattenLoss on device vertices against their NumPy mirrors (which are pinned to the reference's
    laplacian_loss.py / flatten_loss.py in tests/test_host_reference.py)."""
    v, f = jr.synthetic.uv_sphere(52, 27)                                     # the demo's 1 352-vertex class template
    r = np.random.default_rng(12)
    x = np.stack([v * 0.5 + r.normal(0, 0.01, v.shape), v * 0.4 + r.normal(0, 0.02, v.shape),
                  v * np.asarray([0.5, 0.2, 0.3])]).astype(np.float32)
    for loss in (jr.LaplacianLoss(v, f, average=average), jr.FlattenLoss(f, average=average)):
        val_d, g_d = loss.value_and_grad(ctx.array(x))
        want_v, want_g = np.asarray(loss(x), np.float64), loss.backward(x)
        assert close(np.asarray(val_d.numpy(), np.float64), want_v, 2e-5, 0), (type(loss).__name__, val_d.numpy(), want_v)
        assert g_d.shape == x.shape
        assert close(g_d.numpy(), want_g, 1e-4, 2e-5 * np.abs(want_g).max()), type(loss).__name__
        assert close(np.asarray(loss(ctx.array(x)).numpy(), np.float64), want_v, 2e-5, 0)
        assert close(loss.backward(ctx.array(x)).numpy(), want_g, 1e-4, 2e-5 * np.abs(want_g).max())


def test_all_zero_view_does_not_divide_by_zero(ctx):
    p = np.zeros((2, 16, 16), np.float32)
    iou, g = jr.neg_iou_loss_and_grad(ctx.array(p), ctx.array(p))
    assert np.array_equal(iou.numpy(), np.zeros(2, np.float32)) and np.isfinite(g.numpy()).all()


def _renderer(views, size=64):
    r = jr.Renderer(image_size=size, sigma_val=1e-4, aggr_func_rgb='hard', camera_mode='look_at', viewing_angle=15,
                    dr_type='softras')
    r.transform.set_eyes_from_angles(*ring(views, seed=11))
    return r


def test_renderer_chain_device_vs_host(ctx):
    """render_mesh / grad_vertices with ONE device vertex set against the host chain.  The camera kernel and NumPy's
    matmul round differently in the last place, and faces seen edge-on at the limb amplify one ulp into a visibly
    different pixel - so the host chain rasterises the DEVICE camera's output (the camera itself is held to 3e-6
    above): the images must then agree bit for bit, the gradient up to the order of the float atomics."""
    v, f = jr.synthetic.uv_sphere(36, 19)
    v = (v * np.asarray([0.5, 0.35, 0.45], np.float32))[None]
    B, nv = 6, v.shape[1]
    target = (np.random.default_rng(1).uniform(0, 1, (B, 64, 64)) > 0.5).astype(np.float32)
    rh, rd = _renderer(B), _renderer(B)
    sil_d = rd.render_mesh(jr.Mesh(ctx.array(v), f), mode='silhouettes')
    assert isinstance(sil_d, _ffi.DeviceArray) and sil_d.shape == (B, 64, 64)
    cam = rd.transform.transformer(ctx.array(v)).numpy()                       # [B,nv,3]
    fb = np.repeat(f[None], B, 0)
    sil_h = rh.rasterizer(jr.Mesh(cam, fb), 'silhouettes').numpy()
    assert np.array_equal(sil_d.numpy(), sil_h.reshape(B, 64, 64))
    # the whole host chain (its own camera) differs only where an edge-on face flips: a handful of pixels
    sil_hh = rh.render_mesh(jr.Mesh(np.repeat(v, B, 0), fb), mode='silhouettes').numpy().reshape(B, 64, 64)
    assert (np.abs(sil_hh - sil_h.reshape(B, 64, 64)) > 2e-3).mean() <= 2e-3
    rh.rasterizer(jr.Mesh(cam, fb), 'silhouettes')
    g_h = jr.neg_iou_loss_backward(sil_h.reshape(B, 64, 64), target)
    gfv_h, _ = rh.rasterizer.backward(grad_silhouettes=g_h.reshape(B, 1, 64, 64))
    gndc_h = face_vertices_backward(gfv_h.numpy().reshape(B, -1, 3, 3), fb, nv)
    gv_h = rh.transform.transformer.backward(gndc_h, np.repeat(v, B, 0)).astype(np.float64).sum(0, keepdims=True)
    _, g_d = jr.neg_iou_loss_and_grad(sil_d, target)
    gv_d = rd.grad_vertices(grad_silhouettes=g_d)
    assert isinstance(gv_d, _ffi.DeviceArray) and gv_d.shape == (1, nv, 3)
    assert np.abs(gv_d.numpy() - gv_h).max() <= 1e-4 * np.abs(gv_h).max()


def test_rgb_mode_with_device_vertices_and_textures(ctx):
    """Textured surface render from device vertices: one texture block broadcast over the views."""
    v, f = jr.synthetic.uv_sphere(24, 13)
    tex = np.random.default_rng(2).uniform(0, 1, (1, f.shape[0], 4, 3)).astype(np.float32)
    B = 3
    rh, rd = _renderer(B), _renderer(B)
    md = rd.transform(jr.Mesh(ctx.array(v[None] * 0.5), f, textures=tex))
    got = rd.rasterizer(md, 'rgb').numpy()
    want = rh.rasterizer(jr.Mesh(md.vertices.numpy(), np.repeat(f[None], B, 0), textures=np.repeat(tex, B, 0)), 'rgb').numpy()
    assert got.shape == want.shape == (B, 3, 64, 64) and np.array_equal(got, want)


def test_vertex_colours_broadcast_over_the_views(ctx):
    v, f = jr.synthetic.uv_sphere(24, 13)
    col = np.random.default_rng(4).uniform(0, 1, (1, v.shape[0], 3)).astype(np.float32)
    B = 3
    rh, rd = _renderer(B), _renderer(B)
    for r in (rh, rd):
        r.set_texture_mode('vertex')
    md = rd.transform(jr.Mesh(ctx.array(v[None] * 0.5), f, textures=col, texture_type='vertex'))
    got = rd.rasterizer(md, 'rgb').numpy()
    mh = jr.Mesh(md.vertices.numpy(), np.repeat(f[None], B, 0), textures=np.repeat(col, B, 0), texture_type='vertex')
    assert np.array_equal(got, rh.rasterizer(mh, 'rgb').numpy())


def test_lit_rgb_render_from_device_vertices(ctx):
    """render_mesh(mode='rgb'): lighting runs on the host from a download of the face vertices (normals are O(NF) host
    work) and rewrites the textures, the rest of the chain stays on the device.  Per-view vertex sets here."""
    v, f = jr.synthetic.uv_sphere(24, 13)
    B = 2
    vb = np.stack([v * 0.5, v * np.asarray([0.4, 0.5, 0.45], np.float32)]).astype(np.float32)
    tex = np.random.default_rng(5).uniform(0, 1, (B, f.shape[0], 1, 3)).astype(np.float32)
    rh, rd = _renderer(B), _renderer(B)
    want = rh.render_mesh(jr.Mesh(vb, np.repeat(f[None], B, 0), textures=tex), mode='rgb').numpy()
    got = rd.render_mesh(jr.Mesh(ctx.array(vb), np.repeat(f[None], B, 0), textures=tex), mode='rgb').numpy()
    assert got.shape == want.shape
    # same lighting (host, same inputs); the cameras differ in the last place: a handful of edge pixels
    assert (np.abs(got - want) > 2e-3).mean() <= 5e-3 and np.abs(got - want).mean() <= 1e-4


def test_demo2_front_ends_give_the_same_loss_curve():
    spec = importlib.util.spec_from_file_location("demo2", os.path.join(os.path.dirname(GOLD), "..", "examples", "demo2_deform.py"))
    demo2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo2)
    common = ["-b", "16", "--iters", "12", "--quiet"]
    dev = demo2.main(common + ["--front-end", "device"])
    host = demo2.main(common + ["--front-end", "host"])
    assert dev[-1] < dev[0] - 0.02
    assert np.abs(np.asarray(dev) - np.asarray(host)).max() <= 2e-3, (dev, host)


def test_demo2_as_one_hip_graph_retraces_the_launched_loop():
    """--graph (round 5): the iteration number on the device (jr_adam_step_counted, jr_scalar_accumulate_at, jr_counter_add), the
    third iteration RECORDED with jr_graph_begin / jr_graph_end - the forward launches against the pool and launch history of the
    iteration before, nothing waits for the GPU - and replayed for the rest: the loss curve of the launched loop (float atomics
    apart), and jr_graph_check finds every replayed forward inside the captured pool."""
    spec = importlib.util.spec_from_file_location("demo2", os.path.join(os.path.dirname(GOLD), "..", "examples", "demo2_deform.py"))
    demo2 = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo2)
    common = ["-b", "16", "--iters", "30", "--quiet", "--front-end", "device"]
    plain = demo2.main(common)
    graph = demo2.main(common + ["--graph"])
    assert len(graph) == len(plain) == 30 and graph[-1] < graph[0] - 0.05
    assert np.abs(np.asarray(graph) - np.asarray(plain)).max() <= 2e-3, (graph, plain)


def test_graph_capture_rules(ctx):
    """Inside a capture an allocation must come from the cache, and a counted Adam step equals the host-numbered one."""
    rng = np.random.default_rng(3)
    p0 = rng.normal(size=(1, 300, 3)).astype(np.float32)
    grads = [rng.normal(size=p0.shape).astype(np.float32) for _ in range(4)]
    pa, pb = ctx.array(p0), ctx.array(p0)
    oa, ob = jr.Adam([pa], 0.01, betas=(0.5, 0.99)), jr.Adam([pb], 0.01, betas=(0.5, 0.99))
    it = ctx.array(np.zeros(1, np.int32))
    for g in grads:
        gd = ctx.array(g)
        oa.step([gd])
        ob.step([gd], iteration=it)
        ctx.counter_add(it, 1)
    assert int(it.numpy()[0]) == 4
    assert np.abs(pa.numpy() - pb.numpy()).max() <= 2e-7 * np.abs(pa.numpy()).max()
    hist = ctx.zeros((5, 3))
    ctx.scalar_accumulate(hist, 1, ctx.array(np.arange(6, dtype=np.float32)), iteration=it, stride=3)     # row 4, column 1
    h = hist.numpy()
    assert h[4, 1] == 15.0 and np.count_nonzero(h) == 1
    with pytest.raises(RuntimeError, match="capture"):
        with ctx.capture():
            ctx.empty((7, 1234567))                       # a size nobody has freed: the capture refuses to call hipMalloc
    ctx.synchronize()                                     # (the aborted capture left the stream usable)
    assert np.array_equal(hist.numpy(), h)


def test_graph_pins_the_blocks_its_capture_touched(ctx):
    """a graph bakes device addresses into its nodes.  Blocks the allocator handed out (or took back) while the
    capture was open must stay out of circulation - not handed to later jr_malloc calls, not hipFree()d by jr_ctx_trim - until
    the graph is destroyed, even after their Python owners died; a replay then still lands in them."""
    n = 54321                                            # a size class of its own
    src = ctx.array(np.arange(n, dtype=np.float32))
    warm = ctx.empty((n,)); warm_ptr = warm.ptr; del warm          # one cached block of the class for the capture to find
    with ctx.capture() as g:
        tmp = src.clone()                                # library-internal style temporary: allocated inside the capture ...
        out_ptr = tmp.ptr
        del tmp                                          # ... and dropped inside it
    g.keep(src)
    assert out_ptr == warm_ptr
    # the freed temporary is parked: allocations of its size class get other memory, trim does not free it
    others = [ctx.empty((n,)).zero_() for _ in range(3)]
    assert all(o.ptr != out_ptr for o in others)
    del others
    ctx.trim()
    src.copy_from_host(np.arange(n, dtype=np.float32)[::-1].copy())
    g.launch(); g.check()
    look = _ffi.DeviceArray(ctx, out_ptr, (n,), np.float32, owner=g)       # (a non-owning look at the parked block)
    assert np.array_equal(look.numpy(), np.arange(n, dtype=np.float32)[::-1])
    del look
    g.close()
    again = ctx.empty((n,))                              # back in circulation once the graph is gone
    assert again.ptr == out_ptr
    # a graph captured before the context's scratch was reallocated refuses to replay
    it = ctx.array(np.zeros(1, np.int32))
    with ctx.capture() as g2:
        ctx.counter_add(it, 1)
    g2.keep(it)
    g2.launch(); g2.check()
    fv, tex = jr.synthetic.sphere_views(280, 1)
    from jrender_amd.renderer.dr.softras.soft_rasterize import SoftRasterizeFunction
    SoftRasterizeFunction(image_size=2440, ctx=ctx).execute(ctx.array(fv), ctx.array(tex))   # more bins than any earlier test of this context: the bin arrays grow
    with pytest.raises(RuntimeError, match="reallocated"):
        g2.launch()
    g2.close()
    assert int(it.numpy()[0]) == 1


@pytest.mark.parametrize("front_end", ["device", "host"])
def test_demo2_two_ranks_follow_the_one_rank_curve(front_end, tmp_path):
    """BASELINE configs[3] shards the views over the ranks: two ranks (sharing this box's GPU, so the exchange runs over
    the host communicator; with a GPU each it is RCCL on the same device buffer) render 8 views each, all-reduce the
    [1,nv,3] vertex gradient and the IoU sum, and must retrace the single-rank loss curve."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(GOLD), "..", "examples", "demo2_deform.py")
    hist = {}
    for n in (1, 2):
        out = str(tmp_path / ("h%d.npy" % n))
        p = subprocess.run([sys.executable, script, "-b", "16", "--iters", "10", "--quiet", "--front-end", front_end,
                            "--gpus", str(n), "--history-out", out], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
        hist[n] = np.load(out)
    assert hist[1][-1] < hist[1][0] - 0.02
    assert np.abs(hist[1] - hist[2]).max() <= 1e-3, (hist[1], hist[2])


# ---- round 5: the optimiser side of the loop on the device (demo2-deform.py:17-40, :72, :85-88) ----
def test_deform_model_device_vs_host(ctx):
    """jr_deform_vertices_forward / _backward against the NumPy restatement of Model.execute (demo2-deform.py:35-41):
    element-wise float steps to a few ulp, the centre's gradient (a sum over the vertices) to 1e-6 of its scale."""
    v, f = jr.synthetic.uv_sphere(52, 27)
    v = np.concatenate([v, [[0.0, 0.3, -0.9], [1.9, 0.0, 0.0]]]).astype(np.float32)     # a zero coordinate; |t| close to 1
    r = np.random.default_rng(21)
    host, dev = jr.DeformModel(v, f), jr.DeformModel(v, f, ctx=ctx)
    disp = r.normal(0, 0.3, host.displace.shape).astype(np.float32)
    cen = r.normal(0, 0.2, (1, 1, 3)).astype(np.float32)
    host.displace[...] = disp; host.center[...] = cen
    dev.displace.copy_from_host(disp); dev.center.copy_from_host(cen)
    want = host.forward()
    got = dev.forward()
    assert isinstance(got, _ffi.DeviceArray) and got.shape == want.shape
    assert close(got.numpy(), want, 3e-6, 1e-7)
    g = [r.uniform(-1, 1, want.shape).astype(np.float32) for _ in range(3)]
    for terms in ((), ((0.03, g[1]),), ((0.03, g[1]), (0.0003, g[2]))):
        w_disp, w_cen = host.backward(g[0], *terms)
        d_disp, d_cen = dev.backward(ctx.array(g[0]), *[(w, ctx.array(x)) for w, x in terms])
        assert close(d_disp.numpy(), w_disp, 2e-5, 1e-6 * np.abs(w_disp).max())
        assert close(d_cen.numpy(), w_cen, 1e-5, 1e-6 * np.abs(w_cen).max()), (d_cen.numpy(), w_cen)
    # the two-stage sum leaves its accumulators cleared: a second call gives the same bits
    a = dev.backward(ctx.array(g[0]))[1].numpy()
    b = dev.backward(ctx.array(g[0]))[1].numpy()
    assert np.array_equal(a, b)


def test_adam_on_the_device_retraces_the_numpy_restatement(ctx):
    """jr_adam_step performs the float operations of jrender_amd/optim.py's NumPy path in the same order: parameters and
    moments agree to the last bit or two over several steps (betas of demo2-deform.py:72; with and without weight decay)."""
    r = np.random.default_rng(22)
    for wd in (0.0, 0.01):
        p0 = [r.normal(0, 1, (1, 1354, 3)).astype(np.float32), r.normal(0, 1, (1, 1, 3)).astype(np.float32)]
        host = [p.copy() for p in p0]
        dev = [ctx.array(p) for p in p0]
        oh = jr.Adam(host, 0.01, betas=(0.5, 0.99), weight_decay=wd)
        od = jr.Adam(dev, 0.01, betas=(0.5, 0.99), weight_decay=wd)
        for step in range(6):
            grads = [r.normal(0, 10.0 ** r.integers(-6, 2), p.shape).astype(np.float32) for p in p0]
            oh.step(grads)
            od.step([ctx.array(g) for g in grads])
        for h, d, mh, md in zip(host, dev, oh.m, od.m):
            assert close(d.numpy(), h, 3e-7, 1e-9), np.abs(d.numpy() - h).max()
            assert close(md.numpy(), mh, 3e-7, 1e-12)
    with pytest.raises(TypeError):
        jr.Adam([np.zeros(3, np.float64)])


def test_scalar_accumulate_keeps_loss_terms_on_the_device(ctx):
    r = np.random.default_rng(23)
    hist = ctx.zeros((4, 3))
    x = r.uniform(0, 1, 1000).astype(np.float32)
    ctx.scalar_accumulate(hist, 3 * 2 + 1, ctx.array(x))
    ctx.scalar_accumulate(hist, 0, ctx.array(x[:7]), scale=-0.5, bias=1.0)
    ctx.scalar_accumulate(hist, 0, ctx.array(x[:3]), scale=2.0, accumulate=True)
    h = hist.numpy()
    assert abs(h[2, 1] - x.astype(np.float64).sum()) <= 1e-6 * x.sum()
    assert abs(h[0, 0] - (1.0 - 0.5 * x[:7].astype(np.float64).sum() + 2.0 * x[:3].astype(np.float64).sum())) <= 1e-6
    assert np.count_nonzero(h) == 2
    with pytest.raises(IndexError):
        ctx.scalar_accumulate(hist, 12, ctx.array(x))


def test_mesh_regularisers_and_shared_gradient_at_39k_faces(ctx, tmp_path):
    """VERDICT r4 next #7b / weak 10: the regularisers ran ONE workgroup per mesh (fine at 1 352 vertices, a wall at the
    headline mesh).  At 19 502 vertices / 58 500 edge pairs the grid kernels + two-stage loss sums must agree with the
    mirrors like at demo size, twice in a row (the accumulators clear themselves), and the [nv,3] shared-vertex gradient
    (234 KB) goes through a one-rank RCCL all-reduce on the device buffer."""
    from jrender_amd import comm as jcomm
    from jrender_amd.parallel import shared_vertex_gradient
    v, f = jr.synthetic.sphere_mesh(39000)
    r = np.random.default_rng(24)
    x = (v * 0.5 + r.normal(0, 0.002, v.shape)).astype(np.float32)[None]
    lap, flat = jr.LaplacianLoss(v, f), jr.FlattenLoss(f)
    for loss in (lap, flat):
        want_v, want_g = np.asarray(loss(x), np.float64), loss.backward(x)
        for _ in range(2):
            val_d, g_d = loss.value_and_grad(ctx.array(x))
            assert close(np.asarray(val_d.numpy(), np.float64), want_v, 3e-5, 0), (type(loss).__name__, val_d.numpy(), want_v)
            assert close(g_d.numpy(), want_g, 1e-4, 3e-5 * np.abs(want_g).max()), type(loss).__name__
    gf = r.uniform(-1, 1, (2, f.shape[0], 3, 3)).astype(np.float32)
    gv = shared_vertex_gradient(ctx.array(gf), f, v.shape[0])
    assert gv.shape == (v.shape[0], 3) and gv.nbytes == 234024
    before = gv.numpy()
    cm = jcomm.RcclCommunicator(ctx, 0, 1, path=str(tmp_path / "rdzv"))
    try:
        after = cm.all_reduce_sum(gv).numpy()
    finally:
        cm.close()
    assert np.array_equal(before, after)
