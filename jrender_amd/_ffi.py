"""ctypes binding of libjrender_hip.so (the C ABI in include/jrender_hip.h).

This is the whole FFI: no Jittor, PyTorch or Triton.  ``Context`` owns one GPU
(one process per GPU), ``DeviceArray`` is a typed view of device memory with the
small part of the ``jt.Var`` surface the renderer's callers use (``.shape``,
``.dtype``, ``.numpy()``).  The product path has NO CPU fallback: if the HIP
library is missing or no GPU is visible, calls raise.
"""
import contextlib
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# JRENDER_LIB: another build of the same library (tools/ablate/ builds variants next to the product .so)
LIB_PATH = os.environ.get("JRENDER_LIB") or os.path.join(_HERE, "csrc", "libjrender_hip.so")

_lib = None
_lock = threading.Lock()

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)

# name -> (restype, argtypes); kept in one table so tests can check it against the header
_SCALARS_FWD = [C.c_int] * 5 + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float] + [C.c_int] * 4
SIGNATURES = {
    "jr_last_error": (C.c_char_p, []),
    "jr_version": (C.c_char_p, []),
    "jr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "jr_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "jr_ctx_destroy": (C.c_int, [C.c_void_p]),
    "jr_ctx_device": (C.c_int, [C.c_void_p]),
    "jr_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "jr_malloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "jr_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jr_ctx_trim": (C.c_int, [C.c_void_p]),
    "jr_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "jr_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "jr_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "jr_memcpy2d_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "jr_memset": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "jr_synchronize": (C.c_int, [C.c_void_p]),
    "jr_event_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "jr_event_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jr_event_record": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jr_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
    "jr_softras_forward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 6 + _SCALARS_FWD + [c_float_p]),
    "jr_softras_backward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 9 + _SCALARS_FWD),
    "jr_softras_backward_ex": (C.c_int, [C.c_void_p] + [C.c_void_p] * 9 + _SCALARS_FWD + [C.c_uint64]),
    "jr_softras_forward_token": (C.c_uint64, [C.c_void_p]),
    "jr_face_vertices_forward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3),
    "jr_face_vertices_backward": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3),
    "jr_face_vertices_backward_shared": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3),
    "jr_camera_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float]),
    "jr_camera_backward": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float]),
    "jr_face_camera_backward_shared": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float]),
    "jr_laplacian_loss": (C.c_int, [C.c_void_p] * 11 + [C.c_int] * 2 + [C.c_float]),
    "jr_flatten_loss": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 3 + [C.c_float] * 2),
    "jr_neg_iou_loss": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 2 + [C.c_float]),
    "jr_deform_vertices_forward": (C.c_int, [C.c_void_p] * 5 + [C.c_int]),
    "jr_deform_vertices_backward": (C.c_int, [C.c_void_p] * 5 + [C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int]),
    "jr_adam_step": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t] + [C.c_double] * 5 + [C.c_int]),
    "jr_scalar_accumulate": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_float, C.c_int]),
    "jr_adam_step_counted": (C.c_int, [C.c_void_p] * 5 + [C.c_size_t] + [C.c_double] * 5 + [C.c_void_p]),
    "jr_scalar_accumulate_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int]),
    "jr_counter_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "jr_graph_begin": (C.c_int, [C.c_void_p]),
    "jr_graph_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "jr_graph_abort": (C.c_int, [C.c_void_p]),
    "jr_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jr_graph_check": (C.c_int, [C.c_void_p]),
    "jr_graph_destroy": (C.c_int, [C.c_void_p, C.c_void_p]),
    "jr_avgpool2x2_forward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3),
    "jr_avgpool2x2_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3),
    "jr_n3mr_forward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 11 + [C.c_int] * 4 + [C.c_float] * 3 + [c_float_p] + [C.c_int] * 3),
    "jr_n3mr_backward": (C.c_int, [C.c_void_p] + [C.c_void_p] * 14 + [C.c_int] * 4 + [C.c_float] + [C.c_int] * 3),
    "jr_n3mr_image_forward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5),
    "jr_n3mr_image_backward": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 5),
    "jr_selftest_division": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "jr_selftest_reciprocal": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "jr_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "jr_profile_collect": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "jr_softras_last_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "jr_softras_last_launch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "jr_softras_set_launch_policy": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "jr_softras_set_bin_size": (C.c_int, [C.c_void_p, C.c_int]),
    "jr_softras_bin_size": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "jr_softras_set_precise_colour": (C.c_int, [C.c_void_p, C.c_int]),
    "jr_debug_section_clocks": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "jr_comm_unique_id": (C.c_int, [C.c_void_p]),
    "jr_comm_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "jr_comm_destroy": (C.c_int, [C.c_void_p]),
    "jr_comm_rank": (C.c_int, [C.c_void_p]),
    "jr_comm_size": (C.c_int, [C.c_void_p]),
    "jr_comm_all_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "jr_comm_all_gather_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]),
    "jr_comm_all_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "jr_comm_all_reduce_host_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]),
    "jr_comm_barrier": (C.c_int, [C.c_void_p]),
}


def load():
    """Load libjrender_hip.so (built by ``python -m jrender_amd._build`` /
    ``__graft_entry__.build()``).  Raises if it is missing — there is no fallback."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "jrender_amd: %s is missing; build it with `python -m jrender_amd._build` "
                    "(needs hipcc).  There is no CPU fallback." % LIB_PATH)
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                try:
                    fn = getattr(lib, name)
                except AttributeError:
                    # JRENDER_LIB: an A/B build of ANOTHER commit (tools/ablate: libjrender_hip_base.so) may predate an
                    # entry point; the product library must export everything the header declares (tests/test_cabi.py)
                    if os.environ.get("JRENDER_LIB"):
                        continue
                    raise
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise RuntimeError("jrender_hip: " + load().jr_last_error().decode("utf-8", "replace"))


def device_count():
    n = C.c_int(0)
    _check(load().jr_device_count(C.byref(n)))
    return n.value


class DeviceArray:
    """Contiguous row-major device buffer with shape/dtype.  Freed on GC."""

    __slots__ = ("ctx", "ptr", "shape", "dtype", "_owner", "__weakref__")

    def __init__(self, ctx, ptr, shape, dtype, owner=None):
        self.ctx, self.ptr = ctx, ptr
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self._owner = owner            # None: this object owns ptr; else keeps the owner alive (views)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def ndim(self):
        return len(self.shape)

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        if -1 in shape:
            known = -int(np.prod(shape, dtype=np.int64))
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError("cannot reshape %s into %s" % (self.shape, shape))
        return DeviceArray(self.ctx, self.ptr, shape, self.dtype, owner=self if self._owner is None else self._owner)

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            _check(load().jr_memcpy_d2h(self.ctx.handle, out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def copy_from_host(self, arr):
        arr = np.ascontiguousarray(arr, self.dtype)
        if arr.size != self.size:
            raise ValueError("size mismatch: %s vs %s" % (arr.shape, self.shape))
        if arr.nbytes:
            _check(load().jr_memcpy_h2d(self.ctx.handle, self.ptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))
        return self

    def clone(self):
        out = self.ctx.empty(self.shape, self.dtype)
        if self.nbytes:
            _check(load().jr_memcpy_d2d(self.ctx.handle, out.ptr, self.ptr, self.nbytes))
        return out

    def zero_(self):
        _check(load().jr_memset(self.ctx.handle, self.ptr, 0, self.nbytes))
        return self

    def view(self, lo, hi):
        """Rows [lo, hi) of the leading axis as a non-owning view."""
        lo, hi = int(lo), int(hi)
        if not (0 <= lo <= hi <= self.shape[0]):
            raise IndexError("rows [%d, %d) of %s" % (lo, hi, self.shape))
        row = self.nbytes // self.shape[0] if self.shape[0] else 0
        return DeviceArray(self.ctx, (self.ptr or 0) + lo * row, (hi - lo,) + self.shape[1:], self.dtype,
                           owner=self if self._owner is None else self._owner)

    @property
    def __cuda_array_interface__(self):
        # zero-copy hand-off to anything that understands the protocol (tests, interop)
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (int(self.ptr), False),
                "version": 3, "strides": None}

    def __del__(self):
        try:
            if self._owner is None and self.ptr and self.ctx is not None and self.ctx.handle:
                load().jr_free(self.ctx.handle, self.ptr)
        except Exception:
            pass
        self.ptr = None

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, gpu=%d)" % (self.shape, self.dtype, self.ctx.device)


class Graph:
    """A recorded launch sequence (Context.capture)."""

    def __init__(self, ctx):
        self.ctx, self.handle, self._keep = ctx, None, []

    def __enter__(self):
        _check(load().jr_graph_begin(self.ctx.handle))
        return self

    def __exit__(self, et, ev, tb):
        if et is not None:
            load().jr_graph_abort(self.ctx.handle)
            return False
        h = C.c_void_p()
        _check(load().jr_graph_end(self.ctx.handle, C.byref(h)))
        self.handle = h
        return False

    def keep(self, *objects):
        """Objects whose device buffers the graph reads or writes at replay and that must therefore outlive it."""
        self._keep.extend(objects)
        return objects[0] if len(objects) == 1 else objects

    def launch(self):
        _check(load().jr_graph_launch(self.ctx.handle, self.handle))

    def check(self):
        _check(load().jr_graph_check(self.ctx.handle))

    def close(self):
        if self.handle is not None:
            load().jr_graph_destroy(self.ctx.handle, self.handle)
            self.handle = None
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU: HIP stream + scratch arena (jr_ctx).  ``Context.default()`` picks the GPU from
    LOCAL_RANK (one process per GPU; set by bench.py's own launcher or by torch.distributed.run) or GPU 0."""

    _default = None

    def __init__(self, device=0):
        self.handle = None
        h = C.c_void_p()
        _check(load().jr_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    @classmethod
    def default(cls):
        if cls._default is None:
            dev = int(os.environ.get("JRENDER_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = device_count()
            if n < 1:
                raise RuntimeError("jrender_amd: no HIP device visible (there is no CPU fallback)")
            cls._default = cls(dev % n)
        return cls._default

    # ---- memory ----
    def empty(self, shape, dtype=np.float32):
        if isinstance(shape, int):
            shape = (shape,)
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = C.c_void_p()
        _check(load().jr_malloc(self.handle, nbytes, C.byref(p)))
        return DeviceArray(self, p.value, shape, dtype)

    def zeros(self, shape, dtype=np.float32):
        return self.empty(shape, dtype).zero_()

    def array(self, arr, dtype=None):
        if isinstance(arr, DeviceArray):
            if dtype is not None and np.dtype(dtype) != arr.dtype:
                raise TypeError("dtype conversion of device arrays is not supported")
            return arr
        arr = np.ascontiguousarray(arr, dtype)
        return self.empty(arr.shape, arr.dtype).copy_from_host(arr)

    def trim(self):
        """Return the allocator cache to the driver."""
        _check(load().jr_ctx_trim(self.handle))

    def synchronize(self):
        _check(load().jr_synchronize(self.handle))

    # ---- events ----
    def event(self):
        e = C.c_void_p()
        _check(load().jr_event_create(self.handle, C.byref(e)))
        return e

    def record(self, e):
        _check(load().jr_event_record(self.handle, e))

    def elapsed_ms(self, start, stop):
        ms = C.c_float(0)
        _check(load().jr_event_elapsed_ms(self.handle, start, stop, C.byref(ms)))
        return ms.value

    PHASES = ("bin_count", "bin_fill_sort", "fwd_raster", "bwd_raster")

    def profile_enable(self, on=True):
        _check(load().jr_profile_enable(self.handle, int(bool(on))))

    def profile_collect(self):
        """-> {phase: (total_ms, brackets)} since the previous collect (synchronises)."""
        ms = (C.c_double * 4)()
        n = (C.c_int64 * 4)()
        _check(load().jr_profile_collect(self.handle, ms, n))
        return {name: (ms[i], n[i]) for i, name in enumerate(self.PHASES)}

    def selftest_division(self, n=1 << 31, seed=1):
        """Bit mismatches between the kernels' reciprocal-refinement quotient and IEEE a / b."""
        bad = C.c_uint64(0)
        _check(load().jr_selftest_division(self.handle, int(n), int(seed), C.byref(bad)))
        return bad.value

    def selftest_reciprocal(self):
        """Bit mismatches of the Newton-refined v_rcp_f32 against IEEE 1.0f/x, exhaustive over 1.36e9 floats."""
        bad = C.c_uint64(0)
        _check(load().jr_selftest_reciprocal(self.handle, C.byref(bad)))
        return bad.value

    def section_clocks(self):
        """Instrumented builds only: shader-clock totals per kernel section since the previous call."""
        a = (C.c_uint64 * 20)()
        _check(load().jr_debug_section_clocks(self.handle, a))
        return list(a)

    def last_stats(self):
        s = (C.c_int64 * 4)()
        _check(load().jr_softras_last_stats(self.handle, s))
        return dict(bin_face_pairs=s[0], nonempty_bins=s[1], max_faces_in_bin=s[2], bins_per_image=s[3])

    def last_launch(self):
        """Which paths the last launches took (jr_softras_last_launch)."""
        s = (C.c_int64 * 4)()
        _check(load().jr_softras_last_launch(self.handle, s))
        return dict(four_wavefront_kernel=bool(s[0]), heavy_bins=int(s[1]), wavefronts_per_workgroup=int(s[2]),
                    heavy_min_faces=int(s[3]))

    def set_launch_policy(self, heavy_min_faces=-1, heavy_waves=0):
        """Which kernel organisation renders heavy tiles (never what they compute): bins listing more than
        ``heavy_min_faces`` faces get a workgroup of ``heavy_waves`` (4 / 8; 0 = automatic) wavefronts per tile;
        ``heavy_min_faces`` < 0 restores the default, 0 switches the multi-wavefront tiles off."""
        _check(load().jr_softras_set_launch_policy(self.handle, int(heavy_min_faces), int(heavy_waves)))

    def scalar_accumulate(self, dst, index, src, scale=1.0, bias=0.0, accumulate=False, iteration=None, stride=0):
        """dst[index] = (accumulate ? dst[index] : 0) + bias + scale * sum(src) on the device (jr_scalar_accumulate): loss
        terms stay on the GPU - e.g. one slot of a history array per iteration - until the caller reads them.  With
        ``iteration`` (a one-element int32 DeviceArray) the slot is index + stride * iteration[0], read on the device
        (jr_scalar_accumulate_at: what a launch recorded into a graph needs - the caller keeps the slot inside dst)."""
        if dst.dtype != np.float32 or src.dtype != np.float32:
            raise TypeError("scalar_accumulate works on float32 arrays")
        if not 0 <= int(index) < dst.size:
            raise IndexError("index %d outside the %d elements of dst" % (index, dst.size))
        if iteration is not None:
            if iteration.dtype != np.int32 or iteration.size != 1:
                raise TypeError("iteration must be a one-element int32 DeviceArray")
            _check(load().jr_scalar_accumulate_at(self.handle, C.c_void_p(dst.ptr + 4 * int(index)), int(stride), iteration.ptr,
                                                  src.ptr, int(src.size), float(scale), float(bias), int(bool(accumulate))))
            return
        _check(load().jr_scalar_accumulate(self.handle, C.c_void_p(dst.ptr + 4 * int(index)), src.ptr, int(src.size),
                                           float(scale), float(bias), int(bool(accumulate))))

    def counter_add(self, counter, delta=1):
        """counter[0] += delta on the device (jr_counter_add): the iteration number of a loop that runs as a graph."""
        if counter.dtype != np.int32 or counter.size != 1:
            raise TypeError("counter must be a one-element int32 DeviceArray")
        _check(load().jr_counter_add(self.handle, counter.ptr, int(delta)))

    def capture(self):
        """``with ctx.capture() as g: <device calls>`` records the calls on this context into ONE HIP graph instead of running
        them (jr_graph_begin / jr_graph_end, include/jrender_hip.h: no host transfers or waits inside, every buffer from the
        allocator's cache - run the sequence once or twice first -, iteration numbers on the device); ``g.launch()``
        replays it, ``g.check()`` waits and verifies that the replayed forwards stayed inside the captured pool.
        Temporaries created (or dropped) inside the ``with`` block are pinned to the graph by the allocator and only return
        to its cache at ``g.close()``; arrays that exist BEFORE the block and are used inside it must be kept alive by the
        caller (``g.keep(...)``).  A later, larger call outside the graph that makes the library reallocate its scratch
        outdates the graph: ``g.launch()`` raises and the sequence has to be captured again."""
        return Graph(self)

    def set_bin_size(self, bin_size=0):
        """Screen-bin size in pixels for the next launches (the reference operator's ``bin_size``): 0 = by image size,
        else rounded up to 8, 16 or 32.  Results are bit-identical for every value."""
        _check(load().jr_softras_set_bin_size(self.handle, int(bin_size)))
        self._bin_size = int(bin_size)

    def bin_size(self, image_size=0, batch=1, num_faces=0):
        """Bin size a launch of ``batch`` views of ``num_faces`` faces at ``image_size`` would use now; image_size 0: the one the
        workspace's lists were built with."""
        return int(load().jr_softras_bin_size(self.handle, int(image_size), int(batch), int(num_faces)))

    def set_precise_colour(self, on=True):
        """Forward colour path in the reference's own arithmetic (jr_softras_set_precise_colour): element-wise 1e-4 gradients
        at +15 % forward time.  Sticky; ``SoftRasterizeFunction(precise_colour=True)`` sets it for its own launches only."""
        _check(load().jr_softras_set_precise_colour(self.handle, int(bool(on))))
        self._precise = bool(on)

    @contextlib.contextmanager
    def precise_colour_scope(self, on):
        if on is None:
            yield
            return
        prev = getattr(self, "_precise", False)
        self.set_precise_colour(on)
        try:
            yield
        finally:
            self.set_precise_colour(prev)

    @contextlib.contextmanager
    def bin_size_scope(self, bin_size):
        """``with ctx.bin_size_scope(16): ...`` - an operator's own ``bin_size`` for its launches; 0 / None: no change."""
        if not bin_size:
            yield
            return
        prev = getattr(self, "_bin_size", None)
        if prev is None:
            prev = 0 if not os.environ.get("JR_BIN_SIZE") else max(int(os.environ["JR_BIN_SIZE"]), 0)
        self.set_bin_size(bin_size)
        try:
            yield
        finally:
            self.set_bin_size(prev)

    def close(self):
        if self.handle:
            load().jr_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        # DeviceArrays may outlive the interpreter's module teardown order; leak rather than crash
        pass
