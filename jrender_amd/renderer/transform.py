"""Camera transforms — NumPy host-side mirror of jrender/renderer/transform/.

O(nv) float32 math that produces the hot path's input (NDC x,y in [-1,1],
z = camera-space depth).  Names, argument meaning and error behaviour follow the
reference: look_at (look_at.py:3-39), look (look.py:3-54), perspective
(perspective.py:4-17), orthogonal (orthogonal.py:3-16), projection
(projection.py:3-48), get_points_from_angles (utils/get_points_from_angles.py:4-22),
Transform / LookAt / Look / Projection (transform.py:10-135).

``*_backward`` functions are the hand-written vector-Jacobian products the
reference obtained from Jittor autograd (needed for mesh-deformation loops).

Device-resident vertices: ``LookAt`` / ``Look`` given a ``DeviceArray`` [VB,nv,3] run the HIP camera kernel
(``jr_camera_forward`` / ``jr_camera_backward``; only the O(B) rotation matrices are host work) and return a
``DeviceArray`` [B,nv,3] — VB = 1 broadcasts one vertex set over the B eyes like the reference's own
``vertices - eye[:, None, :]`` does, and its backward is the sum over the views.
"""
import math

import numpy as np

from .. import _ffi

F32 = np.float32


def _is_device(x):
    return isinstance(x, _ffi.DeviceArray)


class _CameraOnDevice:
    """eye [B,3] / rotation [B,9] of the views on the device (re-uploaded only when they change) + the two launches."""

    def __init__(self):
        self._key, self._eye_d, self._rot_d = None, None, None

    def _views(self, ctx, eye, rot):
        eye = np.ascontiguousarray(eye, F32).reshape(-1, 3)
        rot = np.ascontiguousarray(rot, F32).reshape(-1, 9)
        key = (id(ctx), eye.tobytes(), rot.tobytes())
        if key != self._key:
            self._eye_d, self._rot_d, self._key = ctx.array(eye), ctx.array(rot), key
        return eye.shape[0]

    def forward(self, vertices, eye, rot, kind, param):
        if vertices.ndim != 3 or vertices.shape[2] != 3 or vertices.dtype != F32:
            raise ValueError('vertices Tensor should have 3 dimensions')
        ctx = vertices.ctx
        B = self._views(ctx, eye, rot)
        VB, NV = vertices.shape[:2]
        if VB != 1 and VB != B:
            raise ValueError("vertices batch %d does not match %d eyes" % (VB, B))
        out = ctx.empty((B, NV, 3), F32)
        _ffi._check(_ffi.load().jr_camera_forward(ctx.handle, vertices.ptr, self._eye_d.ptr, self._rot_d.ptr, out.ptr,
                                                  B, VB, NV, int(kind), float(param)))
        return out

    def backward_from_faces(self, grad_face_vertices, faces_dev, vertices, eye, rot, kind, param):
        """[B,NF,3,3] NDC-space face gradients -> [1,nv,3] for ONE shared vertex set: scatter + VJP in one launch."""
        ctx = vertices.ctx
        B = self._views(ctx, eye, rot)
        NV, NF = vertices.shape[1], faces_dev.size // 3
        if vertices.shape[0] != 1 or grad_face_vertices.size != B * NF * 9:
            raise ValueError("backward_from_faces: one shared vertex set and [%d, %d, 3, 3] gradients expected" % (B, NF))
        gv = ctx.empty((1, NV, 3), F32)
        _ffi._check(_ffi.load().jr_face_camera_backward_shared(ctx.handle, grad_face_vertices.ptr, faces_dev.ptr, vertices.ptr,
                                                               self._eye_d.ptr, self._rot_d.ptr, gv.ptr, B, NV, NF,
                                                               int(kind), float(param)))
        return gv

    def backward(self, grad_out, vertices, eye, rot, kind, param):
        ctx = vertices.ctx
        B = self._views(ctx, eye, rot)
        VB, NV = vertices.shape[:2]
        g = grad_out if _is_device(grad_out) else ctx.array(np.asarray(grad_out, F32))
        if g.size != B * NV * 3:
            raise ValueError("grad_out must be [%d, %d, 3], got %s" % (B, NV, g.shape))
        gv = ctx.empty((VB, NV, 3), F32)
        _ffi._check(_ffi.load().jr_camera_backward(ctx.handle, g.ptr, vertices.ptr, self._eye_d.ptr, self._rot_d.ptr,
                                                   gv.ptr, B, VB, NV, int(kind), float(param)))
        return gv


def _normalize(v, eps=1e-5, axis=-1):
    n = np.sqrt(np.sum(v * v, axis=axis, keepdims=True, dtype=F32))
    return (v / np.maximum(n, F32(eps))).astype(F32)


def _as_batch(x, batch_size):
    x = np.asarray(x, F32)
    if x.ndim == 1:
        x = np.broadcast_to(x, (batch_size,) + x.shape)
    return x


def get_points_from_angles(distance, elevation, azimuth, degrees=True):
    """get_points_from_angles.py:4-22 (scalar -> tuple, arrays -> [N,3])."""
    if isinstance(distance, (float, int)):
        if degrees:
            elevation = math.radians(elevation)
            azimuth = math.radians(azimuth)
        return (distance * math.cos(elevation) * math.sin(azimuth),
                distance * math.sin(elevation),
                -distance * math.cos(elevation) * math.cos(azimuth))
    distance = np.asarray(distance, F32)
    elevation = np.asarray(elevation, F32)
    azimuth = np.asarray(azimuth, F32)
    if degrees:
        elevation = F32(math.pi / 180.) * elevation
        azimuth = F32(math.pi / 180.) * azimuth
    return np.stack([distance * np.cos(elevation) * np.sin(azimuth),
                     distance * np.sin(elevation),
                     -distance * np.cos(elevation) * np.cos(azimuth)], axis=0).T.astype(F32)


def _look_at_rotation(eye, at, up, batch_size):
    eye, at, up = _as_batch(eye, batch_size), _as_batch(at, batch_size), _as_batch(up, batch_size)
    z_axis = _normalize(at - eye)
    x_axis = _normalize(np.cross(up, z_axis))
    y_axis = _normalize(np.cross(z_axis, x_axis))
    return eye, np.stack([x_axis, y_axis, z_axis], axis=1).astype(F32)   # [bs,3,3] rows = axes


def look_at(vertices, eye, at=(0, 0, 0), up=(0, 1, 0)):
    """look_at.py:3-39: rotate/translate so that the z axis is (at - eye)."""
    vertices = np.asarray(vertices, F32)
    if vertices.ndim != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    eye, r = _look_at_rotation(eye, at, up, vertices.shape[0])
    v = vertices - eye[:, None, :]
    return np.matmul(v, r.transpose(0, 2, 1)).astype(F32)


def look_at_backward(grad_out, eye, at=(0, 0, 0), up=(0, 1, 0)):
    """VJP of look_at w.r.t. vertices (eye/at/up are constants)."""
    grad_out = np.asarray(grad_out, F32)
    _, r = _look_at_rotation(eye, at, up, grad_out.shape[0])
    return np.matmul(grad_out, r).astype(F32)


def look(vertices, eye, direction=(0, 1, 0), up=None, coordinate="right"):
    """look.py:3-54."""
    vertices = np.asarray(vertices, F32)
    if vertices.ndim != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    eye, r = _look_rotation(eye, direction, up, coordinate, vertices.shape[0])
    v = vertices - eye[:, None, :]
    return np.matmul(v, r.transpose(0, 2, 1)).astype(F32)


def _look_rotation(eye, direction, up, coordinate, batch_size):
    direction = np.asarray(direction, F32)
    up = np.asarray([0, 1, 0] if up is None else up, F32)
    z_axis = _normalize(direction, axis=0)
    up = _normalize(up, axis=0)
    if abs(float(np.sum(up * z_axis))) > 1 - 1e-4:
        raise ValueError("camera_direction and camera_up can not be the same")
    bs = batch_size
    eye, z_axis, up = _as_batch(eye, bs), _as_batch(z_axis, bs), _as_batch(up, bs)
    if coordinate == "right":
        x_axis = _normalize(np.cross(up, z_axis))
        y_axis = _normalize(np.cross(z_axis, x_axis))
    elif coordinate == "left":
        x_axis = _normalize(np.cross(z_axis, up))
        y_axis = _normalize(np.cross(x_axis, z_axis))
    else:
        raise ValueError("coordinate must be 'right' or 'left'")
    return eye, np.stack([x_axis, y_axis, z_axis], axis=1).astype(F32)


def perspective(vertices, angle=30.):
    """perspective.py:4-17: x,y /= z * tan(angle); z unchanged."""
    vertices = np.asarray(vertices, F32)
    if vertices.ndim != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    width = np.tan(F32(angle / 180 * math.pi)).astype(F32)
    z = vertices[:, :, 2]
    x = vertices[:, :, 0] / z / width
    y = vertices[:, :, 1] / z / width
    return np.stack([x, y, z], axis=2).astype(F32)


def perspective_backward(grad_out, vertices, angle=30.):
    """VJP of perspective w.r.t. its input vertices."""
    grad_out = np.asarray(grad_out, F32)
    vertices = np.asarray(vertices, F32)
    width = np.tan(F32(angle / 180 * math.pi)).astype(F32)
    z = vertices[:, :, 2]
    gx, gy, gz = grad_out[:, :, 0], grad_out[:, :, 1], grad_out[:, :, 2]
    dx = gx / z / width
    dy = gy / z / width
    dz = gz - (gx * vertices[:, :, 0] + gy * vertices[:, :, 1]) / (z * z) / width
    return np.stack([dx, dy, dz], axis=2).astype(F32)


def orthogonal(vertices, scale):
    """orthogonal.py:3-16."""
    vertices = np.asarray(vertices, F32)
    if vertices.ndim != 3:
        raise ValueError('vertices Tensor should have 3 dimensions')
    s = F32(scale)
    return np.stack([vertices[:, :, 0] * s, vertices[:, :, 1] * s, vertices[:, :, 2]], axis=2).astype(F32)


def projection(vertices, K, R, t, dist_coeffs, orig_size, eps=1e-9):
    """projection.py:3-48 (OpenCV-style intrinsics/extrinsics + distortion)."""
    vertices = np.asarray(vertices, F32)
    K, R, t = np.asarray(K, F32), np.asarray(R, F32), np.asarray(t, F32)
    dist_coeffs = np.asarray(dist_coeffs, F32)
    v = np.matmul(vertices, R.transpose(0, 2, 1)[0]) + t
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    x_ = x / (z + F32(eps))
    y_ = y / (z + F32(eps))
    k1, k2, p1, p2, k3 = (dist_coeffs[:, i][:, None] for i in range(5))
    x_2, y_2 = x_ * x_, y_ * y_
    r = np.sqrt(x_2 + y_2)
    r2 = r * r
    r4 = r2 * r2
    r6 = r4 * r2
    tmp = k1 * r2 + k2 * r4 + k3 * r6 + 1
    x__ = x_ * tmp + 2 * p1 * x_ * y_ + p2 * (r2 + 2 * x_2)
    y__ = y_ * tmp + p1 * (r2 + 2 * y_2) + 2 * p2 * x_ * y_
    v = np.stack([x__, y__, np.ones_like(z)], axis=-1)
    v = np.matmul(v, K.transpose(0, 2, 1)[0])
    u, vv = v[:, :, 0], v[:, :, 1]
    vv = orig_size - vv
    u = 2 * (u - orig_size / 2.) / orig_size
    vv = 2 * (vv - orig_size / 2.) / orig_size
    return np.stack([u, vv, z], axis=-1).astype(F32)


class Projection:
    """transform.py:10-34."""

    def __init__(self, K, R, t, dist_coeffs=None, orig_size=512):
        self.K, self.R, self.t = np.asarray(K, F32), np.asarray(R, F32), np.asarray(t, F32)
        self.dist_coeffs = dist_coeffs
        self.orig_size = orig_size
        self._eye = None
        if dist_coeffs is None:
            self.dist_coeffs = np.zeros((self.K.shape[0], 5), F32)

    def __call__(self, vertices):
        return projection(vertices, self.K, self.R, self.t, self.dist_coeffs, self.orig_size)


class LookAt:
    """transform.py:37-56."""

    def __init__(self, perspective=True, viewing_angle=30, viewing_scale=1.0, eye=None):
        self.perspective = perspective
        self.viewing_angle = viewing_angle
        self.viewing_scale = viewing_scale
        self._eye = eye
        if self._eye is None:
            self._eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]

    def _projection_kind(self):
        if self.perspective:
            return 1, float(np.tan(F32(self.viewing_angle / 180 * math.pi)).astype(F32))
        return 2, float(F32(self.viewing_scale))

    def _device_views(self, vertices):
        # eyes [B,3] decide the number of views; a single eye follows the vertices' batch (look_at.py:24-30)
        nviews = np.asarray(self._eye).shape[0] if np.ndim(self._eye) == 2 else vertices.shape[0]
        key = (nviews, np.asarray(self._eye, F32).tobytes())
        if getattr(self, "_views_key", None) != key:           # the rotations only change with the eyes
            self._views_key, self._views = key, _look_at_rotation(self._eye, (0, 0, 0), (0, 1, 0), nviews)
        return self._views

    def __call__(self, vertices):
        if _is_device(vertices):
            eye, rot = self._device_views(vertices)
            self._dev = getattr(self, "_dev", None) or _CameraOnDevice()
            return self._dev.forward(vertices, eye, rot, *self._projection_kind())
        vertices = look_at(vertices, self._eye)
        if self.perspective:
            return perspective(vertices, angle=self.viewing_angle)
        return orthogonal(vertices, scale=self.viewing_scale)

    def backward_from_faces(self, grad_face_vertices, faces_dev, vertices_in):
        """Device only, one shared vertex set [1,nv,3]: face-vertex gradients [B,NF,3,3] -> [1,nv,3] (the scatter-add
        of face_vertices_backward and ``backward`` fused: the VJP is linear, so it commutes with the sums)."""
        eye, rot = self._device_views(vertices_in)
        self._dev = getattr(self, "_dev", None) or _CameraOnDevice()
        return self._dev.backward_from_faces(grad_face_vertices, faces_dev, vertices_in, eye, rot, *self._projection_kind())

    def backward(self, grad_out, vertices_in):
        """VJP w.r.t. the world-space vertices given the same input as __call__."""
        if _is_device(vertices_in):
            eye, rot = self._device_views(vertices_in)
            self._dev = getattr(self, "_dev", None) or _CameraOnDevice()
            return self._dev.backward(grad_out, vertices_in, eye, rot, *self._projection_kind())
        cam = look_at(vertices_in, self._eye)
        if self.perspective:
            g = perspective_backward(grad_out, cam, angle=self.viewing_angle)
        else:
            s = F32(self.viewing_scale)
            g = np.asarray(grad_out, F32) * np.asarray([s, s, 1], F32)
        return look_at_backward(g, self._eye)


class Look:
    """transform.py:59-82."""

    def __init__(self, camera_direction=(0, 0, 1), perspective=True, viewing_angle=30,
                 viewing_scale=1.0, eye=None, up=(0, 1, 0), coordinate="right"):
        self.perspective = perspective
        self.viewing_angle = viewing_angle
        self.viewing_scale = viewing_scale
        self._eye = eye
        self.camera_direction = camera_direction
        self.up = up
        self.coordinate = coordinate
        if self._eye is None:
            self._eye = [0, 0, -(1. / math.tan(math.radians(self.viewing_angle)) + 1)]

    def __call__(self, vertices):
        if _is_device(vertices):
            nviews = np.asarray(self._eye).shape[0] if np.ndim(self._eye) == 2 else vertices.shape[0]
            eye, rot = _look_rotation(self._eye, self.camera_direction, self.up, self.coordinate, nviews)
            self._dev = getattr(self, "_dev", None) or _CameraOnDevice()
            kind, param = LookAt._projection_kind(self)
            return self._dev.forward(vertices, eye, rot, kind, param)
        vertices = look(vertices, self._eye, self.camera_direction, up=self.up,
                        coordinate=self.coordinate)
        if self.perspective:
            return perspective(vertices, angle=self.viewing_angle)
        return orthogonal(vertices, scale=self.viewing_scale)


class Transform:
    """transform.py:83-135.  NB ``__call__`` MUTATES ``mesh.vertices`` like the
    reference (transform.py:105-107); callers ``mesh.reset_()`` between renders."""

    def __init__(self, camera_mode='projection', K=None, R=None, t=None, dist_coeffs=None,
                 orig_size=512, perspective=True, viewing_angle=30, viewing_scale=1.0,
                 eye=None, camera_direction=(0, 0, 1), up=(0, 1, 0), coordinate="right"):
        self.camera_mode = camera_mode
        if camera_mode == 'projection':
            self.transformer = Projection(K, R, t, dist_coeffs, orig_size)
        elif camera_mode == 'look':
            self.transformer = Look(camera_direction, perspective, viewing_angle, viewing_scale,
                                    eye, up, coordinate)
        elif camera_mode == 'look_at':
            self.transformer = LookAt(perspective, viewing_angle, viewing_scale, eye)
        else:
            raise ValueError('Camera mode has to be one of projection, look or look_at')
        self.eye = eye
        self.camera_direction = camera_direction
        self.viewing_angle = viewing_angle
        self.up = up
        self.coordinate = coordinate

    def __call__(self, mesh):
        mesh.vertices = self.transformer(mesh.vertices)
        return mesh

    execute = __call__

    def tranpos(self, pos):
        return self.transformer(pos)

    def set_eyes_from_angles(self, distances, elevations, azimuths):
        if self.camera_mode not in ['look', 'look_at']:
            raise ValueError('Projection does not need to set eyes')
        self.transformer._eye = get_points_from_angles(distances, elevations, azimuths)

    def set_eyes(self, eyes):
        if self.camera_mode not in ['look', 'look_at']:
            raise ValueError('Projection does not need to set eyes')
        self.transformer._eye = eyes

    def view_transform(self, vertices):
        if self.camera_mode == 'look_at':
            vertices = look_at(vertices, self.eye)
        elif self.camera_mode == 'look':
            vertices = look(vertices, self.eye, self.camera_direction, up=self.up,
                            coordinate=self.coordinate)
        return vertices

    def projection_transform(self, vertices):
        return perspective(vertices, self.viewing_angle)

    @property
    def eyes(self):
        return self.transformer._eye
