"""N3mrRasterizer — host-side mirror of jrender/renderer/dr/n3mr/rasterizer.py (N3R:9-105)."""
import numpy as np

from .n3mr import DEFAULT_EPS, DEFAULT_FAR, DEFAULT_NEAR, RasterizeRGBAD
from ....structures.mesh import face_vertices as vertices_to_faces

__all__ = ["N3mrRasterizer", "vertices_to_faces"]


class N3mrRasterizer:
    def __init__(self, image_size=256, anti_aliasing=True, background_color=[0, 0, 0], fill_back=True,
                 near=0.1, far=100):
        self.image_size = image_size
        self.anti_aliasing = anti_aliasing
        self.background_color = background_color
        self.fill_back = fill_back
        self.near = near
        self.far = far
        self.rasterizer_eps = 1e-3

    def execute(self, mesh, mode=None):
        vertices, faces, textures = mesh.vertices, mesh.faces, mesh.textures
        if mode is None:
            return self.render(vertices, faces, textures)
        elif mode == 'rgb':
            return self.render_rgb(vertices, faces, textures)
        elif mode == 'silhouettes':
            return self.render_silhouettes(vertices, faces)
        elif mode == 'depth':
            return self.render_depth(vertices, faces)
        raise ValueError("mode should be one of None, 'silhouettes' or 'depth'")

    __call__ = execute

    def _fill_back(self, faces, textures=None):
        if self.fill_back:                                                               # N3R:63-64, :83-85
            faces = np.concatenate((faces, faces[:, :, ::-1]), axis=1)
            if textures is not None:
                textures = np.concatenate((textures, textures.transpose(0, 1, 4, 3, 2, 5)), axis=1)
        return faces, textures

    def _run(self, vertices, faces, textures, rgb, alpha, depth):
        faces, textures = self._fill_back(faces, textures if rgb else None)
        # The reference's render_silhouettes / render_depth call rasterize_silhouettes / rasterize_depth with
        # (faces, image_size, anti_aliasing) only (N3R:61-80): those modes run with the module defaults near=0.1,
        # far=100, eps=1e-4 whatever the rasterizer was constructed with; self.near / self.far /
        # self.rasterizer_eps (1e-3) only reach render_rgb and render (N3R:82-105).  eps enters the backward as
        # dist +/- eps, so the silhouette / depth gradients depend on it.
        if rgb:
            near, far, eps = self.near, self.far, self.rasterizer_eps
        else:
            near, far, eps = DEFAULT_NEAR, DEFAULT_FAR, DEFAULT_EPS
        self._op = RasterizeRGBAD(self.image_size, self.anti_aliasing, near, far, eps,
                                  self.background_color if rgb else None, rgb, alpha, depth)
        return self._op(vertices_to_faces(vertices, faces), textures)

    def render_silhouettes(self, vertices, faces):
        return self._run(vertices, faces, None, False, True, False)['alpha']

    def render_depth(self, vertices, faces):
        return self._run(vertices, faces, None, False, False, True)['depth']

    def render_rgb(self, vertices, faces, textures):
        return self._run(vertices, faces, textures, True, False, False)['rgb']

    def render(self, vertices, faces, textures):
        out = self._run(vertices, faces, textures, True, True, True)
        return out['rgb'], out['depth'], out['alpha']

    def backward(self, grad_rgb=None, grad_silhouettes=None, grad_depth=None):
        """Image gradients of the last render -> (grad_face_vertices [B,NF',3,3], grad_textures or None) on the
        device; NF' counts the back faces that fill_back appended (the caller folds them back)."""
        if getattr(self, "_op", None) is None:
            raise RuntimeError("backward before a render")
        return self._op.backward(grad_rgb, grad_silhouettes, grad_depth)
