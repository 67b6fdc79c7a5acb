"""N3mrRasterizer — host-side mirror of jrender/renderer/dr/n3mr/rasterizer.py (N3R:9-105)."""
import numpy as np

from .n3mr import rasterize, rasterize_depth, rasterize_rgbad, rasterize_silhouettes
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

    def render_silhouettes(self, vertices, faces):
        faces, _ = self._fill_back(faces)
        return rasterize_silhouettes(vertices_to_faces(vertices, faces), self.image_size, self.anti_aliasing)

    def render_depth(self, vertices, faces):
        faces, _ = self._fill_back(faces)
        return rasterize_depth(vertices_to_faces(vertices, faces), self.image_size, self.anti_aliasing)

    def render_rgb(self, vertices, faces, textures):
        faces, textures = self._fill_back(faces, textures)
        return rasterize(vertices_to_faces(vertices, faces), textures, self.image_size, self.anti_aliasing,
                         self.near, self.far, self.rasterizer_eps, self.background_color)

    def render(self, vertices, faces, textures):
        faces, textures = self._fill_back(faces, textures)
        out = rasterize_rgbad(vertices_to_faces(vertices, faces), textures, self.image_size, self.anti_aliasing,
                              self.near, self.far, self.rasterizer_eps, self.background_color)
        return out['rgb'], out['depth'], out['alpha']
