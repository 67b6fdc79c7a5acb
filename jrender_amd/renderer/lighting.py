"""Lighting — NumPy host-side mirror of jrender/renderer/lighting/ (lighting.py:157-223,
directional_lighting.py:5-145, ambient_lighting.py:4-10).

O(NF) elementwise shading folded into the textures BEFORE rasterisation: ambient + one
directional light, Lambert or Cook-Torrance (GGX / Smith / Schlick) when ``mesh.with_specular``.
Like the reference, ``Lighting.__call__`` MUTATES ``mesh.textures`` (lighting.py:203-204).
Normal-map, SSS and G-buffer modes are outside the accelerated SoftRas path.
"""
import numpy as np

F32 = np.float32


def _relu(x):
    return np.maximum(x, 0)


def _normalize(v, axis, eps=1e-12):
    n = np.sqrt(np.sum(v * v, axis=axis, keepdims=True))
    return v / np.maximum(n, eps)


def ambient_lighting(light, light_intensity=0.5, light_color=(1, 1, 1)):
    light_color = np.asarray(light_color, F32)
    if light_color.ndim == 1:
        light_color = light_color[None]
    return light + F32(light_intensity) * light_color[:, None]


def GGX(N, H, roughness):
    a = roughness * roughness
    a2 = a * a
    NdotH = _relu(np.sum(N * H, axis=2))
    NdotH2 = (NdotH * NdotH)[:, :, None]
    denom = NdotH2 * (a2 - 1.0) + 1.0
    return a2 / (3.1415 * denom * denom)


def SchlickGGX(NdotV, roughness):
    r = roughness + 1.0
    k = (r * r) / 8.0
    NdotV = NdotV[:, :, None]
    return NdotV / (NdotV * (1.0 - k) + k)


def GeometrySmith(N, V, L, roughness):
    return SchlickGGX(_relu(np.sum(N * L, axis=2)), roughness) * SchlickGGX(_relu(np.sum(N * V, axis=2)), roughness)


def fresnelSchlick(cosTheta, F0):
    return F0 + (1.0 - F0) * np.power(1.0 - cosTheta, 5)[:, :, None]


def directional_lighting(diffuseLight, specularLight, normals, light_intensity=0.5, light_color=(1, 1, 1),
                         light_direction=(0, 1, 0), positions=None, eye=None, with_specular=False,
                         metallic_textures=None, roughness_textures=None):
    """directional_lighting.py:54-145 for per-face / per-vertex normals [B,N,3]."""
    light_color = np.asarray(light_color, F32)
    light_direction = _normalize(np.asarray(light_direction, F32), 0)
    if light_color.ndim == 1:
        light_color = light_color[None]
    if light_direction.ndim == 1:
        light_direction = light_direction[None]
    cosine = _relu(np.sum(normals * light_direction, axis=2))
    if with_specular and metallic_textures is not None and metallic_textures.ndim == 4:
        total = metallic_textures.shape[2] * 1.0
        metallic_textures = np.sum(metallic_textures, axis=2) / total
        roughness_textures = np.sum(roughness_textures, axis=2) / total
    elif with_specular and metallic_textures is not None and metallic_textures.ndim == 6:
        metallic_textures = metallic_textures.mean(axis=(2, 3, 4))          # n3mr cube textures
        roughness_textures = roughness_textures.mean(axis=(2, 3, 4))
    if with_specular and eye is not None and positions is not None and metallic_textures is not None \
            and roughness_textures is not None:
        eye = np.asarray(eye, F32)
        if eye.ndim == 1:
            eye = eye[None]
        if eye.ndim == 2:
            eye = eye[:, None]
        N = normals
        V = _normalize(eye - positions, 2)
        L = light_direction
        H = _normalize(V + L, 2)
        metallic, roughness = metallic_textures, roughness_textures
        F0 = np.asarray((0.4, 0.4, 0.4), F32)[None, None] * (1 - metallic) + \
            np.asarray((1.0, 1.0, 1.0), F32)[None, None] * metallic
        radiance = F32(light_intensity) * (light_color[:, None] * cosine[:, :, None])
        NDF = GGX(N, H, roughness)
        G = GeometrySmith(N, V, L, roughness)
        F = fresnelSchlick(_relu(np.sum(H * V, axis=2)), F0)
        KD = (1.0 - F) * (1.0 - metallic)
        diffuseLight = diffuseLight + KD * radiance
        denominator = (4.0 * _relu(np.sum(N * V, axis=2)) * _relu(np.sum(N * L, axis=2)))[:, :, None]
        specular = NDF * G * F / np.maximum(denominator, 0.01)
        specularLight = specularLight + specular * radiance
    else:
        diffuseLight = diffuseLight + F32(light_intensity) * (light_color[:, None] * cosine[:, :, None])
    return [diffuseLight.astype(F32), specularLight.astype(F32)]


def lighting(faces, textures, intensity_ambient=0.5, intensity_directional=0.5, color_ambient=(1, 1, 1),
             color_directional=(1, 1, 1), direction=(0, 1, 0)):
    """The legacy functional lighting of NMR cube textures (lighting.py:14-54): faces [B,NF,3,3], textures
    [B,NF,t,t,t,3] scaled in place by ambient + Lambert light of the face normal (v0-v1) x (v2-v1)."""
    faces = np.asarray(faces, F32)
    bs, nf = faces.shape[:2]
    color_ambient, color_directional = np.asarray(color_ambient, F32), np.asarray(color_directional, F32)
    direction = np.asarray(direction, F32)
    if color_ambient.ndim == 1:
        color_ambient = color_ambient[None]
    if color_directional.ndim == 1:
        color_directional = color_directional[None]
    if direction.ndim == 1:
        direction = direction[None]
    light = np.zeros((bs, nf, 3), F32)
    if intensity_ambient != 0:
        light = light + F32(intensity_ambient) * color_ambient[:, None]
    if intensity_directional != 0:
        f = faces.reshape(bs * nf, 3, 3)
        normals = _normalize(np.cross(f[:, 0] - f[:, 1], f[:, 2] - f[:, 1]), 1, eps=1e-5).reshape(bs, nf, 3)
        cos = _relu(np.sum(normals * direction[:, None], axis=2))
        light = light + F32(intensity_directional) * (color_directional[:, None] * cos[:, :, None])
    textures *= light[:, :, None, None, None].astype(F32)
    return textures


class AmbientLighting:
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1)):
        self.light_intensity = light_intensity
        self.light_color = light_color

    def __call__(self, light):
        return ambient_lighting(light, self.light_intensity, self.light_color)


class DirectionalLighting:
    def __init__(self, light_intensity=0.5, light_color=(1, 1, 1), light_direction=(0, 1, 0)):
        self.light_intensity = light_intensity
        self.light_color = light_color
        self.light_direction = light_direction

    def __call__(self, diffuseLight, specularLight, normals, positions=None, eye=None,
                 with_specular=False, metallic_textures=None, roughness_textures=None):
        return directional_lighting(diffuseLight, specularLight, normals, self.light_intensity,
                                    self.light_color, self.light_direction, positions, eye,
                                    with_specular, metallic_textures, roughness_textures)


class Lighting:
    def __init__(self, light_mode='surface', intensity_ambient=0.5, color_ambient=[1, 1, 1],
                 intensity_directionals=0.5, color_directionals=[1, 1, 1], directions=[0, 1, 0],
                 Gbuffer='None', transform=None):
        if light_mode not in ['surface', 'vertex']:
            raise ValueError('Lighting mode only support surface and vertex')
        if Gbuffer not in ('None', None):
            raise NotImplementedError("G-buffer lighting modes belong to render2 (out of scope)")
        self.Gbuffer = Gbuffer
        self.transform = transform
        self.light_mode = light_mode
        self.ambient = AmbientLighting(intensity_ambient, color_ambient)
        self.directionals = [DirectionalLighting(intensity_directionals, color_directionals, directions)]

    def __call__(self, mesh, eyes=None):
        """lighting.py:177-223."""
        # d(lit)/d(textures) belongs to THIS call or to none: Renderer.grad_textures must not multiply by the mask of
        # an earlier render (vertex mode and unlit modes leave it at None and grad_textures raises "render first")
        self._last = None
        if self.light_mode == 'surface':
            diffuse = self.ambient(np.zeros(mesh.faces.shape, F32))
            specular = np.zeros(mesh.faces.shape, F32)
            centres = np.sum(mesh.face_vertices, axis=2) / F32(3.0)
            for d in self.directionals:
                diffuse, specular = d(diffuse, specular, mesh.surface_normals, centres, eyes,
                                      mesh.with_specular, mesh.metallic_textures, mesh.roughness_textures)
            diffuse, specular = diffuse[:, :, None], specular[:, :, None]
            if mesh.textures.ndim == 6:
                diffuse, specular = diffuse[:, :, None, None], specular[:, :, None, None]
            if mesh.textures.ndim in (4, 6):
                pre = mesh.textures * diffuse + np.ones_like(mesh.textures) * specular
                # d(lit)/d(textures), kept for Renderer.grad_textures (autograd in the reference)
                self._last = {"dlit": (np.broadcast_to(diffuse, pre.shape) * ((pre > 0.0) & (pre < 1.0))).astype(F32)}
                mesh.textures = np.clip(pre, 0.0, 1.0)
        elif self.light_mode == 'vertex':
            diffuse = self.ambient(np.zeros(mesh.vertices.shape, F32))
            specular = np.zeros(mesh.vertices.shape, F32)
            for d in self.directionals:
                diffuse, specular = d(diffuse, specular, mesh.vertex_normals, mesh.vertices, eyes,
                                      mesh.with_specular, mesh.metallic_textures, mesh.roughness_textures)
            # lighting.py:212-218 handles 4-D and 6-D textures only; per-vertex colours [B,NV,3] (what Mesh
            # stores for texture_type='vertex') match neither branch and stay UNLIT in the reference — kept.
            if mesh.textures.ndim == 4:
                mesh.textures = np.clip(mesh.textures * diffuse[:, :, None] +
                                        np.ones_like(mesh.textures) * specular[:, :, None], 0.0, 1.0)
            elif mesh.textures.ndim == 6:
                mesh.textures = np.clip(mesh.textures * diffuse[:, :, None, None, None] +
                                        np.ones_like(mesh.textures) * specular[:, :, None, None, None], 0.0, 1.0)
        return mesh

    execute = __call__
