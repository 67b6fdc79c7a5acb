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


def directional_lighting_backward(g_diffuse, g_specular, normals, light_intensity=0.5, light_color=(1, 1, 1),
                                  light_direction=(0, 1, 0), positions=None, eye=None, metallic_textures=None,
                                  roughness_textures=None):
    """VJP of the Cook-Torrance branch of ``directional_lighting`` (directional_lighting.py:86-130) with respect to the
    per-face metallic and roughness values: upstream gradients of the two lights [B,N,3] -> (g_metallic, g_roughness) in the
    shape of the texture arguments ([B,N,T,1]: every texel gets 1/T of the face's gradient, the forward averages them,
    directional_lighting.py:70-73).  What demo5-optim_metallic_textures.py / demo6-optim_roughness_textures.py get from Jittor's
    autograd.  Same expressions as the forward, differentiated term by term:
        F0 = 0.4 (1 - m) + m              F = F0 + (1 - F0) p5,  p5 = (1 - relu(H.V))^5          dF/dm = 0.6 (1 - p5)
        KD = (1 - F)(1 - m)               dKD/dm = -(1 - m) dF/dm - (1 - F)
        NDF = a2 / (3.1415 d^2),  a2 = r^4,  d = NdotH^2 (a2 - 1) + 1                           dNDF/dr = 4 r^3 (1 / (3.1415 d^2) - 2 a2 NdotH^2 / (3.1415 d^3))
        S(x) = x / (x (1 - k) + k),  k = (r + 1)^2 / 8                                           dS/dr = -x (1 - x) / (x (1 - k) + k)^2 * (r + 1) / 4
        specular = NDF S(N.L) S(N.V) F / max(4 relu(N.V) relu(N.L), 0.01)
    """
    dt = np.result_type(np.asarray(normals).dtype, F32)
    light_color = np.asarray(light_color, dt)
    light_direction = _normalize(np.asarray(light_direction, dt), 0)
    if light_color.ndim == 1:
        light_color = light_color[None]
    if light_direction.ndim == 1:
        light_direction = light_direction[None]
    m_in, r_in = np.asarray(metallic_textures), np.asarray(roughness_textures)
    if m_in.ndim == 4:
        metallic, roughness = np.sum(m_in, axis=2) / (m_in.shape[2] * 1.0), np.sum(r_in, axis=2) / (r_in.shape[2] * 1.0)
    elif m_in.ndim == 6:
        metallic, roughness = m_in.mean(axis=(2, 3, 4)), r_in.mean(axis=(2, 3, 4))
    else:
        metallic, roughness = m_in, r_in
    eye = np.asarray(eye, dt)
    if eye.ndim == 1:
        eye = eye[None]
    if eye.ndim == 2:
        eye = eye[:, None]
    N, L = normals, light_direction
    V = _normalize(eye - positions, 2)
    H = _normalize(V + L, 2)
    cosine = _relu(np.sum(N * L, axis=2))
    radiance = light_intensity * (light_color[:, None] * cosine[:, :, None])
    NdotH2 = (_relu(np.sum(N * H, axis=2)) ** 2)[:, :, None]
    NdotV, NdotL = _relu(np.sum(N * V, axis=2))[:, :, None], _relu(np.sum(N * L, axis=2))[:, :, None]
    p5 = np.power(1.0 - _relu(np.sum(H * V, axis=2)), 5)[:, :, None]
    F0 = 0.4 * (1 - metallic) + 1.0 * metallic
    F = F0 + (1.0 - F0) * p5
    dF_dm = 0.6 * (1.0 - p5)
    dKD_dm = -(1.0 - metallic) * dF_dm - (1.0 - F)
    a2 = (roughness * roughness) ** 2
    d = NdotH2 * (a2 - 1.0) + 1.0
    NDF = a2 / (3.1415 * d * d)
    dNDF_dr = 4.0 * roughness ** 3 * (1.0 / (3.1415 * d * d) - 2.0 * a2 * NdotH2 / (3.1415 * d ** 3))
    k = (roughness + 1.0) ** 2 / 8.0
    dk_dr = (roughness + 1.0) / 4.0

    def S(x):
        den = x * (1.0 - k) + k
        return x / den, -x * (1.0 - x) / (den * den) * dk_dr

    SL, dSL = S(NdotL)
    SV, dSV = S(NdotV)
    G, dG_dr = SL * SV, dSL * SV + SL * dSV
    den = np.maximum(4.0 * NdotV * NdotL, 0.01)
    gs_rad = g_specular * radiance
    g_m = np.sum(g_diffuse * radiance * dKD_dm + gs_rad * (NDF * G / den) * dF_dm, axis=2, keepdims=True)
    g_r = np.sum(gs_rad * (F / den) * (dNDF_dr * G + NDF * dG_dr), axis=2, keepdims=True)

    def spread(g, like):
        if like.ndim == 4:
            return np.broadcast_to((g / (like.shape[2] * 1.0))[:, :, None], like.shape).astype(like.dtype if like.dtype.kind == 'f' else F32)
        if like.ndim == 6:
            n = like.shape[2] * like.shape[3] * like.shape[4] * 1.0
            return np.broadcast_to((g / n)[:, :, None, None, None], like.shape).astype(like.dtype if like.dtype.kind == 'f' else F32)
        return g.astype(like.dtype if like.dtype.kind == 'f' else F32)

    return spread(g_m, m_in), spread(g_r, r_in)


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

    def backward(self, g_diffuse, g_specular, normals, positions, eye, metallic_textures, roughness_textures):
        return directional_lighting_backward(g_diffuse, g_specular, normals, self.light_intensity, self.light_color,
                                             self.light_direction, positions, eye, metallic_textures, roughness_textures)


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
                open_ = (pre > 0.0) & (pre < 1.0)
                # d(lit)/d(textures), kept for Renderer.grad_textures (autograd in the reference) ...
                self._last = {"dlit": (np.broadcast_to(diffuse, pre.shape) * open_).astype(F32)}
                # ... and what the VJP with respect to metallic / roughness needs (Renderer.grad_material; demo5 / demo6)
                if mesh.with_specular and eyes is not None and mesh.metallic_textures is not None and mesh.roughness_textures is not None:
                    self._last.update(open=open_, textures=np.array(mesh.textures, F32), normals=np.array(mesh.surface_normals, F32),
                                      centres=np.array(centres, F32), eyes=np.array(eyes, F32),
                                      metallic=np.array(mesh.metallic_textures, F32), roughness=np.array(mesh.roughness_textures, F32))
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

    def backward_material(self, grad_lit):
        """d(loss)/d(metallic_textures), d(loss)/d(roughness_textures) of the last 'surface' call with a specular mesh, for an
        upstream gradient of the LIT textures [B,NF,T,3] (lighting.py:203-204: lit = clip(textures * diffuse + specular, 0, 1);
        the reference differentiates this with Jittor's autograd in demo5 / demo6)."""
        L = getattr(self, "_last", None)
        if L is None or "metallic" not in L:
            raise RuntimeError("backward_material: call the lighting in 'surface' mode on a mesh with metallic / roughness textures first")
        g = np.asarray(grad_lit, F32).reshape(L["open"].shape) * L["open"]
        red = tuple(range(2, g.ndim - 1))                         # the texel axes
        g_diffuse = np.sum(g * L["textures"], axis=red)           # [B,NF,3]
        g_specular = np.sum(g, axis=red)
        gm = np.zeros_like(L["metallic"]); gr = np.zeros_like(L["roughness"])
        for d in self.directionals:
            a, b = d.backward(g_diffuse, g_specular, L["normals"], L["centres"], L["eyes"], L["metallic"], L["roughness"])
            gm += a; gr += b
        return gm.astype(F32), gr.astype(F32)
