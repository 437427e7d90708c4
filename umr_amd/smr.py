"""nnutils.smr.SoftRenderer on MI355X.

Same constructor, methods and return values as the reference wrapper (nnutils/smr.py:49-87, which
drives external/SoftRas sr.SoftRenderer: renderer.py:48-98, lighting.py:50-57, transform.py:41-48,
rasterizer.py:42-55), so loss code written against the reference runs unchanged:

    imgs [N,4,H,H], p2f [N,F,2], aggr [N,2,2H,2H] = SoftRenderer(H, 'softmax')(verts, faces, cams, textures)

Execution: one projection + gather + surface-lighting kernel, one face-setup + one tiled raster kernel with the
2x2 anti-aliasing pool fused; backward is the mirror image.  No CPU path.
"""
import torch

from . import functional as UF


class SoftRenderer(torch.nn.Module):
    def __init__(self, img_size=256, render_type='softmax', background_color=[0, 0, 0], sigma_val=1e-5,
                 gamma_val=1e-4, dist_eps=1e-10, anti_aliasing=True):
        super(SoftRenderer, self).__init__()
        self.img_size = img_size
        self.render_type = render_type
        self.background_color = list(background_color)
        self.sigma_val, self.gamma_val, self.dist_eps = sigma_val, gamma_val, dist_eps
        self.anti_aliasing = anti_aliasing
        # sr.SoftRenderer defaults (renderer.py:58-60) with smr.py:63's brighter ambient
        self.light_intensity_ambient = 0.8
        self.light_intensity_directional = 0.5
        self.light_color = [1., 1., 1.]
        self.light_direction = [0., 1., 0.]
        self.eye_z = -2.732          # smr.py:60
        self.offset_z = 5.           # smr.py:66
        self.near, self.far, self.eps = 1., 100., 1e-3   # renderer.py:48-49
        self.need_p2f = True         # set False to skip the p2f accumulators (callers that discard them)
        # set True when only imgs[:, 3] is consumed (mask / GAN-view renders): the silhouette-only kernels run,
        # rgb channels of the returned image hold the background colour, p2f is zeros and aggr is None
        self.alpha_only = False
        # hard renderer only: set True when only aggr (the face-id / depth planes) is consumed (TexCycle): the
        # visibility-only kernel runs and the returned image is None
        self.ids_only = False

    def ambient_light_only(self):
        """smr.py:68-71."""
        self.light_intensity_ambient = 1
        self.light_intensity_directional = 0

    def set_bgcolor(self, color):
        """smr.py:73-74."""
        self.background_color = list(color)

    def project_points(self, verts, cams):
        """smr.py:76-78 -> [N,V,2]."""
        return UF.project_points(verts, cams, 2, 0.0)

    def _const(self, ref, values):
        """Small device constant (light colour / direction, background), uploaded once per (device, value): building
        it from a Python list on every call is a pageable host-to-device copy that synchronises the stream and cannot
        be captured in a HIP graph."""
        key = (ref.device, tuple(float(v) for v in values))
        cache = self.__dict__.setdefault("_const_cache", {})
        if key not in cache:
            cache[key] = torch.tensor(key[1], dtype=torch.float32, device=ref.device)
        return cache[key]

    def silhouettes(self, vertices, faces, cams):
        """Alpha channel of forward() only -> [N,S,S] (S = img_size): what the mask / unseen-view renders consume
        (loss_utils.py:266, train_s1.py:200,236).  Same mesh-group convention as forward(): cams [N,7] may hold several
        views per mesh (view n renders mesh n // (N / M)), e.g. the seen and the rotated camera of each image interleaved,
        so both silhouettes of a training step come out of ONE launch per direction."""
        faces = faces.int().contiguous()
        _, face_out, _ = UF.project_faces(vertices, cams, faces, self.offset_z, self.eye_z, False)
        size = self.img_size * (2 if self.anti_aliasing else 1)
        return UF.silhouette(face_out, size, self.near, self.far, True, self.eps, self.sigma_val,
                                           self.dist_eps, self.gamma_val, self.anti_aliasing)

    def forward(self, vertices, faces, cams, textures=None, with_visibility=False, detach_rgb_geometry=False, lean_state=False):
        """vertices [N,V,3] float, faces [N,F,3] integer, cams [N,7] = [s,tx,ty,qw,qx,qy,qz],
        textures None | [N,F,TS,3].
        K camera hypotheses per mesh without the reference's x K repeats (loss_utils.py:260-262, 303-306): pass
        vertices / faces as [N/K,...] and / or textures as [N/K,F,TS,3] with cams [N,7] ordered mesh-major
        (view n = mesh n // K); outputs are per view, gradients come back summed over the K views.
        with_visibility (soft-max renders): a 4th return value, the aggrs_info [N,2,IS,IS] = (nearest depth, face id | -1) a
        SoftRenderer(img_size, 'hard') would return for the same mesh and cameras -- train_s1.py:217-224 renders both; here
        the z-buffer falls out of the textured render's own kernel visits.
        detach_rgb_geometry (soft-max renders with ambient-only lighting): gradients of imgs[:, 0:3] reach the textures only,
        as if vertices and cams had been passed detached, while imgs[:, 3] keeps its gradient to vertices and cams -- the mask
        render (train_s1.py:199, loss_utils.py:265) and the textured render of the same views with detached geometry (:217,
        :313) as ONE render.
        lean_state (with detach_rgb_geometry and anti_aliasing): for callers that read imgs, p2f and the visible-face ids only --
        `aggr` comes back None and the 4th value is the id plane [N,IS,IS] alone (UF.soft_rasterize lean_state)."""
        faces = faces.int().contiguous()                                  # smr.py:81
        N = cams.shape[0]
        if self.ids_only and self.render_type == 'hard':
            with torch.no_grad():
                _, face_out, _ = UF.project_faces(vertices, cams, faces, self.offset_z, self.eye_z, False)
                size = self.img_size * (2 if self.anti_aliasing else 1)
                aggr = UF.visibility(face_out, size, self.near, self.far, True, self.eps, self.sigma_val, self.dist_eps,
                                     self.gamma_val)
            return None, aggr.new_zeros(N, faces.shape[1], 2), aggr                # hard p2f is identically 0
        if self.alpha_only:
            alpha = self.silhouettes(vertices, faces, cams)
            S = alpha.shape[1]
            bg = self._const(alpha, self.background_color).view(1, 3, 1, 1).expand(N, 3, S, S)
            imgs = torch.cat([bg, alpha.unsqueeze(1)], dim=1)
            return imgs, alpha.new_zeros(N, faces.shape[1], 2), None
        directional = self.light_intensity_directional != 0
        if detach_rgb_geometry and (directional or self.render_type != 'softmax'):
            raise RuntimeError("detach_rgb_geometry: soft-max renders with ambient-only lighting (the per-face directional "
                               "light depends on the vertices)")
        # lighting.py:50-57: with a directional term the per-face light comes out of the projection kernel (normals of
        # the projected faces); ambient-only lighting is a constant factor
        light = (self.light_intensity_ambient, self.light_intensity_directional, self.light_color,
                 self.light_direction) if directional else None
        _, face_out, face_light = UF.project_faces(vertices, cams, faces, self.offset_z, self.eye_z, False,
                                                                light)
        F = faces.shape[1]
        if directional:
            # lighting.py:56: textures * light[:, :, None, :]; textures=None means all-ones (mesh.py:46-50), i.e. the
            # light itself is the texture.  Per-view lighting needs per-view texels: shared sets broadcast here.
            lit = face_light[:, :, None, :]
            if textures is None:
                textures = lit
            else:
                if textures.shape[0] != N:
                    textures = textures.repeat_interleave(N // textures.shape[0], dim=0)
                textures = textures * lit
        else:
            if textures is None:                                          # mesh.py:46-50
                textures = self._const(vertices, [1.0] * (F * 3)).view(1, F, 1, 3)
            if self.light_intensity_ambient != 1 or any(c != 1 for c in self.light_color):
                textures = textures * (self.light_intensity_ambient * self._const(textures, self.light_color))
        size = self.img_size * (2 if self.anti_aliasing else 1)          # rasterizer.py:43
        if with_visibility and self.render_type != 'softmax':
            raise RuntimeError("with_visibility: only for render_type 'softmax'")
        return UF.soft_rasterize(face_out, textures, size, self.background_color, self.near, self.far, True,
                                 self.eps, self.sigma_val, 'euclidean', self.dist_eps, self.gamma_val,
                                 self.render_type, 'prod', 'surface', pool=self.anti_aliasing,
                                 need_p2f=self.need_p2f, want_visibility=with_visibility,
                                 detach_rgb_geometry=detach_rgb_geometry, lean_state=lean_state)
