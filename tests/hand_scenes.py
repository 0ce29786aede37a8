"""Scene descriptions assembled by hand (ctypes structures of include/rt_capi.h), small enough that the value of a
pixel can be derived on paper.  Used by the known-answer tests of the oracle (CPU) and of the HIP path (GPU)."""
import ctypes as C

import numpy as np

from cudaraytracer_amd import _native as N

DIFFUSE, GLASS, MIRROR = 0, 1, 2  # RT_MAT_*
BG = (0.2, 0.7, 0.8)


def texture(pixels, mode=1):
    """pixels: [h][w] of (r, g, b) bytes -> (RtTexture, keep-alive buffer).  mode 1 = CLAMP, 0 = WRAP."""
    a = np.asarray(pixels, dtype=np.uint8)
    h, w = a.shape[:2]
    rgba = np.concatenate([a, np.full((h, w, 1), 255, dtype=np.uint8)], axis=2).copy()
    buf = (C.c_ubyte * rgba.size).from_buffer_copy(rgba.tobytes())
    return N.RtTexture(w, h, C.cast(buf, C.POINTER(C.c_ubyte)), mode), buf


WHITE = [[(255, 255, 255)]]
BLACK = [[(0, 0, 0)]]
FLAT_NORMAL = [[(128, 128, 255)]]


class Mesh:
    """One mesh: triangles as lists of three vertices {pos, uv, normal, tangent, bitangent}; tangent and bitangent default
    to zero, so that the shading normal is the interpolated vertex normal whatever the normal map holds.  `nodes`: None = all triangles in one leaf,
    else a list of (bbmin, bbmax, offset, primCnt, axis) in the reference's depth-first order."""

    def __init__(self, triangles, material=DIFFUSE, ior=1.3, ks=0.0, spec_exp=25.0, diffuse=WHITE, specular=BLACK, normals=FLAT_NORMAL,
                 nodes=None, mode=1, transform_inverse=None, world_box=None, device_bvh=False):
        self.triangles, self.material, self.ior, self.ks, self.spec_exp = triangles, material, ior, ks, spec_exp
        self.maps = (diffuse, normals, specular)
        self.nodes, self.mode = nodes, mode
        # row-major, row-vector convention (cudaTracer/Mat.cuh:127-149); with it, world_box = (min, max) of the shape
        self.transform_inverse, self.world_box = transform_inverse, world_box
        self.device_bvh = device_bvh  # no nodes at all: rt_scene_upload builds the BVH on the GPU


def tri(p0, p1, p2, uv=((0, 0), (1, 0), (0, 1)), normal=(0, 0, -1), tangent=(0, 0, 0), bitangent=(0, 0, 0)):
    return [{"pos": p, "uv": t, "normal": normal, "tangent": tangent, "bitangent": bitangent} for p, t in zip((p0, p1, p2), uv)]


class Scene:
    def __init__(self, w, h, meshes=(), spheres=(), lights=(), max_depth=0, bias=1e-4, gamma=0, camera=None, origin=(0, 0, 0)):
        self.keep = []
        self.width, self.height = w, h
        nm, ns = len(meshes), len(spheres)
        self.meshes = (N.RtMesh * max(nm, 1))()
        self.shapes = (N.RtShape * (nm + ns))()
        lo = np.full(3, np.finfo(np.float32).max, dtype=np.float32)
        hi = -lo
        for mi, m in enumerate(meshes):
            nt = len(m.triangles)
            verts = (N.RtVertex * (3 * nt))()
            tris = (N.RtTriangle * nt)()
            pos = []
            for t, corners in enumerate(m.triangles):
                for k, c in enumerate(corners):
                    v = verts[3 * t + k]
                    v.position[:] = c["pos"]
                    v.texCoord[:] = c["uv"]
                    v.normal[:] = c["normal"]
                    v.tangent[:] = c.get("tangent", (0, 0, 0))
                    v.bitangent[:] = c.get("bitangent", (0, 0, 0))
                    tris[t].i[k] = 3 * t + k
                    pos.append(c["pos"])
            pos = np.asarray(pos, dtype=np.float32)
            b0, b1 = pos.min(0), pos.max(0)
            spec = m.nodes if m.nodes is not None else [(b0.tolist(), b1.tolist(), 0, nt, 0)]
            nodes = (N.RtLinearNode * len(spec))()
            for i, (mn, mx, off, cnt, axis) in enumerate(spec):
                nodes[i].bbmin[:], nodes[i].bbmax[:] = mn, mx
                nodes[i].offset, nodes[i].primCnt, nodes[i].axis = off, cnt, axis
            dm = self.meshes[mi]
            dm.triangleCnt, dm.triangles, dm.vertexCnt, dm.vertices = nt, tris, 3 * nt, verts
            dm.nodeCnt, dm.nodes = len(spec), nodes
            if m.device_bvh:
                dm.nodeCnt, dm.nodes = 0, None
            for name, px in zip(("diffuse", "normals", "specular"), m.maps):
                tex, buf = texture(px, m.mode)
                setattr(dm, name, tex)
                self.keep.append(buf)
            ti = np.eye(4, dtype=np.float32) if m.transform_inverse is None else np.asarray(m.transform_inverse, dtype=np.float32)
            dm.transformInverse[:] = ti.ravel().tolist()
            if m.world_box is not None:
                b0, b1 = np.float32(m.world_box[0]), np.float32(m.world_box[1])
            s = self.shapes[mi]
            s.kind, s.meshIdx, s.tag = 1, mi, mi
            s.material.materialType, s.material.ior, s.material.Kd, s.material.Ks, s.material.specularExponent = m.material, m.ior, 0.8, m.ks, m.spec_exp
            s.material.diffuseColor[:] = [0.2, 0.2, 0.2]
            s.bbmin[:], s.bbmax[:] = b0.tolist(), b1.tolist()
            lo, hi = np.minimum(lo, b0), np.maximum(hi, b1)
            self.keep += [verts, tris, nodes]
        for si, (c, r, mat, ior, color) in enumerate(spheres):
            s = self.shapes[nm + si]
            s.kind, s.meshIdx, s.tag = 0, -1, nm + si
            s.center[:] = c
            s.radius, s.radius2 = r, float(np.float32(r) * np.float32(r))
            s.material.materialType, s.material.ior, s.material.Kd, s.material.Ks, s.material.specularExponent = mat, ior, 0.8, 0.0, 25.0
            s.material.diffuseColor[:] = color
            b0, b1 = np.float32(c) - np.float32(r), np.float32(c) + np.float32(r)
            s.bbmin[:], s.bbmax[:] = b0.tolist(), b1.tolist()
            lo, hi = np.minimum(lo, b0), np.maximum(hi, b1)
        self.nodes = (N.RtLinearNode * 1)()
        self.nodes[0].bbmin[:], self.nodes[0].bbmax[:] = lo.tolist(), hi.tolist()
        self.nodes[0].offset, self.nodes[0].primCnt = 0, nm + ns
        self.lights = (N.RtLight * max(len(lights), 1))()
        for i, (p, intensity, power) in enumerate(lights):
            self.lights[i].position[:] = p
            self.lights[i].intensity[:] = intensity
            self.lights[i].power = power
        self.desc = N.RtSceneDesc(self.lights if lights else None, len(lights), self.shapes, nm + ns, self.meshes if nm else None, nm,
                                  self.nodes, 1)
        self.opts = N.RtDeviceOpts()
        o = self.opts
        o.imageAspectRatio, o.scale = w / h, 1.0
        o.width, o.height = w, h
        o.backgroundColor[:] = BG
        o.shadowBias, o.blockSize, o.maxDepth, o.samples = bias, 4, max_depth, 1
        o.gammaCorrection, o.useSceneBvh, o.useMeshBvhs = gamma, 1, 1
        self.camera = N.RtCamera()
        self.camera.orig[:] = origin
        self.camera.camera[:] = (np.eye(4, dtype=np.float32) if camera is None else np.asarray(camera, dtype=np.float32)).ravel().tolist()

    def render_oracle(self, oracle):
        rgba, keys, _ = oracle.render(C.pointer(self.desc), C.pointer(self.opts), C.pointer(self.camera), self.width, self.height, threads=1,
                                      shadow_any_hit=True)
        return rgba, keys

    def render_gpu(self, ctx):
        w, h = self.width, self.height
        dscene = ctx.upload(C.pointer(self.desc))
        surf, pitch = ctx.surface(w, h)
        keys = ctx.alloc(w * h * 16)
        try:
            ctx.render(dscene, self.opts, C.pointer(self.camera), surf, pitch, keys)
            ctx.sync()
            return ctx.download_surface(surf, pitch, w, h), ctx.download(keys, (h, w, 4), np.int32)
        finally:
            ctx.free(surf)
            ctx.free(keys)
            ctx.free_scene(dscene)


# ---- the probes: (scene, what the centre pixel / the key plane must be), derived by hand -----------------------------
# A 1x1 frame has ONE ray: ndc = 2 * 0.5 - 1 = 0, direction (0, 0, 1) exactly.  Gamma is off, so a pixel is the sum
# the reference's diffuseAndGlossy forms.  With the light AT the camera (origin), a surface facing the camera at
# distance t, specular map black:  pixel = intensity * power / t * lambert * diffuse texel, lambert = 1.

FRONT = ((-3, -3, 5), (-3, 6, 5), (6, -3, 5))  # counter-clockwise seen from the origin: det = e1 . (d x e2) = 81 > 0, centroid (0, 0, 5)
BACK = (FRONT[0], FRONT[2], FRONT[1])
LIGHT_AT_CAMERA = [((0, 0, 0), (1, 1, 1), 5.0)]


# relative tolerance of a probe's pixel where the derivation's numbers are not exact in fp32 (default 2e-6)
RTOL = {"refract_oblique_incidence_follows_snell": 1e-5}


def probes():
    """name -> (Scene, expected centre pixel rgb or None, expected centre hit key or None)."""
    P = {}
    # Moller-Trumbore through the centroid: t = 5, so pixel = 1 * 5 / 5 * 1 * white = 1 (cudaTracer/Mesh.cuh:182-212)
    P["triangle_centroid"] = (Scene(1, 1, [Mesh([tri(*FRONT)])], lights=LIGHT_AT_CAMERA), (1.0, 1.0, 1.0), (0, 0, 1, 2))
    # the same triangle wound the other way is culled (det < 0): the ray leaves the scene
    P["triangle_backface"] = (Scene(1, 1, [Mesh([tri(*BACK)])], lights=LIGHT_AT_CAMERA), BG, (-1, -1, -1, -1))
    # sphere of radius 1 at z = 5 from outside: nearest root t = 4, normal (0, 0, -1): pixel = 5 / 4 * colour (Sphere.cuh:41-54)
    P["sphere_outside"] = (Scene(1, 1, spheres=[((0, 0, 5), 1.0, DIFFUSE, 1.3, (0.4, 0.2, 0.1))], lights=LIGHT_AT_CAMERA),
                           (0.5, 0.25, 0.125), (0, -1, -1, -1))
    # from the centre of the sphere: the negative root is dropped, t = 1 at (0, 0, 6); the normal points away from every
    # light that the surface can see, so the pixel is black - but it is a hit, not the background
    P["sphere_inside"] = (Scene(1, 1, spheres=[((0, 0, 5), 1.0, DIFFUSE, 1.3, (0.4, 0.2, 0.1))], lights=LIGHT_AT_CAMERA, origin=(0, 0, 5)),
                          (0.0, 0.0, 0.0), (0, -1, -1, -1))
    # sphere behind the ray origin: both roots negative
    P["sphere_behind"] = (Scene(1, 1, spheres=[((0, 0, 5), 1.0, DIFFUSE, 1.3, (0.4, 0.2, 0.1))], lights=LIGHT_AT_CAMERA, origin=(0, 0, 7)),
                          BG, (-1, -1, -1, -1))
    # A ray that lies IN a face of a box misses that box (cudaTracer/BoundingBox.cuh:115-139): the plane distance is
    # 0 * inf = NaN, which stays in tMin (near face: "tyMin > tMin" is false) and fails the final "tMin < ray.tMax".
    # The triangle's left edge is on x = 0, the ray runs along x = 0: the reference never tests the triangle.
    P["ray_in_box_face"] = (Scene(1, 1, [Mesh([tri((0, -3, 5), (0, 6, 5), (6, -3, 5))])], lights=LIGHT_AT_CAMERA), BG, (-1, -1, -1, -1))
    # mirror at normal incidence: the reflected ray leaves the scene, pixel = background * ((n - 1) / (n + 1))^2 = bg * 0.04
    # for n = 1.5 (cudaTracer/tracer.cu:67-89, 134-142)
    P["fresnel_normal_incidence"] = (Scene(1, 1, [Mesh([tri(*FRONT)], material=MIRROR, ior=1.5)], lights=LIGHT_AT_CAMERA, max_depth=1),
                                     tuple(0.04 * c for c in BG), (0, 0, 1, 2))
    # the same mirror with the shading normal along the ray (cos > 0: "leaving the denser medium") and the ray turned
    # 60 degrees about y: sin_t = 1.5 * sin 60 > 1, total internal reflection, kr = 1: pixel = background
    c60, s60 = 0.5, float(np.sqrt(0.75))
    roty = [[c60, 0, -s60, 0], [0, 1, 0, 0], [s60, 0, c60, 0], [0, 0, 0, 1]]  # row-vector convention: (0, 0, 1) -> (s60, 0, c60)
    big = ((-30, -30, 5), (-30, 60, 5), (60, -30, 5))
    P["fresnel_total_internal_reflection"] = (Scene(1, 1, [Mesh([tri(*big, normal=(0, 0, 1))], material=MIRROR, ior=1.5)], lights=LIGHT_AT_CAMERA,
                                                    max_depth=1, camera=roty), BG, (0, 0, 1, 2))
    # glass at normal incidence: the refracted ray keeps its direction (tracer.cu:102-115) and reaches a small lit sphere
    # behind the pane.  So does the "reflected" ray: the reference reflects the NORMAL about the view direction there
    # (tracer.cu:119, operands swapped), which at normal incidence points forward too.  Both children return the sphere's
    # radiance 5 / 2.5 * colour (light at z = 7, sphere front at z = 9.5): pixel = sphere * 0.04 + sphere * 0.96
    P["refract_normal_incidence"] = (Scene(1, 1, [Mesh([tri(*FRONT)], material=GLASS, ior=1.5)],
                                           spheres=[((0, 0, 10), 0.5, DIFFUSE, 1.3, (0.1, 0.2, 0.3))],
                                           lights=[((0, 0, 7), (1, 1, 1), 5.0)], max_depth=1),
                                     (0.2, 0.4, 0.6), (0, 0, 1, 2))
    # shadow rays accept a hit at t == tMax (Mesh.cuh:203-205): with bias 0 the shadow ray starts on the surface at
    # (0, 0, 5), the light is exactly 3 away at (0, 0, 2), and a triangle facing the surface lies in the plane z = 2:
    # every product of its Moller-Trumbore test is exact, t = 48 / 16 = 3 == tMax: in shadow, pixel black.  (That
    # triangle is back-facing for the primary ray, which therefore passes through it.)  Moved to z = 1.75 the triangle
    # is behind the light (t = 3.25 > tMax): lit, pixel = 5 / 3.
    # (A second, far-off triangle at z + 0.5 makes the occluder mesh's box start before the light: a box that begins
    # exactly at the reach fails the box test's strict "entry < tMax", cudaTracer/BoundingBox.cuh:138 - see the third probe.)
    occl = lambda z: [tri((-1, -1, z), (3, -1, z), (-1, 3, z)),  # noqa: E731  det = 16 > 0 for direction (0, 0, -1)
                      tri((10, 10, z + 0.5), (11, 10, z + 0.5), (10, 11, z + 0.5))]
    light = [((0, 0, 2), (1, 1, 1), 5.0)]
    P["shadow_hit_at_reach_is_accepted"] = (Scene(1, 1, [Mesh([tri(*FRONT)]), Mesh(occl(2.0))], lights=light, bias=0.0), (0.0, 0.0, 0.0), (0, 0, 1, 2))
    P["shadow_occluder_beyond_the_light"] = (Scene(1, 1, [Mesh([tri(*FRONT)]), Mesh(occl(1.75))], lights=light, bias=0.0),
                                             (5.0 / 3, 5.0 / 3, 5.0 / 3), (0, 0, 1, 2))
    # the same occluder alone in its mesh: the mesh's root box is flat at z = 2, its entry distance 3 is not < tMax = 3,
    # the box is skipped and the triangle never tested: lit
    P["shadow_box_starting_at_the_reach_is_skipped"] = (Scene(1, 1, [Mesh([tri(*FRONT)]), Mesh(occl(2.0)[:1])], lights=light, bias=0.0),
                                                        (5.0 / 3, 5.0 / 3, 5.0 / 3), (0, 0, 1, 2))
    # two identical triangles in one leaf: equal t, the later one replaces the earlier (t <= tMax, Mesh.cuh:203)
    P["equal_t_later_triangle_wins"] = (Scene(1, 1, [Mesh([tri(*FRONT), tri(*FRONT)])], lights=LIGHT_AT_CAMERA), (1.0, 1.0, 1.0), (0, 3, 4, 5))

    # ---- shading (cudaTracer/tracer.cu:144-200).  The hit is always (0, 0, 5) with n = (0, 0, -1), dir = (0, 0, 1).
    # Blinn-Phong with the light at the camera: L = (0, 0, -1), half vector = normalize(L - dir) = (0, 0, -1), its dot
    # with n is 1, 1^25 = 1; light component = 1 * 5 / 5 = 1.  Black diffuse map, specular map red channel 128:
    # pixel = 0 + 1 * (128 / 255) * 1 on all three channels (the mask is a scalar: Mesh.cuh getSpecularMask takes [X])
    P["specular_mask_scales_the_highlight"] = (Scene(1, 1, [Mesh([tri(*FRONT)], diffuse=BLACK, specular=[[(128, 7, 9)]])], lights=LIGHT_AT_CAMERA),
                                               (128 / 255,) * 3, (0, 0, 1, 2))
    # light at (3, 0, 1): L = (3, 0, -4) / 5, lambert = 0.8, light component 5 / 5 = 1; half vector = normalize((0.6, 0, -1.8)),
    # (h . n)^2 = 3.24 / 3.6 = 0.9 with shininess 2; white maps: pixel = 1 * 0.8 * 1 + 1 * 1 * 0.9 = 1.7
    P["lambert_and_blinn_phong_off_axis"] = (Scene(1, 1, [Mesh([tri(*FRONT)], specular=WHITE, spec_exp=2.0)], lights=[((3, 0, 1), (1, 1, 1), 5.0)]),
                                             (1.7, 1.7, 1.7), (0, 0, 1, 2))
    # two lights, summed in table order, each falling off with 1 / distance (not its square): at the camera 5 / 5 = 1,
    # at (0, 0, 2.5) with intensity (1, 0.5, 0.25): 5 / 2.5 = 2 times the intensity: pixel = (1 + 2, 1 + 1, 1 + 0.5)
    P["two_lights_with_linear_falloff"] = (Scene(1, 1, [Mesh([tri(*FRONT)])], lights=[((0, 0, 0), (1, 1, 1), 5.0), ((0, 0, 2.5), (1, 0.5, 0.25), 5.0)]),
                                           (3.0, 2.0, 1.5), (0, 0, 1, 2))
    # gamma (cudaTracer/Main.cu:86-92): pixel = radiance^(1 / 2.2); radiance = the diffuse texel 64 / 255
    P["gamma_is_a_power_of_one_over_2_2"] = (Scene(1, 1, [Mesh([tri(*FRONT)], diffuse=[[(64, 64, 64)]])], lights=LIGHT_AT_CAMERA, gamma=1),
                                             ((64 / 255) ** (1 / 2.2),) * 3, (0, 0, 1, 2))
    # normal map (cudaTracer/Mesh.cuh:73-95): texel (255, 255, .) -> (1, 1) after * 2 - 1; n' = normalize(n + 1 * t + 1 * b) with
    # t = (1, 0, 0), b = (0, 1, 0): (1, 1, -1) / sqrt 3; light at the camera: lambert = 1 / sqrt 3
    P["normal_map_tilts_the_normal"] = (Scene(1, 1, [Mesh([tri(*FRONT, tangent=(1, 0, 0), bitangent=(0, 1, 0))], normals=[[(255, 255, 255)]])],
                                              lights=LIGHT_AT_CAMERA), (3 ** -0.5,) * 3, (0, 0, 1, 2))
    # A mesh transform that scales through w (Mat44::scale(s) puts 1 / s into [3][3], cudaTracer/Mat.cuh:381-386):
    # Mesh::intersect divides the ORIGIN by w and leaves the direction alone (Mesh.cuh:115-117, Mat.cuh:127-149).  With
    # w = 2 and the ray from (0, 0, -5): object-space origin (0, 0, -2.5), direction (0, 0, 1); a triangle in the object's
    # plane z = 2.5 is hit at t = 5, which is also the world distance: world hit point (0, 0, 0), light at the camera 5
    # away: pixel = 5 / 5 = 1.  (Were the direction scaled too, or the origin not divided, t would be 7.5 or 2.5.)
    lower = tuple((x, y, 2.5) for x, y, _ in FRONT)
    P["mesh_transform_divides_the_origin_by_w_only"] = (
        Scene(1, 1, [Mesh([tri(*lower)], transform_inverse=np.diag([1.0, 1.0, 1.0, 2.0]))], lights=[((0, 0, -5), (1, 1, 1), 5.0)], origin=(0, 0, -5)),
        (1.0, 1.0, 1.0), (0, 0, 1, 2))
    # two identical spheres in the one scene leaf: equal t, the sphere test rejects only t0 > tMax (Sphere.cuh:48-51), so the
    # shape tested later wins (Scene.cuh:47-50): the pixel has the second sphere's colour, 5 / 4 * colour
    P["equal_t_later_sphere_wins"] = (Scene(1, 1, spheres=[((0, 0, 5), 1.0, DIFFUSE, 1.3, (0.4, 0.2, 0.1)), ((0, 0, 5), 1.0, DIFFUSE, 1.3, (0.1, 0.2, 0.4))],
                                            lights=LIGHT_AT_CAMERA), (0.125, 0.25, 0.5), (1, -1, -1, -1))
    # depth limit (tracer.cu:16-18): with maxDepth 0 the mirror's reflected ray is at depth 1 and returns the background
    # without looking at the scene - a lit sphere sits right in its way (the mirror faces the camera: the reflected ray
    # comes straight back) - times kr = 0.04
    P["depth_limit_returns_the_background"] = (Scene(1, 1, [Mesh([tri(*FRONT)], material=MIRROR, ior=1.5)],
                                                     spheres=[((0, 0, 2), 0.5, DIFFUSE, 1.3, (1.0, 1.0, 1.0))], lights=[((0, 0, 3.5), (1, 1, 1), 5.0)],
                                                     max_depth=0, origin=(0, 0, 3)),
                                               tuple(0.04 * c for c in BG), (0, 0, 1, 2))
    # the shadow ray's origin is biased to the side the VIEW direction comes from (tracer.cu:175,202-208), not the light's:
    # shading normal (0, 0, 1) pointing away from the camera, light behind the triangle at (0, 0, 10): lambert = 1, but
    # dir . n > 0 puts the origin at z = 5 - bias, in front of the triangle, and the ray towards +z meets the triangle
    # itself (front-facing for that direction, t = bias >= 0): in shadow, black
    P["shadow_origin_follows_the_view_side"] = (Scene(1, 1, [Mesh([tri(*FRONT, normal=(0, 0, 1))])], lights=[((0, 0, 10), (1, 1, 1), 5.0)]),
                                                (0.0, 0.0, 0.0), (0, 0, 1, 2))
    # mirror turned 45 degrees about y, shading normal (-1, 0, -1) / sqrt 2: reflected direction dir - 2 (dir . n) n = (-1, 0, 0)
    # (tracer.cu:134-142, Vec.cuh:249-251), which meets a sphere of radius 1 at (-6, 0, 5) in (-5, 0, 5).  The sphere is lit by a
    # light at (-2, 0, 5), 3 away along its normal: radiance = 5 / 3 * colour.  (A light AT the mirror point would be hidden
    # by the mirror itself, which is front-facing for the sphere's shadow ray.)  Fresnel at 45 degrees into n = 1.5
    # (tracer.cu:67-89, the textbook equations): cos_t = sqrt(1 - 0.5 / 2.25); kr = (Rs^2 + Rp^2) / 2
    ci, ct = 0.5 ** 0.5, (1 - 0.5 / 2.25) ** 0.5
    rs, rp = (1.5 * ci - ct) / (1.5 * ci + ct), (ci - 1.5 * ct) / (ci + 1.5 * ct)
    kr45 = (rs * rs + rp * rp) / 2
    slanted = ((-3, -3, 8), (-3, 6, 8), (6, -3, -1))  # the plane x + z = 5 through (0, 0, 5); det > 0 for rays along +z
    r2 = 0.5 ** 0.5
    P["mirror_at_45_degrees"] = (Scene(1, 1, [Mesh([tri(*slanted, normal=(-r2, 0, -r2))], material=MIRROR, ior=1.5)],
                                       spheres=[((-6, 0, 5), 1.0, DIFFUSE, 1.3, (0.8, 0.4, 0.2))], lights=[((-2, 0, 5), (1, 1, 1), 5.0)], max_depth=1),
                                 tuple(kr45 * 5 / 3 * c for c in (0.8, 0.4, 0.2)), (0, 0, 1, 2))
    # Barycentric weights (cudaTracer/Mesh.cuh:50-56, 101-109, 182-212): with v0 = (-1, -2, 5), e1 = (0, 4, 0), e2 = (4, 0, 0) the ray hits
    # v0 + 0.5 e1 + 0.25 e2, so u = 0.5 belongs to vertex 1, v = 0.25 to vertex 2 and 0.25 to vertex 0.  Texture coordinates
    # (0, 0), (1, 0.5), (0.5, 0.5) give st = (0.625, 0.375), the centre of texel x = 2, y = 1 of a 4 x 4 map (with u and v
    # exchanged: (0.5, 0.375), between two texels).
    skew = ((-1, -2, 5), (-1, 2, 5), (3, -2, 5))
    px = [[(10 * x + 60 * y + 5, 255 - 50 * x, 40 * y) for x in range(4)] for y in range(4)]
    P["texcoords_follow_the_barycentric_weights"] = (Scene(1, 1, [Mesh([tri(*skew, uv=((0, 0), (1, 0.5), (0.5, 0.5)))], diffuse=px)],
                                                           lights=LIGHT_AT_CAMERA), tuple(c / 255 for c in px[1][2]), (0, 0, 1, 2))
    # the same triangle with vertex normals (0, 0.8, -0.6), (0.6, 0, -0.8), (0, 0, -1): n = 0.25 n0 + 0.5 n1 + 0.25 n2 =
    # (0.3, 0.2, -0.8), normalised; light at the camera: lambert = 0.8 / sqrt 0.77 (with u and v exchanged: 0.85 / sqrt 0.785)
    corners = tri(*skew)
    for c, n in zip(corners, ((0, 0.8, -0.6), (0.6, 0, -0.8), (0, 0, -1))):
        c["normal"] = n
    P["vertex_normals_follow_the_barycentric_weights"] = (Scene(1, 1, [Mesh([corners])], lights=LIGHT_AT_CAMERA), (0.8 / 0.77 ** 0.5,) * 3, (0, 0, 1, 2))
    # Shadow rays against spheres (cudaTracer/Sphere.cuh:9-54 with the shadow ray's reach, tracer.cu:176-184): surface point
    # (0, 0, 5), light at (3, 0, 1), 5 away along (0.6, 0, -0.8): lambert 0.8, light component 5 / 5.  A sphere of radius 0.5
    # half way, at (1.5, 0, 3), blocks it (the primary ray passes 1.5 from its centre): black.  The same sphere beside the
    # segment, or on the line but 7.5 from the surface (beyond the light): pixel = 0.8.
    side_light = [((3, 0, 1), (1, 1, 1), 5.0)]
    ball = lambda c: [(c, 0.5, DIFFUSE, 1.3, (1.0, 1.0, 1.0))]  # noqa: E731
    P["sphere_between_surface_and_light_shadows"] = (Scene(1, 1, [Mesh([tri(*FRONT)])], spheres=ball((1.5, 0, 3)), lights=side_light),
                                                     (0.0, 0.0, 0.0), (0, 0, 1, 2))
    P["sphere_beside_the_shadow_ray"] = (Scene(1, 1, [Mesh([tri(*FRONT)])], spheres=ball((1.5, 2, 3)), lights=side_light), (0.8, 0.8, 0.8), (0, 0, 1, 2))
    P["sphere_beyond_the_light"] = (Scene(1, 1, [Mesh([tri(*FRONT)])], spheres=ball((4.5, 0, -1)), lights=side_light), (0.8, 0.8, 0.8), (0, 0, 1, 2))
    # glass at oblique incidence (tracer.cu:102-115): the view direction turned about y to (0.6, 0, 0.8) meets the pane z = 5 in
    # P = (3.75, 0, 5) with cos_i = 0.8; Snell into n = 1.5: sin_t = 0.4, the refracted direction T = (0.4, 0, sqrt 0.84).  A
    # sphere of radius 0.5 sits 6 along T, a light 3 along T: the sphere's near point is 2.5 from the light along its
    # normal, radiance 5 / 2.5 * colour.  The "reflected" ray of a glass surface is the NORMAL reflected about the view
    # direction (tracer.cu:119): (0, 0, -1) - 2 (-0.8) (0.6, 0, 0.8) = (0.96, 0, 0.28), which starts a bias in front of the pane,
    # meets the pane again at once, and at depth 2 > maxDepth both of that hit's children are the background.  Fresnel for
    # cos_i = 0.8, cos_t = sqrt 0.84 (tracer.cu:67-89): pixel = kr * background + (1 - kr) * 2 * colour
    ci, ct = 0.8, 0.84 ** 0.5
    rs, rp = (1.5 * ci - ct) / (1.5 * ci + ct), (ci - 1.5 * ct) / (ci + 1.5 * ct)
    kr = (rs * rs + rp * rp) / 2
    rot = [[0.8, 0, -0.6, 0], [0, 1, 0, 0], [0.6, 0, 0.8, 0], [0, 0, 0, 1]]  # row-vector convention: (0, 0, 1) -> (0.6, 0, 0.8)
    hit, T = np.array([3.75, 0.0, 5.0]), np.array([0.4, 0.0, ct])
    colour = (0.1, 0.2, 0.3)
    P["refract_oblique_incidence_follows_snell"] = (
        Scene(1, 1, [Mesh([tri(*big)], material=GLASS, ior=1.5)], spheres=[(tuple(hit + 6 * T), 0.5, DIFFUSE, 1.3, colour)],
              lights=[(tuple(hit + 3 * T), (1, 1, 1), 5.0)], max_depth=1, camera=rot),
        tuple(kr * b + (1 - kr) * 2 * c for b, c in zip(BG, colour)), (0, 0, 1, 2))
    return P


def camera_scene():
    """Pixel -> primary ray (cudaTracer/Main.cu:57-76): a 4 x 2 frame with aspect ratio 2 and scale 1 sends the ray of pixel
    (x, y) through ((2 (x + 0.5) / 4 - 1) * 2, 1 - 2 (y + 0.5) / 2, 1): the top-right pixel (3, 0) looks along (1.5, 0.5, 1).
    A sphere of radius 0.1 at (6, 2, 4) lies on that ray and on no other (the neighbours pass (2, 2, 4) and (6, -2, 4)).
    Light at the camera: the pixel is 5 / t * colour with t = |(6, 2, 4)| - 0.1.
    Returns (scene, (x, y) of the one pixel that sees the sphere, its rgb)."""
    colour = (0.8, 0.4, 0.2)
    t = 56.0 ** 0.5 - 0.1
    return Scene(4, 2, spheres=[((6, 2, 4), 0.1, DIFFUSE, 1.3, colour)], lights=LIGHT_AT_CAMERA), (3, 0), tuple(5 / t * c for c in colour)


def bilinear_scene(u, v, mode=1):
    """A camera-facing triangle whose three corners carry the SAME texture coordinate, so st = (u, v) at the hit; 2x2
    diffuse map; light at the camera, t = 5, power 5: the pixel is the filtered texel (cudaTracer/Texture.cuh:33-69)."""
    px = [[(255, 0, 0), (0, 255, 0)], [(0, 0, 255), (255, 255, 255)]]
    return Scene(1, 1, [Mesh([tri(*FRONT, uv=((u, v),) * 3)], diffuse=px, mode=mode)], lights=LIGHT_AT_CAMERA)


# (u, v, mode) -> expected rgb: texel centres sit at (x + 0.5) / 2
BILINEAR = [
    ((0.25, 0.25, 1), (1, 0, 0)), ((0.75, 0.25, 1), (0, 1, 0)), ((0.25, 0.75, 1), (0, 0, 1)), ((0.75, 0.75, 1), (1, 1, 1)),  # centres
    ((0.5, 0.25, 1), (0.5, 0.5, 0)), ((0.25, 0.5, 1), (0.5, 0, 0.5)), ((0.5, 0.5, 1), (0.5, 0.5, 0.5)),  # midway: the mean
    ((0.0, 0.0, 1), (1, 0, 0)), ((1.0, 1.0, 1), (1, 1, 1)), ((1.0, 0.0, 1), (0, 1, 0)),  # CLAMP repeats the edge texel
    ((0.0, 0.25, 0), (0.5, 0.5, 0)),  # WRAP: the left neighbour of texel 0 is texel 1
    ((1.25, 0.25, 0), (1, 0, 0)), ((1.75, 1.75, 0), (1, 1, 1)),  # WRAP beyond 1: the map repeats (texel centres of the second period)
    ((1.25, 0.25, 1), (0, 1, 0)),  # CLAMP beyond 1: the last texel of the row
]


def flip_scene(w=8, h=8):
    """Two identical camera-facing triangles in two leaves under an interior node split along x.  Both leaves' boxes
    (z from 4 to 6, so that they are entered before the hit at z = 5) are hit by every ray, both triangles at the same
    t: the triangle visited LAST wins.  cudaTracer/Mesh.cuh:155-161 visits the first child first for rays going towards
    +x and the second child first for rays going towards -x, so pixels right of the centre report the second triangle
    (vertices 3, 4, 5), pixels left of it the first (0, 1, 2)."""
    wide = ((-30, -30, 5), (-30, 60, 5), (60, -30, 5))
    lo, hi = [-30, -30, 4], [60, 60, 6]
    nodes = [(lo, hi, 2, 0, 0), (lo, hi, 0, 1, 0), (lo, hi, 1, 1, 0)]
    return Scene(w, h, [Mesh([tri(*wide), tri(*wide)], nodes=nodes)], lights=LIGHT_AT_CAMERA)
