"""Synthetic scenes and cameras for the BASELINE.json configs.

Everything is generated from fixed seeds with a vectorised PCG32 (same generator as
src/core/rng.rs:5-48 of the reference), so inputs are identical on every box.

A scene is a plain dict of numpy arrays in the C-ABI layout of include/pbrt_hip.h:
  positions (nv,3) f32, indices (nt,3) i32, tri_material (nt,) i32, materials (nm,) PbrtMaterial,
  tri_light (nt,) i32, lights (nl,) PbrtLight.
"""
import numpy as np

MATERIAL_DTYPE = np.dtype([("type", "<i4"), ("kd", "<f4", 3), ("kt", "<f4", 3), ("eta", "<f4")])
LIGHT_DTYPE = np.dtype([("type", "<i4"), ("L", "<f4", 3), ("prim", "<i4"), ("two_sided", "<i4"),
                        ("n_samples", "<i4"), ("pad", "<i4"), ("pos", "<f4", 3), ("cos_total_width", "<f4"),
                        ("cos_falloff_start", "<f4"), ("world_to_light", "<f4", 9), ("pad2", "<i4", 2)])
CAMERA_DTYPE = np.dtype([("camera_to_world", "<f4", 16), ("raster_to_camera", "<f4", 16),
                         ("lens_radius", "<f4"), ("focal_distance", "<f4"),
                         ("shutter_open", "<f4"), ("shutter_close", "<f4"), ("kind", "<i4"), ("pad", "<i4", 3)])
CAMERA_PERSPECTIVE, CAMERA_ORTHOGRAPHIC, CAMERA_ENVIRONMENT = 0, 1, 2
RAY_DTYPE = np.dtype([("o", "<f4", 3), ("d", "<f4", 3), ("t_max", "<f4"), ("time", "<f4")])
HIT_DTYPE = np.dtype([("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("prim_id", "<i4"),
                      ("instance_id", "<i4"), ("pad", "<i4", 2)])
INSTANCE_DTYPE = np.dtype([("to_world", "<f4", 16), ("to_object", "<f4", 16), ("material", "<i4"), ("pad", "<i4", 3)])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("offset", "<i4"), ("n_primitives", "<u2"),
                       ("axis", "u1"), ("pad", "u1")])
assert MATERIAL_DTYPE.itemsize == 32 and LIGHT_DTYPE.itemsize == 96
assert RAY_DTYPE.itemsize == 32 and HIT_DTYPE.itemsize == 32 and NODE_DTYPE.itemsize == 32

MAT_NONE, MAT_MATTE, MAT_MIRROR, MAT_GLASS = 0, 1, 2, 3
LIGHT_DIFFUSE_AREA, LIGHT_INFINITE, LIGHT_POINT, LIGHT_SPOT, LIGHT_DISTANT = 0, 1, 2, 3, 4

_PCG32_MULT = np.uint64(0x5851F42D4C957F2D)
_PCG32_DEFAULT_STATE = np.uint64(0x853C49E6748FEA9B)


def pcg32_u32(sequence, n):
    """First n outputs of RNG::new(sequence) (src/core/rng.rs:15-35), vectorised by LCG jump-ahead."""
    with np.errstate(over="ignore"):
        inc = np.uint64((int(sequence) << 1 | 1) & 0xFFFFFFFFFFFFFFFF)
        # set_sequence: state = 0; step; state += default; step
        s0 = np.uint64(0) * _PCG32_MULT + inc
        s0 = (s0 + _PCG32_DEFAULT_STATE) * _PCG32_MULT + inc
        # state_k = A_k * s0 + inc * S_k, A_k = mult^k, S_k = sum_{j<k} mult^j
        a = np.empty(n, dtype=np.uint64)
        a[0] = 1
        if n > 1:
            a[1:] = _PCG32_MULT
            a = np.multiply.accumulate(a)
        s = np.empty(n, dtype=np.uint64)
        s[0] = 0
        if n > 1:
            s[1:] = np.add.accumulate(a[:-1])
        old = a * s0 + inc * s
        xorshifted = (((old >> np.uint64(18)) ^ old) >> np.uint64(27)).astype(np.uint32)
        rot = (old >> np.uint64(59)).astype(np.uint32)
        return (xorshifted >> rot) | (xorshifted << ((~rot + np.uint32(1)) & np.uint32(31)))


def pcg32_float(sequence, n):
    """uniform_float (src/core/rng.rs:46-48)."""
    u = pcg32_u32(sequence, n)
    f = u.astype(np.float32) * np.float32(2.3283064365386963e-10)
    return np.minimum(f, np.float32(1.0) - np.float32(np.finfo(np.float32).eps))


def _materials(rows):
    m = np.zeros(len(rows), dtype=MATERIAL_DTYPE)
    for i, (t, kd, kt, eta) in enumerate(rows):
        m[i] = (t, kd, kt, eta)
    return m


def _lights(rows):
    """rows: (type, L, prim, two_sided, n_samples) tuples or ready LIGHT_DTYPE records (point_light() ...)."""
    l = np.zeros(len(rows), dtype=LIGHT_DTYPE)
    for i, row in enumerate(rows):
        if isinstance(row, np.void) or (isinstance(row, np.ndarray) and row.dtype == LIGHT_DTYPE):
            l[i] = row
            continue
        t, L, prim, two_sided, ns = row
        l[i]["type"], l[i]["L"], l[i]["prim"], l[i]["two_sided"], l[i]["n_samples"] = t, L, prim, two_sided, ns
    return l


def point_light(p_light, intensity):
    """PointLight::new (lights/point.rs:26-40) with light_to_world = translate(p_light)."""
    l = np.zeros((), dtype=LIGHT_DTYPE)
    l["type"], l["L"], l["prim"], l["n_samples"], l["pos"] = LIGHT_POINT, intensity, -1, 1, p_light
    return l


def spot_light(p_from, p_to, intensity, total_width=30.0, falloff_start=25.0):
    """SpotLight::new (lights/spot.rs:29-47); light_to_world = the scene-file form (api.cpp: look from `from`
    down +z towards `to`): rotation with z = normalize(to - from), then translate(from)."""
    p_from, p_to = np.asarray(p_from, dtype=np.float64), np.asarray(p_to, dtype=np.float64)
    z = (p_to - p_from) / np.linalg.norm(p_to - p_from)
    x = np.cross(z, [1.0, 0.0, 0.0]) if abs(z[0]) < 0.9 else np.cross(z, [0.0, 1.0, 0.0])
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    l = np.zeros((), dtype=LIGHT_DTYPE)
    l["type"], l["L"], l["prim"], l["n_samples"], l["pos"] = LIGHT_SPOT, intensity, -1, 1, p_from
    l["cos_total_width"] = np.cos(np.radians(total_width))
    l["cos_falloff_start"] = np.cos(np.radians(falloff_start))
    l["world_to_light"] = np.stack([x, y, z]).reshape(9)       # rows = light axes in world space (rigid: inverse = transpose)
    return l


def distant_light(w_from_surface_to_light, radiance):
    """DistantLight::new (lights/distant.rs:29-46): w_light = normalize(light_to_world * w)."""
    w = np.asarray(w_from_surface_to_light, dtype=np.float64)
    l = np.zeros((), dtype=LIGHT_DTYPE)
    l["type"], l["L"], l["prim"], l["n_samples"], l["pos"] = LIGHT_DISTANT, radiance, -1, 1, w / np.linalg.norm(w)
    return l


def _quad(a, b, c, d):
    """Two triangles (a,b,c), (a,c,d)."""
    return [a, b, c, d], [(0, 1, 2), (0, 2, 3)]


def _box(corners_floor, height):
    """Closed box: 4 floor corners (counter-clockwise seen from above), extruded by `height` in y."""
    f = [np.array(c, dtype=np.float64) for c in corners_floor]
    t = [c + np.array([0.0, height, 0.0]) for c in f]
    quads = [
        (t[0], t[1], t[2], t[3]),  # top
        (f[3], f[2], f[1], f[0]),  # bottom
        (f[0], f[1], t[1], t[0]),
        (f[1], f[2], t[2], t[1]),
        (f[2], f[3], t[3], t[2]),
        (f[3], f[0], t[0], t[3]),
    ]
    return quads


def cornell_box():
    """Config 2: Cornell box, 36 triangles (5 walls x2, light x2, short box 12, tall box 12);
    matte white/red/green; emitter Le = (17, 12, 4) facing down."""
    white, red, green = (0.73, 0.73, 0.73), (0.65, 0.05, 0.05), (0.12, 0.45, 0.15)
    S = 555.0
    quads = []  # (quad corners, material id, emissive)
    quads.append((((0, 0, 0), (S, 0, 0), (S, 0, S), (0, 0, S)), 0, False))      # floor
    quads.append((((0, S, 0), (0, S, S), (S, S, S), (S, S, 0)), 0, False))      # ceiling
    quads.append((((0, 0, S), (S, 0, S), (S, S, S), (0, S, S)), 0, False))      # back wall
    quads.append((((S, 0, 0), (S, S, 0), (S, S, S), (S, 0, S)), 1, False))      # left (red) wall at x = S
    quads.append((((0, 0, 0), (0, 0, S), (0, S, S), (0, S, 0)), 2, False))      # right (green) wall at x = 0
    # light: winding chosen so the geometric normal normalize(dp02 x dp12) points down (-y)
    quads.append((((213, 554, 227), (343, 554, 227), (343, 554, 332), (213, 554, 332)), 0, True))
    short = _box([(130, 0, 65), (82, 0, 225), (240, 0, 272), (290, 0, 114)], 165.0)
    tall = _box([(265, 0, 296), (314, 0, 456), (472, 0, 406), (423, 0, 247)], 330.0)
    for q in short + tall:
        quads.append((q, 0, False))
    positions, indices, tri_mat, tri_light, lights = [], [], [], [], []
    for q, mat, emissive in quads:
        base = len(positions)
        verts, tris = _quad(*q)
        positions.extend(verts)
        for t in tris:
            if emissive:
                tri_light.append(len(lights))
                lights.append((LIGHT_DIFFUSE_AREA, (17.0, 12.0, 4.0), len(indices), 0, 1))
            else:
                tri_light.append(-1)
            indices.append([base + t[0], base + t[1], base + t[2]])
            tri_mat.append(mat)
    scene = dict(
        positions=np.asarray(positions, dtype=np.float32),
        indices=np.asarray(indices, dtype=np.int32),
        tri_material=np.asarray(tri_mat, dtype=np.int32),
        materials=_materials([(MAT_MATTE, white, (0, 0, 0), 1.0), (MAT_MATTE, red, (0, 0, 0), 1.0),
                              (MAT_MATTE, green, (0, 0, 0), 1.0)]),
        tri_light=np.asarray(tri_light, dtype=np.int32),
        lights=_lights(lights),
    )
    assert scene["indices"].shape[0] == 36
    return scene


def _assemble(quads, materials):
    """quads: (corners a b c d, material id, Le or None, two_sided). The geometric normal normalize(dp02 x dp12)
    (triangle.rs:232) of corners going +x then +z in the plane y = const points to -y (as the Cornell light does)."""
    positions, indices, tri_mat, tri_light, lights = [], [], [], [], []
    for q, mat, le, two_sided in quads:
        base = len(positions)
        verts, tris = _quad(*q)
        positions.extend(verts)
        for t in tris:
            if le is not None:
                tri_light.append(len(lights))
                lights.append((LIGHT_DIFFUSE_AREA, le, len(indices), int(two_sided), 1))
            else:
                tri_light.append(-1)
            indices.append([base + t[0], base + t[1], base + t[2]])
            tri_mat.append(mat)
    return dict(positions=np.asarray(positions, dtype=np.float32), indices=np.asarray(indices, dtype=np.int32),
                tri_material=np.asarray(tri_mat, dtype=np.int32), materials=_materials(materials),
                tri_light=np.asarray(tri_light, dtype=np.int32), lights=_lights(lights))


def glass_slab_scene(eta=1.5, thickness=0.2):
    """Closed-form pin of the specular chain (tests/test_oracle_render.py): a glass slab (FresnelSpecular, Kr = Kt = 1) in
    y in [0, thickness], outward normals; a RED one-sided emitter below it facing up (what a camera above the slab sees
    THROUGH it), a GREEN one facing down above the camera (what it sees reflected IN it). Black emitters, no other light."""
    e = 200.0
    quads = [(q, 0, None, False) for q in _box([(-e, 0, -e), (-e, 0, e), (e, 0, e), (e, 0, -e)], thickness)]
    quads.append((((-e, -1, -e), (-e, -1, e), (e, -1, e), (e, -1, -e)), 1, (1.0, 0.0, 0.0), False))     # faces +y
    quads.append((((-e, 9, -e), (e, 9, -e), (e, 9, e), (-e, 9, e)), 1, (0.0, 1.0, 0.0), False))         # faces -y
    return _assemble(quads, [(MAT_GLASS, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), eta), (MAT_MATTE, (0, 0, 0), (0, 0, 0), 1.0)])


def glass_slab_camera(theta_deg, width, height):
    """Looks down at the slab's top face at `theta_deg` from its normal, from y = 4 (below the green emitter)."""
    t = np.radians(theta_deg)
    eye = np.array([-4.0 * np.tan(t), 4.0 + 0.2, 0.0])
    return perspective_camera(tuple(eye), (0.0, 0.2, 0.0), (0, 0, 1), 0.2, width, height)


def mirror_corridor_scene(kr=0.9, Le=3.0):
    """Two facing mirrors x = -1 and x = +1 (SpecularReflection, FresnelNoOp, Kr = kr) and a one-sided emitter that closes the
    corridor at z = 10: a ray from the origin with lateral travel D = 10 tan(phi) is reflected floor((D + 1) / 2) times."""
    h = 100.0
    quads = [(((-1, -h, -2), (-1, h, -2), (-1, h, 14), (-1, -h, 14)), 0, None, False),
             (((1, -h, -2), (1, -h, 14), (1, h, 14), (1, h, -2)), 0, None, False),
             (((-1, -h, 10), (-1, h, 10), (1, h, 10), (1, -h, 10)), 1, (Le, Le, Le), True)]
    return _assemble(quads, [(MAT_MIRROR, (kr, kr, kr), (0, 0, 0), 1.0), (MAT_MATTE, (0, 0, 0), (0, 0, 0), 1.0)])


def mirror_corridor_camera(lateral_travel, width, height):
    return perspective_camera((0, 0, 0), (lateral_travel, 0.0, 10.0), (0, 1, 0), 0.05, width, height)


def two_unequal_lights_scene(rho=0.6):
    """A Lambertian floor under two very unequal one-sided emitters facing down: a large dim one (4 x 4 at y = 3, Le 1) and a
    small bright one (0.2 x 0.2 at y = 1, off to the side, Le 4000: ten times the other's power) — every light-pick strategy must
    find the same mean."""
    e = 100.0
    quads = [(((-e, 0, -e), (-e, 0, e), (e, 0, e), (e, 0, -e)), 0, None, False),
             (((-2, 3, -2), (2, 3, -2), (2, 3, 2), (-2, 3, 2)), 1, (1.0, 1.0, 1.0), False),
             (((1.4, 1, -0.1), (1.6, 1, -0.1), (1.6, 1, 0.1), (1.4, 1, 0.1)), 1, (4000.0, 4000.0, 4000.0), False)]
    return _assemble(quads, [(MAT_MATTE, (rho, rho, rho), (0, 0, 0), 1.0), (MAT_MATTE, (0, 0, 0), (0, 0, 0), 1.0)])


def two_unequal_lights_camera(width, height):
    return perspective_camera((-5.0, 2.5, 0.0), (0.7, 0.0, 0.0), (0, 1, 0), 28.0, width, height)


def with_lights(scene, extra, keep_existing=True):
    """Copy of `scene` with the LIGHT_DTYPE records `extra` appended to (or replacing) its light table."""
    out = dict(scene)
    old = scene["lights"] if keep_existing else scene["lights"][:0]
    out["lights"] = np.concatenate([old, _lights(list(extra))]) if len(extra) else old.copy()
    if not keep_existing:
        out["tri_light"] = np.full_like(scene["tri_light"], -1)
    return out


def with_vertex_shading(scene, seq=5, normals=True, uvs=True, flip_fraction=0.3, tangents=False):
    """Copy of `scene` (unshared vertices: random_triangles / mixed_materials) with per-vertex shading normals
    (the geometric normal bent by up to ~35 degrees, `flip_fraction` of the triangles with all three pointing to
    the back side) and / or per-vertex uvs (random, a few triangles with a degenerate uv map)."""
    out = dict(scene)
    pos, idx = scene["positions"], scene["indices"]
    tri = pos[idx].astype(np.float64)
    ng = np.cross(tri[:, 0] - tri[:, 2], tri[:, 1] - tri[:, 2])
    ng /= np.maximum(np.linalg.norm(ng, axis=1, keepdims=True), 1e-30)
    u = pcg32_float(seq, idx.shape[0] * 16).reshape(idx.shape[0], 16).astype(np.float64)
    if normals:
        n = np.zeros((pos.shape[0], 3), dtype=np.float64)
        for k in range(3):
            bent = ng + 0.7 * (u[:, 3 * k:3 * k + 3] - 0.5)
            n[idx[:, k]] = bent / np.linalg.norm(bent, axis=1, keepdims=True)
        flip = u[:, 9] < flip_fraction
        for k in range(3):
            n[idx[flip, k]] *= -1.0
        zero = u[:, 15] < 0.01                                   # a few all-zero normals: ns falls back to n
        for k in range(3):
            n[idx[zero, k]] = 0.0
        out["normals"] = n.astype(np.float32)
    if tangents:                                                  # random directions; a few zero (fallback to dpdu)
        t = pcg32_float(seq + 100, pos.shape[0] * 3).reshape(-1, 3).astype(np.float64) - 0.5
        t[pcg32_float(seq + 200, pos.shape[0]) < 0.03] = 0.0
        out["tangents"] = t.astype(np.float32)
    if uvs:
        uv = np.zeros((pos.shape[0], 2), dtype=np.float64)
        for k in range(3):
            uv[idx[:, k]] = u[:, 10 + 0:10 + 2] * (k + 1) + u[:, 12:14] * (k == 1) - u[:, 13:15] * (k == 2)
        degenerate = u[:, 9] > 0.97
        for k in range(3):
            uv[idx[degenerate, k]] = 0.25                            # determinant 0: the coordinate_system fallback
        out["uvs"] = uv.astype(np.float32)
    return out


def cornell_delta_lights():
    """Point, spot and distant lights placed inside the Cornell box (intensities sized to its 555-unit scale)."""
    return [point_light((278.0, 450.0, 279.0), (4.0e5, 3.5e5, 3.0e5)),
            spot_light((100.0, 500.0, 100.0), (300.0, 0.0, 300.0), (9.0e5, 9.0e5, 6.0e5), 35.0, 20.0),
            distant_light((0.3, 1.0, -0.8), (1.5, 1.5, 2.0))]


def cornell_camera(width, height):
    return perspective_camera((278.0, 273.0, -800.0), (278.0, 273.0, 0.0), (0.0, 1.0, 0.0), 39.3, width, height)


def random_triangles(n_tris=1_000_000, seq=1, extent=1.0, size=0.01, env_L=(1.0, 1.0, 1.0), rho=0.5):
    """Config 3/4: n triangles, centres U[-extent,extent]^3, each vertex = centre + U[-size,size]^3;
    all matte rho; one constant white infinite light."""
    u = pcg32_float(seq, n_tris * 12).reshape(n_tris, 4, 3)
    centre = (u[:, 0, :] * np.float32(2.0) - np.float32(1.0)) * np.float32(extent)
    off = (u[:, 1:, :] * np.float32(2.0) - np.float32(1.0)) * np.float32(size)
    positions = (centre[:, None, :] + off).reshape(n_tris * 3, 3).astype(np.float32)
    indices = np.arange(n_tris * 3, dtype=np.int32).reshape(n_tris, 3)
    return dict(
        positions=positions,
        indices=indices,
        tri_material=np.zeros(n_tris, dtype=np.int32),
        materials=_materials([(MAT_MATTE, (rho, rho, rho), (0, 0, 0), 1.0)]),
        tri_light=np.full(n_tris, -1, dtype=np.int32),
        lights=_lights([(LIGHT_INFINITE, env_L, -1, 0, 1)]),
    )


def random_triangles_camera(width, height):
    return perspective_camera((0.0, 0.0, 3.5), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, width, height)


def mixed_materials_scene(n_tris=20000, seq=7):
    """Small stress scene with matte / mirror / glass triangles, an area light and an env light
    (exercises every BxDF and both light types; a scaled-down stand-in for config 5's material mix)."""
    sc = random_triangles(n_tris, seq=seq, extent=1.0, size=0.08, env_L=(0.6, 0.7, 0.9))
    tri_material = (np.arange(n_tris) % 3).astype(np.int32)
    materials = _materials([
        (MAT_MATTE, (0.6, 0.5, 0.4), (0, 0, 0), 1.0),
        (MAT_MIRROR, (0.9, 0.9, 0.9), (0, 0, 0), 1.0),
        (MAT_GLASS, (1.0, 1.0, 1.0), (0.95, 0.95, 0.95), 1.5),
    ])
    # a two-triangle emitter above the cloud
    quad = np.array([[-0.5, 1.6, -0.5], [0.5, 1.6, -0.5], [0.5, 1.6, 0.5], [-0.5, 1.6, 0.5]], dtype=np.float32)
    base = sc["positions"].shape[0]
    positions = np.concatenate([sc["positions"], quad])
    indices = np.concatenate([sc["indices"], np.array([[base, base + 1, base + 2], [base, base + 2, base + 3]],
                                                      dtype=np.int32)])
    tri_material = np.concatenate([tri_material, np.zeros(2, dtype=np.int32)])
    tri_light = np.concatenate([sc["tri_light"], np.array([1, 2], dtype=np.int32)])
    lights = _lights([(LIGHT_INFINITE, (0.6, 0.7, 0.9), -1, 0, 1),
                      (LIGHT_DIFFUSE_AREA, (20.0, 18.0, 15.0), n_tris, 1, 1),
                      (LIGHT_DIFFUSE_AREA, (20.0, 18.0, 15.0), n_tris + 1, 1, 1)])
    return dict(positions=positions, indices=indices, tri_material=tri_material, materials=materials,
                tri_light=tri_light, lights=lights)


def _random_rigid(u):
    """Rotation (uniform quaternion from 3 uniforms, Shoemake) + translation from 3 more; float64 4x4."""
    u1, u2, u3 = u[0], u[1], u[2]
    q = np.array([np.sqrt(1 - u1) * np.sin(2 * np.pi * u2), np.sqrt(1 - u1) * np.cos(2 * np.pi * u2),
                  np.sqrt(u1) * np.sin(2 * np.pi * u3), np.sqrt(u1) * np.cos(2 * np.pi * u3)])
    x, y, z, w = q
    r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    m = np.eye(4)
    m[:3, :3] = r
    return m


def instanced_scene(n_base_tris=10_000, n_instances=1000, seq_mesh=2, seq_xf=3, extent=4.0, base_extent=0.3,
                    tri_size=0.02, env_L=(1.0, 1.0, 1.0)):
    """Config 5: a base cloud of triangles (object space) x rigid instances placed in [-extent, extent]^3;
    material of instance i = i % 3 -> matte / mirror / glass(eta 1.5); one constant env light.
    scene["instances"]: (n, 2, 4, 4) float32 {to_world, to_object} (the inverse is computed in float64)."""
    base = random_triangles(n_base_tris, seq=seq_mesh, extent=base_extent, size=tri_size, env_L=env_L)
    u = pcg32_float(seq_xf, n_instances * 6).astype(np.float64).reshape(n_instances, 6)
    inst = np.zeros((n_instances, 2, 4, 4), dtype=np.float32)
    for i in range(n_instances):
        m = _random_rigid(u[i, :3])
        m[:3, 3] = (u[i, 3:] * 2 - 1) * extent
        inst[i, 0] = m.astype(np.float32)
        inst[i, 1] = np.linalg.inv(m).astype(np.float32)
    inst[:, :, 3, :] = (0, 0, 0, 1)
    materials = _materials([
        (MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0),
        (MAT_MIRROR, (0.9, 0.9, 0.9), (0, 0, 0), 1.0),
        (MAT_GLASS, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 1.5),
    ])
    return dict(positions=base["positions"], indices=base["indices"], materials=materials,
                instances=inst, instance_material=(np.arange(n_instances) % 3).astype(np.int32),
                lights=_lights([(LIGHT_INFINITE, env_L, -1, 0, 1)]),
                tri_material=base["tri_material"], tri_light=base["tri_light"])


def two_level_scene(n_instances=90, seq_xf=13, extent=2.0, env_L=(0.2, 0.25, 0.3)):
    """A general two-level scene (primitive.rs:105-159): three different object aggregates (triangle clouds of different
    size and density), instances of them cycling 0, 1, 2 with matte / mirror / glass by instance, and world-space
    geometry beside them: a floor quad and an emitting quad (one-sided DiffuseAreaLight, two triangles) above the cloud,
    plus a dim constant environment. scene["objects"]: list of {positions, indices, tri_material}; scene["instance_object"];
    scene["world"]: {positions, indices, tri_material, tri_light} (lights' `prim` counts the world triangles)."""
    objs = []
    for k, (n, ext, size, sq) in enumerate([(1500, 0.25, 0.05, 21), (600, 0.18, 0.09, 22), (2500, 0.3, 0.03, 23)]):
        b = random_triangles(n, seq=sq, extent=ext, size=size)
        objs.append(dict(positions=b["positions"], indices=b["indices"], tri_material=np.full(n, k % 3, dtype=np.int32)))
    u = pcg32_float(seq_xf, n_instances * 6).astype(np.float64).reshape(n_instances, 6)
    inst = np.zeros((n_instances, 2, 4, 4), dtype=np.float32)
    for i in range(n_instances):
        m = _random_rigid(u[i, :3])
        m[:3, 3] = (u[i, 3:] * 2 - 1) * extent
        inst[i, 0] = m.astype(np.float32)
        inst[i, 1] = np.linalg.inv(m).astype(np.float32)
    inst[:, :, 3, :] = (0, 0, 0, 1)
    e = extent * 1.6
    fy, ly, lw = -extent * 1.3, extent * 1.4, extent * 0.5
    wp = np.array([[-e, fy, -e], [e, fy, -e], [e, fy, e], [-e, fy, e],
                   [-lw, ly, -lw], [lw, ly, -lw], [lw, ly, lw], [-lw, ly, lw]], dtype=np.float32)
    wi = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7]], dtype=np.int32)   # floor faces +y, light faces -y
    materials = _materials([
        (MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0),
        (MAT_MIRROR, (0.9, 0.9, 0.9), (0, 0, 0), 1.0),
        (MAT_GLASS, (1.0, 1.0, 1.0), (1.0, 1.0, 1.0), 1.5),
        (MAT_MATTE, (0.7, 0.6, 0.5), (0, 0, 0), 1.0),
    ])
    lights = _lights([(LIGHT_DIFFUSE_AREA, (12.0, 11.0, 9.0), 2, 0, 1), (LIGHT_DIFFUSE_AREA, (12.0, 11.0, 9.0), 3, 0, 1),
                      (LIGHT_INFINITE, env_L, -1, 0, 1)])
    return dict(objects=objs, instances=inst, instance_object=(np.arange(n_instances) % 3).astype(np.int32),
                instance_material=np.where(np.arange(n_instances) % 4 == 3, -1, np.arange(n_instances) % 3).astype(np.int32),
                world=dict(positions=wp, indices=wi, tri_material=np.array([3, 3, 0, 0], dtype=np.int32),
                           tri_light=np.array([-1, -1, 0, 1], dtype=np.int32)),
                materials=materials, lights=lights)


def two_level_camera(width, height, extent=2.0):
    return perspective_camera((0.3 * extent, 0.4 * extent, 3.4 * extent), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 42.0, width, height)


def combined_two_level_mesh(scene):
    """Every object's triangles, object after object, then the world triangles: (positions, indices, tri_material,
    tri_light, obj_tri_offset[n_objects + 1]); what the oracle's and the library's two-level creation work on."""
    pos, idx, mat, lgt, off = [], [], [], [], [0]
    v0 = 0
    for o in scene["objects"]:
        pos.append(np.asarray(o["positions"], dtype=np.float32))
        idx.append(np.asarray(o["indices"], dtype=np.int32) + v0)
        mat.append(np.asarray(o["tri_material"], dtype=np.int32))
        lgt.append(np.full(len(o["indices"]), -1, dtype=np.int32))
        v0 += len(o["positions"])
        off.append(off[-1] + len(o["indices"]))
    w = scene["world"]
    if len(w["indices"]):
        pos.append(np.asarray(w["positions"], dtype=np.float32))
        idx.append(np.asarray(w["indices"], dtype=np.int32) + v0)
        mat.append(np.asarray(w["tri_material"], dtype=np.int32))
        lgt.append(np.asarray(w["tri_light"], dtype=np.int32))
    return (np.ascontiguousarray(np.concatenate(pos)), np.ascontiguousarray(np.concatenate(idx)),
            np.ascontiguousarray(np.concatenate(mat)), np.ascontiguousarray(np.concatenate(lgt)), np.array(off, dtype=np.int32))


def world_space_instances(scene):
    """The same geometry without TransformedPrimitives: one world-space TriangleMesh per instance, vertices put
    through `Transform * Point3f` (transform.rs:351-370, float32, the operation order of the reference) as
    TriangleMesh::new does for a shape that is not instanced (triangle.rs: object_to_world applied to `p`).
    n_instances x n_base triangles in one flat scene: 10 M triangles = 1.1 GB of triangles + nodes, which is what
    288 GB of HBM is for. Materials follow the instance's material override."""
    inst = scene["instances"]
    base_p, base_i = scene["positions"], scene["indices"]
    nv, nt, n = base_p.shape[0], base_i.shape[0], inst.shape[0]
    pos = np.empty((n, nv, 3), dtype=np.float32)
    x, y, z = base_p[:, 0], base_p[:, 1], base_p[:, 2]
    for i in range(n):
        m = inst[i, 0]
        for r in range(3):
            pos[i, :, r] = m[r, 0] * x + m[r, 1] * y + m[r, 2] * z + m[r, 3]     # w' = 1 for an affine matrix
    idx = (base_i[None, :, :] + (np.arange(n, dtype=np.int32) * nv)[:, None, None]).astype(np.int32)
    im = scene["instance_material"]
    mat = np.where(im[:, None] >= 0, im[:, None], scene["tri_material"][None, :]).astype(np.int32)
    return dict(positions=pos.reshape(-1, 3), indices=idx.reshape(-1, 3), tri_material=mat.reshape(-1),
                materials=scene["materials"], tri_light=np.full(n * nt, -1, dtype=np.int32), lights=scene["lights"])


def instanced_camera(width, height, extent=4.0):
    return perspective_camera((0.0, 0.0, 3.2 * extent), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, width, height)


def sphere_scene():
    """Config 1 (CPU oracle only): unit matte sphere (rho 0.5) at the origin under a one-sided 2x2 quad
    area light at y = 3 (Le = 10, facing down). scene["spheres"]: (n, 8) {c.xyz, radius, material, light, 0, 0}."""
    quad = np.array([[-1, 3, -1], [1, 3, -1], [1, 3, 1], [-1, 3, 1]], dtype=np.float32)
    return dict(
        positions=quad,
        indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32),
        tri_material=np.zeros(2, dtype=np.int32),
        materials=_materials([(MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
        tri_light=np.array([0, 1], dtype=np.int32),
        lights=_lights([(LIGHT_DIFFUSE_AREA, (10.0, 10.0, 10.0), 0, 0, 1), (LIGHT_DIFFUSE_AREA, (10.0, 10.0, 10.0), 1, 0, 1)]),
        spheres=np.array([[0, 0, 0, 1.0, 0, -1, 0, 0]], dtype=np.float32),
    )


def sphere_camera(width, height):
    return perspective_camera((0.0, 1.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 45.0, width, height)


def furnace_scene(rho=0.5, Le=1.0):
    """White-furnace check: a closed matte cube (rho) lit only by a constant env light cannot be
    reached by it, so use the open form: a single matte quad under a constant environment.
    Radiance leaving a Lambertian surface under uniform illumination Le is rho * Le exactly."""
    quad = np.array([[-1e3, 0, -1e3], [-1e3, 0, 1e3], [1e3, 0, 1e3], [1e3, 0, -1e3]], dtype=np.float32)
    return dict(
        positions=quad,
        indices=np.array([[0, 1, 2], [0, 2, 3]], dtype=np.int32),
        tri_material=np.zeros(2, dtype=np.int32),
        materials=_materials([(MAT_MATTE, (rho, rho, rho), (0, 0, 0), 1.0)]),
        tri_light=np.full(2, -1, dtype=np.int32),
        lights=_lights([(LIGHT_INFINITE, (Le, Le, Le), -1, 0, 1)]),
    )


# ---------------------------------------------------------------------------------------
# Camera matrices: Transform::look_at / perspective / PerspectiveCamera::new
# (src/core/transform.rs:510-566, src/cameras/perspective.rs:34-82), computed in float64 and
# rounded once; both the oracle and the HIP path receive the same float32 matrices.
# ---------------------------------------------------------------------------------------
def look_at(pos, look, up):
    pos, look, up = (np.asarray(v, dtype=np.float64) for v in (pos, look, up))
    d = look - pos
    d /= np.linalg.norm(d)
    right = np.cross(up / np.linalg.norm(up), d)
    right /= np.linalg.norm(right)
    new_up = np.cross(d, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, new_up, d, pos
    return c2w


def perspective_camera(eye, look, up, fov_deg, width, height, lens_radius=0.0, focal_distance=1e6):
    c2w = look_at(eye, look, up)
    n, f = 1e-2, 1000.0
    persp = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, f / (f - n), -f * n / (f - n)], [0, 0, 1, 0]], dtype=np.float64)
    inv_tan = 1.0 / np.tan(np.radians(fov_deg) / 2.0)
    camera_to_screen = np.diag([inv_tan, inv_tan, 1.0, 1.0]) @ persp
    # pbrt screen window: the shorter axis spans [-1, 1]
    aspect = width / height
    if aspect > 1.0:
        sw = (-aspect, aspect, -1.0, 1.0)
    else:
        sw = (-1.0, 1.0, -1.0 / aspect, 1.0 / aspect)
    screen_to_raster = (np.diag([width, height, 1.0, 1.0]) @
                        np.diag([1.0 / (sw[1] - sw[0]), 1.0 / (sw[2] - sw[3]), 1.0, 1.0]) @
                        _translate(-sw[0], -sw[3], 0.0))
    raster_to_camera = np.linalg.inv(camera_to_screen) @ np.linalg.inv(screen_to_raster)
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    cam["camera_to_world"] = c2w.astype(np.float32).reshape(16)
    cam["raster_to_camera"] = raster_to_camera.astype(np.float32).reshape(16)
    cam["lens_radius"] = lens_radius
    cam["focal_distance"] = focal_distance
    cam["shutter_open"] = 0.0
    cam["shutter_close"] = 1.0
    return cam


def orthographic_camera(eye, look, up, half_height, width, height, lens_radius=0.0, focal_distance=1e6):
    """OrthographicCamera::new (cameras/orthographic.rs:37-80): camera_to_screen = orthographic(0, 1) (z unchanged for
    near 0, far 1), screen window [-a, a] x [-1, 1] scaled by `half_height`."""
    c2w = look_at(eye, look, up)
    aspect = width / height
    sw = (-aspect * half_height, aspect * half_height, -half_height, half_height)
    screen_to_raster = (np.diag([width, height, 1.0, 1.0]) @
                        np.diag([1.0 / (sw[1] - sw[0]), 1.0 / (sw[2] - sw[3]), 1.0, 1.0]) @ _translate(-sw[0], -sw[3], 0.0))
    raster_to_camera = np.linalg.inv(screen_to_raster)        # orthographic(0, 1) is the identity
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    cam["camera_to_world"] = c2w.astype(np.float32).reshape(16)
    cam["raster_to_camera"] = raster_to_camera.astype(np.float32).reshape(16)
    cam["lens_radius"], cam["focal_distance"], cam["shutter_open"], cam["shutter_close"] = lens_radius, focal_distance, 0.0, 1.0
    cam["kind"] = CAMERA_ORTHOGRAPHIC
    return cam


def environment_camera(eye, look, up):
    """EnvironmentCamera::new (cameras/environment.rs:19-29): only camera_to_world matters."""
    cam = np.zeros((), dtype=CAMERA_DTYPE)
    cam["camera_to_world"] = look_at(eye, look, up).astype(np.float32).reshape(16)
    cam["raster_to_camera"] = np.eye(4, dtype=np.float32).reshape(16)
    cam["shutter_close"] = 1.0
    cam["kind"] = CAMERA_ENVIRONMENT
    return cam


def _translate(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def random_rays(n, seq, origin_extent=1.5, t_max=np.inf):
    """n rays with origins U[-e,e]^3 and directions uniform on the sphere (by normalising a
    Gaussian-free cube rejection surrogate: U[-1,1]^3 direction, not normalised — pbrt rays
    need not be unit length, shadow rays are not)."""
    u = pcg32_float(seq, n * 6).reshape(n, 6)
    rays = np.zeros(n, dtype=RAY_DTYPE)
    rays["o"] = (u[:, :3] * 2 - 1) * origin_extent
    d = u[:, 3:] * 2 - 1
    d[np.all(d == 0, axis=1)] = (0, 0, 1)
    rays["d"] = d
    rays["t_max"] = t_max
    rays["time"] = 0
    return rays


def camera_dict_to_floats(cam):
    """37 floats for the oracle's flat entry point."""
    return np.concatenate([cam["camera_to_world"].reshape(-1), cam["raster_to_camera"].reshape(-1),
                           [cam["lens_radius"], cam["focal_distance"], cam["shutter_open"], cam["shutter_close"],
                            float(cam["kind"])]]).astype(np.float32)
