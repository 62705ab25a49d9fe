"""Closed forms for the last stage every result passes through — FilmTile::add_sample / Film::merge_film_tile
(src/core/film.rs:252-295, 93-123, table :52-63) — and for the thin lens (src/cameras/perspective.rs:90-112), written
WITHOUT the oracle or the kernels: numpy float64 for the filter functions, the reference's own float32 index arithmetic for
which table entry a sample selects, PCG32 streams keyed as DESIGN.md section 2 says. Used by tests/test_oracle_render.py
(CPU oracle) and tests/test_gpu_closed_forms.py (HIP path): both must land on THESE numbers."""
import numpy as np

from pbrt_hip import scenes

PI = np.pi


def filter_eval(kind, x, y, rx, ry, a=0.0, b=0.0):
    """Filter::evaluate of src/filters/{boxf,gaussian,mitchell,sinc,triangle}.rs in float64."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if kind == "box":
        return np.ones_like(x * y)
    if kind == "triangle":
        return np.maximum(0.0, rx - np.abs(x)) * np.maximum(0.0, ry - np.abs(y))
    if kind == "gaussian":   # gaussian.rs:24-31: max(0, exp(-alpha d^2) - exp(-alpha r^2))
        g = lambda d, r: np.maximum(0.0, np.exp(-a * d * d) - np.exp(-a * r * r))
        return g(x, rx) * g(y, ry)
    if kind == "mitchell":   # mitchell.rs:25-41 with B = a, C = b
        def m1(t):
            t = np.abs(2.0 * t)
            far = ((-a - 6 * b) * t ** 3 + (6 * a + 30 * b) * t ** 2 + (-12 * a - 48 * b) * t + (8 * a + 24 * b)) / 6.0
            near = ((12 - 9 * a - 6 * b) * t ** 3 + (-18 + 12 * a + 6 * b) * t ** 2 + (6 - 2 * a)) / 6.0
            return np.where(t > 1.0, far, near)
        return m1(x / rx) * m1(y / ry)
    if kind == "lanczos":    # sinc.rs:24-45 with tau = a
        def sinc(t):
            t = np.abs(t)
            return np.where(t < 1e-5, 1.0, np.sin(PI * np.maximum(t, 1e-30)) / (PI * np.maximum(t, 1e-30)))

        def ws(t, r):
            t = np.abs(t)
            return np.where(t > r, 0.0, sinc(t) * sinc(t / a))
        return ws(x, rx) * ws(y, ry)
    raise ValueError(kind)


def analytic_table(kind, rx, ry, a=0.0, b=0.0):
    """Film::new's 16 x 16 table (film.rs:52-63): filter.evaluate at ((x + 0.5) rx / 16, (y + 0.5) ry / 16), row y major."""
    i = (np.arange(16) + 0.5) / 16.0
    return filter_eval(kind, (i * rx)[None, :], (i * ry)[:, None], rx, ry, a, b)


def camera_sample_positions(width, height, spp, seed, rx, ry):
    """p_film of every camera sample of the frame: pixels of the film's SAMPLE bounds (film.rs:76-81), sample s of pixel
    number n (row-major over those bounds) draws from the PCG32 stream seed ^ (n * spp + s); get_camera_sample takes
    p_film = pixel + get_2d() first (sampler.rs:66-73). Returns float32 arrays (x, y) of shape [n_pixels * spp]."""
    x0, y0 = int(np.floor(0.5 - rx)), int(np.floor(0.5 - ry))
    x1, y1 = int(np.ceil(width - 0.5 + rx)), int(np.ceil(height - 0.5 + ry))
    xs, ys = [], []
    n = 0
    for py in range(y0, y1):
        for px in range(x0, x1):
            for s in range(spp):
                u = scenes.pcg32_float(seed ^ (n * spp + s), 2)
                xs.append(np.float32(px) + u[0])
                ys.append(np.float32(py) + u[1])
            n += 1
    return np.array(xs, dtype=np.float32), np.array(ys, dtype=np.float32)


def expected_weight_sums(width, height, px, py, table, rx, ry):
    """FilmTile::add_sample's footprint and table index for every sample (film.rs:263-290, float32 as written; the footprint
    clipped to the film: p1 = min(..., bounds.max), the intended reading of :266), the selected table entries summed per
    pixel in float64. Returns (weight_sum[h, w], n_terms[h, w])."""
    f = np.float32
    wsum = np.zeros((height, width), dtype=np.float64)
    cnt = np.zeros((height, width), dtype=np.int64)
    inv_rx, inv_ry = f(1.0) / f(rx), f(1.0) / f(ry)
    for sx, sy in zip(px, py):
        dx, dy = f(sx) - f(0.5), f(sy) - f(0.5)
        ax0, ax1 = max(int(np.ceil(dx - f(rx))), 0), min(int(np.floor(dx + f(rx))) + 1, width)
        ay0, ay1 = max(int(np.ceil(dy - f(ry))), 0), min(int(np.floor(dy + f(ry))) + 1, height)
        if ax0 >= ax1 or ay0 >= ay1:
            continue
        xi = np.arange(ax0, ax1)
        yi = np.arange(ay0, ay1)
        ifx = np.minimum(15, np.floor(np.abs((xi.astype(np.float32) - dx) * inv_rx * f(16.0))).astype(np.int64))
        ify = np.minimum(15, np.floor(np.abs((yi.astype(np.float32) - dy) * inv_ry * f(16.0))).astype(np.int64))
        wsum[ay0:ay1, ax0:ax1] += table[np.ix_(ify, ifx)]
        cnt[ay0:ay1, ax0:ax1] += 1
    return wsum, cnt


def sky_scene(le=(1.0, 1.0, 1.0)):
    """A constant environment of radiance `le` and one small triangle BEHIND the camera of sky_camera(): every camera ray
    leaves the scene, so every sample carries exactly L = le (path.rs:88-93: the emitted radiance of the environment)."""
    pos = np.array([[0.0, 0.0, 50.0], [0.01, 0.0, 50.0], [0.0, 0.01, 50.0]], dtype=np.float32)
    return dict(positions=pos, indices=np.array([[0, 1, 2]], dtype=np.int32), tri_material=np.zeros(1, dtype=np.int32),
                materials=scenes._materials([(scenes.MAT_MATTE, (0.5, 0.5, 0.5), (0, 0, 0), 1.0)]),
                tri_light=np.full(1, -1, dtype=np.int32), lights=scenes._lights([(scenes.LIGHT_INFINITE, le, -1, 0, 1)]))


def sky_camera(width, height):
    return scenes.perspective_camera((0.0, 0.0, 5.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 40.0, width, height)


def luminance(rgb):
    return 0.212671 * rgb[..., 0] + 0.715160 * rgb[..., 1] + 0.072169 * rgb[..., 2]


# ---- thin lens (perspective.rs:100-106) ----
LENS = dict(fov=30.0, eye=(0.0, 0.0, 0.0), look=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0))


def emitter_scene(depth, half_size, le=(5.0, 5.0, 5.0)):
    """A two-sided square emitter of half-size `half_size` on the optical axis of lens_camera(), `depth` in front of it."""
    s, z = half_size, -depth
    quad = ((-s, -s, z), (s, -s, z), (s, s, z), (-s, s, z))
    return scenes._assemble([(quad, 0, le, True)], [(scenes.MAT_MATTE, (0.0, 0.0, 0.0), (0, 0, 0), 1.0)])


def lens_camera(width, height, lens_radius, focal_distance):
    return scenes.perspective_camera(LENS["eye"], LENS["look"], LENS["up"], LENS["fov"], width, height, lens_radius, focal_distance)


def raster_of_camera_point(cam, p):
    """Where raster_to_camera^-1 puts a camera-space point (the projection PerspectiveCamera::new composes, perspective.rs:34-82)."""
    m = np.linalg.inv(np.asarray(cam["raster_to_camera"], dtype=np.float64).reshape(4, 4))
    # raster_to_camera maps (x, y, 0) onto the near plane: a camera point is first brought there along its ray from the origin
    r2c = np.asarray(cam["raster_to_camera"], dtype=np.float64).reshape(4, 4)
    z_near = (r2c @ np.array([0.0, 0.0, 0.0, 1.0]))
    z_near = z_near[2] / z_near[3]
    q = np.array([p[0] * z_near / p[2], p[1] * z_near / p[2], z_near, 1.0])
    r = m @ q
    return r[:2] / r[3]


def lit_radius_px(rgb, width, height, threshold):
    """Largest distance (pixels, from the image centre to pixel centres) of a pixel brighter than `threshold`."""
    ys, xs = np.nonzero(luminance(rgb) > threshold)
    if len(xs) == 0:
        return 0.0
    return float(np.max(np.hypot(xs + 0.5 - width / 2.0, ys + 0.5 - height / 2.0)))


# ---- orthographic camera (orthographic.rs:82-104) and environment camera (environment.rs:37-56) ----
def offaxis_emitter_scene(cx, cy, depth, half_size, le=(5.0, 5.0, 5.0)):
    """A two-sided square emitter centred at CAMERA-space (cx, cy, depth) of a camera at LENS["eye"] looking along LENS["look"]:
    look_at gives camera +x = world -x, +y = world +y, +z = world -z (transform.rs:510-545), so the world centre is (-cx, cy, -depth)."""
    s, x, y, z = half_size, -cx, cy, -depth
    quad = ((x - s, y - s, z), (x + s, y - s, z), (x + s, y + s, z), (x - s, y + s, z))
    return scenes._assemble([(quad, 0, le, True)], [(scenes.MAT_MATTE, (0.0, 0.0, 0.0), (0, 0, 0), 1.0)])


def ortho_camera(width, height, half_height):
    return scenes.orthographic_camera(LENS["eye"], LENS["look"], LENS["up"], half_height, width, height)


def ortho_raster_of_camera_point(width, height, half_height, cx, cy):
    """raster_to_camera of the orthographic projection inverted by hand: the screen window is [-a hh, a hh] x [-hh, hh]
    (a = width / height), raster y runs downwards; depth plays no part."""
    a = width / height
    return np.array([(cx + a * half_height) / (2 * a * half_height) * width, (half_height - cy) / (2 * half_height) * height])


def env_direction(theta, phi):
    """EnvironmentCamera::generate_ray's direction for film position (x, y): theta = pi y / H, phi = 2 pi x / W."""
    return np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)])


def env_emitter_scene(theta, phi, dist, half_size, le=(5.0, 5.0, 5.0)):
    """A two-sided square emitter facing the origin from direction (theta, phi), `dist` away, its sides along the directions
    of growing phi and growing theta."""
    c = dist * env_direction(theta, phi)
    t_phi = np.array([-np.sin(phi), 0.0, np.cos(phi)])
    t_theta = np.array([np.cos(theta) * np.cos(phi), -np.sin(theta), np.cos(theta) * np.sin(phi)])
    s = half_size
    quad = tuple(tuple(c + a * s * t_phi + b * s * t_theta) for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)))
    return scenes._assemble([(quad, 0, le, True)], [(scenes.MAT_MATTE, (0.0, 0.0, 0.0), (0, 0, 0), 1.0)])


def env_camera():
    return scenes.environment_camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0))   # look_at = identity: camera space is world space


def lit_box(rgb, threshold):
    """(x0, x1, y0, y1) of the pixels brighter than `threshold`, in raster coordinates (pixel i spans [i, i + 1])."""
    ys, xs = np.nonzero(luminance(rgb) > threshold)
    return (float(xs.min()), float(xs.max() + 1), float(ys.min()), float(ys.max() + 1)) if len(xs) else None
