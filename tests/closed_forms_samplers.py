"""What the tabulating samplers promise, written WITHOUT the oracle or the kernels (numpy only): StratifiedSampler puts one sample
into every stratum of every dimension (src/samplers/stratified.rs:44-104 over src/core/sampling.rs's stratified_sample_1d / _2d),
un-jittered samples sit on the stratum centres; ZeroTwoSequenceSampler's first 2^k points are a (0, k, 2)-net in base 2 — one point
in every dyadic box of area 2^-k, whatever its shape — and its 1D dimensions one point per interval of length 2^-k
(src/samplers/zerotwosequence.rs:28-60 over lowdiscrepancy.rs's van_der_corput / sobol_2d; the random digit scramble keeps the
net). Used by tests/test_oracle_kat.py (CPU oracle) and tests/test_gpu_closed_forms.py (HIP path)."""
import numpy as np

ONE_MINUS_EPSILON = np.float32(1.0) - np.float32(2.0 ** -24)


def check_unit_interval(v):
    v = np.asarray(v)
    assert v.dtype == np.float32 and np.all(v >= 0.0) and np.all(v <= ONE_MINUS_EPSILON), (v.min(), v.max())


def check_one_per_interval(v, n):
    """n values, one in each [i / n, (i + 1) / n)."""
    v = np.asarray(v, dtype=np.float64)
    assert v.shape == (n,)
    assert sorted(np.floor(v * n).astype(int)) == list(range(n)), np.sort(np.floor(v * n))


def check_one_per_stratum(p, nx, ny):
    """nx * ny points, one in each of the nx x ny boxes."""
    p = np.asarray(p, dtype=np.float64)
    assert p.shape == (nx * ny, 2)
    cell = np.floor(p[:, 0] * nx).astype(int) + nx * np.floor(p[:, 1] * ny).astype(int)
    assert sorted(cell) == list(range(nx * ny)), np.sort(cell)


def check_stratum_centres(p, nx, ny):
    """jitter off: (x + 0.5) / nx, (y + 0.5) / ny in the reference's float arithmetic, (i + 0.5f) * (1.0f / nx)."""
    p = np.asarray(p, dtype=np.float32)
    f = np.float32
    xs = np.minimum((np.arange(nx, dtype=np.float32) + f(0.5)) * (f(1.0) / f(nx)), ONE_MINUS_EPSILON)
    ys = np.minimum((np.arange(ny, dtype=np.float32) + f(0.5)) * (f(1.0) / f(ny)), ONE_MINUS_EPSILON)
    want = sorted((float(x), float(y)) for y in ys for x in xs)
    assert sorted((float(a), float(b)) for a, b in p) == want


def check_02_net(p, n):
    """n = 2^k points: every dyadic box 2^-a x 2^-b with a + b = k holds exactly one."""
    p = np.asarray(p, dtype=np.float64)
    k = int(np.log2(n))
    assert p.shape == (n, 2) and 2 ** k == n
    for a in range(k + 1):
        b = k - a
        cell = np.floor(p[:, 0] * 2 ** a).astype(int) * 2 ** b + np.floor(p[:, 1] * 2 ** b).astype(int)
        assert sorted(cell) == list(range(n)), (a, b, np.sort(cell))


def radical_inverse(base, i):
    """Phi_base(i) in float64 (lowdiscrepancy.rs:293-320): the digits of i mirrored about the radix point."""
    i = np.asarray(i, dtype=np.int64).copy()
    out, scale = np.zeros(i.shape, dtype=np.float64), 1.0 / base
    while np.any(i > 0):
        out += (i % base) * scale
        i //= base
        scale /= base
    return out


def halton_camera_samples(width, height, spp):
    """HaltonSampler over a width x height film (halton.rs:63-142) from its definition: sample i of the 2D Halton sequence
    (Phi_2(i), Phi_3(i)), scaled by the smallest 2^j >= width and 3^k >= height, lies in pixel (floor x, floor y); in every run of
    2^j 3^k consecutive indices every pixel of that grid is visited once, so the s-th sample of a pixel is the one of the s-th run.
    Returns u[height, width, spp, 2] = the positions inside the pixels (the sampler's first two dimensions) and the indices."""
    sx = 1
    while sx < min(width, 128):
        sx *= 2
    sy = 1
    while sy < min(height, 128):
        sy *= 3
    stride = sx * sy
    i = np.arange(stride * spp, dtype=np.int64)
    x, y = radical_inverse(2, i) * sx, radical_inverse(3, i) * sy
    # Phi_3 in float64 can land a hair under an integer it equals exactly: floor on the exact digits instead
    px = np.zeros(len(i), dtype=np.int64)
    t, m = i.copy(), sx
    while m > 1:
        m //= 2
        px += (t % 2) * m
        t //= 2
    py = np.zeros(len(i), dtype=np.int64)
    t, m = i.copy(), sy
    while m > 1:
        m //= 3
        py += (t % 3) * m
        t //= 3
    u = np.full((height, width, spp, 2), np.nan)
    idx = np.full((height, width, spp), -1, dtype=np.int64)
    s = i // stride
    inside = (px < width) & (py < height)
    u[py[inside], px[inside], s[inside], 0] = (x - px)[inside]
    u[py[inside], px[inside], s[inside], 1] = (y - py)[inside]
    idx[py[inside], px[inside], s[inside]] = i[inside]
    assert not np.isnan(u).any()          # every pixel once per run
    return u, idx
