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
