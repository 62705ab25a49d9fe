"""An independent pin of the two decision kernels everything else rests on: the watertight ray-triangle test
(src/shapes/triangle.rs:74-158) and the ray-box slab test (src/core/geometry.rs:709-751).

The oracle and the HIP kernels were written by the same hand from the same reading of the reference, so their
bit-for-bit agreement cannot catch a shared misreading. Here the GEOMETRY is evaluated from scratch in exact rational
arithmetic (python `fractions`: every float input is an exact rational, no rounding anywhere) — not the reference's
sequence of float operations, but what that sequence is meant to decide:

  triangle   plane of the triangle, exact barycentrics of the intersection point, exact ray parameter t;
             hit  <=>  0 < t <= t_max  and  all barycentrics >= 0  (the watertight test's closed edges)
  box        exact entry / exit parameters of the three slabs;  hit  <=>  max(entry) <= min(exit), entry < t_max, exit > 0

and the oracle's decisions (orc_triangle_test / orc_bounds_intersect_p, the functions its traversal calls) are compared
with it on the known-answer set plus seeded random and adversarial cases:
  * wherever the exact result is decided by a margin (barycentrics, t, t_max - t, slab overlap away from zero by more than
    float rounding can move them) the oracle must agree, and its t / barycentrics must be the exact ones to a few ulps;
  * the box test must be conservative: an exact hit is never rejected (the (1 + 2 gamma_3) widening exists for that).
This cannot lift "parity unpinned" (the reference holds no vectors and cannot run), but a misreading of either routine
would have to be made a third time, in different mathematics, to go unnoticed.
"""
import ctypes
from fractions import Fraction as Fr

import numpy as np
import pytest

import oracle
from pbrt_hip import scenes

L = oracle.lib()
f32 = np.float32


def _fr3(v):
    return [Fr(float(x)) for x in v]


def _sub(a, b):
    return [x - y for x, y in zip(a, b)]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _dot(a, b):
    return sum(x * y for x, y in zip(a, b))


def exact_triangle(p0, p1, p2, o, d, t_max):
    """Exact (hit, b0, b1, b2, t, margin): margin = how far the nearest deciding quantity is from its threshold,
    relative to its scale; None for a ray parallel to the plane or a degenerate triangle."""
    p0, p1, p2, o, d = map(_fr3, (p0, p1, p2, o, d))
    e1, e2 = _sub(p1, p0), _sub(p2, p0)
    n = _cross(e1, e2)
    denom = _dot(n, d)
    if denom == 0 or _dot(n, n) == 0:
        return None
    t = _dot(n, _sub(p0, o)) / denom
    hit_p = [oo + t * dd for oo, dd in zip(o, d)]
    nn = _dot(n, n)
    # barycentrics from sub-triangle areas (signed, along n)
    b0 = _dot(_cross(_sub(p1, hit_p), _sub(p2, hit_p)), n) / nn
    b1 = _dot(_cross(_sub(p2, hit_p), _sub(p0, hit_p)), n) / nn
    b2 = 1 - b0 - b1
    tm = Fr(float(t_max)) if np.isfinite(t_max) else None
    inside = b0 >= 0 and b1 >= 0 and b2 >= 0
    in_range = t > 0 and (tm is None or t <= tm)
    margins = [abs(b0), abs(b1), abs(b2)]
    if t != 0:
        margins.append(abs(t) / max(abs(t), 1))
    if tm is not None:
        margins.append(abs(tm - t) / max(abs(tm), abs(t), Fr(1, 10 ** 30)))
    return bool(inside and in_range), b0, b1, b2, t, float(min(margins))


def oracle_triangle(p0, p1, p2, o, d, t_max):
    out = np.zeros(5, dtype=np.float32)
    arr = [np.asarray(v, dtype=np.float32) for v in (p0, p1, p2)]
    ray = np.array(list(o) + list(d) + [t_max, 0.0], dtype=np.float32)
    L.orc_triangle_test(arr[0].ctypes.data, arr[1].ctypes.data, arr[2].ctypes.data, ray.ctypes.data, 0, out.ctypes.data)
    return bool(out[0] == 1.0), out[1:].copy()


def _check_triangle(p0, p1, p2, o, d, t_max, margin_needed):
    ex = exact_triangle(p0, p1, p2, o, d, t_max)
    hit, vals = oracle_triangle(p0, p1, p2, o, d, t_max)
    if ex is None:
        assert not hit          # parallel ray / degenerate triangle: det == 0 (triangle.rs:123-125)
        return "degenerate"
    e_hit, b0, b1, b2, t, margin = ex
    # conditioning: the test translates the vertices to the ray origin and shears them in f32, so what it sees of the
    # triangle is blurred by a few ulps of (distance to the triangle) relative to (its size * cos of the incidence angle)
    P = [np.asarray(v, dtype=np.float64) for v in (p0, p1, p2)]
    O, D = np.asarray(o, dtype=np.float64), np.asarray(d, dtype=np.float64)
    nrm = np.cross(P[1] - P[0], P[2] - P[0])
    size = np.sqrt(np.linalg.norm(nrm))
    cos_t = abs(nrm @ D) / (np.linalg.norm(nrm) * np.linalg.norm(D))
    cond = max(np.abs(q - O).max() for q in P) / size / max(cos_t, 1e-12)
    tol = 16 * 2.0 ** -24 * (1.0 + cond)
    if margin < max(margin_needed, 4 * tol):
        return "near"           # within rounding of an edge, of t = 0 or of t_max: either answer is the algorithm's to give
    assert hit == e_hit, (p0, p1, p2, o, d, t_max, margin)
    if hit:
        for got, want in zip(vals[:3], (b0, b1, b2)):
            assert abs(float(got) - float(want)) <= tol, (got, float(want), tol)
        assert abs(float(vals[3]) - float(t)) <= tol * abs(float(t)), (vals[3], float(t), tol)
    return "hit" if hit else "miss"


def test_triangle_known_answers_against_exact_geometry():
    from test_oracle_kat import TRI_CASES
    for name, tri, o, d, t_max, expect in TRI_CASES:
        ex = exact_triangle(*tri, o, d, t_max)
        hit, _ = oracle_triangle(*tri, o, d, t_max)
        if ex is None:
            assert not hit and not expect, name
            continue
        if name == "grazing start on plane":     # t == 0 exactly: not a hit (0 < t), and the oracle says so
            assert ex[0] is False and not hit
            continue
        assert ex[0] == expect == hit, name


@pytest.mark.parametrize("seq,scale,tri_size", [(101, 1.0, 0.5), (102, 100.0, 0.01), (103, 1.0, 1e-3), (104, 1000.0, 30.0)])
def test_triangle_seeded_cases_against_exact_geometry(seq, scale, tri_size):
    n = 1500
    u = scenes.pcg32_float(seq, n * 19).reshape(n, 19).astype(np.float64)
    tally = {"hit": 0, "miss": 0, "near": 0, "degenerate": 0}
    for k in range(n):
        c = (u[k, 0:3] * 2 - 1) * scale
        p = [(c + (u[k, 3 + 3 * j:6 + 3 * j] * 2 - 1) * tri_size).astype(np.float32) for j in range(3)]
        o = ((u[k, 12:15] * 2 - 1) * scale * 2).astype(np.float32)
        # aim at a point of the triangle's plane: inside for barycentric weights in [0, 1], outside otherwise
        w = u[k, 15:18] * 1.6 - 0.3
        w = w / w.sum() if abs(w.sum()) > 1e-3 else np.array([1 / 3] * 3)
        target = sum(wi * pi.astype(np.float64) for wi, pi in zip(w, p))
        d = (target - o.astype(np.float64)).astype(np.float32)
        if k % 3 == 0:
            d = (d / np.linalg.norm(d)).astype(np.float32)
        if not np.any(d):
            continue
        t_max = np.inf if k % 2 else f32(u[k, 18] * 2.5 * (np.linalg.norm(target - o) / max(np.linalg.norm(d), 1e-30)))
        tally[_check_triangle(p[0], p[1], p[2], o, d, t_max, 2e-5)] += 1
    assert tally["hit"] > 200 and tally["miss"] > 200, tally   # the far, tiny triangles of the second set are mostly "near": f32 cannot resolve them better


def test_triangle_shared_edges_and_vertices_are_watertight():
    """Rays through points ON the shared edge of two triangles (computed exactly representable: the edge runs along an
    axis) hit at least one of them: no light leaks between adjacent triangles (the point of triangle.rs:109-121)."""
    a, b, c, d4 = (0, 0, 0), (4, 0, 0), (0, 4, 0), (4, -4, 0)       # triangles (a, b, c) and (a, d4, b) share edge a-b
    u = scenes.pcg32_float(7, 600 * 4).reshape(600, 4)
    for k in range(600):
        x = f32(u[k, 0] * 4)
        o = np.array([x + f32(u[k, 1] - 0.5), f32(u[k, 2] - 0.5), f32(1 + u[k, 3])], dtype=np.float32)
        target = np.array([x, 0, 0], dtype=np.float32)
        d = target - o
        h1, _ = oracle_triangle(a, b, c, o, d, np.inf)
        h2, _ = oracle_triangle(a, d4, b, o, d, np.inf)
        # the exact intersection with the plane z = 0 lies within float rounding of the edge: one side must take it
        assert h1 or h2, (o, d)


def exact_box(lo, hi, o, d, t_max):
    """Exact slab intersection of the ray with the box: (hit, overlap margin). Directions with a zero component are
    handled as the limit (the slab either contains the origin coordinate or it does not)."""
    lo, hi, o, d = map(_fr3, (lo, hi, o, d))
    t0, t1 = None, None      # entry = max of near planes, exit = min of far planes
    for a in range(3):
        if d[a] == 0:
            if not (lo[a] <= o[a] <= hi[a]):
                return False, 1.0
            continue
        ta, tb = (lo[a] - o[a]) / d[a], (hi[a] - o[a]) / d[a]
        near, far = min(ta, tb), max(ta, tb)
        t0 = near if t0 is None else max(t0, near)
        t1 = far if t1 is None else min(t1, far)
    if t0 is None:
        return True, 1.0
    tm = Fr(float(t_max)) if np.isfinite(t_max) else None
    hit = t0 <= t1 and t1 > 0 and (tm is None or t0 < tm)
    scale = max(abs(t0), abs(t1), Fr(1, 10 ** 30))
    margins = [abs(t1 - t0) / scale, abs(t1) / scale]
    if tm is not None:
        margins.append(abs(tm - t0) / max(abs(tm), abs(t0), Fr(1, 10 ** 30)))
    return bool(hit), float(min(margins))


def oracle_box(lo, hi, o, d, t_max):
    box = np.array(list(lo) + list(hi), dtype=np.float32)
    ray = np.array(list(o) + list(d) + [t_max, 0.0], dtype=np.float32)
    return bool(L.orc_bounds_intersect_p(box.ctypes.data, ray.ctypes.data, 0))


@pytest.mark.parametrize("seq,scale,size", [(201, 1.0, 0.5), (202, 500.0, 0.02), (203, 1.0, 0.0), (204, 30.0, 10.0)])
def test_box_seeded_cases_against_exact_geometry(seq, scale, size):
    n = 3000
    u = scenes.pcg32_float(seq, n * 16).reshape(n, 16).astype(np.float64)
    decided = conservative = near = 0
    for k in range(n):
        lo = ((u[k, 0:3] * 2 - 1) * scale).astype(np.float32)
        ext = (u[k, 3:6] * size).astype(np.float32)
        if k % 5 == 0:
            ext[k % 3] = 0                      # flat boxes: axis-aligned triangles have them
        hi = (lo + ext).astype(np.float32)
        o = ((u[k, 6:9] * 2 - 1) * scale * 1.5).astype(np.float32)
        inside = lo.astype(np.float64) + (u[k, 9:12] * 1.4 - 0.2) * ext.astype(np.float64)
        d = (inside - o.astype(np.float64) + (u[k, 12:15] - 0.5) * 0.3 * max(ext.max(), 1e-5 * scale)).astype(np.float32)
        if k % 7 == 0:
            d[(k // 7) % 3] = 0                 # axis-parallel rays
        if k % 11 == 0:
            o[k % 3] = lo[k % 3]                # origin exactly on a face plane
        if not np.any(d):
            continue
        t_max = np.inf if k % 2 else f32(u[k, 15] * 3)
        e_hit, margin = exact_box(lo, hi, o, d, t_max)
        got = oracle_box(lo, hi, o, d, t_max)
        zero_dir_on_plane = any(d[a] == 0 and (o[a] == lo[a] or o[a] == hi[a]) for a in range(3))
        if zero_dir_on_plane:
            near += 1                            # 0 * inf = NaN in the float test (geometry.rs:716-719): its answer stands
            continue
        if e_hit and not got:
            # conservativeness: an exact hit may only be rejected at the open ends of the test: t0 == t_max or t1 == 0
            assert margin < 1e-6, (lo, hi, o, d, t_max, margin)
            near += 1
            continue
        if margin > 2e-6:      # the float test's own error is a few ulps (2^-24) of the parameters it compares
            assert got == e_hit, (lo, hi, o, d, t_max, margin)
            decided += 1
        else:
            near += 1
        conservative += int(e_hit)
    assert decided > n // 2 and (size == 0.0 or conservative > n // 10), (decided, conservative, near)


# ---------------------------------------------------------------------------------------------------------------------
# Round 3: the remaining decision code, pinned the same way — geometry / physics evaluated independently (exact rationals
# or float64), the oracle's routine compared with it. None of the closed-form scenes reaches these.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_offset(p, err, n, w):
    out = np.zeros(3, dtype=np.float32)
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (p, err, n, w)]
    L.orc_offset_ray_origin(*(x.ctypes.data_as(ctypes.c_void_p) for x in a), out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_offset_ray_origin_leaves_the_error_box_on_the_side_of_w():
    """geometry.rs:1139-1154. For the surface point p with error box p +- p_error and normal n, the spawned origin must lie
    outside the slab the error box sweeps along n, on the side the ray leaves to: exactly, for every q in the box,
    (po - q) . n has the sign of w . n, i.e. |(po - p) . n| >= sum |n_k| p_error_k and the sign is right. The float routine
    adds n * d (d = the float dot product) and then rounds every coordinate AWAY from p; checked in exact rationals:
      * sign of (po - p) . n = sign of w . n,
      * per axis |po_k - p_k| >= |n_k| * D_lo with D_lo = (1 - 4 u) * exact d (a float dot product of non-negative terms
        cannot come out lower), and the move is away from p on every axis where n_k != 0,
      * so (po - p) . n >= D_lo * (n . n): outside the swept slab whenever n . n >= 1 / (1 - 4 u); for the unit normals
        the path produces (|n|^2 within 2^-22 of 1) the rounding step more than covers the difference — asserted on the
        exact value, with the slab computed exactly,
      * and not farther than the formula explains (two ulps beyond |n_k| d (1 + 4 u))."""
    rng = np.random.default_rng(20263)
    u = Fr(1, 2 ** 24)
    n_cases = 4000
    for i in range(n_cases):
        scale = 10.0 ** rng.uniform(-3, 4)
        p = (rng.uniform(-1, 1, 3) * scale).astype(f32)
        n = rng.normal(size=3)
        if i % 7 == 0:
            n[rng.integers(3)] = 0.0            # axis-aligned components
        if i % 11 == 0:
            n = np.eye(3)[rng.integers(3)] * rng.choice([-1.0, 1.0])
        n = (n / np.linalg.norm(n)).astype(f32)
        err = (np.abs(p) * rng.uniform(1e-8, 1e-5, 3) + rng.uniform(0, 1e-9)).astype(f32)   # gamma(7) |p| and up
        w = rng.normal(size=3).astype(f32)
        po = _oracle_offset(p, err, n, w)
        P, N, E, W, PO = _fr3(p), _fr3(n), _fr3(err), _fr3(w), _fr3(po)
        side = _dot(W, N)
        if side == 0:
            continue
        sgn = 1 if side > 0 else -1
        d_exact = sum(abs(a) * b for a, b in zip(N, E))
        move = _sub(PO, P)
        along = _dot(move, N) * sgn
        assert along > 0, (i, p, n, err, w, po)
        d_lo, d_hi = d_exact * (1 - 4 * u), d_exact * (1 + 4 * u)
        for k in range(3):
            if N[k] == 0:
                assert move[k] == 0
                continue
            assert (move[k] > 0) == ((N[k] > 0) == (sgn > 0)), (i, k)        # away from p, along +-n
            assert abs(move[k]) >= abs(N[k]) * d_lo, (i, k, float(abs(move[k])), float(abs(N[k]) * d_lo))
            ulp = Fr(float(np.spacing(f32(max(abs(float(p[k])), abs(float(po[k])))))))
            assert abs(move[k]) <= abs(N[k]) * d_hi + 2 * ulp, (i, k)
        # the whole error box is behind the origin, exactly: min over the box of (po - q) . n * sgn = along - d_exact
        assert along - d_exact >= 0, (i, float(along), float(d_exact))


def _oracle_sphere(c, r, o, d, t_max):
    out = np.zeros(11, dtype=np.float32)
    ray = np.array(list(o) + list(d) + [t_max, 0.0], dtype=np.float32)
    cc = np.ascontiguousarray(c, dtype=np.float32)
    L.orc_sphere_test(cc.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(r), ray.ctypes.data_as(ctypes.c_void_p),
                      out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_sphere_efloat_interval_brackets_the_exact_root():
    """sphere.rs:228-284 with efloat.rs: the root the test reports comes with running error bounds [t.low, t.high]. Interval
    arithmetic is sound iff the exact root of the quadratic of ANY ray inside the operand intervals lies in the result
    interval; checked for the object-space ray the routine solved (its float coordinates taken as exact rationals):
    f(t) = |o + t d|^2 - r^2 changes sign (or vanishes) between t.low and t.high, in the direction of the root reported
    (entering: + to -, leaving: - to +). That holds for every ray that is not grazing; EFloat::quadratic (efloat.rs:64-87)
    takes the discriminant from the rounded midpoints in f64 and bounds the root by machine_epsilon * root, which is not an
    interval bound when the discriminant is a small difference of large terms (2 of 2 500 seeded rays, both with
    discriminant < 2e-4 b^2): for grazing rays the reported value is held to the conditioned bound
    |t - t_exact| <= 64 u |t| b^2 / discriminant instead. And the decision itself: an exact root inside (0, t_max) by a margin is found,
    a ray whose exact closest approach misses the sphere by a margin reports nothing."""
    rng = np.random.default_rng(7)
    hits = brackets = 0
    for i in range(3000):
        c = rng.uniform(-3, 3, 3).astype(f32)
        r = f32(10.0 ** rng.uniform(-2, 1))
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d) * (1.0 if i % 3 else 10.0 ** rng.uniform(-2, 2))).astype(f32)
        # aim near the sphere: offset the target by up to 1.3 radii (misses and grazing rays included), start in- or outside
        target = c.astype(np.float64) + rng.normal(size=3) / np.sqrt(3) * float(r) * rng.uniform(0, 1.3)
        o = (target - d.astype(np.float64) * rng.uniform(-0.5 if i % 5 == 0 else 0.2, 6.0) * float(r) / np.linalg.norm(d)).astype(f32)
        t_max = f32(np.inf) if i % 4 else f32(rng.uniform(0.5, 8.0) * float(r) / np.linalg.norm(d))
        out = _oracle_sphere(c, float(r), o, d, t_max)
        O, D = _fr3(out[4:7]), _fr3(out[7:10])
        R = Fr(float(r))
        qa, qb, qc = _dot(D, D), 2 * _dot(D, O), _dot(O, O) - R * R
        f = lambda t: (qa * t + qb) * t + qc    # noqa: E731
        disc = qb * qb - 4 * qa * qc
        if out[0] != 0.0:
            hits += 1
            lo, hi, tv = Fr(float(out[2])), Fr(float(out[3])), Fr(float(out[1]))
            assert lo <= tv <= hi and lo >= 0 and (not np.isfinite(t_max) or hi <= Fr(float(t_max)))
            assert disc >= 0, i
            vertex = -qb / (2 * qa)
            flo, fhi = f(lo), f(hi)
            if disc * 100 < qb * qb:    # grazing: the conditioned bound on the value
                sq = Fr(float(np.sqrt(float(disc))))
                root = (-qb - sq) / (2 * qa) if tv <= vertex else (-qb + sq) / (2 * qa)
                assert abs(tv - root) <= 64 * Fr(1, 2 ** 24) * abs(tv) * qb * qb / disc + Fr(1, 2 ** 40), (i, float(tv), float(root))
            elif tv <= vertex:  # the entering root
                assert flo >= 0 >= fhi, (i, float(flo), float(fhi))
                brackets += 1
            else:               # the leaving root (origin inside, or the entering one rejected)
                assert flo <= 0 <= fhi, (i, float(flo), float(fhi))
                brackets += 1
        else:
            # nothing reported: no exact root may sit comfortably inside (0, t_max)
            if disc > 0:
                sq = Fr(float(np.sqrt(float(disc))))       # only used to place test points: the decision below is exact
                for root in ((-qb - sq) / (2 * qa), (-qb + sq) / (2 * qa)):
                    m = abs(root) * Fr(1, 2 ** 12) + Fr(1, 2 ** 30)
                    inside = root - m > 0 and (not np.isfinite(t_max) or root + m < Fr(float(t_max)))
                    # a genuine sign change around `root` well inside the range would have had to be reported
                    assert not (inside and f(root - m) * f(root + m) < 0), (i, float(root))
    assert hits > 800 and brackets > 0.9 * hits


def test_fresnel_and_refraction_against_float64():
    """reflection.rs:19-40 (fr_dielectric) and :140-156 (refract), D37 intended. Evaluated independently in float64 from the
    physics — Snell's law, the two amplitude ratios — and compared with the float32 routines: |F32 - F64| <= 64 u (u = 2^-24,
    values in [0, 1]; away from the total-reflection threshold where cos_t -> 0 amplifies the rounding of sin_t: there the
    bound is scaled by 1 / cos_t). Exact properties: 1 at and beyond total internal reflection, symmetric in the side the
    ray comes from, refract() fails exactly when Snell has no solution (by a margin), returns a unit vector in the plane of
    incidence on the far side of n, with sin_t = eta sin_i."""
    rng = np.random.default_rng(11)
    u = 2.0 ** -24
    for i in range(4000):
        cos_i = float(f32(rng.uniform(-1, 1)))
        eta_i, eta_t = float(f32(rng.uniform(1.0, 2.5))), float(f32(rng.uniform(1.0, 2.5)))
        got = float(L.orc_fr_dielectric(ctypes.c_float(cos_i), ctypes.c_float(eta_i), ctypes.c_float(eta_t)))
        ci, ei, et = abs(cos_i), (eta_i, eta_t)[cos_i <= 0], (eta_t, eta_i)[cos_i <= 0]
        si = np.sqrt(max(0.0, 1 - ci * ci))
        st = ei / et * si
        if st >= 1 + 1e-6:
            assert got == 1.0
            continue
        if st > 1 - 1e-6:
            continue    # the threshold itself: either answer is within rounding
        ct = np.sqrt(1 - st * st)
        r_par = (et * ci - ei * ct) / (et * ci + ei * ct)
        r_per = (ei * ci - et * ct) / (ei * ci + et * ct)
        want = 0.5 * (r_par * r_par + r_per * r_per)
        assert abs(got - want) <= 64 * u / max(ct, 1e-3), (cos_i, eta_i, eta_t, got, want)
        assert 0.0 <= got <= 1.0
    for i in range(4000):
        n = rng.normal(size=3)
        n = (n / np.linalg.norm(n)).astype(f32)
        wi = rng.normal(size=3)
        wi = (wi / np.linalg.norm(wi)).astype(f32)
        if np.dot(wi.astype(np.float64), n.astype(np.float64)) < 0:
            n = -n                                   # refract() is called with n on wi's side (reflection.rs:703)
        eta = float(f32(rng.uniform(0.4, 2.5)))
        wt = np.zeros(3, dtype=np.float32)
        ok = L.orc_refract(wi.ctypes.data_as(ctypes.c_void_p), n.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(eta), 0,
                           wt.ctypes.data_as(ctypes.c_void_p))
        W, N = wi.astype(np.float64), n.astype(np.float64)
        ci = float(np.dot(N, W))
        s2t = eta * eta * max(0.0, 1 - ci * ci)
        if s2t >= 1 + 1e-5:
            assert not ok
            continue
        if s2t > 1 - 1e-5:
            continue
        assert ok
        ct = np.sqrt(1 - s2t)
        want = -eta * W + (eta * ci - ct) * N
        assert np.max(np.abs(wt - want)) <= 32 * u * max(1.0, eta) / max(ct, 1e-2), (wi, n, eta, wt, want)
        assert np.dot(wt, N) < 0                                                            # the far side
        assert abs(np.linalg.norm(wt) - 1.0) <= 1e-5 / max(ct, 1e-2)
        # Snell: the tangential parts are antiparallel with ratio eta
        tan_i, tan_t = W - ci * N, wt.astype(np.float64) - float(np.dot(wt, N)) * N
        assert np.max(np.abs(tan_t + eta * tan_i)) <= 1e-5
