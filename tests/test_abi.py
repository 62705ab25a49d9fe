"""C-ABI library: loads, exports every symbol include/pbrt_hip.h declares, host entry points work
without a GPU, compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import oracle
import pbrt_hip
from pbrt_hip import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    header = open(os.path.join(ROOT, "include", "pbrt_hip.h")).read()
    declared = set(re.findall(r"\b(pbrt_hip_\w+)\s*\(", header))
    assert len(declared) >= 18
    L = pbrt_hip.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/pbrt_hip.h but not exported"
    assert declared == set(pbrt_hip.EXPORTS)


def test_struct_sizes_match_header(tmp_path):
    assert ctypes.sizeof(pbrt_hip.RenderParams) == 128  # 11 x i32, pad, u64 seed, 4 x i32, 2 x f32, pointer, 10 x i32
    # ... and against the header itself: a C99 translation unit prints what the compiler lays out
    src = tmp_path / "layout.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "pbrt_hip.h"\nint main(void) { printf("%zu %zu %zu %zu %zu\\n", '
                   'sizeof(PbrtRenderParams), offsetof(PbrtRenderParams, seed), offsetof(PbrtRenderParams, filter_table), '
                   'offsetof(PbrtRenderParams, samples_per_wave), sizeof(PbrtRenderStats)); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)], text=True).split()]
    P = pbrt_hip.RenderParams
    assert got == [ctypes.sizeof(P), P.seed.offset, P.filter_table.offset, P.samples_per_wave.offset, ctypes.sizeof(pbrt_hip.RenderStats)]
    assert ctypes.sizeof(pbrt_hip.RenderStats) == 48
    assert scenes.CAMERA_DTYPE.itemsize == 160


def test_product_does_not_touch_the_oracle():
    """Nothing under pbrt-rs_amd/ may import, link or execute anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pbrt-rs_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".sh")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "import oracle" not in txt and "oracle/" not in txt.replace(
                    "see oracle o_reflection.h", ""), f
    out = os.popen(f"ldd {pbrt_hip.LIB_PATH}").read()
    assert "oracle" not in out


@pytest.mark.parametrize("split", [pbrt_hip.SPLIT_SAH, pbrt_hip.SPLIT_HLBVH, pbrt_hip.SPLIT_MIDDLE,
                                   pbrt_hip.SPLIT_EQUAL_COUNTS])
@pytest.mark.parametrize("max_prims", [1, 4, 255])
def test_host_bvh_build_equals_oracle(split, max_prims):
    """BVHAccel::new on the host (csrc/host_bvh.cpp) vs the oracle's restatement: identical node array and leaf order."""
    for sc in (scenes.cornell_box(), scenes.random_triangles(20_000, seq=5, size=0.05), scenes.mixed_materials_scene(3000)):
        nodes, order = pbrt_hip.bvh_build(sc["positions"], sc["indices"], max_prims, split)
        osc = oracle.OracleScene(sc, max_prims, split)
        assert nodes.tobytes() == osc.nodes().tobytes()
        assert np.array_equal(order, osc.prim_order())
        # structural invariants of the flat layout (bvh.rs:774-811)
        leaf = nodes["n_primitives"] > 0
        assert leaf.sum() == (len(nodes) + 1) // 2
        assert np.all(nodes["offset"][~leaf] > np.nonzero(~leaf)[0] + 1)   # second child after the first subtree
        assert nodes["n_primitives"].sum() == len(sc["indices"])
        assert sorted(order.tolist()) == list(range(len(sc["indices"])))
        # (leaves may exceed max_prims only when all centroids coincide, bvh.rs:312-326: e.g. the two
        # triangles of an axis-aligned quad have the same bounds, hence the same centroid)
        osc.close()


def test_two_level_build_equals_oracle():
    """Object-level tree + TransformedPrimitive world bounds + top-level tree vs the oracle."""
    sc = scenes.instanced_scene(1500, 37, extent=1.2)
    blas_nodes, blas_order, inst, tlas_nodes, tlas_order = pbrt_hip.build_two_level(sc)
    osc = oracle.OracleScene(sc)
    bn, bo = osc.blas()
    assert blas_nodes.tobytes() == bn.tobytes() and np.array_equal(blas_order, bo)
    assert tlas_nodes.tobytes() == osc.nodes().tobytes() and np.array_equal(tlas_order, osc.prim_order())
    osc.close()


def test_bvh_build_edge_cases():
    nodes, order = pbrt_hip.bvh_build(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    assert len(nodes) == 0 and len(order) == 0                       # bvh.rs:228-230 empty aggregate
    one = scenes.furnace_scene()
    nodes, order = pbrt_hip.bvh_build(one["positions"], one["indices"][:1])
    assert len(nodes) == 1 and nodes[0]["n_primitives"] == 1         # single leaf root
    # identical centroids -> a leaf holding all of them (bvh.rs:312-326)
    p = np.tile(one["positions"][:3], (5, 1))
    idx = np.arange(15, dtype=np.int32).reshape(5, 3)
    nodes, order = pbrt_hip.bvh_build(p, idx, 4)
    assert len(nodes) == 1 and nodes[0]["n_primitives"] == 5
    # bad index -> error status, not a crash
    bad = idx.copy()
    bad[0, 0] = 99
    with pytest.raises(pbrt_hip.PbrtHipError):
        pbrt_hip.bvh_build(p, bad)


def test_tile_partition_covers_bounds_once():
    bounds = (5, 3, 171, 150)
    seen = {}
    for world in (1, 2, 3, 8):
        tiles = [pbrt_hip.tile_partition(bounds, r, world) for r in range(world)]
        allt = np.concatenate(tiles)
        assert len({tuple(t) for t in allt.tolist()}) == len(allt)   # disjoint
        seen[world] = sorted(map(tuple, allt.tolist()))
        sizes = [len(t) for t in tiles]
        assert max(sizes) - min(sizes) <= 1                            # balanced round-robin
    assert seen[1] == seen[2] == seen[3] == seen[8]
    ntx, nty = (171 - 5 + 15) // 16, (150 - 3 + 15) // 16
    assert len(seen[1]) == ntx * nty


def test_failed_context_creation_from_several_threads():
    """pbrt_hip_last_error(NULL) is thread-local (the text of the calling thread's last failed pbrt_hip_context_create): many
    threads failing at once neither crash nor read a torn string. (With a GPU: tests/test_gpu_intersect.py, distinct reasons.)"""
    import threading
    L = pbrt_hip.lib()
    L.pbrt_hip_last_error.restype = ctypes.c_char_p
    bad = []

    def worker():
        for _ in range(500):
            h = ctypes.c_void_p()
            rc = L.pbrt_hip_context_create(1 << 20, ctypes.byref(h))
            txt = L.pbrt_hip_last_error(None)
            if rc == 0 or h.value or txt not in (b"device_id out of range",
                                                 b"no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)"):
                bad.append((rc, txt))
    threads = [threading.Thread(target=worker) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not bad, bad[:3]
    assert L.pbrt_hip_context_is_lost(None) == -1


def test_a_failed_host_allocation_is_a_status_not_an_exception():
    """no exceptions or panics across the boundary. Every status-returning entry point is a function-try-block
    (csrc/abi_guard.h); here the tile deal of a frame of 1.6e9 tiles (25 GB of keys) runs under a 3 GB address-space limit in
    a child process: the call returns PBRT_HIP_ERR_OOM, the process lives on and the library still works."""
    code = (
        "import ctypes, resource, sys\n"
        "sys.path.insert(0, %r)\n"
        "import pbrt_hip\n"
        "L = pbrt_hip.lib()\n"
        "resource.setrlimit(resource.RLIMIT_AS, (3 << 30, 3 << 30))\n"
        "n = ctypes.c_int32()\n"
        "out = (ctypes.c_int32 * 8)()\n"
        "rc = L.pbrt_hip_tile_partition_order(0, 0, 640000, 640000, 0, 1 << 30, 0, out, 4, ctypes.byref(n))\n"
        "assert rc == 4, rc          # PBRT_HIP_ERR_OOM\n"
        "assert len(pbrt_hip.tile_partition((0, 0, 64, 64), 0, 2)) == 8\n"
        "print('survived')\n") % os.path.join(ROOT, "pbrt-rs_amd")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "survived" in r.stdout, r.stderr[-1500:]
    # every status-returning entry point of the library carries the guard
    import glob
    n_guarded = n_entry = 0
    for f in glob.glob(os.path.join(ROOT, "pbrt-rs_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "pbrt-rs_amd", "csrc", "*.cpp")):
        src = open(f).read()
        for m in re.finditer(r'^extern "C" int [^;{]*\{', src, re.M):
            n_entry += 1
            n_guarded += m.group(0).rstrip().endswith("try {")
    assert n_entry >= 45 and n_guarded == n_entry, (n_entry, n_guarded)


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pbrt_hip.PbrtHipError) as e:
        pbrt_hip.Context(0)
    assert "no HIP device" in str(e.value) or "failed" in str(e.value)


def test_comm_without_a_gpu_is_an_error():
    """The RCCL film-merge entry points refuse to start without a device; bad arguments are errors."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pbrt_hip.PbrtHipError) as e:
        pbrt_hip.comm_unique_id()
    assert "no HIP device" in str(e.value) or "RCCL" in str(e.value)
    L = pbrt_hip.lib()
    out = ctypes.c_void_p()
    buf = (ctypes.c_uint8 * 128)()
    assert L.pbrt_hip_comm_create(None, 2, 0, buf, ctypes.byref(out)) == 1 and not out.value      # no context
    assert L.pbrt_hip_film_reduce(None, None, 0, 0) == 1
    L.pbrt_hip_comm_destroy(None)                                                                   # a no-op
    film = ctypes.c_void_p()
    assert L.pbrt_hip_film_create(None, 16, ctypes.byref(film)) == 1 and not film.value             # the ABI's own device film: no context
    assert L.pbrt_hip_film_download(None, None, 16, None) == 1
    L.pbrt_hip_film_destroy(None, None)                                                             # a no-op


def test_null_handles_are_errors_on_the_round4_entry_points():
    """pbrt_hip_context_is_lost / set_wide_build / probe_state_stream (round 4) with no context: error codes, no dereference."""
    L = pbrt_hip.lib()
    assert L.pbrt_hip_context_is_lost(None) == -1
    assert L.pbrt_hip_context_set_wide_build(None, 0) == 1
    rd, wr, ms = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_double()
    assert L.pbrt_hip_probe_state_stream(None, 1 << 20, 700, 1 << 20, 31, ctypes.byref(rd), ctypes.byref(wr), ctypes.byref(ms)) == 1
    assert (pbrt_hip.TRAVERSAL_AUTO, pbrt_hip.WIDE_BUILD_DEVICE, pbrt_hip.WIDE_BUILD_HOST, pbrt_hip.WIDE_BUILD_NONE) == (0, 0, 1, 2)


@pytest.mark.parametrize("kind,rx,ry,a,b", [("box", 0.5, 0.5, 0, 0), ("gaussian", 2.0, 2.0, 2.0, 0),
                                            ("mitchell", 2.0, 2.0, 1 / 3, 1 / 3), ("lanczos", 4.0, 4.0, 3.0, 0),
                                            ("triangle", 2.0, 1.5, 0, 0)])
def test_filter_tables_and_sample_bounds(kind, rx, ry, a, b):
    """Film::new's filter table (film.rs:52-63) and get_sample_bounds: host library == oracle restatement."""
    _, _, t = pbrt_hip.filter_table(kind, rx, ry, a, b)
    assert np.array_equal(t, oracle.filter_table(kind, rx, ry, a, b))
    t = t.reshape(16, 16)
    if kind == "box":
        assert np.all(t == 1.0)
    else:
        assert t[0, 0] == t.max() and t[0, 0] > 0 and abs(t[15, 15]) < 0.05 * t[0, 0]
        if kind in ("gaussian", "triangle"):
            assert np.all(np.diff(t[0]) <= 0) and np.all(np.diff(t[:, 0]) <= 0) and t.min() >= 0
    for w, h in ((64, 48), (1920, 1080)):
        assert pbrt_hip.sample_bounds(w, h, rx, ry) == oracle.sample_bounds(w, h, rx, ry)
    assert pbrt_hip.sample_bounds(64, 48) == (0, 0, 64, 48)
    assert pbrt_hip.sample_bounds(64, 48, 2.0, 2.0) == (-2, -2, 66, 50)


def test_write_pfm_roundtrip(tmp_path):
    rgb = scenes.pcg32_float(3, 7 * 5 * 3).reshape(5, 7, 3)
    path = tmp_path / "img.pfm"
    pbrt_hip.write_pfm(path, rgb)
    raw = open(path, "rb").read()
    header, body = raw.split(b"-1.0\n", 1)
    assert header == b"PF\n7 5\n"
    back = np.frombuffer(body, dtype="<f4").reshape(5, 7, 3)[::-1]
    assert np.array_equal(back, rgb)


def _read_exr(raw):
    """Minimal reader of a single-part uncompressed scanline OpenEXR file -> (attributes, {channel: [h, w] float32})."""
    import struct
    assert struct.unpack("<iI", raw[:8]) == (20000630, 2)
    pos, attrs = 8, {}

    def cstr(p):
        e = raw.index(b"\0", p)
        return raw[p:e].decode(), e + 1

    while raw[pos] != 0:
        name, pos = cstr(pos)
        typ, pos = cstr(pos)
        (size,) = struct.unpack("<i", raw[pos:pos + 4])
        attrs[name] = (typ, raw[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    chans, p, blob = [], 0, attrs["channels"][1]
    while blob[p] != 0:
        e = blob.index(b"\0", p)
        name = blob[p:e].decode()
        ptype, plinear, xs, ys = struct.unpack("<iB3xii", blob[e + 1:e + 17])
        chans.append((name, ptype, xs, ys))
        p = e + 17
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    offsets = struct.unpack(f"<{h}Q", raw[pos:pos + 8 * h])
    out = {c[0]: np.zeros((h, w), dtype=np.float32) for c in chans}
    for row, off in enumerate(offsets):
        y, nbytes = struct.unpack("<ii", raw[off:off + 8])
        assert y == y0 + row and nbytes == 4 * w * len(chans)
        data = np.frombuffer(raw[off + 8:off + 8 + nbytes], dtype="<f4").reshape(len(chans), w)
        for k, c in enumerate(chans):
            out[c[0]][row] = data[k]
    assert offsets[-1] + 8 + 4 * w * len(chans) == len(raw)
    return attrs, chans, out


def test_write_exr_reads_back(tmp_path):
    """pbrt_hip_write_exr: the OpenEXR layout (magic / version, the eight required attributes, alphabetical float
    channels, offset table, one scanline per chunk) and the exact linear values, negative and > 1 included."""
    import struct
    w, h = 37, 23
    rgb = (scenes.pcg32_float(8, w * h * 3).reshape(h, w, 3) * 40.0 - 2.0).astype(np.float32)
    path = tmp_path / "img.exr"
    pbrt_hip.write_exr(path, rgb)
    attrs, chans, planes = _read_exr(open(path, "rb").read())
    assert set(attrs) == {"channels", "compression", "dataWindow", "displayWindow", "lineOrder", "pixelAspectRatio",
                          "screenWindowCenter", "screenWindowWidth"}
    assert chans == [("B", 2, 1, 1), ("G", 2, 1, 1), ("R", 2, 1, 1)]                 # FLOAT, alphabetical
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    assert struct.unpack("<4i", attrs["displayWindow"][1]) == (0, 0, w - 1, h - 1)
    assert struct.unpack("<f", attrs["pixelAspectRatio"][1]) == (1.0,)
    assert np.array_equal(planes["R"], rgb[..., 0]) and np.array_equal(planes["G"], rgb[..., 1])
    assert np.array_equal(planes["B"], rgb[..., 2])
    L = pbrt_hip.lib()
    assert L.pbrt_hip_write_exr(None, None, 1, 1) == 1 and L.pbrt_hip_write_exr(b"/nonexistent/x.exr", rgb.ctypes.data, w, h) == 1


def test_write_png_decodes_back(tmp_path):
    """pbrt_hip_write_png: a valid PNG (signature, chunk CRCs, zlib stream with Adler-32) whose pixels are the
    sRGB-encoded image; decoded here with zlib and checked against the transfer curve."""
    import struct
    import zlib
    w, h = 301, 223                                           # > 65535 raw bytes: several stored deflate blocks
    rgb = (scenes.pcg32_float(5, w * h * 3).reshape(h, w, 3) * 1.2 - 0.1).astype(np.float32)   # some < 0 and > 1
    path = tmp_path / "img.png"
    pbrt_hip.write_png(path, rgb)
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(typ + data) & 0xffffffff
        chunks.append((typ, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
    pix = np.frombuffer(zlib.decompress(chunks[1][1]), dtype=np.uint8).reshape(h, 1 + 3 * w)
    assert np.all(pix[:, 0] == 0)
    v = np.clip(rgb.astype(np.float64), 0.0, None)
    srgb = np.where(v <= 0.0031308, 12.92 * v, 1.055 * np.power(v, 1 / 2.4) - 0.055)
    expect = np.clip(np.floor(255.0 * srgb + 0.5), 0, 255)
    got = pix[:, 1:].reshape(h, w, 3).astype(np.float64)
    assert np.abs(got - expect).max() <= 1                    # float32 pow vs float64
    assert (got == expect).mean() > 0.99


def test_too_deep_tree_is_refused_without_a_gpu_fault():
    """A BVH deeper than the reference's 64-entry stack (bvh.rs:839) is rejected at scene creation. The check runs on
    the host before any device work; without a GPU the call fails earlier on the missing context, so only the
    builder side is exercised here: a degenerate "Middle" split over collinear, exponentially spaced centroids."""
    n = 80
    x = (3.0 ** np.arange(n)).astype(np.float32)      # the midpoint of the centroid bounds peels off one triangle per level
    pos = np.zeros((3 * n, 3), dtype=np.float32)
    pos[0::3, 0], pos[1::3, 0], pos[2::3, 0] = x, x, x
    pos[1::3, 1], pos[2::3, 2] = 1e-3, 1e-3
    idx = np.arange(3 * n, dtype=np.int32).reshape(n, 3)
    nodes, order = pbrt_hip.bvh_build(pos, idx, 1, pbrt_hip.SPLIT_MIDDLE)
    # depth of the flat tree
    depth, stack = 0, [(0, 1)]
    while stack:
        i, d = stack.pop()
        depth = max(depth, d)
        if nodes["n_primitives"][i] == 0:
            stack += [(i + 1, d + 1), (int(nodes["offset"][i]), d + 1)]
    assert depth > 64                      # the scene below must be refused (checked on the GPU box in test_gpu_intersect)


def test_general_two_level_host_trees_equal_oracle():
    """Host builds for the general two-level scene (one tree per object aggregate, the top-level tree over the instances'
    world bounds followed by the world triangles') against the oracle's BVHAccel::new on the same primitives. No GPU."""
    import oracle
    from pbrt_hip import scenes
    sc = scenes.two_level_scene(n_instances=40)
    osc = oracle.OracleScene(sc)
    trees, inst, tlas_nodes, tlas_order = pbrt_hip.build_general_two_level(sc)
    assert tlas_nodes.tobytes() == osc.nodes().tobytes() and np.array_equal(tlas_order, osc.prim_order())
    assert len(tlas_order) == 40 + 4
    for k, (nodes, order) in enumerate(trees):
        on, oo = osc.object_tree(k)
        assert nodes.tobytes() == on.tobytes() and np.array_equal(order, oo)
    osc.close()
