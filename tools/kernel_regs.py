"""VGPR / SGPR / LDS / scratch of the kernels in a built library, from the code objects' metadata notes.
usage: python tools/kernel_regs.py [lib.so] [name filter]"""
import os, re, struct, subprocess, sys, tempfile
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "pbrt-rs_amd", "pbrt_hip", "libpbrt_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
b = open(lib, "rb").read()
for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", b):
    i = m.start()
    n = struct.unpack_from("<Q", b, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p)
        name = b[p + 24:p + 24 + tl]
        p += 24 + tl
        if b"gfx950" not in name or size == 0:
            continue
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(b[i + off:i + off + size])
            f.flush()
            txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
        for blk in txt.split("- .agpr_count")[1:]:
            g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
            if flt in g("name"):
                print(f"{g('name')[:120]:120s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
