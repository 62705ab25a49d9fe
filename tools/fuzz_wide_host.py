"""CPU fuzz of the wide records host-side obligations (tests/test_wide_host.py: structure, visiting order, exact leaf boxes, filter
conservativeness under adversarial rays) on the random shared-vertex meshes of tools/fuzz_intersect.py. usage: python tools/fuzz_wide_host.py N
(needs tests/native/_build/libwide_check.so: run tests/test_wide_host.py once)."""
import sys, ctypes, time, os
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo/pbrt-rs_amd'); sys.path.insert(0,'/root/repo/oracle')
import numpy as np
import test_wide_host as W
import fuzz_intersect as F
import pbrt_hip
L = ctypes.CDLL(W.OUT)
vp, i32 = ctypes.c_void_p, ctypes.c_int32
L.wide_check_structure.argtypes = [vp, i32, vp, i32, vp, ctypes.c_char_p, ctypes.c_int]
L.wide_check_filter.argtypes = [vp, i32, vp, i32, vp, i32, vp]; L.wide_check_filter.restype = ctypes.c_int64
n=int(sys.argv[1]); bad=0; declined=0; t=time.time()
for seed in range(n):
    rng=np.random.default_rng(seed+10_000_000)
    v,i,scale=F.make_mesh(rng)
    mp,split=int(rng.choice([1,2,4])),int(rng.choice([0,1,2,3]))
    nodes,order=pbrt_hip.bvh_build(v,i,mp,split)
    tris=W._tree(dict(positions=v,indices=i),mp,split)[1]
    stats=np.zeros(4,dtype=np.int64); why=ctypes.create_string_buffer(200)
    rc=L.wide_check_structure(nodes.ctypes.data,len(nodes),tris.ctypes.data,len(tris),stats.ctypes.data,why,200)
    if rc==-100: declined+=1; continue
    if rc!=0: bad+=1; print("STRUCTURE", seed, rc, why.value); continue
    rays=W._adversarial_rays(nodes, 600, seed%1000+1)
    counts=np.zeros(4,dtype=np.int64)
    b=L.wide_check_filter(nodes.ctypes.data,len(nodes),tris.ctypes.data,len(tris),rays.ctypes.data,len(rays),counts.ctypes.data)
    if b!=0: bad+=1; print("FILTER", seed, b, counts, len(i), scale, mp, split)
print(n,"meshes", declined,"declined", bad,"bad", round(time.time()-t,1),"s")
