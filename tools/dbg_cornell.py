import sys, numpy as np
sys.path.insert(0,'pbrt-rs_amd'); sys.path.insert(0,'oracle')
import oracle, pbrt_hip
from pbrt_hip import scenes
w=h=512
sc, cam = scenes.cornell_box(), scenes.cornell_camera(512,512)
osc = oracle.OracleScene(sc); ctx = pbrt_hip.Context(0); gsc = pbrt_hip.Scene(ctx, sc)
kw = dict(max_depth=8, rr_threshold=1.0, light_strategy=1, seed=0)
fc,_ = osc.render(scenes.camera_dict_to_floats(cam), w,h,64, n_threads=16, **kw)
fg,_ = gsc.render(cam, w,h,64, **kw)
d = np.abs(fg[...,:3]-fc[...,:3]).max(-1)
rel = d/np.maximum(1.0, np.abs(fc[...,:3]).max(-1))
idx = np.argsort(rel.ravel())[::-1][:8]
print("n pixels > 1e-5 rel:", (rel>1e-5).sum(), " >1e-6:", (rel>1e-6).sum(), " exact-equal pixels:", (d==0).sum(), "of", w*h)
for i in idx:
    y,x = divmod(i, w); print(x,y, fg[y,x], fc[y,x])
y,x = divmod(idx[0], w)
for md in range(0,9):
    k2 = dict(kw); k2["max_depth"]=md
    a,_ = osc.render(scenes.camera_dict_to_floats(cam), w,h,64, bounds=(x,y,x+1,y+1), **k2)
    b,_ = gsc.render(cam, w,h,64, bounds=(x,y,x+1,y+1), **k2)
    print("depth", md, a[y,x], b[y,x])
