import importlib, sys, time, torch
sys.path.insert(0,".")
import bench
g=importlib.import_module("sdsl-lite_amd")
text=bench.synthetic_text(1<<30,1234,torch.device("cuda",0))
csa=g.csa_wt(text=text)
t0=time.time(); b=csa.serialize(32,64,g.capi.LAYOUT_BV_MCL); print("serialize MCL", round(time.time()-t0,2), "s", len(b))
t0=time.time(); b2=csa.serialize(32,64,g.capi.LAYOUT_BV_MCL); print("again", round(time.time()-t0,2), "s", b==b2)
t0=time.time(); b3=csa.serialize(32,64,0); print("scan flavour", round(time.time()-t0,2), "s", len(b3))
