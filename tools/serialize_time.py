import importlib, sys, time, torch
sys.path.insert(0,".")
import bench
g=importlib.import_module("sdsl-lite_amd")
text=bench.synthetic_text(1<<30,1234,torch.device("cuda",0))
csa=g.csa_wt(text=text)
t0=time.time(); b=csa.serialize(32,64,g.capi.LAYOUT_BV_MCL); print("serialize MCL", round(time.time()-t0,2), "s", len(b))
t0=time.time(); b2=csa.serialize(32,64,g.capi.LAYOUT_BV_MCL); print("again", round(time.time()-t0,2), "s", b==b2)
t0=time.time(); b3=csa.serialize(32,64,0); print("scan flavour", round(time.time()-t0,2), "s", len(b3))
csa.close(); del csa
t0=time.time(); c2=g.csa_wt(sdsl_bytes=b, select_is_mcl=True, sa_dens=32, isa_dens=64); torch.cuda.synchronize(); print("load the MCL stream back", round(time.time()-t0,2), "s")
import numpy as np
st=torch.randint(0,(1<<30)-20,(100000,),device="cuda")
p=text[(st.view(-1,1)+torch.arange(20,device="cuda").view(1,20)).reshape(-1)].contiguous()
print("count on the loaded index, all found:", bool((c2.count(p,20)>=1).all()))
