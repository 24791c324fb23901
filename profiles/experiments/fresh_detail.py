import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bsms_gnn_amd as eng
from bsms_gnn_amd import graph
from bench import WORKLOADS, build_mesh, make_cfg
B=8; kind="cylinder"; w=WORKLOADS[kind]; n,c=w["nodes"],w["out_dim"]
gen=torch.Generator().manual_seed(0); pool=[]
for seed in range(2*B):
    pts,m_es,m_ids=build_mesh(kind,seed=seed)
    state,target=torch.randn(n,c,generator=gen),torch.randn(n,c,generator=gen)
    x=torch.cat([state,torch.tensor(pts,dtype=torch.float32),torch.zeros(n,1)],-1)
    sizes=[n]+[len(i) for i in m_ids]
    pool.append([eng.LevelData(torch.tensor(m_es[l]),sizes[l],face=torch.tensor(m_ids[l]) if l<w["levels"] else None,x=x if l==0 else None,y=target if l==0 else None,mask=torch.ones(n,1) if l==0 else None) for l in range(w["levels"]+1)])
perm=torch.Generator().manual_seed(1)
def batch():
    idx=torch.randperm(len(pool),generator=perm)[:B].tolist()
    return eng.collate_variable_meshes([pool[i] for i in idx])
import gc
_g=[0.0,0]
def _cb(phase, info):
    if phase=="start": _g[1]=time.perf_counter()
    else:
        T["gc%d"%info["generation"]]=T.get("gc%d"%info["generation"],0)+time.perf_counter()-_g[1]
        T["gc_collected"]=T.get("gc_collected",0)+info["collected"]*1e-3
gc.callbacks.append(_cb)
T={}
def timed(name, fn):
    def wrap(*a, **k):
        t0=time.perf_counter(); r=fn(*a,**k); T[name]=T.get(name,0)+time.perf_counter()-t0; return r
    return wrap
graph._content_key=timed("content_key",graph._content_key)
graph._upload=timed("upload",graph._upload)
graph._key=timed("_key",graph._key)
graph._host_copy=timed("host_copy",graph._host_copy)
graph._reap=timed("reap",graph._reap)
L=eng._abi.lib()
orig_init=graph.LevelPlan.__init__
graph.LevelPlan.__init__=timed("plan_init",orig_init)
graph.LevelPlan.set_pool=timed("set_pool",graph.LevelPlan.set_pool)
torch.manual_seed(0)
sim=eng.BSMS_Simulator(make_cfg(w)).cuda(); dp=eng.DataParallel(sim)
fs=None
d=[x.to("cuda",intern=True) for x in batch()]
sim(d,False,True)
for it in range(60):
    b=batch()
    T.clear()
    import cProfile, pstats, io
    pr=cProfile.Profile(); pr.enable()
    t0=time.perf_counter(); d=[x.to("cuda",intern=True) for x in b]; t1=time.perf_counter()
    torch.cuda.synchronize(); t2=time.perf_counter()
    dp.step_loss_backward(d,False); t3=time.perf_counter(); torch.cuda.synchronize(); t4=time.perf_counter()
    pr.disable()
    if it>=40 and (t4-t0)>0.05:
        so=io.StringIO(); pstats.Stats(pr,stream=so).sort_stats("tottime").print_stats(6); print("SPIKE", so.getvalue()[-900:])
    if it>=50:
        print(f"to_dev {1e3*(t1-t0):6.2f} sync {1e3*(t2-t1):6.2f} step-host {1e3*(t3-t2):6.2f} step-sync {1e3*(t4-t3):6.2f} | "+" ".join(f"{k} {1e3*v:.2f}" for k,v in T.items()))
