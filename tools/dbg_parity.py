import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch, ctypes as C
import helpers
from arap_flow_amd import opt
from oracle import oracle as orc
orc.set_trig(1)
st=opt.State()
W,H=130,37
pb=helpers.random_problem(W,H,seed=W+H,generic_urshape=True,ncons=max(4,W*H//60))
lib=st.lib
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def ptr(t): return C.c_void_p(t.data_ptr())
d={k:dev(pb[k]) for k in "OAUCM"}
gO=torch.zeros(H,W,2,device='cuda');gA=torch.zeros(H,W,device='cuda');dO=torch.zeros(H,W,2,device='cuda');dA=torch.zeros(H,W,device='cuda')
lib.ArapFlow_EvalJTF(st.handle,W,H,ptr(d['O']),ptr(d['A']),ptr(d['U']),ptr(d['C']),ptr(d['M']),10.0,0.1,ptr(gO),ptr(gA),ptr(dO),ptr(dA))
g32,d32=orc.evalJTF(pb['O'],pb['A'],pb['U'],pb['C'],pb['M'],10.0,0.1,dtype=np.float32)
g=np.concatenate([gO.cpu().numpy(),gA.cpu().numpy()[...,None]],-1); dd=np.concatenate([dO.cpu().numpy(),dA.cpu().numpy()[...,None]],-1)
print('evalJTF bit mismatches g',(g!=g32).sum(),'of',g.size,'max',np.abs(g-g32).max(),' d',(dd!=d32).sum(),np.abs(dd-d32).max())
rng=np.random.default_rng(7); P=rng.normal(size=(H,W,3)).astype(np.float32); P[pb['M']!=0]=0
pO,pA=dev(P[...,:2]),dev(P[...,2]); oO=torch.zeros(H,W,2,device='cuda');oA=torch.zeros(H,W,device='cuda')
lib.ArapFlow_ApplyJTJ(st.handle,W,H,ptr(d['A']),ptr(d['U']),ptr(d['C']),ptr(d['M']),10.0,0.1,ptr(pO),ptr(pA),ptr(oO),ptr(oA))
j32=orc.applyJTJ(pb['A'],pb['U'],pb['C'],pb['M'],10.0,0.1,P,dtype=np.float32)
j=np.concatenate([oO.cpu().numpy(),oA.cpu().numpy()[...,None]],-1)
print('applyJTJ bit mismatches',(j!=j32).sum(),'of',j.size,np.abs(j-j32).max())
cost=C.c_double(); lib.ArapFlow_Cost(st.handle,W,H,ptr(d['O']),ptr(d['A']),ptr(d['U']),ptr(d['C']),ptr(d['M']),10.0,0.1,C.byref(cost))
print('cost',cost.value, orc.cost(pb['O'],pb['A'],pb['U'],pb['C'],pb['M'],10.0,0.1,dtype=np.float32,mode=1))
def gpu_solve(n,l):
    dv={k:torch.from_numpy(pb[k].copy()).cuda() for k in "OAUCM"}
    s=opt.OptSolver(st,(W,H)); pp=opt.NamedParameters()
    for nm,k in [("Offset","O"),("Angle","A"),("UrShape","U"),("Constraints","C"),("Mask","M")]: pp.set(nm,dv[k])
    pp.set("w_fitSqrt",10.0); pp.set("w_regSqrt",0.1); sp=opt.NamedParameters(); sp.set("nIterations",n); sp.set("lIterations",l)
    c=s.solve(sp,pp); out=dv['O'].cpu().numpy(),dv['A'].cpu().numpy(),c; s.close(); return out
for n,l in [(1,1),(1,2),(1,5),(1,20),(1,50),(2,50),(4,50)]:
    O,A,c=gpu_solve(n,l)
    Or,Ar,cs=orc.solve(pb['O'],pb['A'],pb['U'],pb['C'],pb['M'],10.0,0.1,n,l,dtype=np.float32,mode=1,trig=1)
    print(n,l,'O mism',(O!=Or).sum(),'max',np.abs(O-Or).max(),'A mism',(A!=Ar).sum(),np.abs(A-Ar).max(),'rel',helpers.rel_l2(O-pb['O'],Or-pb['O']),'cost',c,cs[-1])
