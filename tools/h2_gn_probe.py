"""numerics probe (round 5): the FCOS tower test case (two towers, stacked first layer) with the channel products on csrc/h2.hip vs the fp32-format
path, per tensor, against fp64"""
import sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import torch.nn.functional as F
from lgd_amd import ops, synth
import common as cm
from oracle import student_oracle as SO
DEV='cuda'
B,C,G,level_hw=3,256,32,[(12,16),(6,7),(2,3)]
def P(shape, seed, lo, hi, scale=1.0):
    return (torch.from_numpy(synth.det_uniform(shape, seed, lo, hi)) * scale).to(DEV).requires_grad_(True)
def run(force, mode):
    xs=[P((B,C,h,w),1700+i,-2.0,3.0) for i,(h,w) in enumerate(level_hw)]
    std=(2.0/(9*C))**0.5
    lay={}
    for n,seed in (("a1",1710),("b1",1720),("a2",1730),("b2",1740)):
        lay[n]=(P((C,C,3,3),seed,-1.0,1.0,std),P((C,),seed+1,-0.1,0.1),P((C,),seed+2,0.5,1.5),P((C,),seed+3,-0.5,0.5))
    fin={n:(P((24,C,3,3),seed,-1.0,1.0,std),P((24,),seed+1,-0.1,0.1)) for n,seed in (("a",1750),("b",1760))}
    gys={n:[torch.from_numpy(synth.det_uniform((B,24,h,w),seed+i,-1.0,1.0)).to(DEV) for i,(h,w) in enumerate(level_hw)] for n,seed in (("a",1770),("b",1780))}
    prev=ops.conv3x3_backend(winograd=True,min_tiles=0,tile=6); ph=ops.h2_backend(True,force=force); was=ops._GN_FUSED_BWD; ops._GN_FUSED_BWD = mode=="one_node"
    if mode == "shared_only":
        (pa,ma),(pb,mb)=ops.conv3x3_gn(xs,[lay["a1"],lay["b1"]],G)
        ya=[F.relu(m*pa[l*B:(l+1)*B,:,0,None,None]+pa[l*B:(l+1)*B,:,1,None,None]) for l,m in enumerate(ma)]
        yb=[F.relu(m*pb[l*B:(l+1)*B,:,0,None,None]+pb[l*B:(l+1)*B,:,1,None,None]) for l,m in enumerate(mb)]
    else:
        (pa,ma),(pb,mb)=ops.conv3x3_gn(xs,[lay["a1"],lay["b1"]],G)
        (pa,ma),=ops.conv3x3_gn(ma,[lay["a2"]],G,pre=pa)
        (pb,mb),=ops.conv3x3_gn(mb,[lay["b2"]],G,pre=pb)
        ya=ops.conv3x3_levels(ma,*fin["a"],pre=pa); yb=ops.conv3x3_levels(mb,*fin["b"],pre=pb)
    g_a = gys["a"] if mode != "shared_only" else [torch.from_numpy(synth.det_uniform(tuple(y.shape),1790+i,-1.0,1.0)).to(DEV) for i,y in enumerate(ya)]
    g_b = gys["b"] if mode != "shared_only" else [torch.from_numpy(synth.det_uniform(tuple(y.shape),1795+i,-1.0,1.0)).to(DEV) for i,y in enumerate(yb)]
    torch.autograd.backward(list(ya)+list(yb),g_a+g_b)
    ops._GN_FUSED_BWD=was; ops.h2_backend(*ph); ops.conv3x3_backend(*prev)
    d=lambda t:t.detach().double().cpu().requires_grad_(True)
    x64=[d(x) for x in xs]; l64={n:tuple(d(t) for t in v) for n,v in lay.items()}; f64={n:tuple(d(t) for t in v) for n,v in fin.items()}
    def tower(x,n):
        ks = ("1",) if mode == "shared_only" else ("1","2")
        for k in ks:
            w,b,ga,be=l64[n+k]; x=SO.group_norm_relu(F.conv2d(x,w,b,1,1),G,ga,be,True)
        return x if mode == "shared_only" else F.conv2d(x,f64[n][0],f64[n][1],1,1)
    ra,rb=[tower(x,"a") for x in x64],[tower(x,"b") for x in x64]
    torch.autograd.backward(ra+rb,[g.double().cpu() for g in g_a+g_b])
    print("force=%s mode=%s: y rel %s" % (force,mode," ".join("%.1e"%cm.rel_err(y,r) for y,r in zip(list(ya)+list(yb),ra+rb))))
    for i,(x,xr) in enumerate(zip(xs,x64)):
        ok,msg=cm.kink_robust_close(x.grad,xr.grad,tol=4e-4,max_outlier_frac=2e-3,max_rel=1e-2); print("   dx level",i,msg)
    print("   dw:", " ".join("%s %.1e"%(n,cm.rel_err(lay[n][0].grad,l64[n][0].grad)) for n in lay if lay[n][0].grad is not None))
for mode in ("shared_only","one_node"):
    for force in (False,True): run(force,mode)
