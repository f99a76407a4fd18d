"""dev probe (CPU, oracle): would IPOPT's nlp_scaling_method gradient-based (nlp_scaling_max_gradient 50, generate_landingCtrller_IPOPT.m:237-238) scale anything on this NLP?
|grad f|inf and the largest row gradient of g at the callers' initial guess, 64 members of three families -> every scale factor is 1."""
import sys, importlib, numpy as np
sys.path.insert(0,'/root/repo')
from oracle import oracle as om
problem = importlib.import_module("landing-controller_amd.problem")
for N,grid,law in ((40,"uniform","main"),(20,"reference","datagen"),(20,"reference","main")):
    O=om.Oracle(N)
    consts = problem.production_constants(law) if grid=="reference" else None
    P,X0,_,_=problem.make_batch(64,N,0.6,seed=100000,consts=consts,dt_grid=grid,law=law)
    gf=[]; gr=[]; nsc=[]
    cj,rj=O.pattern_jac()
    for b in range(64):
        f,g=O.grad_f(X0[b],P[b]); gf.append(np.abs(g).max())
        _,jac=O.jac_g(X0[b],P[b])
        rown=np.zeros(O.ng); 
        cols=np.repeat(np.arange(O.nx), np.diff(cj))
        np.maximum.at(rown, rj, np.abs(jac))
        gr.append(rown.max()); nsc.append((rown>50).sum())
    print(N,law,"|grad f|inf at x0: median %.1f max %.1f -> d_f median %.3f"%(np.median(gf),np.max(gf),np.median(np.minimum(1,50/np.array(gf)))),"| max row gradient: median %.1f max %.1f, rows scaled (of %d): median %d max %d"%(np.median(gr),np.max(gr),O.ng,np.median(nsc),np.max(nsc)))
