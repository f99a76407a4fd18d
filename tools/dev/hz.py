import importlib, sys, numpy as np
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
for N in (16,20,30,40):
    L=capi.LandingLib(N,0)
    for B,seed in ((6,4),(256,11)):
        P,X0,_,_=problem.make_batch(B,N,0.6,seed=seed)
        for fr in (0.5,0.2,0.1,0.05):
            o=L.default_opts(); o.bound_frac=fr; o.max_iter=600
            r=L.solve_host(P,X0,o)
            print(N,B,seed,fr,'conv',(r['status']==0).sum(),'iters med',np.median(r['iters']), np.bincount(r['status'],minlength=3).tolist(), flush=True)
    L.close()
