import importlib, sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
N,B=40,1024
P,X0,q,qd=problem.make_batch(B,N,0.6,seed=20211)
for w in (2,3,4):
    L=capi.LandingLib(N,0,lib_path='tmp_libs/lib_w%d.so'%w)
    for mi in (300,):
        o=L.default_opts(); o.max_iter=mi; o.reset_du=1e9; o.max_resets=8
        r=L.solve_host(P,X0,o)
        t=time.time(); r=L.solve_host(P,X0,o); dt=time.time()-t
        c=r['status']==0
        print('w',w,'max_iter',mi,'converged',c.sum(),'iters mean',r['iters'].mean(),'sec %.3f'%dt, 'nlp/s %.0f'%(c.sum()/dt))
    L.close()
