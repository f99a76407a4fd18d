import importlib, sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
N,B=40,1024
P,X0,q,qd=problem.make_batch(B,N,0.6,seed=20211)
L=capi.LandingLib(N,0)
def run(label, local=0, **kw):
    o=L.default_opts(); o.reserved[0]=local
    for k,v in kw.items(): setattr(o,k,v)
    r=L.solve_host(P,X0,o)
    t=time.time(); r=L.solve_host(P,X0,o); dt=time.time()-t
    c=r['status']==0
    print('%-44s conv %4d  iters mean %.1f med %.0f p90 %.0f  sec %.3f  nlp/s %.0f'%(label,c.sum(),r['iters'].mean(),np.median(r['iters']),np.percentile(r['iters'],90),dt,c.sum()/dt))
run('global delta, max_iter 300', 0, max_iter=300)
run('stage-local, max_iter 300', 1, max_iter=300)
run('stage-local, inc 8', 1, max_iter=300, delta_inc=8.0)
run('stage-local, inc 2', 1, max_iter=300, delta_inc=2.0)
run('stage-local, inc 4 first 100', 1, max_iter=300, delta_inc_first=100.0)
run('stage-local, dec 0.1', 1, max_iter=300, delta_dec=0.1)
run('stage-local, max_iter 3000', 1, max_iter=3000)
