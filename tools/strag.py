import importlib, sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
N,B=40,1024
P,X0,q,qd=problem.make_batch(B,N,0.6,seed=20211)
L=capi.LandingLib(N,0)
def run(label, **kw):
    o=L.default_opts()
    for k,v in kw.items(): setattr(o,k,v)
    r=L.solve_host(P,X0,o)
    t=time.time(); r=L.solve_host(P,X0,o); dt=time.time()-t
    c=r['status']==0
    print('%-40s conv %4d  iters mean %.1f med %.0f p90 %.0f  sec %.3f  nlp/s %.0f'%(label,c.sum(),r['iters'].mean(),np.median(r['iters']),np.percentile(r['iters'],90),dt,c.sum()/dt))
for mi in (150,200,250,300,400,3000):
    run('max_iter %d'%mi, max_iter=mi)
