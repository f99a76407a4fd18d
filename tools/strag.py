import importlib, sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
N,B=40,1024
P,X0,q,qd=problem.make_batch(B,N,0.6,seed=20211)
L=capi.LandingLib(N,0)
prof=torch.zeros(B,16,device='cuda',dtype=torch.float64)
def run(label, r1=0, **kw):
    o=L.default_opts(); o.max_iter=300; o.sticky_delta=r1
    for k,v in kw.items():
        if k=='reserved2': o.restart_period=v
        else: setattr(o,k,v)
    L.lib.landing_set_profile_buffer(L.ctx, prof.data_ptr()); prof.zero_()
    r=L.solve_host(P,X0,o)
    ph=prof.cpu().numpy(); L.lib.landing_set_profile_buffer(L.ctx, None)
    t=time.time(); r=L.solve_host(P,X0,o); dt=time.time()-t
    c=r['status']==0
    print('%-40s conv %4d  iters mean %.1f med %.0f p90 %.0f  fact/iter %.2f  sec %.3f  nlp/s %.0f'%(label,c.sum(),r['iters'].mean(),np.median(r['iters']),np.percentile(r['iters'],90),ph[:,8].sum()/ph[:,10].sum(),dt,c.sum()/dt))
run('default', 0, max_iter=300)
run('sticky', 1, max_iter=300)
run('sticky inc 3', 1, max_iter=300, delta_inc=3.0)
run('sticky inc 2', 1, max_iter=300, delta_inc=2.0)
run('sticky dec .5', 1, max_iter=300, delta_dec=0.5)
run('sticky inc 3 max_iter 250', 1, max_iter=250, delta_inc=3.0)
