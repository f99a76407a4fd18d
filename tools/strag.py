import importlib, sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo')
capi=importlib.import_module("landing-controller_amd.capi"); problem=importlib.import_module("landing-controller_amd.problem")
N,B=40,1024
P,X0,q,qd=problem.make_batch(B,N,0.6,seed=20211)
L=capi.LandingLib(N,0)
def run(label, X, **kw):
    o=L.default_opts(); o.max_iter=300
    for k,v in kw.items(): setattr(o,k,v)
    r=L.solve_host(P,X,o)
    c=r['status']==0
    print('%-44s conv %4d  iters mean %.1f med %.0f p90 %.0f p99conv %.0f'%(label,c.sum(),r['iters'].mean(),np.median(r['iters']),np.percentile(r['iters'],90),np.percentile(r['iters'][c],99)))
    return r
rng=np.random.default_rng(0)
nX=12*(N+1)
r0=run('baseline x0', X0)
for sig in (1e-4,1e-3,1e-2):
    Xp=X0.copy(); Xp[:,nX:]+=sig*rng.normal(size=Xp[:,nX:].shape)
    run('U jitter sigma %g'%sig, Xp)
# forces initial guess: static weight support instead of zero
Xp=X0.copy()
U=Xp[:,nX:].reshape(B,N,24)
U[:,:,12+2::3]+=8.252*9.81/4
run('fz guess = mg/4', Xp)
Xp=X0.copy(); U=Xp[:,nX:].reshape(B,N,24); U[:,N//2:,12+2::3]+=8.252*9.81/4*2
run('fz guess = mg/2 second half', Xp)
# feet guess on the ground (c_z = 0) for the second half of the horizon
Xp=X0.copy(); U=Xp[:,nX:].reshape(B,N,24); U[:,:,2:12:3]=np.maximum(U[:,:,2:12:3],0.0)
run('feet clipped to c_z>=0', Xp)
# X guess: keep initial state constant (no interpolation)
Xp=X0.copy(); X=Xp[:,:nX].reshape(B,N+1,12); X[:,1:,:]=X[:,:1,:]
run('X guess = X0 constant', Xp)
