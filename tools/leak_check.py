import sys; sys.path.insert(0,'.')
import numpy as np, torch
import decomp_amd
rng=np.random.RandomState(0)
free0=None
for rep in range(6):
    for (N,F,K) in [(300,64,5),(2048,512,32),(700,129,9)]:
        y=np.abs(rng.randn(N,F)).astype(np.float32); D=np.abs(rng.randn(K,F)).astype(np.float32)+0.1
        for _ in range(10):
            decomp_amd.nmf.solve(y,D.copy(),tol=0.0,maxiter=30)
            decomp_amd.nmf.solve(y,D.copy(),tol=0.0,maxiter=12,minibatch=50,method='svrmu')
            decomp_amd.lasso.solve(y,D,0.1,method='parallel_cd',maxiter=25)
            decomp_amd.lasso.solve(y,D,0.1,method='admm',maxiter=25)
            decomp_amd.dictionary_learning.solve(y,rng.randn(K,F).astype(np.float32),0.1,minibatch=100,maxiter=3,lasso_method='cd')
    torch.cuda.synchronize()
    free,total=torch.cuda.mem_get_info()
    if free0 is None: free0=free
    print(rep, 'free MiB', free>>20, 'delta vs first', (free0-free)>>20)
