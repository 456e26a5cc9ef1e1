import time, sys
t0=time.perf_counter()
sys.path.insert(0,'.')
import numpy as np
import torch
t1=time.perf_counter()
import decomp_amd
from decomp_amd import _hip
_hip.load()
t2=time.perf_counter()
torch.cuda.init(); torch.zeros(1,device='cuda'); torch.cuda.synchronize()
t3=time.perf_counter()
rng=np.random.RandomState(0)
y=np.abs(rng.randn(256,128)); D=np.abs(rng.randn(8,128))+0.1
decomp_amd.nmf.solve(y,D.copy(),maxiter=5)
t4=time.perf_counter()
decomp_amd.lasso.solve(y,D,0.1,maxiter=5)
t5=time.perf_counter()
decomp_amd.dictionary_learning.solve(y,rng.randn(8,128),0.1,minibatch=64,maxiter=2)
t6=time.perf_counter()
decomp_amd.nmf.solve(y.astype(np.float32),D.astype(np.float32),maxiter=5)
t7=time.perf_counter()
decomp_amd.nmf.solve(y,D.copy(),maxiter=5)
t8=time.perf_counter()
print('import numpy+torch %.2fs | load lib %.2fs | cuda init %.2fs | first nmf f64 %.2fs | first lasso f64 %.2fs | first dl f64 %.2fs | first nmf f32 %.2fs | warm nmf %.4fs' % (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5,t7-t6,t8-t7))
