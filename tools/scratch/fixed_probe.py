import sys
sys.path.insert(0, '/root/repo')
import numpy as np
from kfunca_amd import hip_abi as H
H.set_device(0)
n=4096
rng=np.random.default_rng(0)
x=rng.uniform(-1,1,size=(n,n)).astype(np.float32).view(np.uint32)
bits=((x+0x7FFF+((x>>16)&1))>>16).astype(np.uint16)
A=H.DevBuf.from_numpy(bits); B=H.DevBuf.from_numpy(bits[::-1].copy()); C=H.DevBuf(2*n*n)
def b2b(fn, reps=50):
    fn(); H.device_sync()
    e0,e1=H.Event(),H.Event(); e0.record(None)
    for _ in range(reps): fn()
    e1.record(None); H.device_sync()
    return e0.elapsed_ms(e1)/reps*1e3
for tb,tag in ((1,'NT'),(0,'NN')):
    for K in (4096,):
        f=lambda: H.gemm(H.BF16,0,tb,n,n,K,1.0,A.ptr,4096,B.ptr,4096,0.0,C.ptr,n,0,None,None,0)
        print('gemm',tag,'K',K,'b2b us %.2f'%b2b(f), flush=True)
