import torch, time
from fastdm_amd import _native as N
BF16=torch.bfloat16
for M,K,Nn in [(2,3072,18432*120),(1,3072,18432*19),(1,3072,9216*38),(1,256,3072),(2,3072,3072)]:
    a=torch.randn(M,K,device='cuda').to(BF16); w=(torch.randn(Nn,K,device='cuda',dtype=BF16)); b=torch.zeros(Nn,device='cuda',dtype=BF16)
    out=torch.empty(M,Nn,device='cuda',dtype=BF16)
    f=lambda: N.check(N.lib().fdm_gemm_bf16(out.data_ptr(),a.data_ptr(),w.data_ptr(),b.data_ptr(),M,Nn,K,K,K,Nn,0,None,0,None,0,N.stream()),"g")
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/10
    print(M,K,Nn, f"{dt*1e3:.3f} ms  {Nn*K*2/dt/1e12:.2f} TB/s")
