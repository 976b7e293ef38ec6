import torch, sys, statistics
sys.path.insert(0, ".")
from fastdm_amd.kernel import operators_set as K
BF16=torch.bfloat16
for (B,S,H,hd) in [(2,4429,38,64),(2,4096,10,64),(2,1024,20,64),(1,8704,24,128)]:
    g=torch.Generator().manual_seed(1)
    q,k,v=[(torch.randn(B,S,H*hd,generator=g)).to(BF16).cuda() for _ in range(3)]
    for _ in range(3): K.scaled_dot_product_attention(q,k,v,H,H,hd)
    ev=[]
    for _ in range(20):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); K.scaled_dot_product_attention(q,k,v,H,H,hd); e1.record(); ev.append((e0,e1))
    torch.cuda.synchronize()
    t=statistics.median(a.elapsed_time(b) for a,b in ev)*1e3
    fl=4.0*B*H*S*S*hd
    print(f"B={B} S={S} H={H} hd={hd}: {t:8.1f} us  {fl/t/1e6:7.1f} TF/s  frac {fl/t/1e6/2500:.3f}")
