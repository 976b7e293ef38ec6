"""Time the block-sparse attention kernel on Wan2.2-A14B's radial mask (768x1280x81 frames: 21 latent frames x 3840 tokens, 40 heads x 128)
and print the mask's union / per-row densities.  FDM_ATT_STRETCH=0 runs every wave over the whole union list (the A/B baseline).
usage: PYTHONPATH=. python tools/micro/sparse_attn_bench.py [heads]"""
import sys, time, torch
from fastdm_amd.sparse import SparseAttn
from fastdm_amd.kernel.operators_set import sparse_scaled_dot_product_attention
H = int(sys.argv[1]) if len(sys.argv) > 1 else 40
frames, tpf, hd = 21, 48 * 80, 128
S = frames * tpf
sp = SparseAttn.from_json("examples/sparse/radial_attn_wan.json")
sp.post_init(video_token_num=S, num_frame=frames)
km = sp.kernel_mask(S, "cuda")                       # int8 [S/128, S/64]
nq = km.shape[0]
pair = km.view(nq // 2, 2, -1)
union = pair.max(dim=1).values.float().mean().item()
both = pair.min(dim=1).values.float().mean().item()
print(f"S {S} heads {H}: row density {km.float().mean().item():.4f}, union of a 256-row block {union:.4f}, kept by both rows {both:.4f}")
mask = km[None, None].expand(1, H, -1, -1).contiguous()
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(1, S, H * hd, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))
f = lambda: sparse_scaled_dot_product_attention(q, k, v, num_q_heads=H, num_kv_heads=H, head_dim=hd, is_causal=False, scale=hd ** -0.5, sparse_mask=mask)
o = f()
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 3)
flops = 4.0 * S * S * H * hd * km.float().mean().item()
print(f"sparse attention: median {sorted(ts)[2]:.3f} ms  min {min(ts):.3f}  ({flops / sorted(ts)[2] / 1e9:.1f} TFLOP/s on kept blocks)  checksum {o.float().abs().mean().item():.6f}")
