"""Where does the block-sparse kernel's time per list position go?  Wan2.2-A14B geometry (S = 80640, 40 x 128).  Four runs of one attention call:
dense kernel; masked kernel with an all-ones mask (loop overhead of the masked form); masked kernel with the radial mask; masked kernel with a mask
of the SAME per-row tile sets for both rows of every 256-row block (the radial rows 2b duplicated: no exclusive stretches, same scatter).
usage: PYTHONPATH=. python tools/micro/sparse_overhead_bench.py"""
import torch
from fastdm_amd.sparse import SparseAttn
from fastdm_amd.kernel.operators_set import sparse_scaled_dot_product_attention, scaled_dot_product_attention
H, frames, tpf, hd = 40, 21, 48 * 80, 128
S = frames * tpf
sp = SparseAttn.from_json("examples/sparse/radial_attn_wan.json")
sp.post_init(video_token_num=S, num_frame=frames)
km = sp.kernel_mask(S, "cuda")
g = torch.Generator(device="cuda").manual_seed(1)
q, k, v = (torch.randn(1, S, H * hd, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))


def timed(f, n=3):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[1]


def run(name, m):
    pair = m.view(m.shape[0] // 2, 2, -1)
    union = int(pair.max(dim=1).values.sum().item())      # list positions over all 256-row blocks
    both = int(pair.min(dim=1).values.sum().item())
    mask = m[None, None].expand(1, H, -1, -1).contiguous()
    ms = timed(lambda: sparse_scaled_dot_product_attention(q, k, v, num_q_heads=H, num_kv_heads=H, head_dim=hd, is_causal=False, scale=hd ** -0.5, sparse_mask=mask))
    per_pos = ms * 1e3 / (union * H / 256.0)
    print(f"{name:44s} {ms:8.2f} ms   list positions / block-row pair {union / (m.shape[0] // 2):7.1f}  shared {both / max(union, 1):.3f}   {per_pos:.3f} us per position and CU")


ms = timed(lambda: scaled_dot_product_attention(q, k, v, H, H, hd, scale=hd ** -0.5), n=2)
nt = (S // 256) * (S // 64)
print(f"{'dense kernel':44s} {ms:8.2f} ms   {ms * 1e3 / (nt * H / 256.0):.3f} us per tile step and CU")
run("masked kernel, all-ones mask", torch.ones_like(km))
run("masked kernel, radial mask", km)
dup = km.clone(); dup[1::2] = km[0::2]
run("masked kernel, rows 2b duplicated (no exclusives)", dup)
band = torch.zeros_like(km)
w = int(km.float().mean().item() * km.shape[1])
for r in range(km.shape[0]):
    c = min(max(2 * r - w // 2, 0), km.shape[1] - w)
    band[r, c:c + w] = 1
run("masked kernel, one contiguous band per row", band)
