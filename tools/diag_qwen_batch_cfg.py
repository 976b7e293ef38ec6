import torch, sys
sys.path.insert(0, "/root/repo")
from oracle import qwen as OQ
from fastdm_amd.model_entry import create_model
from fastdm_amd.pipeline import SyntheticQwenPipeline
cfg = OQ.TINY
sd = OQ.synth_state_dict(cfg, seed=11)
kw = dict(patch_size=cfg.patch_size, num_layers=cfg.num_layers, attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
          joint_attention_dim=cfg.joint_attention_dim, in_channels=cfg.in_channels, out_channels=cfg.out_channels)
outs = []
for bc in (False, True):
    m = create_model("qwen", ckpt_path=sd, dtype=torch.bfloat16, quant_type=torch.int8, kernel_backend="hip", **{k: v for k, v in kw.items() if k != "patch_size"}).eval()
    p = SyntheticQwenPipeline(m, seed=3, batch_cfg=bc)
    outs.append(p(num_inference_steps=4, true_cfg_scale=4.0, width=96, height=64, max_sequence_length=7, seed=3))
print("equal:", torch.equal(outs[0], outs[1]), (outs[0].float() - outs[1].float()).abs().max().item())
