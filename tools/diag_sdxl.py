import torch, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import sdxl as OX, flux as OF
from fastdm_amd.model_entry import create_model
BF16=torch.bfloat16; FP8=torch.float8_e4m3fn
# mimic the preceding flux runs
cfg=OF.TINY; sd=OF.synth_state_dict(cfg,seed=7)
arch=dict(in_channels=cfg.in_channels,out_channels=cfg.out_channels,num_layers=cfg.num_layers,num_single_layers=cfg.num_single_layers,attention_head_dim=cfg.attention_head_dim,num_attention_heads=cfg.num_attention_heads,joint_attention_dim=cfg.joint_attention_dim,pooled_projection_dim=cfg.pooled_projection_dim,axes_dims_rope=cfg.axes_dims_rope)
inp={k:v.cuda() for k,v in OF.synth_inputs(cfg,1,4,6,8,seed=7).items()}
for qt in (FP8, torch.int8):
    m=create_model("flux",ckpt_path=sd,dtype=BF16,quant_type=qt,kernel_backend="hip",**arch).eval()
    m(timestep=torch.tensor([0.5]).cuda(),guidance=torch.tensor([3.5]).cuda(),return_dict=False,**inp)
xc=OX.TINY; xsd=OX.synth_state_dict(xc,seed=41); i=OX.synth_inputs(xc,2,16,24,11,seed=41)
for cl in (False, True):
    outs=[]
    for rep in range(2):
        m=create_model("sdxl",ckpt_path=xsd,dtype=BF16,quant_type=FP8,kernel_backend="hip",block_out_channels=xc.block_out_channels,transformer_layers=xc.transformer_layers,mid_layers=xc.mid_layers,channels_last=cl).eval()
        for r in range(2):
            outs.append(m(i["sample"].cuda(),torch.tensor(500.0).cuda(),i["encoder_hidden_states"].cuda(),{k:v.cuda() for k,v in i["added_cond_kwargs"].items()})[0].clone())
    for a in range(1,4):
        d=(outs[a].float()-outs[0].float()).abs()
        print("channels_last",cl,"run",a,"vs 0: max",d.max().item(),"batch0",d[0].max().item(),"batch1",d[1].max().item())
