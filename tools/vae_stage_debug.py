import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vae_oracle as V
from oracle.make_golden_vae import TINY, latents
from vista_amd import synth, ops
from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder
import torch.nn.functional as F
torch.set_grad_enabled(False)
dec = VideoDecoder(video_kernel_size=[3,1,1], **TINY)
shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
sd = synth.seeded_state_dict(shapes, 0); dec.load_state_dict(sd); dec.cuda()
T,H,W = 5,8,16
z = latents(T,H,W,5)
def tok(x, H, W):  # tokens (n,S,C) -> nchw cpu
    n,S,C = x.shape
    return x.float().cpu().view(n,H,W,C).permute(0,3,1,2)
def rel(a,b): return ((a-b).pow(2).sum()/b.pow(2).sum()).sqrt().item()
# oracle stages
r = V._conv(sd,"conv_in",z,1)
h = ops.nchw_to_tokens(z.cuda().float(), 64); h,_,_ = ops.conv3x3(h, dec.packed()["conv_in"], T, H, W)
print("conv_in", rel(tok(h,H,W), r))
def stage(name, fo, fg):
    global r,h
    r = fo(r); h = fg(h)
    print(f"{name:20s} rel {rel(tok(h,H,W), r):.3e}   rms {r.pow(2).mean().sqrt():.3f}")
stage("mid.block_1", lambda x: V.video_resblock(sd,"mid.block_1",x,T), lambda x: dec.mid.block_1(x,None,H,W,timesteps=T))
stage("mid.attn_1", lambda x: V.attn_block(sd,"mid.attn_1",x), lambda x: dec.mid.attn_1(x,H,W))
stage("mid.block_2", lambda x: V.video_resblock(sd,"mid.block_2",x,T), lambda x: dec.mid.block_2(x,None,H,W,timesteps=T))
for lvl in reversed(range(4)):
    for b in range(3):
        stage(f"up.{lvl}.block.{b}", lambda x: V.video_resblock(sd,f"up.{lvl}.block.{b}",x,T), lambda x: dec.up[lvl].block[b](x,None,H,W,timesteps=T))
        # restart-from-reference error of this block alone
        xr = r  # after
    if lvl:
        r = V._conv(sd, f"up.{lvl}.upsample.conv", F.interpolate(r, scale_factor=2.0, mode="nearest"), 1)
        h,H,W = dec.up[lvl].upsample(h,H,W)
        print(f"up.{lvl}.upsample       rel {rel(tok(h,H,W), r):.3e}")
r2 = V._swish(V._gn(sd,"norm_out",r,1e-6)); h2 = ops.groupnorm(h, dec.norm_out.weight, dec.norm_out.bias, 1e-6, silu=True)
print("norm_out", rel(tok(h2,H,W), r2))
ro = V.ae3d_conv(sd,"conv_out",r2,T); ho = dec.conv_out(h2,H,W,timesteps=T)
print("conv_out", rel(ho.cpu(), ro), "rms", ro.pow(2).mean().sqrt().item())
# single-layer error: feed the ORACLE activations (rounded to bf16) into the last stage
hin = r2.permute(0,2,3,1).reshape(T,H*W,-1).to(torch.bfloat16).cuda().contiguous()
print("conv_out alone", rel(dec.conv_out(hin,H,W,timesteps=T).cpu(), ro))
