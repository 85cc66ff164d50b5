"""Network wrappers (reference: vwm/modules/diffusionmodules/wrappers.py:24-40)."""
import torch
import torch.nn as nn

from ...util import repeat_as_img_seq

OPENAIUNETWRAPPER = "vista_amd.modules.diffusionmodules.wrappers.OpenAIWrapper"


class IdentityWrapper(nn.Module):
    def __init__(self, diffusion_model, compile_model: bool = False):
        super().__init__()
        # the reference optionally torch.compile()s here; this build has no tracing compiler: kernels are explicit HIP
        self.diffusion_model = diffusion_model

    def forward(self, *args, **kwargs):
        return self.diffusion_model(*args, **kwargs)


class OpenAIWrapper(IdentityWrapper):
    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, cond_mask: torch.Tensor, num_frames: int, **kwargs) -> torch.Tensor:
        if "concat" in c and num_frames > 1 and c["concat"].shape[0] != x.shape[0]:
            assert c["concat"].shape[0] == x.shape[0] // num_frames, f"{c['concat'].shape} {x.shape}"
            c["concat"] = repeat_as_img_seq(c["concat"], num_frames)  # mutates c like the reference (wrappers.py:30)
        if "concat" in c:
            x = torch.cat((x, c["concat"].to(x.dtype)), dim=1)
        return self.diffusion_model(x, timesteps=t, context=c.get("crossattn", None), y=c.get("vector", None), cond_mask=cond_mask,
                                    num_frames=num_frames, **kwargs)
