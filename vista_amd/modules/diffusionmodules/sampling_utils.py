"""to_d (reference: vwm/modules/diffusionmodules/sampling_utils.py:46-47); the Euler update itself is vk_euler_step."""
from ...util import append_dims


def to_d(x, sigma, denoised):
    return (x - denoised) / append_dims(sigma, x.ndim)
