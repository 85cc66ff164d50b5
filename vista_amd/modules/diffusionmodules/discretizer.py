"""Noise-level schedules of the sampler (what the reference keeps in vwm/modules/diffusionmodules/discretizer.py:16-37).

The schedule is evaluated ONCE per sampling call, on the host in fp32 with the same torch expression order as the reference
(scalar roots in Python floats, one `linspace`, one fused affine map, one tensor power), so every sigma is bit-identical to the
reference's and identical on every rank of a multi-GPU run; only the finished vector moves to the device."""
import torch

from ...util import append_zero


class Discretization:
    """Callable schedule: `disc(n, do_append_zero=True, device=..., flip=False)` -> (n [+1],) fp32 sigmas, largest first."""

    def get_sigmas(self, n, device):
        raise NotImplementedError("a Discretization provides get_sigmas(n, device)")

    def __call__(self, n, do_append_zero=True, device="cpu", flip=False):
        schedule = self.get_sigmas(n, device=device)
        if do_append_zero:
            schedule = append_zero(schedule)  # the trailing sigma = 0 the Euler loop steps onto
        if flip:
            schedule = schedule.flip(0)
        return schedule


class EDMDiscretization(Discretization):
    """Karras et al. (EDM) rho-schedule: sigma_i = (smax^(1/rho) + i/(n-1) * (smin^(1/rho) - smax^(1/rho)))^rho."""

    def __init__(self, sigma_min=0.002, sigma_max=80.0, rho=7.0):
        self.sigma_min, self.sigma_max, self.rho = sigma_min, sigma_max, rho

    def get_sigmas(self, n, device="cpu"):
        root_lo, root_hi = (s ** (1 / self.rho) for s in (self.sigma_min, self.sigma_max))
        t = torch.linspace(0, 1, n)  # host fp32: no device-side linspace / pow drift between back-ends
        return ((root_hi + t * (root_lo - root_hi)) ** self.rho).to(device)
