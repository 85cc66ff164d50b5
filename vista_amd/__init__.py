"""vista_amd -- MI355X-native (gfx950) implementation of Vista's denoising hot path:
EulerEDM sampler loop x spatiotemporal VideoUNet, behind the reference's class / config / state-dict API.
See DESIGN.md and include/vista_hip.h."""
__version__ = "0.1.0"
