"""EulerEDM sampler (reference: vwm/modules/diffusionmodules/sampling.py), MI355X-native.

Same constructor and `__call__(denoiser, x, cond, uc, cond_frame, cond_mask, num_steps)` contract as the reference,
including its in-place scaling of the caller's `x` (sampling.py:36). Two execution paths, identical maths:

  * generic : `denoiser` is any callable `(x, sigma, cond, cond_mask) -> denoised` (e.g. the closure of the reference's
              sample_utils.py:314-315 around this package's Denoiser/OpenAIWrapper). Per step: mask-replace, guider
              prepare (cat), denoiser, guider combine, Euler update -- each elementwise stage one HIP kernel.
  * fused   : `denoiser` is a `FusedDenoiser` (Denoiser + OpenAIWrapper(VideoUNet) of this package). The sigma schedule
              and the EDM coefficients live on the host as floats (no per-step device sync: the reference syncs twice
              per step, sampling.py:109 and video_model.py:457), the UNet is entered token-major, and the step's
              elementwise work is two kernels (vk_sampler_prepare, vk_sampler_update).
"""
import os
from typing import Dict, Union

import torch

from ... import ops
from ...util import append_dims, default, instantiate_from_config
from .guiders import IdentityGuider


class FusedDenoiser:
    """Callable with the reference closure's signature that also exposes its parts to the sampler's fused path."""

    def __init__(self, denoiser, network):
        self.denoiser, self.network = denoiser, network

    def __call__(self, x, sigma, cond, cond_mask):
        return self.denoiser(self.network, x, sigma, cond, cond_mask)


class BaseDiffusionSampler:
    def __init__(self, discretization_config: Union[Dict, None], num_steps: Union[int, None] = None,
                 guider_config: Union[Dict, None] = None, verbose: bool = False, device: str = "cuda"):
        self.num_steps = num_steps
        self.discretization = instantiate_from_config(discretization_config)
        self.guider = instantiate_from_config(default(guider_config, {"target": "vista_amd.modules.diffusionmodules.guiders.IdentityGuider"}))
        self.verbose = verbose
        self.device = device

    def host_sigmas(self, num_steps=None):
        return self.discretization(self.num_steps if num_steps is None else num_steps, device="cpu").float()

    def prepare_sampling_loop(self, x, cond, uc=None, num_steps=None):
        sigmas = self.host_sigmas(num_steps)
        uc = default(uc, cond)
        s0 = float(torch.sqrt(1.0 + sigmas[0] ** 2))
        x.copy_(ops.scale_rows(x.float(), torch.full((x.shape[0],), s0, device=x.device)).to(x.dtype))  # in place, like the reference
        return x, sigmas, len(sigmas), cond, uc

    def denoise(self, x, denoiser, sigma, cond, cond_mask, uc):
        denoised = denoiser(*self.guider.prepare_inputs(x, sigma, cond, cond_mask, uc))
        return self.guider(denoised, sigma)


class SingleStepDiffusionSampler(BaseDiffusionSampler):
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, uc, *args, **kwargs):
        raise NotImplementedError

    def euler_step(self, x, d, dt):
        return x + dt * d


class EulerEDMSampler(SingleStepDiffusionSampler):
    def __init__(self, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.s_churn, self.s_tmin, self.s_tmax, self.s_noise = s_churn, s_tmin, s_tmax, s_noise
        # the churn noise source, `x -> standard normal tensor like x` (reference: torch.randn_like, sampling.py:81). A hook so that a caller
        # -- like do_sample's noise_fn -- can inject its own draws (tests replay the draws recorded from the reference).
        self.noise_fn = None

    # ---- generic path -------------------------------------------------------------------------------------------
    def sampler_step(self, sigma, next_sigma, denoiser, x, cond, cond_mask=None, uc=None, gamma=0.0):
        """sampling.py:78-89. sigma / next_sigma: (N,) device tensors."""
        sigma_hat = sigma * (gamma + 1.0)
        if gamma > 0:  # stochastic churn (never enabled by Vista: s_churn = 0, sample_utils.py:212); torch RNG unless noise_fn is set
            eps = (torch.randn_like(x) if self.noise_fn is None else self.noise_fn(x).to(device=x.device, dtype=x.dtype)) * self.s_noise
            x = x + eps * append_dims(sigma_hat ** 2 - sigma ** 2, x.ndim) ** 0.5
        denoised = self.denoise(x, denoiser, sigma_hat, cond, cond_mask, uc)
        return ops.euler_step(x, denoised.float(), sigma_hat, next_sigma)  # x + (x - denoised)/sigma_hat * (next - sigma_hat)

    @torch.no_grad()
    def __call__(self, denoiser, x, cond, uc=None, cond_frame=None, cond_mask=None, num_steps=None):
        x, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        sig = [float(s) for s in sigmas]
        # the reference evaluates cond_mask.any() on the device; the mask is a host-known 0/1 pattern set by the caller
        replace = cond_mask is not None and cond_frame is not None and bool(cond_mask.detach().cpu().any())
        if cond_mask is None:
            cond_mask = torch.zeros(x.shape[0], device=x.device)
        maskf = cond_mask.float().contiguous()
        if isinstance(denoiser, FusedDenoiser) and not isinstance(self.guider, IdentityGuider) and self.s_churn == 0.0 \
                and self._fused_layout_ok(denoiser, x, cond, uc):
            return self._sample_fused(denoiser, x, cond, uc, cond_frame, maskf, replace, sig)
        n = x.shape[0]
        xw = x.float()
        for i in range(num_sigmas - 1):
            if replace:
                xw = ops.mask_replace(xw, cond_frame.float(), maskf)
            gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sig[i] <= self.s_tmax else 0.0
            s_cur = torch.full((n,), sig[i], device=x.device)
            s_next = torch.full((n,), sig[i + 1], device=x.device)
            xw = self.sampler_step(s_cur, s_next, denoiser, xw, cond, maskf, uc, gamma)
        if replace:
            xw = ops.mask_replace(xw, cond_frame.float(), maskf)
        return xw.to(x.dtype)

    # ---- fused path ---------------------------------------------------------------------------------------------
    @staticmethod
    def _fused_layout_ok(fd, x, cond, uc):
        """The fused kernels hard-code Vista's layout: a 4-channel latent, a 4-channel `concat` conditioning of the same H x W
        (per frame or per clip), `crossattn` / `vector` present and an 8-channel UNet input. Anything else takes the generic path."""
        if x.dim() != 4 or x.shape[1] != 4 or getattr(fd.network.diffusion_model, "in_channels", None) != 8:
            return False
        for d in (cond, uc):
            if d is None or any(k not in d for k in ("concat", "crossattn", "vector")):
                return False
            c = d["concat"]
            if c.dim() != 4 or tuple(c.shape[1:]) != tuple(x.shape[1:]):
                return False
        return True

    def _sample_fused(self, fd, x, cond, uc, cond_frame, maskf, replace, sig):
        """`self.shard` (a vista_amd.parallel.FrameShard, set by the caller on every rank) turns on frame sharding: every
        rank passes the same full-window tensors, works on its own frames and returns the gathered full result."""
        loop = FusedLoop(self, fd, x.float().clone(), cond, uc, cond_frame, maskf, replace, sig, shard=getattr(self, "shard", None))
        for i in range(len(sig) - 1):
            loop.step(i)
        return loop.finish().to(x.dtype)


class FusedLoop:
    """State of one fused sampling run; `step(i)` is exactly one EulerEDMSampler.sampler_step (sampling.py:78-89):
    mask replace -> CFG-doubled UNet forward -> guider combine -> to_d -> Euler update. bench.py times this."""

    def __init__(self, sampler, fd, xw, cond, uc, cond_frame, maskf, replace, sig, shard=None, graph=None, cfg_streams=None):
        from .video_model import CIN_PAD
        self.cin_pad = CIN_PAD
        # cfg_streams (default: env VISTA_CFG_STREAMS, "0"): run the two guidance halves of a step as TWO UNet forwards of n images each
        # instead of one of 2n -- exact, the halves only meet on the batch axis (guiders.py:27-44) -- and, with graph replay, CONCURRENTLY:
        # half 0's graph on the launch stream, half 1's on a side stream, each with a split-K workspace of its own. The chip back-fills
        # one half's ragged last rounds and HBM-bound launches with the other half's work: 159.8 against 163.8 ms per UNet forward pair at
        # BASELINE config 2, while the same two forwards on one stream take 170.2 (profiles/r06_cfg_streams.txt). Eagerly the halves run one
        # after the other (bitwise the concurrent replays: same kernels, same buffers). One GPU only: a frame-shard rank keeps one stream.
        # graph: replay the UNet forward of every step from ONE captured hipGraph (the ~800 launches of a step become one host call; the
        # per-step scalars live in the sampler kernels outside it, the noise level enters through a device tensor). Default: env VISTA_HIPGRAPH=1.
        # Off whenever the forward contains collectives (frame-sharded groups): capturing RCCL is left opt-in (VISTA_HIPGRAPH=force).
        import os
        want = os.environ.get("VISTA_HIPGRAPH", "0") if graph is None else ("1" if graph else "0")
        self._graph_mode = want
        self._graph = None
        self.den, self.unet = fd.denoiser, fd.network.diffusion_model
        self.sig, self.replace, self.shard = sig, replace, shard
        n, _, self.H, self.W = xw.shape
        self.n, self.T = n, self.den.num_frames
        T, dev = self.T, xw.device
        scales = sampler.guider.frame_scales(T).float().repeat(n // T).to(dev)
        if shard is not None and n != T:
            raise NotImplementedError("frame sharding handles one window (b = 1) per call")
        loc = (lambda t: t) if shard is None else (lambda t: shard.take_local_frames(t).contiguous())
        self.xw, self.maskf, self.scales = loc(xw), loc(maskf), loc(scales)

        def both(k):
            a, b = uc[k], cond[k]
            if a.shape[0] != n:
                a, b = a.repeat_interleave(T, 0), b.repeat_interleave(T, 0)
            return a, b
        cu, cc = both("concat")
        self.cu, self.cc = loc(cu.float().contiguous()), loc(cc.float().contiguous())
        self.half = None if shard is None else shard.cfg_half  # CFG x frame hybrid: this rank runs ONE guidance half
        # a 1-rank frame group needs no re-sharding -- unless the shard asks for it (`always_exchange`: the one-rank RCCL test puts every exchange of the
        # forward through the process group although each is a copy to itself)
        self.unet_shard = shard if (shard is not None and (shard.P > 1 or getattr(shard, "always_exchange", False))) else None
        if self.half is None:
            self.ctx2 = torch.cat(both("crossattn"), 0)   # full window (replicated on every rank)
            self.y2 = torch.cat(both("vector"), 0)
            self.mask2 = torch.cat([maskf, maskf])
        else:
            self.ctx2 = both("crossattn")[self.half]
            self.y2 = both("vector")[self.half]
            self.mask2 = maskf
        self.cf = loc(cond_frame.float().contiguous()) if replace else None
        want_cs = os.environ.get("VISTA_CFG_STREAMS", "0") if cfg_streams is None else ("1" if cfg_streams else "0")
        self._split = want_cs == "1" and shard is None
        # EDM coefficients per step on the host (denoiser_scaling.py:51-59): no device round trip inside the loop
        self.coef = [tuple(float(v) for v in self.den.scaling(torch.tensor(s, dtype=torch.float32))) for s in sig[:-1]]

    def _unet(self, net_in, n_ts, c_noise):
        """One UNet forward on the step's input; eager, or the replay of the captured graph (same kernels, same buffers every step)."""
        use = self._graph_mode == "force" or (self._graph_mode == "1" and self.unet_shard is None)
        if self._split:
            return self._unet_halves(net_in, c_noise, use)
        if use and self.unet_shard is not None and getattr(self.unet_shard.comm, "backend", None) == "nccl" \
                and os.environ.get("VISTA_HIPGRAPH_RCCL", "0") != "1":
            # measured on the MI355X test boxes (round 6, one-rank nccl group, torch 2.10 + RCCL 2.26.6): capturing the sharded forward hangs -- under
            # the default capture mode ProcessGroupNCCL's watchdog thread dies with "operation not permitted when stream is capturing", under
            # "thread_local" the capture never returns (profiles/r06_cfg_streams.txt section 6). Refuse instead of hanging a multi-GPU job.
            raise RuntimeError("VISTA_HIPGRAPH=force / bench.py --graph with a frame-sharded forward over RCCL: capturing its collectives hangs on this "
                               "software stack (see DESIGN section 6); run the sharded step eagerly, or set VISTA_HIPGRAPH_RCCL=1 to try anyway")
        if not use:
            ts = torch.full((n_ts,), c_noise, device=self.xw.device)
            return self.unet.forward_tokens(net_in, ts, self.ctx2, self.y2, self.mask2, self.T, self.H, self.W, shard=self.unet_shard)
        g = self._graph
        if g is None:
            g = self._graph = self._graph_for(net_in, n_ts)
        g["in"].copy_(net_in)
        g["ts"].fill_(c_noise)
        g["graph"].replay()
        return g["out"]

    def _unet_halves(self, net_in, c_noise, graph):
        """cfg_streams: the step's forward as two forwards of n images (uncond rows [0, n), cond rows [n, 2n)); returns the (2n, S, C) output."""
        n = self.n
        half = lambda t, h: t[h * n:(h + 1) * n]  # noqa: E731
        if not graph:
            ts = torch.full((n,), c_noise, device=self.xw.device)
            return torch.cat([self.unet.forward_tokens(half(net_in, h), ts, half(self.ctx2, h), half(self.y2, h), half(self.mask2, h), self.T, self.H, self.W)
                              for h in (0, 1)])
        gs = self._graph
        if gs is None:
            gs = self._graph = [self._graph_for(half(net_in, h), n, ctx=half(self.ctx2, h), y=half(self.y2, h), mask=half(self.mask2, h), slot=h) for h in (0, 1)]
        for h in (0, 1):
            gs[h]["in"].copy_(half(net_in, h))
            gs[h]["ts"].fill_(c_noise)
        dev = net_in.device
        side = FusedLoop._CFG_STREAMS.get(dev.index)
        if side is None:
            side = FusedLoop._CFG_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            gs[1]["graph"].replay()
        gs[0]["graph"].replay()
        cur.wait_stream(side)
        return torch.cat([gs[0]["out"], gs[1]["out"]])

    _WARM_STREAMS = {}   # device -> the one side stream every warm-up forward runs on
    _CFG_STREAMS = {}    # device -> the side stream half 1's graph is replayed on (cfg_streams)

    def _graph_for(self, net_in, n_ts, ctx=None, y=None, mask=None, slot=0):
        """The captured forward for this window geometry. Graphs are cached on the UNet (not per FusedLoop: do_sample builds a FusedLoop per
        sampling round) and keyed by everything that fixes the launch sequence; the conditioning tensors are STATIC buffers of the graph,
        refreshed by copy when another run (re-)uses it. Warm-up and capture run inside ops.graph_workspace() -- one split-K workspace per
        device, allocated outside the capture -- and all warm-ups share one side stream."""
        unet = self.unet
        cache = unet.__dict__.setdefault("_hipgraph_cache", {})
        from .. import attention as _att
        if ctx is None:
            ctx, y, mask = self.ctx2, self.y2, self.mask2
        # ... including the module-level switches that decide WHICH kernels forward_tokens launches (bench.py's config-5 side figure and the
        # fp8 tests flip them in-process): a graph captured under other switches must not be replayed
        switches = (tuple(sorted(_att.FP8.items())), _att.FF_FUSED, _att.QKV_SPLIT, _att.Q_LOG2, ops.TILE_CFG, ops.GN_EPI,
                    None if self.unet_shard is None else getattr(self.unet_shard, "a2a_chunks", None))
        key = (tuple(net_in.shape), n_ts, tuple(ctx.shape), tuple(y.shape), tuple(mask.shape), self.T, self.H, self.W,
               None if self.unet_shard is None else id(self.unet_shard), str(net_in.device), switches, slot)
        g = cache.get(key)
        # the captured launches hold raw pointers into the packed weights (owned by the modules' _pk, not by the graph pool): the key covers
        # what Packable itself can see (identity + version counter of every parameter) AND the pack generation, which every
        # invalidate_packed() bumps -- writes through p.data (EMA swap), in-place loads into inference tensors and the load_state_dict
        # post-hooks are invisible to version counters but all go through invalidate_packed
        wkey = (_att.pack_generation(),) + tuple((id(q), _att.Packable._param_version(q)) for q in unet.parameters())   # once per run, not per step
        if g is not None and g["wkey"] != wkey:
            g = None   # parameters were replaced / updated since the capture: its launches point at the old packed weights
        if g is None:
            if ops.PROFILE_ATTN is not None:
                raise RuntimeError("ops.PROFILE_ATTN records timing events around attention launches; clear it before capturing a hipGraph")
            dev = net_in.device
            g = {"in": torch.empty_like(net_in), "ts": torch.empty((n_ts,), device=dev), "ctx": ctx.clone(), "y": y.clone(),
                 "mask": mask.clone(), "wkey": wkey}
            g["in"].copy_(net_in)
            g["ts"].fill_(0.0)
            fwd = lambda: unet.forward_tokens(g["in"], g["ts"], g["ctx"], g["y"], g["mask"], self.T, self.H, self.W, shard=self.unet_shard)  # noqa: E731
            side = FusedLoop._WARM_STREAMS.get(dev.index)
            if side is None:
                side = FusedLoop._WARM_STREAMS[dev.index] = torch.cuda.Stream(device=dev)
            with ops.graph_workspace(slot):   # (slot: the two halves of cfg_streams replay at the same time -- a workspace each)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):   # eager warm-up off the capture: weight packs and caches get built here
                    fwd()
                torch.cuda.current_stream().wait_stream(side)
                # the key is taken AFTER the warm-up: a forward that finds a switch flipped since its packs were built calls invalidate_packed()
                # itself and bumps the generation -- the key taken before it would be stale at once and the next window would re-capture for nothing
                g["wkey"] = (_att.pack_generation(),) + tuple((id(q), _att.Packable._param_version(q)) for q in unet.parameters())
                g["ws"] = ops.graph_workspace_tensor()   # the split-K workspace the captured launches point at lives as long as the graph does
                g["graph"] = torch.cuda.CUDAGraph()
                # a sharded forward holds collectives: ProcessGroupNCCL's watchdog THREAD polls their events, and under the default "global"
                # capture mode any HIP call of another thread is an error while this one captures (first run over a real RCCL group, round 6:
                # "operation not permitted when stream is capturing" from the watchdog, or a hang) -- "thread_local" confines the check to this thread
                mode = "global" if self.unet_shard is None else "thread_local"
                with torch.cuda.graph(g["graph"], capture_error_mode=mode):
                    g["out"] = fwd()
            cache[key] = g
        else:   # another run of the same geometry: its conditioning goes into the graph's static buffers
            g["ctx"].copy_(ctx)
            g["y"].copy_(y)
            g["mask"].copy_(mask)
        return g

    def step(self, i):
        c_skip, c_out, c_in, c_noise = self.coef[i]
        net_in = ops.sampler_prepare(self.xw, self.cf, self.maskf, self.cu, self.cc, self.cin_pad, c_in, self.replace)
        if self.half is None:
            net_out = self._unet(net_in, 2 * self.n, c_noise)
        else:  # run this rank's guidance half, then swap outputs with the partner that owns the same frames of the other half
            tl = self.xw.shape[0]
            mine = self._unet(net_in[self.half * tl:(self.half + 1) * tl], self.n, c_noise)
            net_out = self.shard.exchange_cfg_halves(mine)
        ops.sampler_update(self.xw, net_out, self.scales, c_out, c_skip, self.sig[i], self.sig[i + 1])

    def finish(self):
        if self.replace:
            self.xw = ops.mask_replace(self.xw, self.cf, self.maskf)
        return self.xw if self.shard is None else self.shard.gather_frames(self.xw)
