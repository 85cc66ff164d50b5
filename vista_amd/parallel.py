"""Frame-sharded multi-GPU execution of the denoising step (one process per GPU, RCCL over xGMI).

The reference has no multi-GPU inference path. Frames of the 25-frame window couple in three places per block pair
(temporal self-attention, the 3x1x1 temporal convs and the 5-D GroupNorms of every VideoResBlock.time_stack), so a
single all-gather is not enough. The partition used here:

  spatial half  (2-D ResBlock, spatial transformer block, up/down-sampling, per-image norms, sampler elementwise):
      FRAME-sharded -- rank r owns t_r frames (of both CFG halves), all pixels.           25 -> 4/3/3/3/3/3/3/3 at 8 GPUs
  temporal transformer (VideoTransformerBlock): every op is pointwise in space, so it runs
      PIXEL-sharded -- rank r owns all T frames of S/P pixels; `to_pixels` / `to_frames` re-shard with ONE all-to-all each
      (xGMI is point-to-point: every pair of GPUs exchanges its slice directly, no ring).
  temporal ResBlock (time_stack: GN5d -> 3x1x1 conv -> GN5d -> 3x1x1 conv): stays FRAME-sharded; each conv needs only the
      neighbour ranks' boundary frame (`halo_exchange`, 2 frames per conv instead of a full re-shard) and each 5-D GroupNorm
      a 64-float-per-clip all-reduce of its partial sums. Weights and the (tiny) conditioning tensors are replicated. No collective touches the spatial
half.

Communicators: `DistComm` wraps torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests);
`ThreadComm` runs P ranks as threads of one process (used to validate the sharded numerics on a single GPU).
"""
import threading

import torch


def split_counts(n, parts):
    """n items over `parts` ranks, larger shares first: 25 over 8 -> [4,3,3,3,3,3,3,3]."""
    base, rem = divmod(n, parts)
    return [base + (1 if r < rem else 0) for r in range(parts)]


def offsets(counts):
    o = [0]
    for c in counts:
        o.append(o[-1] + c)
    return o


class SelfComm:
    """Single-rank communicator (a frame-shard group of one rank, e.g. the CFG-only split at 2 GPUs)."""
    rank, world = 0, 1

    def all_to_all(self, recv, send, out_splits, in_splits, async_op=False):
        recv.copy_(send)

    def all_reduce_sum(self, t):
        pass

    def all_gather_list(self, t, counts):
        return [t]


def make_shard(T, world, rank, mode="hybrid", make_group=None):
    """Build the FrameShard of `rank` in a `world`-rank job. mode 'frames': pure frame sharding (both CFG halves per rank).
    mode 'hybrid': CFG halves x frame groups when world is even (ranks [0, world/2) = uncond, the rest = cond).
    make_group(list_of_ranks) -> communicator for that subgroup (must be called collectively, same order on all ranks)."""
    if world == 1:
        return None
    def named(comm, name):
        if comm is not None and hasattr(comm, "name"):
            comm.name = name
        return comm
    if mode == "frames" or world % 2:
        return FrameShard(T, named(make_group(list(range(world))), f"frames[0..{world - 1}]"), B=2)
    half = world // 2
    groups = [named(make_group(list(range(h * half, (h + 1) * half))), f"cfg-half-{h}[{h * half}..{(h + 1) * half - 1}]") if half > 1 else None
              for h in range(2)]
    pairs = [named(make_group([i, i + half]), f"cfg-pair[{i},{i + half}]") for i in range(half)]
    h, i = rank // half, rank % half
    comm = groups[h] if half > 1 else SelfComm()
    return FrameShard(T, comm, B=1, cfg_pair=pairs[i], cfg_half=h)


class CollectiveError(RuntimeError):
    """A collective of the sharded step failed: names the collective, the group and the split lists (first contact with RCCL happens on
    the driver's multi-GPU box, where the only diagnostics are what the process prints)."""


class DistComm:
    """torch.distributed transport. With the gloo backend (CPU tests, and the single-GPU dry run of bench.py's multi-rank
    path) device tensors are staged through host memory; with "nccl" (= RCCL) they go GPU-to-GPU over xGMI.
    Every collective signature the nccl path uses is one of the three methods below (all_to_all_single with per-rank split lists that
    may hold zeros, all_reduce(SUM), all_gather of equally padded tensors) and is exercised over gloo by tests/test_parallel_cpu.py."""

    def __init__(self, group=None, name="world"):
        import torch.distributed as dist
        self.dist, self.group, self.name = dist, group, name
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)  # rank WITHIN the group
        self.backend = dist.get_backend(group)
        self.host_staged = self.backend == "gloo"

    def _fail(self, what, e, **info):
        detail = ", ".join(f"{k}={v}" for k, v in info.items())
        raise CollectiveError(f"[global rank {self.dist.get_rank()}] {what} on group '{self.name}' (backend {self.backend}, "
                              f"group rank {self.rank}/{self.world}) failed: {detail}: {type(e).__name__}: {e}") from e

    def all_to_all(self, recv, send, out_splits, in_splits, async_op=False):
        """async_op (device-to-device backends only): returns the torch.distributed Work handle -- the exchange runs on the backend's own
        stream behind everything enqueued so far, the caller's stream continues and `handle.wait()` orders it after the exchange."""
        if len(out_splits) != self.world or len(in_splits) != self.world or sum(out_splits) != recv.numel() or sum(in_splits) != send.numel():
            raise CollectiveError(f"all_to_all_single on group '{self.name}': split lists do not match the buffers: out_splits {out_splits} "
                                  f"(recv {recv.numel()}), in_splits {in_splits} (send {send.numel()}), group size {self.world}")
        try:
            if self.host_staged and recv.is_cuda:
                r = torch.empty(recv.shape, dtype=recv.dtype)
                self.dist.all_to_all_single(r, send.cpu(), out_splits, in_splits, group=self.group)
                recv.copy_(r)
                return
            return self.dist.all_to_all_single(recv, send, out_splits, in_splits, group=self.group, async_op=bool(async_op))
        except CollectiveError:
            raise
        except Exception as e:  # noqa: BLE001
            self._fail("all_to_all_single", e, dtype=recv.dtype, out_splits=out_splits, in_splits=in_splits, device=recv.device)

    def all_reduce_sum(self, t):
        try:
            if self.host_staged and t.is_cuda:
                c = t.cpu()
                self.dist.all_reduce(c, group=self.group)
                t.copy_(c)
                return
            self.dist.all_reduce(t, group=self.group)
        except Exception as e:  # noqa: BLE001
            self._fail("all_reduce(SUM)", e, dtype=t.dtype, numel=t.numel(), device=t.device)

    def all_gather_list(self, t, counts):
        mx = max(counts)  # pad to equal sizes: uneven all_gather is not portable across backends
        dev = t.device
        try:
            src = t.cpu() if (self.host_staged and t.is_cuda) else t
            pad = torch.zeros((mx,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            pad[:src.shape[0]] = src
            outs = [torch.empty_like(pad) for _ in counts]
            self.dist.all_gather(outs, pad, group=self.group)
            return [o[:c].to(dev) for o, c in zip(outs, counts)]
        except Exception as e:  # noqa: BLE001
            self._fail("all_gather (padded list)", e, dtype=t.dtype, counts=counts, shape=tuple(t.shape), device=dev)


class ThreadGroups:
    """Factory of ThreadComm sub-communicators for a job of thread ranks: `make(rank)` returns the make_group callback of
    that rank (same contract as the torch.distributed one: called collectively, returns None to non-members)."""

    def __init__(self):
        self.lock = threading.Lock()
        self.shared = {}

    def make(self, rank):
        def make_group(ranks):
            key = tuple(ranks)
            with self.lock:
                if key not in self.shared:
                    self.shared[key] = ThreadComm.Shared(len(ranks))
            return ThreadComm(self.shared[key], ranks.index(rank)) if rank in ranks else None
        return make_group

    def abort(self):
        for sh in self.shared.values():
            sh.barrier.abort()


class ThreadComm:
    """In-process emulation: P threads, one per rank, exchange through shared slots guarded by a barrier."""

    class Shared:
        def __init__(self, world):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def _exchange(self, payload):
        sh = self.shared
        sh.slots[self.rank] = payload
        t = payload[0] if isinstance(payload, tuple) else payload
        if t.is_cuda:
            torch.cuda.synchronize()
        sh.barrier.wait()
        got = list(sh.slots)
        sh.barrier.wait()
        return got

    def all_to_all(self, recv, send, out_splits, in_splits, async_op=False):
        got = self._exchange((send, in_splits))
        pos = 0
        for q in range(self.world):
            s_q, splits_q = got[q]
            o = offsets(splits_q)
            chunk = s_q[o[self.rank]:o[self.rank + 1]]
            assert chunk.numel() == out_splits[q]
            recv[pos:pos + out_splits[q]].copy_(chunk)
            pos += out_splits[q]
        if recv.is_cuda:
            torch.cuda.synchronize()
        self.shared.barrier.wait()

    def all_reduce_sum(self, t):
        got = self._exchange(t.clone())
        acc = got[0].clone()
        for q in range(1, self.world):  # fixed order: every rank computes the identical sum
            acc += got[q]
        t.copy_(acc)
        if t.is_cuda:
            torch.cuda.synchronize()
        self.shared.barrier.wait()

    def all_gather_list(self, t, counts):
        got = self._exchange(t.clone())
        return [g.clone() for g in got]


def _rows(t, index):
    """Row gather of an exchange plan; None = the identity (the tensor itself, made contiguous if it is not)."""
    return t.contiguous() if index is None else t.index_select(0, index)


class FrameShard:
    """Frame / pixel partition of one sampling window and the exchanges between the two layouts.

    `cfg_pair` (optional) turns on the CFG x frame hybrid: the classifier-free-guidance halves (uncond / cond) never
    interact inside the UNet, so world = 2 x P ranks run them on two independent frame-shard groups of P ranks (this
    object then describes ONE half: B = 1, `comm` = that half's group) and the two ranks that own the same frames
    exchange their (t_local, S, 4) network outputs once per step through `cfg_pair` (a 2-rank communicator,
    index 0 = uncond, 1 = cond). 25 frames on 8 GPUs: 2 x (7/6/6/6) instead of 4/3/3/3/3/3/3/3."""

    def __init__(self, T, comm, B=2, cfg_pair=None, cfg_half=None):
        self.T, self.comm, self.B = T, comm, B
        self.cfg_pair, self.cfg_half = cfg_pair, cfg_half
        if cfg_pair is not None:
            assert B == 1 and cfg_half in (0, 1)
        self.P, self.rank = comm.world, comm.rank
        if self.P > T:
            raise ValueError(f"cannot shard {T} frames over {self.P} ranks")
        self.t_counts = split_counts(T, self.P)
        self.t_off = offsets(self.t_counts)
        self.t_local = self.t_counts[self.rank]
        self._plans = {}
        # a one-rank group needs no exchange and FusedLoop skips the sharded forward for it -- unless this is set: then every re-shard, halo and statistics
        # exchange of the forward is issued on the communicator although each is a copy to itself (the one-rank RCCL test on a one-GPU box)
        self.always_exchange = False
        # VISTA_A2A_CHUNKS = n > 1 (opt-in, default 1): the temporal block runs on n pixel sub-ranges of the rank's slice in turn and each
        # sub-range's way back to the frame layout is its own all-to-all, started asynchronously -- the exchange of sub-range i runs under
        # the compute of sub-range i + 1 (SURVEY 8e "overlap"). Same result bit for bit: the temporal block is pointwise in space.
        import os
        self.a2a_chunks = max(1, int(os.environ.get("VISTA_A2A_CHUNKS", "1")))

    # global image ids (b*T + t) of this rank's frames, in local (b, t_local) order
    def local_image_ids(self):
        t0 = self.t_off[self.rank]
        return [b * self.T + t0 + i for b in range(self.B) for i in range(self.t_local)]

    def pixel_counts(self, S):
        return split_counts(S, self.P)

    # ---- exchanges. Packing / unpacking is one row gather each (cached int64 index tensors), so an exchange is
    #      gather -> all-to-all -> gather regardless of the number of ranks.
    def _plan(self, S, device):
        key = (S, str(device))
        if key in self._plans:
            return self._plans[key]
        B, P, r, T, t_l = self.B, self.P, self.rank, self.T, self.t_local
        sc = self.pixel_counts(S)
        so = offsets(sc)
        s_r = sc[r]
        ar = torch.arange
        # frames -> pixels: send buffer = for q: rows (b, t_local, s in slice q) of x viewed (B*t_l*S, C)
        pack_fp = torch.cat([((ar(B)[:, None, None] * t_l + ar(t_l)[None, :, None]) * S + (so[q] + ar(sc[q]))[None, None, :]).reshape(-1)
                             for q in range(P)])
        # received = for q: (B, t_q, s_r); output rows (b, t_global, s) of (B*T*s_r, C) gather from the received buffer
        ro = offsets([B * self.t_counts[q] * s_r for q in range(P)])
        unpack_fp = torch.empty(B, T, s_r, dtype=torch.int64)
        for q in range(P):
            tq = self.t_counts[q]
            unpack_fp[:, self.t_off[q]:self.t_off[q + 1]] = ro[q] + ((ar(B)[:, None, None] * tq + ar(tq)[None, :, None]) * s_r + ar(s_r)[None, None, :])
        # pixels -> frames: send buffer = for q: rows (b, t in q's frames, s) of y viewed (B*T*s_r, C)
        pack_pf = torch.cat([((ar(B)[:, None, None] * T + (self.t_off[q] + ar(self.t_counts[q]))[None, :, None]) * s_r + ar(s_r)[None, None, :]).reshape(-1)
                             for q in range(P)])
        ro2 = offsets([B * t_l * sc[q] for q in range(P)])
        unpack_pf = torch.empty(B, t_l, S, dtype=torch.int64)
        for q in range(P):
            unpack_pf[:, :, so[q]:so[q + 1]] = ro2[q] + ((ar(B)[:, None, None] * t_l + ar(t_l)[None, :, None]) * sc[q] + ar(sc[q])[None, None, :])
        # A gather whose index is the identity is skipped (None): with one clip per group (B = 1: the CFG x frame hybrid, the 8-GPU default)
        # the per-peer blocks of the pixel-sharded side are whole frame ranges, already in global frame order -- the receive buffer of
        # to_pixels IS the result and the send buffer of to_frames IS the input; only the frame-sharded side needs its row gather.
        def ident(ix):
            ix = ix.reshape(-1)
            return None if torch.equal(ix, ar(ix.numel())) else ix.to(device)
        plan = {"sc": sc, "s_r": s_r,
                "pack_fp": ident(pack_fp), "unpack_fp": ident(unpack_fp),
                "pack_pf": ident(pack_pf), "unpack_pf": ident(unpack_pf)}
        self._plans[key] = plan
        return plan

    def exchange_splits(self, S, C, rank=None):
        """Element counts per peer of the three all_to_all_single calls of a block pair, for `rank` (default: this rank):
        {"to_pixels": (in_splits, out_splits), "to_frames": (...), "halo": (...)}. Pure function of (T, P, B, S, C, rank): the CPU tests check
        that every rank's out_splits[q] equals rank q's in_splits[r] for all world sizes and all four UNet levels."""
        r = self.rank if rank is None else rank
        B, P = self.B, self.P
        sc = self.pixel_counts(S)
        t_r, s_r = self.t_counts[r], sc[r]
        fs = B * S * C
        halo_in = [fs if abs(q - r) == 1 else 0 for q in range(P)]  # one boundary frame to each existing neighbour, nothing to the others
        return {"to_pixels": ([B * t_r * sc[q] * C for q in range(P)], [B * self.t_counts[q] * s_r * C for q in range(P)]),
                "to_frames": ([B * self.t_counts[q] * s_r * C for q in range(P)], [B * t_r * sc[q] * C for q in range(P)]),
                "halo": (halo_in, list(halo_in))}

    def selfcheck(self, device, levels=((9216, 320), (2304, 640), (576, 1280), (144, 1280)), C=8, log=None):
        """Plumbing pass run by bench.py before any model work when N > 1: every collective signature of a sharded step, with the real
        split structure of each UNet level (S tokens; C scaled down so the whole pass moves < 1 MB) and VALUE checks, synchronising after
        each call so that an asynchronous RCCL failure is reported against the collective that caused it."""
        def sync():
            if torch.device(device).type == "cuda":
                torch.cuda.synchronize()

        def step(name, fn):
            try:
                fn()
                sync()
            except CollectiveError:
                raise
            except Exception as e:  # noqa: BLE001
                raise CollectiveError(f"selfcheck step '{name}' failed on frame-shard rank {self.rank}/{self.P}: {type(e).__name__}: {e}") from e
            if log:
                log(f"selfcheck ok: {name}")
        B, T, r = self.B, self.T, self.rank
        t0 = self.t_off[r]

        def stats():
            s = torch.full((B * 64,), float(r + 1), dtype=torch.float32, device=device)  # the 5-D GroupNorm's 64 floats per clip
            self.all_reduce_sum(s)
            assert float(s[0]) == self.P * (self.P + 1) / 2, "all_reduce value"
        step(f"all_reduce_sum {B * 64 * 4} B", stats)

        def chunks_agree():
            # VISTA_A2A_CHUNKS is read per rank from its own environment and decides how many all_to_all_single calls a block issues: ranks
            # that disagree would issue different collective sequences and hang. Sum and sum of squares pin "all equal" with one all-reduce.
            v = torch.tensor([float(self.a2a_chunks), float(self.a2a_chunks) ** 2] + [0.0] * (B * 64 - 2), dtype=torch.float32, device=device)
            self.all_reduce_sum(v)
            if float(v[0]) != self.P * self.a2a_chunks or float(v[1]) != self.P * self.a2a_chunks ** 2:
                raise CollectiveError(f"VISTA_A2A_CHUNKS differs between the ranks of a frame-shard group (this rank: {self.a2a_chunks}, "
                                      f"group mean {float(v[0]) / self.P:.2f}): every rank must issue the same collective sequence")
        step("VISTA_A2A_CHUNKS agrees across the group", chunks_agree)
        for S, _ in levels:
            def roundtrip(S=S):
                # value = global (b, t, s) id, so a mis-routed chunk is caught, not just a size mismatch
                ids = ((torch.arange(B)[:, None, None] * T + (t0 + torch.arange(self.t_local))[None, :, None]) * S + torch.arange(S)[None, None, :])
                x = (ids.reshape(B * self.t_local, S, 1).float() + torch.arange(C).float() / 16).to(device)  # exact in fp32 (ids < 2^24)
                xp = self.to_pixels(x)
                so = offsets(self.pixel_counts(S))
                want = ((torch.arange(B)[:, None, None] * T + torch.arange(T)[None, :, None]) * S + (so[r] + torch.arange(so[r + 1] - so[r]))[None, None, :])
                assert torch.equal(xp[..., 0].cpu().float().reshape(-1), want.reshape(-1).float()), "to_pixels routed a chunk to the wrong place"
                assert torch.equal(self.to_frames(xp, S), x), "to_frames(to_pixels(x)) != x"
                xb = (x % 251).to(torch.bfloat16)  # the dtype the real step moves
                assert torch.equal(self.to_frames(self.to_pixels(xb), S), xb), "bf16 round trip"
                nch = min(self.a2a_chunks, min(self.pixel_counts(S)))
                if nch > 1:   # the opt-in chunked, asynchronous way back (to_frames_begin / to_frames_end) on the same data
                    xp2 = self.to_pixels(xb)
                    out = torch.empty_like(xb)
                    pend = [self.to_frames_begin(xp2[:, lo:hi].contiguous(), S, nch, ci) for ci, (lo, hi) in enumerate(self.pixel_chunks(S, nch))]
                    for pnd in pend:
                        self.to_frames_end(pnd, out)
                    assert torch.equal(out, xb), "chunked to_frames round trip"
            step(f"frame<->pixel all_to_all_single, S={S} (pixel slices {self.pixel_counts(S)})", roundtrip)

            def halo(S=S):
                h = torch.zeros((B * self.t_local, S, C), dtype=torch.bfloat16, device=device)
                h.view(B, self.t_local, S, C)[:] = (t0 + torch.arange(self.t_local, device=device)).to(torch.bfloat16)[None, :, None, None]
                prev, nxt = self.halo_exchange(h)
                assert (prev is None) == (r == 0) and (nxt is None) == (r == self.P - 1)
                assert prev is None or float(prev[0, 0, 0]) == t0 - 1
                assert nxt is None or float(nxt[0, 0, 0]) == t0 + self.t_local
            step(f"halo all_to_all_single with zero-length splits, S={S}", halo)

        def gather():
            g = self.gather_frames(torch.full((self.t_local, 3), float(r), device=device))
            assert g.shape[0] == T and g[:, 0].tolist() == [float(q) for q in range(self.P) for _ in range(self.t_counts[q])]
        step(f"all_gather of unequal frame counts {self.t_counts}", gather)
        if self.cfg_pair is not None:
            def pair():
                both = self.exchange_cfg_halves(torch.full((self.t_local, 4, 2), float(self.cfg_half), device=device))
                assert both.shape[0] == 2 * self.t_local and float(both[0, 0, 0]) == 0.0 and float(both[-1, 0, 0]) == 1.0
            step("cfg-pair all_gather", pair)

    def to_pixels(self, x):
        """(B*t_local, S, C) frame-sharded -> (B*T, S_r, C) pixel-sharded (frames in global order)."""
        B, P = self.B, self.P
        n, S, C = x.shape
        assert n == B * self.t_local
        pl = self._plan(S, x.device)
        sc, s_r = pl["sc"], pl["s_r"]
        send = _rows(x.reshape(n * S, C), pl["pack_fp"])
        in_splits, out_splits = self.exchange_splits(S, C)["to_pixels"]
        recv = torch.empty(sum(out_splits), dtype=x.dtype, device=x.device)
        self.comm.all_to_all(recv, send.reshape(-1), out_splits, in_splits)
        return _rows(recv.view(-1, C), pl["unpack_fp"]).view(B * self.T, s_r, C)

    def to_frames(self, y, S):
        """(B*T, S_r, C) pixel-sharded -> (B*t_local, S, C) frame-sharded."""
        B, P = self.B, self.P
        n, s_r, C = y.shape
        assert n == B * self.T
        pl = self._plan(S, y.device)
        sc = pl["sc"]
        assert s_r == pl["s_r"]
        send = _rows(y.reshape(n * s_r, C), pl["pack_pf"])
        in_splits, out_splits = self.exchange_splits(S, C)["to_frames"]
        recv = torch.empty(sum(out_splits), dtype=y.dtype, device=y.device)
        self.comm.all_to_all(recv, send.reshape(-1), out_splits, in_splits)
        return _rows(recv.view(-1, C), pl["unpack_pf"]).view(B * self.t_local, S, C)

    # ---- chunked way back (pixels -> frames) for compute / transport overlap
    def pixel_chunks(self, S, chunks):
        """[(lo, hi)] sub-ranges of THIS rank's pixel slice; every rank cuts its own slice by the same rule (split_counts)."""
        o = offsets(split_counts(self.pixel_counts(S)[self.rank], chunks))
        return [(o[c], o[c + 1]) for c in range(chunks)]

    def _chunk_plan(self, S, chunks, c, device):
        key = (S, chunks, c, str(device))
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        B, P, T, t_l = self.B, self.P, self.T, self.t_local
        sc = self.pixel_counts(S)
        so = offsets(sc)
        sub = [split_counts(sc[q], chunks) for q in range(P)]          # every rank's cut of its own slice
        cc = [sub[q][c] for q in range(P)]                              # width of sub-range c on rank q
        co = [offsets(sub[q])[c] for q in range(P)]                     # its offset inside q's slice
        n_c = cc[self.rank]
        ar = torch.arange
        # send: for q, rows (b, t in q's frames, s) of y viewed (B*T*n_c, C)
        pack = torch.cat([((ar(B)[:, None, None] * T + (self.t_off[q] + ar(self.t_counts[q]))[None, :, None]) * n_c + ar(n_c)[None, None, :]).reshape(-1)
                          for q in range(P)]) if n_c else torch.empty(0, dtype=torch.int64)
        # received: for q, rows (b, t_local, s in q's sub-range) -> rows of the frame-sharded result viewed (B*t_l*S, C)
        dest = torch.cat([((ar(B)[:, None, None] * t_l + ar(t_l)[None, :, None]) * S + (so[q] + co[q] + ar(cc[q]))[None, None, :]).reshape(-1)
                          for q in range(P)])
        pl = {"n_c": n_c, "pack": pack.to(device), "dest": dest.to(device),
              "in_rows": [B * self.t_counts[q] * n_c for q in range(P)], "out_rows": [B * t_l * cc[q] for q in range(P)]}
        self._plans[key] = pl
        return pl

    def to_frames_begin(self, y, S, chunks, c):
        """y: (B*T, n_c, C) -- sub-range c (of `chunks`) of this rank's pixel slice, all frames. Packs it and STARTS its all-to-all towards the
        frame layout (asynchronously where the transport can); returns the pending exchange for to_frames_end."""
        n, n_c, C = y.shape
        pl = self._chunk_plan(S, chunks, c, y.device)
        assert n == self.B * self.T and n_c == pl["n_c"]
        send = y.reshape(n * n_c, C).index_select(0, pl["pack"])
        recv = torch.empty(sum(pl["out_rows"]) * C, dtype=y.dtype, device=y.device)
        work = self.comm.all_to_all(recv, send.reshape(-1), [r * C for r in pl["out_rows"]], [r * C for r in pl["in_rows"]], async_op=True)
        return (work, recv, send, pl, C)  # (send is kept alive until the exchange has been waited for)

    def to_frames_end(self, pending, out):
        """Waits for a to_frames_begin exchange and scatters it into out (B*t_local, S, C)."""
        work, recv, _send, pl, C = pending
        if work is not None and hasattr(work, "wait"):
            work.wait()
        out.view(-1, C).index_copy_(0, pl["dest"], recv.view(-1, C))
        return out

    def halo_exchange(self, h):
        """Frame-sharded temporal conv support: h (B*t_local, S, C) -> (prev, next), each (B, S, C): the neighbour ranks' last /
        first frame, or None at the ends of the window (= the conv's zero padding). One sparse all-to-all (only the two
        neighbour splits are non-empty), 2 frames instead of the (P-1)/P of the whole activation a re-shard moves."""
        B, t_l, r, P = self.B, self.t_local, self.rank, self.P
        n, S, C = h.shape
        assert n == B * t_l
        h4 = h.view(B, t_l, S, C)
        fs = B * S * C
        in_splits, out_splits = self.exchange_splits(S, C)["halo"]
        parts = []
        if r > 0:
            parts.append(h4[:, 0].reshape(-1))
        if r < P - 1:
            parts.append(h4[:, t_l - 1].reshape(-1))
        send = torch.cat(parts) if parts else h.new_empty(0)
        recv = torch.empty(sum(out_splits), dtype=h.dtype, device=h.device)
        self.comm.all_to_all(recv, send, out_splits, in_splits)
        prev = recv[:fs].view(B, S, C) if r > 0 else None
        nxt = recv[(fs if r > 0 else 0):(fs if r > 0 else 0) + fs].view(B, S, C) if r < P - 1 else None
        return prev, nxt

    def all_reduce_sum(self, t):
        self.comm.all_reduce_sum(t)

    def gather_frames(self, x_local):
        """(t_local, ...) per rank -> (T, ...) on every rank (final latents)."""
        return torch.cat(self.comm.all_gather_list(x_local, self.t_counts), 0)

    def take_local_rows(self, full):
        """Rows of a (B*T, ...) replicated tensor that belong to this rank's images, in local order."""
        key = str(full.device)
        cache = self.__dict__.setdefault("_local_rows_idx", {})
        if key not in cache:  # built once: torch.tensor(list, device=cuda) is a blocking host->device copy (6 ms mid-step)
            cache[key] = torch.tensor(self.local_image_ids(), device=full.device)
        return full.index_select(0, cache[key])

    def exchange_cfg_halves(self, net_out_half):
        """Hybrid mode: (t_local, S, C) output of this rank's CFG half -> (2*t_local, S, C) [uncond; cond] on both partners."""
        halves = self.cfg_pair.all_gather_list(net_out_half.contiguous(), [net_out_half.shape[0]] * 2)
        return torch.cat(halves, 0)

    def take_local_frames(self, full):
        """(T, ...) -> (t_local, ...)."""
        return full[self.t_off[self.rank]:self.t_off[self.rank + 1]]
