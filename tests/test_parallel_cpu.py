"""Multi-GPU partition / exchange logic of vista_amd.parallel on CPU: in-process thread ranks and real 2-process gloo.
(The sharded numerics of the whole UNet are validated on the GPU box with thread ranks: tests/test_parallel_gpu.py.)"""
import os
import threading

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vista_amd.parallel import DistComm, FrameShard, ThreadComm, offsets, split_counts


def _check_chunked_way_back(shard, X, chunks):
    """to_frames_begin / to_frames_end over pixel sub-ranges == the one-shot to_frames, bit for bit."""
    B, T, S, C = X.shape
    r = shard.rank
    so = offsets(shard.pixel_counts(S))
    y = X[:, :, so[r]:so[r + 1]].reshape(B * T, -1, C).contiguous()          # this rank's pixel-sharded tensor
    want = shard.to_frames(y, S)
    out = torch.full_like(want, float("nan"))
    parts = shard.pixel_chunks(S, chunks)
    assert parts[0][0] == 0 and parts[-1][1] == y.shape[1] and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    pending = [shard.to_frames_begin(y[:, lo:hi].contiguous(), S, chunks, c) for c, (lo, hi) in enumerate(parts)]
    for p in pending:
        shard.to_frames_end(p, out)
    assert torch.equal(out, want), "chunked pixels->frames must equal the one-shot exchange"


@pytest.mark.parametrize("P,T,S,chunks", [(2, 5, 16, 2), (3, 7, 10, 3), (4, 25, 18, 2), (8, 25, 144, 2), (4, 25, 37, 3)])
def test_chunked_way_back_thread_ranks(P, T, S, chunks):
    B, C = 2, 8
    X = torch.randn(B, T, S, C)
    shared = ThreadComm.Shared(P)
    errs = []

    def run(rank):
        try:
            _check_chunked_way_back(FrameShard(T, ThreadComm(shared, rank), B=B), X, chunks)
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]


def _gloo_chunk_worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        X = torch.randn(2, 7, 11, 8)
        sh = FrameShard(7, DistComm(), B=2)
        for chunks in (2, 3):
            _check_chunked_way_back(sh, X, chunks)   # async_op=True Work handles over real processes (gloo), 11 pixels over 3 ranks: 4/4/3
    finally:
        dist.destroy_process_group()


def test_chunked_way_back_gloo_three_processes():
    port = 29900 + (os.getpid() % 1500)
    mp.spawn(_gloo_chunk_worker, args=(3, port), nprocs=3, join=True)


def test_split_counts_matches_baseline_partition():
    assert split_counts(25, 8) == [4, 3, 3, 3, 3, 3, 3, 3]  # BASELINE.json config 3
    assert split_counts(25, 4) == [7, 6, 6, 6] and split_counts(25, 2) == [13, 12] and split_counts(25, 1) == [25]
    assert offsets([4, 3, 3]) == [0, 4, 7, 10]
    with pytest.raises(ValueError):
        FrameShard(3, type("C", (), {"world": 4, "rank": 0})())


def _check_rank(shard, X):
    """X: the global (B, T, S, C) tensor, identical on all ranks."""
    B, T, S, C = X.shape
    r, P = shard.rank, shard.P
    t0, t1 = shard.t_off[r], shard.t_off[r + 1]
    x_f = X[:, t0:t1].reshape(B * (t1 - t0), S, C).contiguous()
    so = offsets(shard.pixel_counts(S))
    x_p = shard.to_pixels(x_f)
    assert torch.equal(x_p, X[:, :, so[r]:so[r + 1]].reshape(B * T, -1, C)), "to_pixels must yield all frames of the rank's pixel slice"
    back = shard.to_frames(x_p, S)
    assert torch.equal(back, x_f), "to_frames(to_pixels(x)) must be the identity"
    s = torch.full((4,), float(r + 1))
    shard.all_reduce_sum(s)
    assert torch.equal(s, torch.full((4,), float(P * (P + 1) // 2)))
    full = torch.arange(B * T).float()[:, None].repeat(1, 3)
    loc = shard.take_local_rows(full)
    assert loc[:, 0].tolist() == [float(b * T + t) for b in range(B) for t in range(t0, t1)]
    g = shard.gather_frames(X[0, t0:t1, :, 0].contiguous())
    assert torch.equal(g, X[0, :, :, 0])


@pytest.mark.parametrize("P,T,S", [(2, 5, 16), (3, 7, 10), (8, 25, 144), (4, 25, 18)])
def test_frame_pixel_exchange_thread_ranks(P, T, S):
    B, C = 2, 8
    X = torch.randn(B, T, S, C)
    shared = ThreadComm.Shared(P)
    errs = []

    def run(rank):
        try:
            _check_rank(FrameShard(T, ThreadComm(shared, rank), B=B), X)
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


@pytest.mark.parametrize("P,T,S", [(4, 25, 18), (4, 13, 144), (3, 7, 10)])
def test_one_clip_groups_skip_the_pixel_side_gathers(P, T, S):
    """B = 1 (one CFG half per frame-shard group: the 8-GPU hybrid layout): the per-peer blocks of the pixel-sharded side are whole frame
    ranges in global order, so the receive buffer of to_pixels is the result and the input of to_frames is the send buffer -- the plan holds no
    index for them (round 4; two of the four row gathers of a temporal block's round trip) and the exchange is still exact. B = 2 keeps all four."""
    C = 8
    for B in (1, 2):
        X = torch.randn(B, T, S, C)
        shared = ThreadComm.Shared(P)
        errs, plans = [], {}

        def run(rank, B=B, X=X, shared=shared, errs=errs, plans=plans):
            try:
                sh = FrameShard(T, ThreadComm(shared, rank), B=B)
                _check_rank(sh, X)
                plans[rank] = sh._plan(S, torch.device("cpu"))
            except Exception as e:  # noqa: BLE001
                errs.append((rank, repr(e)))
                shared.barrier.abort()

        th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for r in range(P):
            pl = plans[r]
            assert (pl["unpack_fp"] is None) == (B == 1) and (pl["pack_pf"] is None) == (B == 1), (B, r)
            assert pl["pack_fp"] is not None and pl["unpack_pf"] is not None   # the frame-sharded side always permutes (P > 1)


def _gloo_worker(rank, world, port, T, S):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        X = torch.randn(2, T, S, 8)
        _check_rank(FrameShard(T, DistComm(), B=2), X)
    finally:
        dist.destroy_process_group()


def test_frame_pixel_exchange_gloo_two_processes():
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker, args=(2, port, 5, 16), nprocs=2, join=True)


def test_make_shard_hybrid_layout_threads():
    """8 ranks -> 2 CFG halves x 4 frame ranks (7/6/6/6); partner pairs (i, i+4); exchanges work inside each half."""
    from vista_amd.parallel import ThreadGroups, make_shard
    world, T, S, C = 8, 25, 16, 4
    groups = ThreadGroups()
    info, errs = [None] * world, []
    X = torch.randn(1, T, S, C)

    def run(rank):
        try:
            sh = make_shard(T, world, rank, mode="hybrid", make_group=groups.make(rank))
            assert sh.B == 1 and sh.P == 4 and sh.t_counts == [7, 6, 6, 6] and sh.cfg_half == rank // 4 and sh.rank == rank % 4
            t0, t1 = sh.t_off[sh.rank], sh.t_off[sh.rank + 1]
            x_f = X[:, t0:t1].reshape(t1 - t0, S, C).contiguous()
            assert torch.equal(sh.to_frames(sh.to_pixels(x_f), S), x_f)
            both = sh.exchange_cfg_halves(torch.full((t1 - t0, 2, 1), float(sh.cfg_half)))
            assert both[:t1 - t0].eq(0).all() and both[t1 - t0:].eq(1).all()
            info[rank] = sh.local_image_ids()
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            groups.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]
    assert info[0] == list(range(0, 7)) and info[4] == list(range(0, 7)) and info[1] == list(range(7, 13))
    assert make_shard(T, 1, 0) is None


@pytest.mark.parametrize("P,T", [(2, 5), (4, 7), (5, 5)])
def test_halo_exchange_thread_ranks(P, T):
    """Neighbour boundary frames: prev = last frame of rank r-1, next = first frame of rank r+1, None at the window ends."""
    B, S, C = 2, 6, 4
    X = torch.randn(B, T, S, C)
    shared = ThreadComm.Shared(P)
    errs = []

    def run(rank):
        try:
            sh = FrameShard(T, ThreadComm(shared, rank), B=B)
            t0, t1 = sh.t_off[rank], sh.t_off[rank + 1]
            prev, nxt = sh.halo_exchange(X[:, t0:t1].reshape(B * (t1 - t0), S, C).contiguous())
            assert (prev is None) == (rank == 0) and (nxt is None) == (rank == P - 1)
            if prev is not None:
                assert torch.equal(prev, X[:, t0 - 1])
            if nxt is not None:
                assert torch.equal(nxt, X[:, t1])
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs[0]


@pytest.mark.parametrize("mode_B", [("frames", 2), ("hybrid", 1)])
def test_exchange_split_lists_are_globally_consistent_at_every_level(mode_B):
    """for every group size 1..8 and the four UNet levels (S = 9216 / 2304 / 576 / 144, most of them NOT divisible by P),
    what rank r expects from rank q (out_splits[q]) is what q sends to r (in_splits[r]), the lists cover the buffers exactly, and the
    halo lists are zero except for existing neighbours."""
    _, B = mode_B
    T = 25
    for P in range(1, 9):
        shards = [FrameShard(T, type("C", (), {"world": P, "rank": r})(), B=B) for r in range(P)]
        for S, C in ((9216, 320), (2304, 640), (576, 1280), (144, 1280)):
            sp = [sh.exchange_splits(S, C) for sh in shards]
            sc = shards[0].pixel_counts(S)
            assert sum(sc) == S and max(sc) - min(sc) <= 1
            for r in range(P):
                for kind in ("to_pixels", "to_frames", "halo"):
                    ins, outs = sp[r][kind]
                    assert len(ins) == P and len(outs) == P
                    for q in range(P):
                        assert outs[q] == sp[q][kind][0][r], (P, S, kind, r, q)
                assert sum(sp[r]["to_pixels"][0]) == B * shards[r].t_local * S * C        # the whole local frame-sharded tensor leaves
                assert sum(sp[r]["to_pixels"][1]) == B * T * sc[r] * C                    # the whole pixel-sharded tensor arrives
                assert sp[r]["to_frames"] == (sp[r]["to_pixels"][1], sp[r]["to_pixels"][0])
                assert [i for i, v in enumerate(sp[r]["halo"][0]) if v] == [q for q in (r - 1, r + 1) if 0 <= q < P]
                # another rank's lists through the `rank` argument
                assert shards[0].exchange_splits(S, C, rank=r) == sp[r]


def _gloo_selfcheck_worker(rank, world, port, mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vista_amd.parallel import make_shard

        def make_group(ranks):
            g = dist.new_group(ranks=ranks)
            return DistComm(g) if rank in ranks else None
        sh = make_shard(25, world, rank, mode=mode, make_group=make_group)
        seen = []
        sh.selfcheck("cpu", log=seen.append)
        assert any("zero-length" in m for m in seen) and any("all_gather" in m for m in seen)
        # a wrong split list is refused by name before it reaches the transport
        from vista_amd.parallel import CollectiveError
        if sh.P > 1:
            with pytest.raises(CollectiveError, match="split lists do not match"):
                sh.comm.all_to_all(torch.empty(4), torch.empty(4), [4] + [0] * (sh.P - 1), [3] + [0] * (sh.P - 1))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(3, "frames"), (4, "hybrid")])
def test_selfcheck_over_gloo_processes(world, mode):
    """Every collective signature the RCCL path uses, as real multi-process collectives: 3 ranks in 'frames' mode (the middle rank's halo
    exchange has zero-length splits on both sides of the list; frames 9/8/8, pixel slices 48/48/48 .. 3072 each), 4 ranks in 'hybrid'
    mode (two 2-rank frame groups + the cfg-pair gathers)."""
    port = 29700 + (os.getpid() % 1500) + world
    mp.spawn(_gloo_selfcheck_worker, args=(world, port, mode), nprocs=world, join=True)
