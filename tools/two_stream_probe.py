"""Probe (round 6): do the two classifier-free-guidance halves of a step run faster as TWO concurrent hipGraphs (N = 25 images each, one
stream each) than as one N = 50 forward?  The halves never interact inside the UNet (guiders.py:27-44 concatenates them on the batch axis,
every kernel is per-image), so the split is exact; what it could buy is chip-level overlap of one half's HBM-bound launches (GroupNorm,
epilogues, temporal attention) with the other half's MFMA-bound ones and back-filling of ragged last rounds, against smaller launches.
    python tools/two_stream_probe.py [--steps 6]
Prints ms per UNet forward pair for: one N=50 graph; two N=25 graphs replayed back to back on ONE stream; the same two on TWO streams;
and checks that the concurrent replays are bitwise the serial ones (separate split-K workspaces)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--model-channels", type=int, default=320)
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--latent-h", type=int, default=72)
    ap.add_argument("--latent-w", type=int, default=128)
    args = ap.parse_args()
    import bench
    from vista_amd import _lib, ops, synth
    from vista_amd.modules.diffusionmodules.video_model import CIN_PAD
    _lib.load()
    T, H, W = args.frames, args.latent_h, args.latent_w
    net = bench.build_model(args.model_channels)
    w = synth.window_inputs(T=T, H=H, W=W, seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    tok = (torch.randn((2 * T, H * W, CIN_PAD), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    tok[..., 8:] = 0
    ctx = torch.cat([w["uc"]["crossattn"], w["c"]["crossattn"]], 0).cuda()
    y = torch.cat([w["uc"]["vector"], w["c"]["vector"]], 0).cuda()
    if ctx.shape[0] != 2 * T:
        ctx, y = ctx.repeat_interleave(T, 0), y.repeat_interleave(T, 0)
    mask = w["cond_mask"].cuda().float()
    mask2 = torch.cat([mask, mask])

    def capture(sl):
        st = {"in": tok[sl].clone(), "ts": torch.full((sl.stop - sl.start,), 0.7, device="cuda"), "ctx": ctx[sl].clone(), "y": y[sl].clone(),
              "mask": mask2[sl].clone()}
        fwd = lambda: net.forward_tokens(st["in"], st["ts"], st["ctx"], st["y"], st["mask"], T, H, W)  # noqa: E731
        ops._GRAPH_TLS.ws.clear()   # a workspace of its own for every graph of this probe: two of them run at the same time
        with ops.graph_workspace():
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fwd()
            torch.cuda.current_stream().wait_stream(side)
            st["ws"] = ops.graph_workspace_tensor()
            st["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["graph"]):
                st["out"] = fwd()
        return st

    full = capture(slice(0, 2 * T))
    h0, h1 = capture(slice(0, T)), capture(slice(T, 2 * T))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()

    def run_full():
        full["graph"].replay()

    def run_serial():
        h0["graph"].replay()
        h1["graph"].replay()

    def run_two(delay_cycles=0):
        cur = torch.cuda.current_stream()
        sa.wait_stream(cur)
        sb.wait_stream(cur)
        with torch.cuda.stream(sa):
            h0["graph"].replay()
        with torch.cuda.stream(sb):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)   # half 1 starts late: do the halves overlap better out of phase?
            h1["graph"].replay()
        cur.wait_stream(sa)
        cur.wait_stream(sb)

    def timeit(fn):
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                fn()
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / args.steps)
        return best * 1e3

    run_full()
    torch.cuda.synchronize()
    ref = full["out"].clone()
    run_serial()
    torch.cuda.synchronize()
    ser = torch.cat([h0["out"], h1["out"]]).clone()
    run_two()
    torch.cuda.synchronize()
    two = torch.cat([h0["out"], h1["out"]]).clone()
    rel = ((ser.float() - ref.float()).norm() / ref.float().norm()).item()
    print(f"halves vs full batch: rel-L2 {rel:.3e} (bitwise {torch.equal(ser, ref)}); concurrent vs serial halves bitwise: {torch.equal(two, ser)}")
    for rnd in range(2):
        print(f"round {rnd}: one N={2 * T} graph {timeit(run_full):8.2f} ms | two N={T} graphs, one stream {timeit(run_serial):8.2f} ms | "
              f"two streams {timeit(run_two):8.2f} ms", flush=True)
    lo, hi = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)   # does a priority difference between the halves change anything?

    def run_prio():
        cur = torch.cuda.current_stream()
        lo.wait_stream(cur)
        hi.wait_stream(cur)
        with torch.cuda.stream(hi):
            h0["graph"].replay()
        with torch.cuda.stream(lo):
            h1["graph"].replay()
        cur.wait_stream(lo)
        cur.wait_stream(hi)
    for rnd in range(2):
        print(f"two streams, equal priority {timeit(run_two):8.2f} ms | half 0 on a high-priority stream {timeit(run_prio):8.2f} ms", flush=True)
    for us in (100, 300, 1000, 3000):   # torch.cuda._sleep counts ~100 MHz ticks on ROCm builds (wall time is what is printed)
        print(f"two streams, half 1 delayed by _sleep({us * 100}): {timeit(lambda: run_two(us * 100)):8.2f} ms", flush=True)


if __name__ == "__main__":
    main()
