"""Times the temporal VAE decode of one 25-frame 576x1024 window (BASELINE configs 1/4: `decode_first_stage`, 14 frames at a
time with 3 overlapping) on cuda:0 with the shipped decoder_config and seeded random weights.

    python tools/vae_bench.py [--frames 25] [--latent-h 72] [--latent-w 128] [--reps 3]
Prints one JSON line: seconds per window decode, frames/s, and the conv-FLOP rate.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vista_amd import synth  # noqa: E402
from vista_amd.models.diffusion import decode_first_stage  # noqa: E402
from vista_amd.modules.autoencoding.temporal_ae import VideoDecoder  # noqa: E402

SHIPPED = dict(attn_type="vanilla", double_z=True, z_channels=4, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=[1, 2, 4, 4],
               num_res_blocks=2, attn_resolutions=[], dropout=0.0, video_kernel_size=[3, 1, 1])  # configs/inference/vista.yaml:170-184


def decoder_flops(dec, frames, h, w):
    """2*M*N*K over every conv / projection of one `frames`-frame clip (attention matmuls included)."""
    total, H, W = 0.0, h, w
    def conv(cin, cout, k, hh, ww):
        return 2.0 * frames * hh * ww * cout * cin * k
    def vres(cin, cout, hh, ww):
        f = conv(cin, cout, 9, hh, ww) + conv(cout, cout, 9, hh, ww) + 2 * conv(cout, cout, 3, hh, ww)
        return f + (conv(cin, cout, 1, hh, ww) if cin != cout else 0.0)
    ch, mult = dec.ch, [1, 2, 4, 4]
    c = ch * mult[-1]
    total += conv(4, c, 9, H, W) + 2 * vres(c, c, H, W) + 4 * conv(c, c, 1, H, W) + frames * 4.0 * (H * W) ** 2 * c
    for lvl in reversed(range(4)):
        co = ch * mult[lvl]
        for _ in range(3):
            total += vres(c, co, H, W)
            c = co
        if lvl:
            H, W = 2 * H, 2 * W
            total += conv(c, c, 9, H, W)
    return total + conv(c, 3, 9, H, W) + conv(3, 3, 3, H, W)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=25)
    ap.add_argument("--latent-h", type=int, default=72)
    ap.add_argument("--latent-w", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    torch.cuda.set_device(0)
    dec = VideoDecoder(**SHIPPED)
    shapes = {k: tuple(v.shape) for k, v in dec.state_dict().items()}
    dec.load_state_dict(synth.seeded_state_dict(shapes, 0))
    dec.cuda()
    z = (synth.seeded_tensor("vae.zb", (a.frames, 4, a.latent_h, a.latent_w), 1) * 0.18215).cuda()
    out = decode_first_stage(dec, z)  # warm-up: packs weights
    torch.cuda.synchronize()
    assert out.shape == (a.frames, 3, 8 * a.latent_h, 8 * a.latent_w) and torch.isfinite(out).all()
    ts = []
    for _ in range(a.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        decode_first_stage(dec, z)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    best = min(ts)
    clips, n, ov = [], 14, 3
    rest = a.frames - ov
    while rest > 0:
        clips.append(min(n - ov, rest) + ov)
        rest -= n - ov
    fl = sum(decoder_flops(dec, c, a.latent_h, a.latent_w) for c in clips)
    print(json.dumps({"metric": "vae_decode_window_seconds", "value": round(best, 4), "frames": a.frames, "clips": clips,
                      "frames_per_s": round(a.frames / best, 2), "tflops": round(fl / best / 1e12, 1), "flops_per_window": fl,
                      "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2), "all_s": [round(t, 4) for t in ts]}))


if __name__ == "__main__":
    main()
