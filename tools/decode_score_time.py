"""Decoder + signal score at the benchmark's shape (32 x 512 x 3999 frames, win 32 / hop 16): the decoder launch with the
moments epilogue (ps_free_decode_moments_f32) against decoder + ps_wave_moments_f64, HIP events on torch's stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from puresound_amd import hip as H  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    n, c, t, win, hop = 32, 512, 3999, 32, 16
    g = torch.Generator(device="cpu").manual_seed(1)
    feats = H.pad_rows(torch.rand(n, c, t, generator=g).to(dev))
    mask = H.pad_rows(torch.rand(n, c, t, generator=g).to(dev))
    w = (torch.rand(c, 1, win, generator=g) - 0.5).to(dev)
    lout = (t - 1) * hop + win
    ref = torch.rand(n, lout, generator=g).to(dev)
    out = torch.empty(n, lout, device=dev)
    dec = timed(lambda: H.free_decode(feats, t, w, hop, mask, "relu", "linear", out))
    two = timed(lambda: H.wave_moments(H.free_decode(feats, t, w, hop, mask, "relu", "linear", out), ref))
    one = timed(lambda: H.free_decode_moments(feats, t, w, hop, ref, mask, "relu", "linear", out))
    print(f"decoder alone {dec:.1f} us | decoder + ps_wave_moments_f64 {two:.1f} us | decoder with the moments epilogue "
          f"{one:.1f} us (both scored forms include the torch sum over the partial slots)")


if __name__ == "__main__":
    main()
