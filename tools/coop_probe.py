"""Per-step time and barrier mode of the cooperative LSTM on a SkiM-shaped launch (GPU box):
  python tools/coop_probe.py [N Q steps H D]     default 32 27 150 256 1 (the segment LSTM of tse_skim_v2_causal)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from puresound_amd import hip as H  # noqa: E402


def main():
    n, q, steps, hid, d = (int(v) for v in (sys.argv[1:6] if len(sys.argv) >= 6 else (32, 27, 150, 256, 1)))
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    t = q * steps
    ldt = H.padded_frames(t)
    gx = (torch.rand(n, ldt, d * 4 * hid, generator=g) - 0.5).to(dev)
    whh = ((torch.rand(d, hid, 4 * hid, generator=g) - 0.5) * 0.2).to(dev)
    img, scale = H.pack_whh_h256(whh)
    for coop in (False, True):
        H.COOP_LSTM = coop
        H._COOP_LAST[0] = None
        for _ in range(2):
            out, _ = H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            out, _ = H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        line = f"{'cooperative' if coop else 'streamed   '}: {ms:.3f} ms per launch = {ms / steps * 1e3:.2f} us per step"
        if coop and H._COOP_LAST[0] is not None:
            groups = (n * q + 15) // 16
            masks = H.coop_lstm_xcd_masks(d, groups, hid).tolist()
            one = sum(1 for m in masks if bin(m).count("1") == 1)
            line += (f"; {groups * d} clusters x {H._coop_slices(d, groups, hid)} slices, {one} of them behind one L2 (light barrier), "
                     f"error word {H.coop_lstm_error_word(d, groups, hid)}")
        print(line, flush=True)
        ref = out if not coop else ref
        if coop:
            print("max |coop - streamed| =", float((out[..., :t] - ref[..., :t]).abs().max()))
    # where a step's time goes: the same launch with parts removed (ps_debug_flags bits 24..27; results are wrong by design)
    from puresound_amd import _abi
    H.COOP_LSTM = True
    for bits, what in ((1, "no h' output stores"), (2, "no MFMAs"), (4, "no exchange loads"), (8, "no barriers"), (15, "none of them")):
        old = _abi.lib().ps_debug_flags(bits << 24)
        try:
            for _ in range(2):
                H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)
            b.record()
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        print(f"  {what:22s} {a.elapsed_time(b) / 5 / steps * 1e3:6.2f} us per step", flush=True)


if __name__ == "__main__":
    main()
