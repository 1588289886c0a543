"""Round 4 (GPU): the bf16 residual stream of BASELINE config 3 and the register-B fp16x2 GEMM's eligibility rules."""
import numpy as np
import pytest
import torch

import cases
from conftest import rel_max
from detweights import det_state_dict, det_wave
from oracle import separator_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def PA():
    import puresound_amd.nnet as PA
    return PA


@pytest.fixture(scope="module")
def H():
    import puresound_amd.hip as H
    return H


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = np.random.Generator(np.random.Philox(key=seed))
    return torch.tensor(g.uniform(lo, hi, shape), dtype=torch.float32)


def _l2rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("n,k,m,t,flags", [(2, 256, 512, 300, 0), (8, 256, 512, 3999, 0), (8, 256, 512, 3999, 1 << 27),
                                           (3, 64, 96, 700, 0)])
def test_bf16_gemm_with_bf16_residual_and_output_rows(dev, n, k, m, t, flags):
    """ps_conv1x1_bf16_io with y_bf16 and a residual: the residual is bf16 rows too (out_conv of a block whose residual
    stream is stored in bf16), on the persistent kernel (large launches) and the one-tile-per-workgroup kernel (bit 27 /
    small launches).  Reference: fp64 product of the bf16-rounded operands + the bf16 residual, rounded to bf16 once."""
    from puresound_amd import _abi, hip as H
    x, w, b = _rand((n, k, t), 401), _rand((m, k), 402, -0.1, 0.1), _rand((m,), 403)
    res = _rand((n, m, t), 404)
    gamma, beta, slope = _rand((k,), 405, 0.5, 1.5), _rand((k,), 406, -0.2, 0.2), torch.tensor([0.25])
    stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
    keep = (stats, gamma.to(dev), beta.to(dev), slope.to(dev))
    pro = H.make_prologue(_abi.PS_NORM_GLOBAL, True, keep[0], k * t, 1e-8, keep[1], keep[2], keep[3])
    xd = H.pad_rows(x.to(dev)).to(torch.bfloat16)
    rd = H.pad_rows(res.to(dev)).to(torch.bfloat16)
    old = _abi.lib().ps_debug_flags(flags)
    try:
        y, _ = H.conv1x1_bf16(xd, t, H.pack_wt_bf16(w.to(dev), 1), m, pro, b.to(dev), None, rd, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
    finally:
        _abi.lib().ps_debug_flags(old)
    assert y.dtype == torch.bfloat16
    xb = xd[..., :t].float().cpu().double()
    a = O.prelu(O.glob_ln(xb, gamma.double(), beta.double()), slope.double())   # statistics are those of the fp32 x: close
    mean, var = x.double().mean((1, 2), keepdim=True), x.double().var((1, 2), unbiased=False, keepdim=True)
    a = (xb - mean) / torch.sqrt(var + 1e-8) * gamma.double().reshape(1, -1, 1) + beta.double().reshape(1, -1, 1)
    a = torch.where(a >= 0, a, 0.25 * a).to(torch.bfloat16).double()
    ref = torch.matmul(w.to(torch.bfloat16).double(), a) + b.double().reshape(1, -1, 1) + rd[..., :t].float().cpu().double()
    got = y[..., :t].float().cpu().double()
    assert torch.isfinite(got).all()
    assert _l2rel(got.numpy(), ref.numpy()) < 6e-3       # one bf16 rounding of the result: 2^-9 relative, rms ~ 2e-3
    assert rel_max(got.numpy(), ref.numpy()) < 2e-2


@pytest.mark.parametrize("size", ["short", "full"])
def test_config3_with_the_residual_stream_in_bf16(PA, dev, size):
    """BASELINE configs[2] as it names its arithmetic -- bf16 storage, fp32 accumulate -- for EVERY activation row of the
    TCN stacks (hidden maps and residual stream; masker and speaker net): l2-rel <= 3e-2 against the fp32 oracle
    (SURVEY 8d), and the switch `stream_bf16 = False` gives back round 3's fp32 residual stream."""
    name = "cfg3_short"
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    model.masker.set_gemm_precision("bf16")
    tcns = [m for m in model.speaker_net if hasattr(m, "gemm_precision")]
    for m in tcns:
        m.gemm_precision = "bf16"
    b, L = (2, 4000) if size == "short" else (32, 64000)
    noisy, enroll = det_wave(301, b, L), det_wave(302, b, L)
    out = model.inference(noisy.to(dev), enroll.to(dev))
    assert out.shape == (b, L) and torch.isfinite(out).all()
    pick = 1 if size == "short" else 5
    ref = O.inference(noisy[pick:pick + 1], sd, cases.oracle_cfg(name), enroll[pick:pick + 1])
    e_stream = _l2rel(out[pick:pick + 1].cpu().numpy(), ref.numpy())
    assert e_stream < 3e-2
    for m in [mm for st in model.masker.tcn_list for mm in st] + tcns:
        m.stream_bf16 = False
    out2 = model.inference(noisy.to(dev), enroll.to(dev))
    e_hidden = _l2rel(out2[pick:pick + 1].cpu().numpy(), ref.numpy())
    assert e_hidden < 3e-2 and not torch.equal(out, out2)
    print(f"cfg3 {size}: l2-rel bf16 stream {e_stream:.2e}, fp32 residual stream {e_hidden:.2e}")


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "atten"])
def test_mha_self_atten_layer_matches_reference_golden(PA, dev, golden_dir, name):
    """MhaSelfAttenLayer called on its own, incl. improved=True (lobe/attention.py:170-183: an LSTM in place of the first
    feed-forward Linear), which round 3 refused; and the reference's own AttributeError for improved + position_encoding."""
    import os
    c = cases.CASES[name]
    g = dict(np.load(os.path.join(golden_dir, name + ".npz")))
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    y = model(torch.tensor(g["x"]).to(dev), causal=c["causal"])
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < 1e-4
    if c["kw"]["improved"]:
        bad = PA.MhaSelfAttenLayer(*c["args"], improved=True, position_encoding=True).eval().to(dev)
        with pytest.raises(AttributeError):
            bad(torch.tensor(g["x"]).to(dev))


# ------------------------------------------------------------------------------------------------
# frame-major gate pre-activations + the 16-sequence fp16x2 recurrence at H = 128 (DPCRN's bottleneck LSTMs)
# ------------------------------------------------------------------------------------------------
def _rand4(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


def test_conv1x1_f16x2_frame_major_is_the_row_major_result_transposed(H, dev):
    """ps_conv1x1_f16x2_fmajor_f32 writes y [N][ldt][M]: bit for bit ps_conv1x1_f16x2_f32's [N][M][ldt] (same kernel, same
    accumulation order, only the epilogue's addresses differ), pad frames included; M = 512 and 1024, two utterances."""
    from puresound_amd import _abi
    n, k, t = 2, 128, 300
    x = H.pad_rows(_rand4((n, k, t), 401).to(dev))
    ldt = x.shape[-1]
    old = _abi.lib().ps_debug_flags(1 << 28)   # the register-B kernel at any size
    try:
        for m in (512, 1024):
            w = _rand4((m, k), 402 + m, -0.3, 0.3).to(dev)
            b = _rand4((m,), 403, -0.5, 0.5).to(dev)
            wf, we = H.pack_wt_f16x2(w)
            assert H.conv1x1_f16x2_fmajor_ok(n, k, m, t, ldt)
            amax = H.absmax(x, t)
            y_rm, _, _ = H.conv1x1_f16x2(x, t, wf, we, m, None, b, x_amax=amax)
            y_fm = H.conv1x1_f16x2_fmajor(x, t, wf, we, m, b, x_amax=amax)
            torch.cuda.synchronize()
            assert y_fm.shape == (n, ldt, m) and y_fm.stride(1) == H.fmajor_ld(m)
            assert torch.equal(y_fm.transpose(1, 2), y_rm), m
            ref = torch.einsum("mk,nkt->nmt", w.double().cpu(), x[..., :t].double().cpu()) + b.double().cpu()[None, :, None]
            assert rel_max(y_rm[..., :t].cpu().numpy(), ref.numpy()) < 2e-6
    finally:
        _abi.lib().ps_debug_flags(old)
    # a launch that fills the chip: 512 workgroups, the four m-tiles of a frame tile on four workgroups of one XCD
    n2, t2, m2 = 8, 2000, 1024
    x2 = H.pad_rows(_rand4((n2, k, t2), 405).to(dev))
    w2 = _rand4((m2, k), 406, -0.3, 0.3).to(dev)
    wf2, we2 = H.pack_wt_f16x2(w2)
    am2 = H.absmax(x2, t2)
    y_rm, _, _ = H.conv1x1_f16x2(x2, t2, wf2, we2, m2, None, None, x_amax=am2)
    y_fm = H.conv1x1_f16x2_fmajor(x2, t2, wf2, we2, m2, None, x_amax=am2)
    cov = (t2 + 127) // 128 * 128   # (the GEMM writes whole 128-frame tiles up to T; rows are padded to an odd tile count)
    assert torch.equal(y_fm.transpose(1, 2)[..., :cov], y_rm[..., :cov])
    ref = torch.einsum("mk,nkt->nmt", w2.double().cpu(), x2[..., :t2].double().cpu())
    assert rel_max(y_rm[..., :t2].cpu().numpy(), ref.numpy()) < 2e-6
    assert not H.conv1x1_f16x2_fmajor_ok(n, 100, 512, t, ldt)      # K not a multiple of 32
    assert not H.conv1x1_f16x2_fmajor_ok(n, 128, 384, t, ldt)      # M not a multiple of 256


@pytest.mark.parametrize("mode,bi,n,f,t,wscale", [
    ("intra", True, 2, 9, 37, 1.0),      # DPCRN's intra pass: sequences = frames (ragged last block of 16), steps = rows
    ("intra", False, 3, 5, 16, 1.5),     # exactly one block per utterance, saturating recurrent weights
    ("inter", False, 2, 9, 37, 1.0),     # inter pass: sequences = rows, 37 consecutive frames (partial last step group)
    ("inter", True, 1, 20, 24, 1.0),     # consecutive frames in both directions (whole step groups), two blocks
    ("inter", False, 2, 3, 130, 1e-3),   # more steps than the ring, vanishing recurrent weights, second row of frames
])
def test_lstm_frame_major_f16x2_kernel(H, dev, mode, bi, n, f, t, wscale):
    """ps_lstm_fmajor_f16x2_f32 (H = 128, 16 sequences per workgroup, LDS-DMA ring) against the oracle's LSTM and the
    fp32 channel-major kernel on the same pre-activations."""
    import torch.nn as nn
    from oracle import dualpath_oracle as DP
    from puresound_amd.nnet._plans import lstm_plan
    hid, c = 128, 12
    m = nn.LSTM(c, hid, num_layers=1, bidirectional=bi, batch_first=True)
    sd = {k: _rand(tuple(v.shape), 410 + i, -0.4, 0.4) for i, (k, v) in enumerate(m.state_dict().items())}
    m.load_state_dict(sd)
    if wscale != 1.0:
        sd = {kk: (v * wscale if "weight_hh" in kk else v) for kk, v in sd.items()}
        m.load_state_dict(sd)
    d = 2 if bi else 1
    x = _rand4((n, c, f, t), 411)
    if mode == "intra":   # one sequence per frame, steps along the rows
        ref, _ = DP.lstm(x.permute(0, 3, 2, 1).reshape(n * t, f, c), sd, "", bi)
    else:
        ref, _ = DP.lstm(x.permute(0, 2, 3, 1).reshape(n * f, t, c), sd, "", bi)
    p = lstm_plan(m.to(dev), torch.device(dev))
    xp = H.pad_rows(x.reshape(n, c * f, t).to(dev)).view(n, c, f, -1)
    ld = xp.shape[-1]
    frames = (f - 1) * ld + t
    gx, _ = H.conv1x1(xp.view(n, c, f * ld), frames, p["wih"], p["rows"], None, p["bias"])
    gx_fm = torch.zeros(n, f * ld, H.fmajor_ld(d * 4 * hid), device=dev)[..., :d * 4 * hid]   # frames as padded rows
    gx_fm.copy_(gx.transpose(1, 2))
    q, qs, steps, ss = (t, 1, f, ld) if mode == "intra" else (f, ld, t, 1)
    assert H.lstm_fmajor_ok(n, f * ld, hid, d, q, qs, steps, ss)
    base, _ = H.lstm(gx, p["whh_t"], hid, d, q, qs, steps, ss)
    hout = H.lstm_fmajor(gx_fm, p["whh_t"], hid, d, q, qs, steps, ss)
    torch.cuda.synchronize()

    def seqs(h):
        h4 = h.view(n, d * hid, f, ld)[..., :t].cpu()
        return (h4.permute(0, 3, 2, 1).reshape(n * t, f, -1) if mode == "intra" else h4.permute(0, 2, 3, 1).reshape(n * f, t, -1))
    tol = 2e-5 if steps < 64 else 5e-5
    assert rel_max(seqs(base).numpy(), ref.numpy()) < tol
    e = rel_max(seqs(hout).numpy(), ref.numpy())
    assert e < tol, e
    assert rel_max(seqs(hout).numpy(), seqs(base).numpy()) < tol / 2
    if mode == "inter" and t % 4:    # the padded step group of a row is written as zeros
        pad = hout.view(n, d * hid, f, ld)[..., t:(t + 3) // 4 * 4]
        assert float(pad.abs().max()) == 0.0


def test_lstm_frame_major_refuses_what_it_does_not_cover(H, dev):
    gx = torch.zeros(1, 128, 256, device=dev)
    whh = torch.zeros(1, 64, 256, device=dev)
    assert not H.lstm_fmajor_ok(1, 128, 64, 1, 4, 1, 8, 4)
    with pytest.raises(RuntimeError, match="ps_lstm_fmajor_ok"):
        H.lstm_fmajor(gx, whh, 64, 1, 4, 1, 8, 4)


def test_dpcrn_preset_takes_the_frame_major_recurrence(PA, dev):
    """ns_dpcrn_v0_causal in the fp16x2 arithmetic: with and without the frame-major LSTM path the enhanced waveform agrees
    to fp32 class with the exact-fp32 run (the GEMM needs a full-size batch to qualify: 32 x 1 s here)."""
    import cases
    from detweights import det_state_dict
    from puresound_amd.nnet import _plans
    model = cases.build(PA.NS, "ns_dpcrn_short").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision("fp32")
    g = torch.Generator().manual_seed(77)
    x = ((torch.rand(32, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    ref = model.inference(x)
    model.masker.set_gemm_precision("fp16x2")
    outs = {}
    for on in (True, False):
        old = _plans.FMAJOR_LSTM
        _plans.FMAJOR_LSTM = on
        try:
            outs[on] = model.inference(x)
        finally:
            _plans.FMAJOR_LSTM = old
    for on, y in outs.items():
        err = float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref))
        assert err < 2e-5, (on, err)
    assert not torch.equal(outs[True], outs[False])   # (the two paths are different kernels)


@pytest.mark.parametrize("n,k,t,flags", [(2, 128, 300, 1 << 28), (3, 256, 1000, 1 << 28), (16, 256, 2000, 0)])
def test_gemm_with_layernorm_epilogue(H, dev, n, k, t, flags):
    """ps_conv1x1_f16x2_ln_f32: y = res + LayerNorm_128(W x + b) in one launch (the projection behind a recurrence, |x| < 1)
    against float64 and against the two-launch path it replaces (fp16x2 GEMM, then ps_chan_layernorm_f32)."""
    from puresound_amd import _abi
    import torch.nn.functional as F
    c = 128
    x = _rand4((n, k, t), 501, -0.95, 0.95)
    w, b = _rand4((c, k), 502, -0.3, 0.3), _rand4((c,), 503, -0.5, 0.5)
    g, be = _rand4((c,), 504, 0.5, 1.5), _rand4((c,), 505, -0.3, 0.3)
    res = _rand4((n, c, t), 506)
    p = torch.einsum("mk,nkt->nmt", w.double(), x.double()) + b.double()[None, :, None]
    ref = res.double() + F.layer_norm(p.transpose(1, 2), (c,), g.double(), be.double(), 1e-5).transpose(1, 2)
    xp, rp = H.pad_rows(x.to(dev)), H.pad_rows(res.to(dev))
    w256 = torch.zeros(256, k)
    w256[:c] = w
    wf, we = H.pack_wt_f16x2(w256.to(dev))
    wf1, we1 = H.pack_wt_f16x2(w.to(dev))
    old = _abi.lib().ps_debug_flags(flags)
    try:
        assert H.conv1x1_f16x2_ln_ok(n, k, c, t)
        y = H.conv1x1_f16x2_ln(xp, t, wf, we, c, b.to(dev), g.to(dev), be.to(dev), 1e-5, rp, x_bound=1.0)
        y_nores = H.conv1x1_f16x2_ln(xp, t, wf, we, c, None, g.to(dev), be.to(dev), 1e-5, None, x_bound=1.0)
        p2, _, _ = H.conv1x1_f16x2(xp, t, wf1, we1, c, None, b.to(dev), x_bound=1.0)
        y2 = H.chan_layernorm(p2, t, g.to(dev), be.to(dev), 1e-5, res=rp)
        torch.cuda.synchronize()
    finally:
        _abi.lib().ps_debug_flags(old)
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 5e-6
    assert rel_max(y[..., :t].cpu().numpy(), y2[..., :t].cpu().numpy()) < 5e-6
    p0 = torch.einsum("mk,nkt->nmt", w.double(), x.double())
    ref0 = F.layer_norm(p0.transpose(1, 2), (c,), g.double(), be.double(), 1e-5).transpose(1, 2)
    assert rel_max(y_nores[..., :t].cpu().numpy(), ref0.numpy()) < 5e-6
    assert not H.conv1x1_f16x2_ln_ok(n, k, 64, t) and not H.conv1x1_f16x2_ln_ok(n, 100, c, t)


def test_dparn_preset_in_the_fp16x2_arithmetic(PA, dev):
    """ns_dparn_v0_causal with the masker in the fp16x2 arithmetic -- the attention layers' five GEMMs with the LayerNorms as
    epilogues, ranges from LayerNorm bounds and the GEMMs' own maxima, the frame-major recurrence -- against the exact-fp32
    forward of the same model: fp32 class (32 x 1 s: the launches need a full-size batch to qualify)."""
    model = cases.build(PA.NS, "ns_dparn_short").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    g = torch.Generator().manual_seed(78)
    x = ((torch.rand(32, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    model.masker.set_gemm_precision("fp32")
    ref = model.inference(x)
    model.masker.set_gemm_precision("fp16x2")
    y = model.inference(x)
    err = float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref))
    print("ns_dparn fp16x2 vs fp32: l2-rel", err)
    assert err < 2e-5, err
    assert not torch.equal(y, ref)


def test_unfold2d_batches_beyond_the_grid_limit(H, dev):
    """ps_unfold2d_f32 at (utterances x tap rows) > 65535 -- tse_unet_tcn_v0 at the benchmark's batch of 32 has 81,920 and was
    refused before round 4: the launch is split at utterance boundaries, every utterance equal to its own single launch."""
    n, c1, c2, f, t = 60, 100, 28, 3, 5
    kf, kt, sf = 5, 2, 1
    x1, x2 = _rand4((n, c1, f, t), 601), _rand4((n, c2, f, t), 602)
    pad = lambda v: H.pad_rows(v.reshape(v.shape[0], -1, t).to(dev)).view(v.shape[0], v.shape[1], f, -1)  # noqa: E731
    assert n * (c1 + c2) * kf * kt > 65535
    big = H.unfold2d(pad(x1), pad(x2), t, f, kf, kt, sf, 1, 1, kf // 2, kt - 1, False)
    for i in (0, 50, 51, 52, 59):
        one = H.unfold2d(pad(x1[i:i + 1]), pad(x2[i:i + 1]), t, f, kf, kt, sf, 1, 1, kf // 2, kt - 1, False)
        assert torch.equal(big[i:i + 1], one), i
    assert float(big.abs().max()) > 0


@pytest.mark.parametrize("bi,n,s,k,shift,wscale,hid", [
    (True, 2, 9, 8, 0, 1.0, 256),      # 18 sequences (a ragged second block), even segment length: 8-byte h' stores, both directions
    (False, 3, 5, 7, 1, 1.0, 256),     # odd segment length (4-byte stores), MemLSTM's shifted state hand-over, one direction
    (True, 1, 20, 6, 0, 1.5, 256),     # saturating recurrent weights
    (False, 2, 3, 30, 0, 1e-3, 256),   # vanishing recurrent weights, more steps
    (True, 3, 1, 90, 0, 1.0, 192),     # H = 192 (6 waves): one sequence per utterance over 90 frames, both directions
    (False, 2, 4, 9, 0, 1.0, 192),
    (True, 32, 1, 300, 0, 1.0, 192),   # the speaker LSTM of tse_skim_v1 in small: 32 sequences (two groups), 300 steps
])
@pytest.mark.parametrize("coop", [False, True, "agent-scope fences", "scattered", "4-byte stores"])
def test_lstm_h256_streamed_weights_kernel(H, dev, bi, n, s, k, shift, wscale, hid, coop):
    """ps_lstm_fmajor_h256_f16x2_f32 (H = 256: W_hh streamed from its packed image, frame-major pre-activations, initial and
    final states) on SkiM's segment layout (q = segment, k consecutive frames each) against the oracle's LSTM and the generic
    fp32 kernel -- and the same launches on ps_lstm_fmajor_coop_f16x2_f32 (W_hh resident in the registers of H / 32 CUs per
    group of 16 sequences, h' exchanged through L2 every step; its error word must stay 0)."""
    from puresound_amd import _abi
    old_coop = H.COOP_LSTM
    H.COOP_LSTM = bool(coop)
    H._COOP_LAST[0] = None
    # bit 19: the memory model's agent-scope fences at every barrier; bit 18: a cluster's slices on consecutive workgroup ids,
    # i.e. on different XCDs -- the kernel must notice (more than one bit in the cluster's mask) and take those fences itself
    # bit 20: 4-byte h' stores where the launch qualifies for 8-byte ones
    old_flags = _abi.lib().ps_debug_flags({"agent-scope fences": 1 << 19, "scattered": 1 << 18, "4-byte stores": 1 << 20}.get(coop, 0))
    try:
        _lstm_h256_case(H, dev, bi, n, s, k, shift, wscale, hid)
        if coop:
            assert H._COOP_LAST[0] is not None, "the cooperative kernel did not run"
            d, groups = 2 if bi else 1, (n * s + 15) // 16
            assert H.coop_lstm_error_word(d, groups, hid) == 0
            ids = H.coop_lstm_xcd_ids(d, groups, hid)   # the XCD every slice ran on, and per cluster the OR of 1 << XCD
            masks = H.coop_lstm_xcd_masks(d, groups, hid)
            for row, m in zip(ids.tolist(), masks.tolist()):
                assert m == sum(1 << x for x in set(row)), (row, m)
            if coop == "scattered":
                assert all(len(set(row)) > 1 for row in ids.tolist()), ids    # (different XCDs: the heavy barrier was taken)
            elif coop is True or coop == "4-byte stores":
                assert all(len(set(row)) == 1 for row in ids.tolist()), ids   # alone on the chip: one XCD per cluster
        else:
            assert H._COOP_LAST[0] is None
    finally:
        H.COOP_LSTM = old_coop
        _abi.lib().ps_debug_flags(old_flags)


def _lstm_h256_case(H, dev, bi, n, s, k, shift, wscale, hid):
    import torch.nn as nn
    from oracle import dualpath_oracle as DP
    from puresound_amd.nnet._plans import lstm_plan
    c = 12
    m = nn.LSTM(c, hid, num_layers=1, bidirectional=bi, batch_first=True)
    sd = {kk: _rand(tuple(v.shape), 430 + i, -0.25, 0.25) * (wscale if "weight_hh" in kk else 1.0)
          for i, (kk, v) in enumerate(m.state_dict().items())}
    m.load_state_dict(sd)
    d = 2 if bi else 1
    x = _rand4((n, c, s * k), 431)
    seqs = x.transpose(1, 2).reshape(n * s, k, c)
    h0 = _rand4((d, n * s, hid), 432, -0.5, 0.5)
    c0 = _rand4((d, n * s, hid), 433, -0.5, 0.5)
    p = lstm_plan(m.to(dev), torch.device(dev))
    t = s * k
    xp = H.pad_rows(x.to(dev))
    ldt = xp.shape[-1]
    gx, _ = H.conv1x1(xp, t, p["wih"], p["rows"], None, p["bias"])
    gx_fm = gx.transpose(1, 2).contiguous()
    to_state = lambda v: H.pad_rows(v.reshape(d, n, s, hid).permute(1, 0, 3, 2).reshape(n, d * hid, s).to(dev))  # noqa: E731
    back = lambda v: v[..., :s].cpu().reshape(n, d, hid, s).permute(1, 0, 3, 2).reshape(d, n * s, hid)  # noqa: E731
    img, scale = H.pack_whh_h256(p["whh_t"])
    base, (bh, bc) = H.lstm(gx, p["whh_t"], hid, d, s, k, k, 1, to_state(h0), to_state(c0), want_state=True, state_shift=shift)
    hout, (hl, cl) = H.lstm_fmajor_h256(gx_fm, img, scale, d, s, k, k, 1, to_state(h0), to_state(c0), want_state=True,
                                        state_shift=shift)
    torch.cuda.synchronize()
    tol = 2e-5 if k < 20 else (1e-4 if k < 200 else 3e-4)
    e = (rel_max(hout[..., :t].cpu().numpy(), base[..., :t].cpu().numpy()), rel_max(back(hl).numpy(), back(bh).numpy()),
         rel_max(back(cl).numpy(), back(bc).numpy()))
    assert max(e) < tol, e
    if shift == 0:   # (the oracle's LSTM takes per-sequence initial states directly)
        ref, (hn, cn) = DP.lstm(seqs, sd, "", bi, (h0, c0))
        got = hout[..., :t].cpu().transpose(1, 2).reshape(n * s, k, d * hid)
        e = (rel_max(got.numpy(), ref.numpy()), rel_max(back(hl).numpy(), hn.numpy()), rel_max(back(cl).numpy(), cn.numpy()))
        assert max(e) < tol, e


@pytest.mark.parametrize("transposed,m,c1,c2,amp", [(False, 40, 20, 12, 1.0), (True, 100, 30, 34, 1.0), (True, 32, 16, 16, 1.0),
                                                    (False, 64, 40, 0, 300.0), (True, 128, 64, 64, 1e-3)])
def test_conv2d_f16x2_kernel(H, dev, transposed, m, c1, c2, amp):
    """ps_conv2d_f16x2_f32 against float64 Conv2d / ConvTranspose2d (stride 1 in time, 2 / 1 in frequency), two sources, PReLU
    epilogue, statistics for a gLN; activations of very different magnitudes (the kernel finds their range itself) -- and
    against the fp32 implicit GEMM: fp32 class."""
    import torch.nn.functional as F
    n, f, t = 2, 11, 150
    x1 = _rand4((n, c1, f, t), 651) * amp
    x1[:, :, :, 40:44] *= 50.0            # a burst: the scale has to fall in the middle of some waves' K loops
    x2 = _rand4((n, c2, f, t), 652) * amp if c2 else None
    x = torch.cat([x1, x2], 1) if c2 else x1
    kf, kt, sf = 3, 2, (1 if transposed else 2)
    b, slope = _rand4((m,), 653) * amp, torch.tensor([0.2])
    pad = lambda v: H.pad_rows(v.reshape(n, -1, t).to(dev)).view(n, v.shape[1], f, -1)  # noqa: E731
    if not transposed:
        w = _rand4((m, c1 + c2, kf, kt), 654, -0.3, 0.3)
        ref = F.conv2d(F.pad(x.double(), (kt - 1, 0, kf // 2, kf // 2)), w.double(), b.double(), stride=(sf, 1))
        w2, shift = w.reshape(m, -1), kt - 1
    else:
        w = _rand4((c1 + c2, m, kf, kt), 654, -0.3, 0.3)
        op = sf - kf + 2 * (kf // 2)
        ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=(sf, 1), padding=(kf // 2, 0),
                                 output_padding=(op, 0))[..., (kt - 1):]
        w2, shift = w.permute(1, 0, 2, 3).reshape(m, -1), kt - 1
    pre = ref
    ref = torch.where(ref >= 0, ref, 0.2 * ref)
    img, w_exp = H.pack_conv2d_f16x2(w2.contiguous().to(dev))
    args = (m, t, ref.shape[2], kf, kt, sf, 1, 1, kf // 2, shift, transposed)
    y = H.conv2d_f16x2(pad(x1), None if x2 is None else pad(x2), img, w_exp, b.to(dev), *args, "prelu", slope.to(dev))
    y32 = H.conv2d(pad(x1), None if x2 is None else pad(x2), H.pack_wt(w2.contiguous().to(dev)), b.to(dev), *args, "prelu",
                   slope.to(dev))
    yr, stats = H.conv2d_f16x2(pad(x1), None if x2 is None else pad(x2), img, w_exp, b.to(dev), *args, want_stats=True)
    torch.cuda.synchronize()
    e = rel_max(y[..., :t].cpu().numpy(), ref.numpy())
    e32 = rel_max(y32[..., :t].cpu().numpy(), ref.numpy())
    print("conv2d fp16x2", e, "fp32 kernel", e32)
    assert e < 5e-6 and e < 4 * e32 + 1e-6
    assert float(y[..., t:].abs().max()) == 0.0
    assert rel_max(yr[..., :t].cpu().numpy(), pre.numpy()) < 5e-6
    tot = stats.sum(dim=1).cpu()
    assert torch.allclose(tot[:, 0], pre.sum(dim=(1, 2, 3)), rtol=1e-5, atol=1e-3 * float(pre.abs().max()))
    assert torch.allclose(tot[:, 1], (pre ** 2).sum(dim=(1, 2, 3)), rtol=1e-5)


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "single_rnn"])
def test_single_rnn_cells_match_reference_golden(PA, dev, name):
    """SingleRNN(rnn_type = LSTM | GRU | RNN) on its own against the reference's output (ps_rnn_f32 for the GRU / Elman cells)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name + ".npz"))
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    y = model(torch.tensor(g["x"]).to(dev))
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < 1e-4


@pytest.mark.parametrize("intra_skip,inter_skip", [(True, True), (False, True), (True, False), (False, False)])
@pytest.mark.parametrize("kind", ["dpcrn", "dparn"])
def test_dual_path_blocks_with_skip_flags(PA, dev, kind, intra_skip, inter_skip):
    """DPRNNblock2D / DPARNblock2D.forward(x, intra_skip, inter_skip) (dpcrn.py:34-81, dparn.py:55-108) against the oracle's
    restatement with the same flags (the oracle itself is pinned by the golden cases with both skips on)."""
    from oracle import unet_oracle as UO
    n, ch, f, t = 2, 16, 9, 21
    blk = (PA.NS.DPRNNblock2D(input_size=ch, hidden_size=12) if kind == "dpcrn"
           else PA.NS.DPARNblock2D(input_size=ch, hidden_size=12, nhead=2)).eval()
    sd = det_state_dict(blk)
    blk.load_state_dict(sd)
    blk.to(dev)
    x = _rand4((n, ch, f, t), 711)
    sdd = {k: v.double() for k, v in sd.items()}
    ref = (UO.dprnn_block2d(x.double(), sdd, "", intra_skip, inter_skip) if kind == "dpcrn"
           else UO.dparn_block2d(x.double(), sdd, "", 2, intra_skip, inter_skip))
    y = blk(x.to(dev), intra_skip, inter_skip)
    assert rel_max(y.cpu().numpy(), ref.numpy()) < 1e-4


def test_graphed_inference_does_not_freeze_specaugment(PA, dev):
    """tse_skim_v2_causal masks its speaker features at random in eval mode (as the reference's SpecAugment does): a captured
    hipGraph would replay one draw for ever, so GraphedInference runs such a model eagerly -- same seed, same result as
    model.inference; a model without stochastic layers still replays a graph, bit for bit the eager forward."""
    from puresound_amd.graphs import GraphedInference
    g = torch.Generator().manual_seed(9)
    noisy = ((torch.rand(2, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    enroll = ((torch.rand(2, 16000, generator=g) * 2 - 1) * 0.5).to(dev)
    for name, graphs in (("tse_skim_v2_short", 0), ("tse_skim_causal_short", 1)):
        model = cases.build(PA.NS, name).eval()
        model.load_state_dict(det_state_dict(model))
        model.to(dev)
        fast = GraphedInference(model)
        for seed in (0, 1):
            torch.manual_seed(seed)
            ref = model.inference(noisy, enroll)
            torch.manual_seed(seed)
            out = fast(noisy, enroll)
            assert torch.equal(out, ref), (name, seed)
        assert len(fast._graphs) == graphs, name


# ------------------------------------------------------------------------------------------------
# the signal score's moments as the decoder's epilogue (ps_free_decode_moments_f32)
# ------------------------------------------------------------------------------------------------
def _aligned(ref, length):
    """_align_waveform's rule for the reference (base_nn.py:398-412) in numpy."""
    have = ref.shape[-1]
    if have < length:
        return np.concatenate([np.zeros(ref.shape[:-1] + (length - have,), ref.dtype), ref], -1)
    return ref[..., :length]


def _moments_np(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return np.stack([a.sum(-1), b.sum(-1), (a * a).sum(-1), (b * b).sum(-1), (a * b).sum(-1)], -1)


@pytest.mark.parametrize("n,c,t,dref,mask_act,out_mode", [
    (3, 64, 200, 0, "relu", "none"), (3, 64, 200, -37, "sigmoid", "sigmoid"), (2, 32, 1031, 50, "linear", "linear"),
    (5, 512, 3999, -16, "relu", "none"), (2, 16, 64, -1000, "linear", "none"), (2, 24, 200, -5, "relu", "none"),
    (2, 64, 40, 3, "relu", "none")])
def test_decoder_leaves_the_score_moments_behind(H, dev, n, c, t, dref, mask_act, out_mode):
    """The waveform is bit-identical to ps_free_decode_ws_f32's; the moments equal an fp64 numpy pass over that waveform and
    the reference aligned by base_nn.py:398-412 (shorter: left-padded, longer: cut).  The last two shapes (C % 16 != 0, T <
    64) are outside the fused kernel: the wrapper then decodes and calls ps_wave_moments_f64."""
    from puresound_amd import _abi
    win, hop = 32, 16
    lout = (t - 1) * hop + win
    feats, mask = H.pad_rows(_rand((n, c, t), 801).to(dev)), H.pad_rows(_rand((n, c, t), 802).to(dev))
    w = _rand((c, 1, win), 803, -0.2, 0.2).to(dev)
    wide = _rand((n, lout + dref + 9), 804).to(dev)
    ref = wide[:, 5:5 + lout + dref]  # rows of a wider buffer (ldr > ref_len)
    fused = _abi.lib().ps_free_decode_moments_parts(n, c, t, feats.shape[-1], win, hop) > 0
    assert fused == (c % 16 == 0 and t >= 64)
    want = H.free_decode(feats, t, w, hop, mask, mask_act, out_mode)
    out, m = H.free_decode_moments(feats, t, w, hop, ref, mask, mask_act, out_mode)
    assert torch.equal(out, want)
    ref_m = _moments_np(want.cpu().numpy(), _aligned(ref.cpu().numpy(), lout))
    np.testing.assert_allclose(m.cpu().numpy(), ref_m, rtol=1e-11, atol=1e-9)
    if fused:  # the same call again: one writer per slot, so bit-identical moments
        _, m2 = H.free_decode_moments(feats, t, w, hop, ref, mask, mask_act, out_mode)
        assert torch.equal(m, m2)
        with pytest.raises(RuntimeError):
            H.free_decode_moments(feats, t, w, hop, ref[:1], mask, mask_act, out_mode)


@pytest.mark.parametrize("name,dref,lanes", [("cfg2_short", -100, 1), ("cfg2_short", 64, 1), ("cfg3_short", 0, 1),
                                             ("cfg2_short", -100, 2)])
def test_inference_scored_equals_inference_then_the_oracle_score(PA, dev, name, dref, lanes):
    """wrapper.inference_scored = inference() followed by _align_waveform + SDRLoss (oracle/loss_oracle.py restating
    loss/sdr.py:104-183), for a shorter / equal / longer reference, with inactive rows, and over two HIP stream lanes."""
    from oracle import loss_oracle as LO
    from puresound_amd.nnet.loss.sdr import SDRLoss
    c = cases.CASES[name]
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    b = 16 if lanes > 1 else c["B"]
    model.hip_streams = lanes
    noisy = det_wave(c["seed"], b, c["L"]).to(dev)
    enroll = det_wave(c["seed"] + 1, b, c["L_enroll"]).to(dev) if "L_enroll" in c else None
    want = model.inference(noisy, enroll)
    clean = (0.7 * want.cpu() + 0.05 * det_wave(c["seed"] + 2, b, want.shape[-1]))
    clean = clean[:, :want.shape[-1] + dref] if dref <= 0 else torch.nn.functional.pad(clean, (0, dref))
    labels = torch.zeros(b, dtype=torch.bool)
    labels[-1] = True
    for mode in ("sisnr", "sdsdr", "tsdr"):
        enh, score = model.inference_scored(noisy, clean.to(dev), enroll, SDRLoss.init_mode(mode, reduction=False),
                                            labels.to(dev))
        assert torch.equal(enh, want)
        e, r = LO.align_waveform_single(want.cpu(), clean)
        ref_score = LO.sdr_loss(e, r, reduction=False, inactive_labels=labels, **LO.mode_flags(mode))
        np.testing.assert_allclose(score.cpu().numpy(), ref_score.numpy(), atol=2e-3, rtol=0)
    # default loss: the wrapper's loss_func_wav, else SI-SNR with the mean reduction
    _, score = model.inference_scored(noisy, clean.to(dev), enroll)
    e, r = LO.align_waveform_single(want.cpu(), clean)
    np.testing.assert_allclose(float(score), float(LO.sdr_loss(e, r, reduction=True, **LO.mode_flags("sisnr"))), atol=2e-3)


def test_simo_forward_scores_from_the_decoder_launch(PA, dev):
    """SiMoTaskWrapModule.forward on the learned filterbank (win 32, hop 16): the loss comes from the decoder's moments;
    equal to inference() + the oracle's align + SDRLoss; a longer reference fails as in the reference (base_nn.py:885-887)."""
    from oracle import loss_oracle as LO
    c = cases.CASES["simo_free"]
    enc = cases.build_encoder(PA.NS, dict(kind="free", win=32, hop=16, C=32))
    masker = cases.build_simo_masker(PA.NS, dict(c, enc=dict(kind="free", win=32, hop=16, C=32),
                                                 masker=dict(c["masker"], input_dim=32)))
    model = PA.NS.SiMoTaskWrapModule(encoder=enc, masker=masker, verbose=False,
                                     loss_func_wav=PA.NS.SDRLoss.init_mode("sisnr", reduction=False), **c["wrap"]).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    b, heads, length = 3, c["heads"], 3000
    noisy = det_wave(91, b, length).to(dev)
    wav = model.inference(noisy)
    assert wav.shape == (b, heads, (length - 32) // 16 * 16 + 32)
    from puresound_amd import _abi
    assert _abi.lib().ps_free_decode_moments_parts(b * heads, 32, (length - 32) // 16 + 1, 256, 32, 16) > 0
    ref_clean = (0.5 * wav.cpu() + 0.1 * det_wave(92, b * heads, wav.shape[-1]).reshape(wav.shape))[..., :-40]
    labels = torch.zeros(b, heads, dtype=torch.bool)
    labels[1, 0] = True
    got = model(noisy, ref_clean.to(dev), labels.to(dev)).cpu().numpy()
    e, r = LO.align_waveform(wav.cpu(), ref_clean)
    want = LO.sdr_loss(e.reshape(b * heads, -1), r.reshape(b * heads, -1), reduction=False,
                       inactive_labels=labels.reshape(-1), **LO.mode_flags("sisnr"))
    np.testing.assert_allclose(got, want.numpy(), atol=2e-3, rtol=0)
    with pytest.raises(RuntimeError, match="must match the size"):
        model(noisy, torch.nn.functional.pad(ref_clean, (0, 100)).to(dev), labels.to(dev))


# ------------------------------------------------------------------------------------------------
# streaming wavefront: the cells of an anti-diagonal as one launch per kernel (ps_*_cells_f32)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("t", [4, 37, 64])
def test_cells_launches_equal_separate_launches(H, dev, t):
    """ps_film_conv_cells_f32 / ps_lstm_gates_cell_cells_f32 / ps_proj_layernorm_cells_f32 on three cells with different
    weights, inputs and states: bit-identical to three separate ps_film_conv_f32 / ... launches."""
    c, hid, ncell = 12, 8, 3
    dv = lambda shape, seed, lo=-1.0, hi=1.0: _rand(shape, seed, lo, hi).to(dev)  # noqa: E731
    # FiLM
    xs = [H.pad_rows(dv((1, c, t), 900 + i)) for i in range(ncell)]
    wts = [H.pack_wt(dv((2 * c, c), 910 + i, -0.3, 0.3)) for i in range(ncell)]
    ress = [H.pad_rows(dv((1, 2 * c, t), 920 + i)) if i != 1 else None for i in range(ncell)]
    want = [H.film_conv(xs[i], t, wts[i], ress[i]) for i in range(ncell)]
    outs = [torch.zeros_like(xs[0]) for _ in range(ncell)]
    H.film_conv_cells([(xs[i], wts[i], ress[i], outs[i]) for i in range(ncell)], t)
    for a, b in zip(outs, want):
        assert torch.equal(a[..., :t], b[..., :t])
    # gates + cell
    xhs = [H.pad_rows(dv((1, c + hid, t), 930 + i)) for i in range(ncell)]
    ws = [H.pack_wt(dv((4 * hid, c + hid), 940 + i, -0.3, 0.3)) for i in range(ncell)]
    bs = [dv((4 * hid,), 950 + i) for i in range(ncell)]
    c0 = [H.pad_rows(dv((1, hid, t), 960 + i)) for i in range(ncell)]
    c_ref, h_ref = [x.clone() for x in c0], [torch.zeros_like(x) for x in c0]
    for i in range(ncell):
        H.lstm_gates_cell(xhs[i], t, ws[i], bs[i], c_ref[i], h_ref[i], hid)
    c_got, h_got = [x.clone() for x in c0], [torch.zeros_like(x) for x in c0]
    H.lstm_gates_cell_cells([(xhs[i], ws[i], bs[i], c_got[i], h_got[i]) for i in range(ncell)], t, hid)
    for i in range(ncell):
        assert torch.equal(c_got[i][..., :t], c_ref[i][..., :t]) and torch.equal(h_got[i][..., :t], h_ref[i][..., :t])
    # projection + LayerNorm + residual; the last cell without the second norm (the last block of the wavefront)
    for m, k in ((12, 8), (128, 256), (200, 20)):
        hx = [H.pad_rows(dv((1, k, t), 970 + i)) for i in range(ncell)]
        res = [H.pad_rows(dv((1, m, t), 980 + i)) for i in range(ncell)]
        wp = [H.pack_wt(dv((m, k), 990 + i, -0.3, 0.3)) for i in range(ncell)]
        vec = [[dv((m,), 1000 + 10 * i + j, 0.5, 1.5) for j in range(5)] for i in range(ncell)]
        cells, refs = [], []
        for i in range(ncell):
            bp, g1, b1, g2, b2 = vec[i]
            norm2 = (g2, b2, 1e-5) if i + 1 < ncell else None
            cp = torch.zeros_like(hx[i])
            refs.append(H.proj_layernorm(hx[i], t, wp[i], bp, m, g1, b1, 1e-5, res[i], norm2, x_copy=cp) + (cp,))
            cells.append(dict(x=hx[i], wt=wp[i], bias=bp, gamma=g1, beta=b1, eps=1e-5, res=res[i], norm2=norm2,
                              y=torch.zeros_like(res[i]), y2=torch.zeros_like(res[i]), x_copy=torch.zeros_like(hx[i])))
        H.proj_layernorm_cells(cells, t, m)
        for cell, (y, y2, cp) in zip(cells, refs):
            assert torch.equal(cell["y"][..., :t], y[..., :t]) and torch.equal(cell["x_copy"][..., :t], cp[..., :t])
            if y2 is not None:
                assert torch.equal(cell["y2"][..., :t], y2[..., :t])
    with pytest.raises(RuntimeError, match="ncells"):
        H.film_conv_cells([(xs[0], wts[0], None, outs[0])] * 9, t)


def test_wavefront_in_one_launch_per_diagonal_equals_one_branch_per_block(dev):
    """A chunk's wavefront with the cells of an anti-diagonal in one launch per kernel (round 4: a linear chain of launches) and
    with one graph branch per block (round 3): the same bits, across a Mem-LSTM update."""
    from puresound_amd.streaming.demo import DemoTseNet
    net = DemoTseNet().eval()
    net.load_state_dict(det_state_dict(net))
    net.to(dev)
    b, n_chunks = 6, 10
    wav = det_wave(521, b, 320 * n_chunks).to(dev)
    emb = torch.rand(b, 192, generator=torch.Generator().manual_seed(522)).to(dev)
    outs = []
    for batched in (True, False):
        net._batched_cells = batched
        net.init_streams(b, use_graph=True)
        pre = None
        for i in range(n_chunks):
            pre = net.streaming_inference_chunk(wav[:, i * 320:(i + 1) * 320], emb, pre)
        outs.append(pre.clone())
    assert torch.equal(outs[0], outs[1])


# ------------------------------------------------------------------------------------------------
# fp16x2 behind a folded BatchNorm (bN1d blocks): measured maxima mapped through the norm's scale / shift
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,h,t,dil,left", [(3, 40, 3000, 1, 2), (2, 512, 3999, 8, 16), (2, 64, 1500, 128, 256),
                                            (2, 33, 700, 2, 2)])
def test_dwconv_leaves_the_maxima_of_its_output(H, dev, n, h, t, dil, left):
    """ps_dwconv_amax_f32: the same rows as ps_dwconv_f32 and partial maxima whose maximum is max |y| over the valid frames (affine prologue + PReLU in front, as inside a bN1d block)."""
    from puresound_amd import _abi
    x = H.pad_rows(_rand((n, h, t), 1101).to(dev) * 3)
    w, b = _rand((h, 1, 3), 1102).to(dev), _rand((h,), 1103).to(dev)
    keep = (_rand((h,), 1104, 0.2, 2.0).to(dev), _rand((h,), 1105).to(dev), torch.tensor([0.3], device=dev))
    pro = H.make_prologue(_abi.PS_NORM_AFFINE, True, None, 0.0, 0.0, *keep)
    want, _ = H.dwconv(x, t, w, b, dil, left, pro)
    got, amax = H.dwconv(x, t, w, b, dil, left, pro, want_amax=True)
    # (another build of the same kernel: the compiler contracts the tap sum into FMAs differently, one rounding apart)
    assert rel_max(got[..., :t].cpu().numpy(), want[..., :t].cpu().numpy()) < 1e-6
    assert amax.shape == (n, _abi.lib().ps_dwconv_stats_parts(h, t))
    assert torch.equal(amax.max(1).values, got[..., :t].abs().amax((1, 2)))
    with pytest.raises(RuntimeError, match="wave-private"):
        H.dwconv(x, t, _rand((h, 1, 5), 1106).to(dev), b, 1, 2, pro, want_amax=True)


@pytest.mark.parametrize("scale", [1.0, 1e-3, 2e3])
@pytest.mark.parametrize("n,k,m,t", [(2, 64, 48, 300), (8, 512, 256, 3999), (4, 256, 512, 2000)])
def test_fp16x2_gemm_behind_an_affine_norm_takes_mapped_maxima(H, dev, n, k, m, t, scale):
    """ps_conv1x1_f16x2_f32 with a per-channel affine prologue + PReLU and the input's maxima mapped by (max |scale| f,
    max |shift| f), f = max(1, |slope|): as close to fp64 as the exact-fp32 kernel, for inputs of very different size."""
    from puresound_amd import _abi
    x = (_rand((n, k, t), 1111) + 0.1) * scale
    w, b = _rand((m, k), 1112, -0.2, 0.2), _rand((m,), 1113)
    sc, sh = _rand((k,), 1114, -2.0, 2.0) / scale, _rand((k,), 1115, -0.5, 0.5)
    for slope in (0.25, -1.7):
        a = x.double() * sc.double().reshape(1, -1, 1) + sh.double().reshape(1, -1, 1)
        a = torch.where(a >= 0, a, a * slope)
        ref = torch.matmul(w.double(), a) + b.double().reshape(1, -1, 1)
        xd = H.pad_rows(x.to(dev))
        keep = (sc.to(dev), sh.to(dev), torch.tensor([slope], device=dev))
        pro = H.make_prologue(_abi.PS_NORM_AFFINE, True, None, 0.0, 0.0, *keep)
        y32, _ = H.conv1x1(xd, t, H.pack_wt(w.to(dev)), m, pro, b.to(dev))
        wf, we = H.pack_wt_f16x2(w.to(dev))
        f = max(1.0, abs(slope))
        y, _, _ = H.conv1x1_f16x2(xd, t, wf, we, m, pro, b.to(dev), x_amax=H.absmax(xd, t),
                                  amax_map=(float(sc.abs().max()) * f, float(sh.abs().max()) * f))
        rms = float(ref.pow(2).mean().sqrt())
        e32 = float((y32[..., :t].cpu().double() - ref).pow(2).mean().sqrt()) / rms
        err = float((y[..., :t].cpu().double() - ref).pow(2).mean().sqrt()) / rms
        assert torch.isfinite(y[..., :t]).all()
        assert err < 1.5 * e32 + 1e-8, (slope, err, e32)


def test_causal_bn_preset_runs_its_blocks_in_fp16x2(PA, dev, golden_dir):
    """td_tse_conv_tasnet_v0_causal (bN1d TCN blocks): with the default arithmetic every such block plans gemm_planes = 2;
    the result matches the reference's golden vector as before and the exact-fp32 arithmetic to 1e-5."""
    import puresound_amd.nnet.conv_tasnet as CT
    name = "cfg3_causal_short"
    c = cases.CASES[name]
    g = np.load(f"{golden_dir}/{name}.npz")
    noisy = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]).to(dev)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    bn = [m for m in model.masker.modules() if isinstance(m, CT.TCN) and isinstance(m.dconv[0].depthwise[1], torch.nn.BatchNorm1d)]
    assert len(bn) == 24 and all(m.gemm_planes_for_plan() == 2 for m in bn)
    got = model.inference(noisy, enroll)
    model.set_gemm_precision("fp32")
    exact = model.inference(noisy, enroll)
    assert _l2rel(got.cpu().numpy(), exact.cpu().numpy()) < 1e-5
    assert rel_max(got.cpu().numpy(), g["wav"]) < 2e-5


@pytest.mark.parametrize("scale", [1e-3, 1.0, 1e6])
@pytest.mark.parametrize("dilation", [1, 8, 128])
def test_bn_block_in_fp16x2_over_input_scales(PA, dev, scale, dilation):
    """One causal bN1d block with checkpoint-like weights ("wild": BatchNorm gains over 2^-4 .. 2^4, PReLU slopes up to 3,
    weight rows spanning 2^18) on inputs of very different size: the fp16x2 arithmetic (maxima measured by the depthwise /
    pointwise kernels, mapped through the norm) equals the exact-fp32 arithmetic to fp32 rounding.  (A whole STACK of such
    blocks with such weights diverges in every arithmetic -- nothing normalises -- so the preset test uses plain weights.)"""
    import puresound_amd.nnet.conv_tasnet as CT
    blk = CT.TCN(256, 512, 3, dilation, causal=True, tcn_norm="bN1d", dconv_norm="bN1d").eval()
    blk.load_state_dict(det_state_dict(blk, mode="wild"))
    blk.to(dev)
    x = (_rand((3, 256, 1500), 1201) * scale).to(dev)
    outs = {}
    for prec in ("fp32", "fp16x2"):
        blk.gemm_precision = prec
        blk._plan = None
        assert blk.gemm_planes_for_plan() == (2 if prec == "fp16x2" else 0)
        outs[prec] = blk(x)
    assert torch.isfinite(outs["fp16x2"]).all()
    assert rel_max(outs["fp16x2"].cpu().numpy(), outs["fp32"].cpu().numpy()) < 4e-6


def test_cooperative_lstm_gives_up_loudly(H, dev):
    """A group barrier that one slice never reaches (ps_debug_flags bit 17) runs into its bound instead of spinning for
    ever: the launch ends, the error word is set and the result is NaN from the first step on -- and the next launch is fine."""
    import torch.nn as nn
    from puresound_amd import _abi
    from puresound_amd.nnet._plans import lstm_plan
    hid, c, n, k = 256, 12, 4, 6
    m = nn.LSTM(c, hid, num_layers=1, bidirectional=False, batch_first=True)
    m.load_state_dict({kk: _rand(tuple(v.shape), 1300 + i, -0.2, 0.2) for i, (kk, v) in enumerate(m.state_dict().items())})
    p = lstm_plan(m.to(dev), torch.device(dev))
    xp = H.pad_rows(_rand((n, c, k), 1310).to(dev))
    gx, _ = H.conv1x1(xp, k, p["wih"], p["rows"], None, p["bias"])
    gx_fm = gx.transpose(1, 2).contiguous()
    img, scale = H.pack_whh_h256(p["whh_t"])
    assert H.COOP_LSTM
    good, _ = H.lstm_fmajor_h256(gx_fm, img, scale, 1, 1, k, k, 1)
    assert H.coop_lstm_error_word(1, 1, hid) == 0 and torch.isfinite(good[..., :k]).all()
    old = _abi.lib().ps_debug_flags(1 << 17)
    try:
        bad, _ = H.lstm_fmajor_h256(gx_fm, img, scale, 1, 1, k, k, 1)
        torch.cuda.synchronize()
    finally:
        _abi.lib().ps_debug_flags(old)
    assert H.coop_lstm_error_word(1, 1, hid) == 1
    assert torch.isnan(bad[..., :k]).all()
    again, _ = H.lstm_fmajor_h256(gx_fm, img, scale, 1, 1, k, k, 1)
    assert H.coop_lstm_error_word(1, 1, hid) == 0 and torch.equal(again[..., :k], good[..., :k])


@pytest.mark.parametrize("flags,what", [(0, "light barrier, 8-byte stores"), (1 << 20, "light barrier, 4-byte stores"),
                                        (1 << 19, "agent-scope fences")])
@pytest.mark.parametrize("d", [1, 2])
def test_cooperative_lstm_many_groups_back_to_back(H, dev, flags, what, d):
    """SkiM's segment-LSTM launch (864 sequences x 150 steps, H = 256: 54 groups x 4 slices on 216 CUs) eight times back to
    back without a synchronisation in between, every launch against the streamed-weight kernel.  This is the test the light
    barrier's missing vmcnt(0) failed once in a few launches (the counter overtook the h' stores: 5e-6 .. 4e-2 off)."""
    from puresound_amd import _abi
    # d = 2 (the non-causal presets: 108 clusters do not fit the chip at once): one launch per direction, 50 steps here
    n, q, steps, hid = 32, 27, (150 if d == 1 else 50), 256
    t = q * steps
    g = torch.Generator().manual_seed(3)
    gx = (torch.rand(n, H.padded_frames(t), d * 4 * hid, generator=g) - 0.5).to(dev)
    whh = ((torch.rand(d, hid, 4 * hid, generator=g) - 0.5) * 0.2).to(dev)
    img, scale = H.pack_whh_h256(whh)
    old_coop = H.COOP_LSTM
    try:
        H.COOP_LSTM = False
        ref, _ = H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)
        H.COOP_LSTM = True
        old = _abi.lib().ps_debug_flags(flags)
        try:
            outs = [H.lstm_fmajor_h256(gx, img, scale, d, q, steps, steps, 1)[0] for _ in range(8)]
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        groups = (n * q + 15) // 16
        assert H.coop_lstm_error_word(d, groups, hid) == 0
        diffs = [float((o[..., :t] - ref[..., :t]).abs().max()) for o in outs]
        assert max(diffs) < 2e-6, (what, diffs)
        assert all(torch.equal(outs[0][..., :t], o[..., :t]) for o in outs[1:]), what   # (frames past t are never written)
    finally:
        H.COOP_LSTM = old_coop


def test_two_lanes_keep_off_the_cooperative_lstm(PA, H, dev):
    """hip_streams = 2 on a SkiM preset: two lanes must not launch the cooperative LSTM side by side (their workgroups would
    wait for CUs the other lane's workgroups hold): the lanes run the streamed kernel, the result is that of one lane."""
    name = "tse_skim_v0_short"
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    noisy = det_wave(77, 16, 16000).to(dev)
    enroll = det_wave(78, 16, 16000).to(dev)
    assert H.COOP_LSTM
    one = model.inference(noisy, enroll)
    H._COOP_LAST[0] = None
    model.hip_streams = 2
    two = model.inference(noisy, enroll)
    assert H._COOP_LAST[0] is None and H.COOP_LSTM     # no cooperative launch inside the lanes; the switch is back on
    assert torch.isfinite(two).all()
    assert rel_max(two.cpu().numpy(), one.cpu().numpy()) < 2e-5
