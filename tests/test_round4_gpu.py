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
