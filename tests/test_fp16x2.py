"""The "fp16x2" arithmetic of the 1x1-conv GEMMs (ps_conv1x1_f16x2_f32): two fp16 terms per operand, three products,
operands brought into fp16's range by powers of two.  The claims under test: its result is as close to an fp64 product
as the exact-fp32 MFMA kernel's at any input scale once the caller hands over the range (a bound behind a norm, the
producer's maxima for raw rows); it fails loudly (inf / NaN), not silently, beyond its range; the partial maxima it
leaves for the next consumer are exact; through the model it meets the reference's golden vector like the fp32 paths."""
import os

import numpy as np
import pytest
import torch

import cases
from conftest import rel_max
from detweights import det_state_dict, det_wave
from oracle import separator_oracle as O


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(*shape, generator=g) * (hi - lo) + lo


# ---- host side (no GPU) -----------------------------------------------------------------------------------------------
def test_weight_packer_two_fp16_planes_reconstruct_the_weight():
    from puresound_amd import hip as H
    for scale in (1.0, 3e-5, 700.0):
        w = _rand((300, 40), 5) * scale
        img, w_exp = H.pack_wt_f16x2(w)
        assert img.dtype == torch.float16 and tuple(img.shape) == (2, 3, 2, 256, 16)
        assert 2.0 ** 13 <= float(w.abs().max()) * 2.0 ** w_exp < 2.0 ** 14
        planes = img.float().permute(2, 0, 3, 1, 4).reshape(2, 512, 48)  # [plane][row][k]
        back = (planes[0] + planes[1])[:300, :40] * 2.0 ** -w_exp
        # two 11-bit terms: 2^-22 relative, or the subnormal step 2^-24 / 2^w_exp for the smallest entries
        err = (back - w).abs()
        assert float((err - w.abs() * 2.0 ** -22).max()) <= 2.0 ** -24 * 2.0 ** -w_exp
        assert float(planes[:, 300:].abs().max()) == 0.0 and float(planes[:, :, 40:].abs().max()) == 0.0
    img, w_exp = H.pack_wt_f16x2(torch.zeros(4, 4))
    assert w_exp == 0 and float(img.abs().max()) == 0.0
    with pytest.raises(ValueError):
        H.pack_wt_f16x2(torch.full((2, 2), float("inf")))


def test_fp16x2_blocks_need_a_range_for_their_normalised_values():
    import puresound_amd.nnet.conv_tasnet as CT
    assert CT.GEMM_PLANES["fp16x2"] == 2
    blk = CT.TCN(16, 8, 3, 1, causal=True, tcn_norm="cLN", dconv_norm="cLN").eval()
    blk.gemm_precision = "fp16x2"
    assert blk.gemm_planes_for_plan() == 3  # per-frame norms, run stage by stage -> the three-plane bf16 split
    blk = CT.TCN(16, 8, 3, 1, causal=True, tcn_norm="bN1d", dconv_norm="bN1d").eval()
    blk.gemm_precision = "fp16x2"
    assert blk.gemm_planes_for_plan() == 2  # folded BatchNorm: the producers' measured maxima through its scale / shift
    blk = CT.TCN(16, 8, 3, 1).eval()
    blk.gemm_precision = "fp16x2"
    assert blk.gemm_planes_for_plan() == 2  # global norms: a bound on the normalised values


# ---- device ------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    return torch.device("cuda:0")


def _case(n, k, m, t, mode, scale, dev):
    from puresound_amd import _abi, hip as H
    x = (_rand((n, k, t), 121) + 0.2) * scale
    w, b = _rand((m, k), 122, -0.2, 0.2), _rand((m,), 123)
    gamma, beta, slope = _rand((k,), 124, 0.5, 1.5), _rand((k,), 125, -0.2, 0.2), torch.tensor([0.2])
    a, pro, keep, kw = x.double(), None, None, {}
    xd = H.pad_rows(x.to(dev))
    if mode.startswith("norm"):
        a = O.prelu(O.glob_ln(a, gamma.double(), beta.double()), slope.double())
        stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
        keep = (stats, gamma.to(dev), beta.to(dev), slope.to(dev))
        pro = H.make_prologue(_abi.PS_NORM_GLOBAL, True, keep[0], k * t, 1e-8, keep[1], keep[2], keep[3])
        kw = dict(x_bound=float(gamma.abs().max()) * (k * t) ** 0.5 + float(beta.abs().max()))
    else:
        kw = dict(x_amax=H.absmax(xd, t))
    ref = torch.matmul(w.double(), a) + b.double().reshape(1, -1, 1)
    res = _rand((n, m, t), 126) if mode == "norm_res" else None
    if res is not None:
        ref = ref + res.double()
    return xd, w, b, pro, keep, kw, res, ref


@pytest.mark.gpu
@pytest.mark.parametrize("scale", [1.0, 1e-4, 3e3])
@pytest.mark.parametrize("n,k,m,t,mode", [(2, 24, 12, 77, "plain"), (2, 256, 256, 500, "norm_stats"),
                                          (1, 256, 512, 300, "norm_res"), (2, 512, 256, 129, "stats"),
                                          (9, 64, 256, 3800, "norm_stats"), (5, 48, 512, 3700, "norm_res"),
                                          (9, 40, 200, 3800, "plain"),
                                          (8, 512, 256, 3999, "stats"), (8, 256, 512, 3999, "norm_res"),
                                          (8, 256, 256, 3999, "norm_stats")])
def test_fp16x2_gemm_is_as_close_to_fp64_as_the_fp32_kernel(dev, n, k, m, t, mode, scale):
    from puresound_amd import _abi, hip as H
    xd, w, b, pro, keep, kw, res, ref = _case(n, k, m, t, mode, scale, dev)
    want = mode in ("norm_stats", "stats")
    resd = None if res is None else H.pad_rows(res.to(dev))
    y32, _ = H.conv1x1(xd, t, H.pack_wt(w.to(dev)), m, pro, b.to(dev), None, resd, want_stats=want)
    rms = float(ref.pow(2).mean().sqrt())
    e32 = float((y32[..., :t].cpu().double() - ref).pow(2).mean().sqrt()) / rms
    wf, we = H.pack_wt_f16x2(w.to(dev))
    first = None
    # 0 = the persistent interleaved kernel where the launch is large enough, bit 27 = the one-tile-per-workgroup
    # kernel (256 x 32 tiles for small grids), bits 27 | 29 = its 256 x 128 tile
    # bit 7 = the one-wave-per-SIMD persistent kernel (conv1x1_f16x2_w1.inc) instead of the interleaved one, bit 28 = a
    # persistent kernel at any launch size (where the shape allows one)
    # bit 22 = the interleaved kernel where the default is the register-B kernel (conv1x1_f16x2_rb.inc: K % 32 == 0, M % 256 == 0)
    for flags in (0, 1 << 27, (1 << 27) | (1 << 29), 128, 1 << 28, 128 | (1 << 28), 1 << 22, (1 << 22) | (1 << 28)):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            y, st, am = H.conv1x1_f16x2(xd, t, wf, we, m, pro, b.to(dev), None, resd, want_stats=want, want_amax=True,
                                        **kw)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        got = y[..., :t].cpu().double()
        assert torch.isfinite(got).all()
        err = float((got - ref).pow(2).mean().sqrt()) / rms
        assert err < 1.5 * e32 + 1e-8, (flags, err, e32)  # rms error: the fp32 kernel's (accumulation rounding) class
        assert rel_max(got.numpy(), ref.numpy()) < 4e-6, flags
        if first is None:
            first = got
        else:
            assert rel_max(got.numpy(), first.numpy()) < 2e-6, flags
        # the maxima handed to the next consumer are exact
        assert torch.equal(am.amax(1).cpu(), y[..., :t].abs().amax((1, 2)).cpu()), flags
        if want:
            s = st.sum(1).cpu().numpy()
            np.testing.assert_allclose(s[:, 0], got.sum((1, 2)).numpy(), rtol=1e-6, atol=1e-3 * max(scale, 1.0))
            np.testing.assert_allclose(s[:, 1], (got * got).sum((1, 2)).numpy(), rtol=1e-5)


@pytest.mark.gpu
def test_fp16x2_default_range_and_its_loud_failure(dev):
    """Without a range from the caller the kernel scales raw rows by 2^-4: fine around unit scale, coarser for tiny
    inputs (absolute resolution 2^-21), and inf / NaN -- never a silently wrong number -- beyond 1e6."""
    from puresound_amd import hip as H
    n, k, m, t = 2, 64, 32, 300
    w, b = _rand((m, k), 1, -0.2, 0.2).to(dev), _rand((m,), 2).to(dev)
    wf, we = H.pack_wt_f16x2(w)
    for scale, tol in ((1.0, 2e-6), (1e-3, 2e-3)):
        x = _rand((n, k, t), 3) * scale
        ref = torch.matmul(w.cpu().double(), x.double())
        y, _, _ = H.conv1x1_f16x2(H.pad_rows(x.to(dev)), t, wf, we, m, None, None)
        assert rel_max(y[..., :t].cpu().double().numpy(), ref.numpy()) < tol
        ya, _, _ = H.conv1x1_f16x2(H.pad_rows(x.to(dev)), t, wf, we, m, None, None, x_amax=H.absmax(H.pad_rows(x.to(dev)), t))
        assert rel_max(ya[..., :t].cpu().double().numpy(), ref.numpy()) < 2e-6
    x = _rand((n, k, t), 4) * 1e7
    y, _, _ = H.conv1x1_f16x2(H.pad_rows(x.to(dev)), t, wf, we, m, None, b)
    assert not torch.isfinite(y[..., :t]).all()
    ya, _, _ = H.conv1x1_f16x2(H.pad_rows(x.to(dev)), t, wf, we, m, None, b, x_amax=H.absmax(H.pad_rows(x.to(dev)), t))
    ref = torch.matmul(w.cpu().double(), x.double()) + b.cpu().double().reshape(1, -1, 1)
    assert rel_max(ya[..., :t].cpu().double().numpy(), ref.numpy()) < 2e-6
    # an all-zero utterance next to a loud one: per-utterance ranges
    x = _rand((n, k, t), 5)
    x[0] = 0.0
    xd = H.pad_rows(x.to(dev))
    ya, _, am = H.conv1x1_f16x2(xd, t, wf, we, m, None, b, x_amax=H.absmax(xd, t), want_amax=True)
    assert torch.equal(ya[0, :, :t].cpu(), b.cpu().reshape(-1, 1).expand(m, t))
    assert float(H.absmax(xd, t)[0].max()) == 0.0


@pytest.mark.gpu
def test_fp16x2_meets_the_reference_like_the_fp32_paths(dev, golden_dir):
    import puresound_amd.nnet as PA
    name = "cfg2_full"
    c = cases.CASES[name]
    g = dict(np.load(os.path.join(golden_dir, f"{name}.npz")))
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    x = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    errs = {}
    for prec in ("fp32", "fp16x2"):
        model.masker.set_gemm_precision(prec)
        feats, t = model.encoder.encode_padded(x)
        mask = model.masker.forward_padded(feats, t)
        pre = model.encoder.decode_padded(feats, t, mask, "relu", "none").cpu().numpy()
        errs[prec] = rel_max(pre, g["wav_preclamp"])
        assert rel_max(model.inference(x).cpu().numpy(), g["wav"]) < 1e-4
    assert errs["fp16x2"] < 1e-5 and errs["fp16x2"] < 2 * errs["fp32"], errs
    # the benchmark's batch (the persistent kernels, the range chain across 24 blocks): against the exact-fp32 run
    xb = det_wave(7, 32, 64000).to(dev)
    model.masker.set_gemm_precision("fp32")
    ref = model.inference(xb)
    model.masker.set_gemm_precision("fp16x2")
    y = model.inference(xb)
    assert torch.isfinite(y).all()
    assert float((y - ref).abs().max()) < 5e-5 and float((y - ref).norm() / ref.norm()) < 1e-5
    # a row does not depend on its neighbours: every range is per utterance
    y1 = model.inference(xb[5:6])
    assert float((y1 - y[5:6]).abs().max()) < 2e-5
    # an utterance 1e3 times quieter goes through the same arithmetic at its own range
    yq = model.inference(xb[:2] * 1e-3)
    model.masker.set_gemm_precision("fp32")
    rq = model.inference(xb[:2] * 1e-3)
    assert float((yq - rq).norm() / rq.norm()) < 1e-5


@pytest.mark.gpu
def test_fp16x2_inside_a_captured_graph(dev):
    import puresound_amd.nnet as PA
    from puresound_amd.graphs import GraphedInference
    model = cases.build(PA.NS, "cfg2_full").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision("fp16x2")
    x = det_wave(3, 1, 16000).to(dev)
    eager = model.inference(x)
    graphed = GraphedInference(model)
    assert torch.equal(graphed(x), eager)
    x2 = det_wave(4, 1, 16000).to(dev) * 0.01  # another range through the same graph
    assert torch.equal(graphed(x2), model.inference(x2))


@pytest.mark.gpu
def test_input_range_may_be_a_loose_bound(dev):
    """ps_conv_tasnet_ranged_f32: the range of the first block's input may be any upper bound (the wrapper passes
    max |wav| x the largest row sum of the encoder's |w|); a bound 16 x too loose costs four bits of headroom, not
    accuracy, and the measured maxima (no range given: one ps_absmax_f32 pass) give the same result to rounding."""
    import puresound_amd.nnet as PA
    from puresound_amd import hip as H
    model = cases.build(PA.NS, "cfg2_full").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    x = det_wave(9, 3, 16000).to(dev)
    x[1] *= 1e-3
    feats, t = model.encoder.encode_padded(x)
    bound = model.encoder.feature_bound(x)
    true_max = feats[..., :t].abs().amax((1, 2))
    assert bound.shape[0] == 3 and bound.dim() == 2   # [N, parts] partial bounds
    assert (bound.amax(1) >= true_max).all() and (bound.amax(1) < 64 * true_max).all()
    model.masker.set_gemm_precision("fp32")
    ref = model.masker.forward_padded(feats, t)[..., :t]
    model.masker.set_gemm_precision("fp16x2")
    outs = [model.masker.forward_padded(feats, t, x_amax=r)[..., :t] for r in (None, bound, bound * 16.0, H.absmax(feats, t))]
    for o in outs:
        assert float((o - ref).norm() / ref.norm()) < 5e-6
    with pytest.raises(ValueError):
        model.masker.forward_padded(feats, t, x_amax=bound[:2])


@pytest.mark.gpu
def test_fp16x2_on_degenerate_inputs(dev):
    """Silence, near-silence, an impulse, a full-scale square wave and a batch mixing them: the ranges are per utterance
    and an all-zero utterance must not turn into 0 * inf; against the exact-fp32 path."""
    import puresound_amd.nnet as PA
    model = cases.build(PA.NS, "cfg2_full").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    n, length = 5, 8000
    x = torch.zeros(n, length)
    x[1] = det_wave(21, 1, length)[0] * 1e-7
    x[2, 4000] = 0.9
    x[3] = torch.sign(torch.sin(torch.arange(length) * 0.05)) * 0.99
    x[4] = det_wave(22, 1, length)[0]
    x = x.to(dev)
    model.masker.set_gemm_precision("fp32")
    ref = model.inference(x)
    model.masker.set_gemm_precision("fp16x2")
    y = model.inference(x)
    assert torch.isfinite(y).all()
    for i in range(n):
        scale = float(ref[i].abs().max())
        err = float((y[i] - ref[i]).abs().max())
        assert err <= 2e-5 * max(scale, 1e-30) + 1e-12, (i, err, scale)
