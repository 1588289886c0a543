"""GPU parity tests: every call goes through the C ABI (libpuresound_hip.so) and is compared with the
CPU oracle on the same seeded inputs and with the committed golden vectors of the imported reference.

Tolerance (north star): max|a-b| / max|b| <= 1e-4 in fp32.
"""
import os

import numpy as np
import pytest
import torch

import cases
from conftest import rel_max
from detweights import det_state_dict, det_wave
from oracle import separator_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def H():
    from puresound_amd import hip
    hip.lib()  # fail loudly if the extension is missing
    return hip


@pytest.fixture(scope="module")
def PA():
    import puresound_amd.nnet as PA
    return PA


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


@pytest.fixture(autouse=True)
def _nan_in_the_allocator_cache(dev):
    """Every GPU test starts with NaN-filled blocks in torch's caching allocator, so that `torch.empty` scratch and
    pad frames hold NaN rather than the zeros of a fresh process: a kernel that lets uninitialised memory reach a
    result fails its parity check instead of passing by luck."""
    junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(16)]  # 16 x 16 MiB
    junk += [torch.full((n,), float("nan"), device=dev) for n in (1 << 10, 1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20)]
    del junk
    yield


def _rand(shape, seed, lo=-1.0, hi=1.0):
    g = np.random.Generator(np.random.Philox(key=seed))
    return torch.tensor(g.uniform(lo, hi, shape), dtype=torch.float32)


# ------------------------------------------------------------------------------------------------
# single kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,length,c,win,hop,relu", [
    (2, 1000, 40, 32, 16, False), (3, 4111, 70, 32, 16, True), (2, 600, 24, 16, 8, False),
    (2, 211, 9, 20, 6, True), (1, 32, 5, 32, 16, False), (1, 64000, 512, 32, 16, False),
    # enough frames for the four-frames-per-thread kernel: T = 4001 (a one-frame tail group), T = 4098 (two), odd C
    (5, 64032, 70, 32, 16, True), (4, 65584, 33, 32, 16, False), (5, 64036, 64, 32, 16, True),
    # tile edges of the matrix-pipe kernel (C % 32 == 0): T = 64, 96, 97, 127, 128 and a ragged last workgroup
    (2, 1040, 32, 32, 16, False), (2, 1552, 64, 32, 16, True), (1, 1568, 128, 32, 16, False), (3, 2048, 96, 32, 16, True),
    (2, 2064, 512, 32, 16, False), (2, 9000, 128, 32, 16, True)])
def test_free_encode(H, dev, n, length, c, win, hop, relu):
    wav = _rand((n, length), 1, -0.5, 0.5)
    w = _rand((c, 1, win), 2, -0.2, 0.2)
    ref = O.free_encode(wav, w, hop, relu)
    from puresound_amd import _abi
    for flags in (0, 1):  # 0: the matrix-pipe kernel where the shape allows it (32 / 16, C % 32 == 0, T >= 64); bit 0: VALU kernel
        old = _abi.lib().ps_debug_flags(flags)
        try:
            feats, t = H.free_encode(wav.to(dev), w.to(dev), hop, relu)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert t == ref.shape[-1] == (length - win) // hop + 1
        assert rel_max(feats[..., :t].cpu().numpy(), ref.numpy()) < 1e-5, flags


@pytest.mark.parametrize("n,c,t,win,hop,mask_act,out_mode", [
    (2, 40, 61, 32, 16, "relu", "linear"), (2, 70, 700, 32, 16, "linear", "none"),
    (3, 24, 300, 16, 8, "sigmoid", "sigmoid"), (2, 9, 33, 20, 6, "relu", "linear"),
    (1, 8, 1, 32, 16, "linear", "none"), (1, 16, 255, 32, 16, "linear", "none"),
    (1, 16, 256, 32, 16, "linear", "none"), (1, 512, 3999, 32, 16, "relu", "linear"),
    # tile edges of the matrix-pipe kernel: T a multiple of 32, one more, one less, an odd channel count
    (2, 64, 64, 32, 16, "relu", "linear"), (2, 33, 96, 32, 16, "linear", "none"), (3, 128, 97, 32, 16, "relu", "linear"),
    (1, 48, 127, 32, 16, "sigmoid", "sigmoid"), (2, 128, 4000, 32, 16, "relu", "linear"),
    # channel-split workgroups (C >= 64): channel counts that do not divide by four, the 16 / 8 filterbank
    (2, 66, 513, 32, 16, "relu", "linear"), (2, 127, 300, 16, 8, "sigmoid", "none")])
def test_free_decode(H, dev, n, c, t, win, hop, mask_act, out_mode):
    feats = _rand((n, c, t), 3)
    mask = _rand((n, c, t), 4)
    w = _rand((c, 1, win), 5, -0.3, 0.3)
    enh = feats * O.get_mask(mask, mask_act)
    ref = O.free_decode(enh, w, hop)
    if out_mode != "none":
        ref = O.output_constrain(ref, out_mode)
    from puresound_amd import _abi
    for flags in (0, 1):  # 0: the matrix-pipe kernel + boundary fix-up where the shape allows it (32 / 16, T >= 64); bit 0: VALU
        old = _abi.lib().ps_debug_flags(flags)
        try:
            out = H.free_decode(H.pad_rows(feats.to(dev)), t, w.to(dev), hop, H.pad_rows(mask.to(dev)), mask_act, out_mode)
            # no-mask variant == module-level FreeEncDec.inverse
            out2 = H.free_decode(H.pad_rows(feats.to(dev)), t, w.to(dev), hop)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert out.shape == ref.shape
        assert rel_max(out.cpu().numpy(), ref.numpy()) < 2e-5, flags
        assert rel_max(out2.cpu().numpy(), O.free_decode(feats, w, hop).numpy()) < 2e-5, flags


@pytest.mark.parametrize("n,k,m,t", [(2, 16, 8, 50), (1, 512, 256, 300), (2, 256, 512, 129), (2, 20, 300, 128),
                                     (1, 33, 70, 1), (3, 256, 256, 1000)])
def test_conv1x1_plain_and_stats(H, dev, n, k, m, t):
    x = _rand((n, k, t), 6)
    w = _rand((m, k), 7, -0.2, 0.2)
    b = _rand((m,), 8)
    bn = _rand((n, m), 9)
    res = _rand((n, m, t), 10)
    ref = torch.matmul(w, x) + b.reshape(1, -1, 1) + bn.reshape(n, m, 1)
    xd, wd = H.pad_rows(x.to(dev)), H.pack_wt(w.to(dev))
    y, _ = H.conv1x1(xd, t, wd, m, None, b.to(dev), bn.to(dev), H.pad_rows(res.to(dev)))
    assert rel_max(y[..., :t].cpu().numpy(), (ref + res).numpy()) < 1e-5
    y, st = H.conv1x1(xd, t, wd, m, None, b.to(dev), bn.to(dev), None, want_stats=True)
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 1e-5
    with pytest.raises(RuntimeError, match="cannot be combined"):
        H.conv1x1(xd, t, wd, m, None, b.to(dev), bn.to(dev), H.pad_rows(res.to(dev)), want_stats=True)
    s = st.sum(1).cpu().numpy()
    ref64 = ref.double()
    np.testing.assert_allclose(s[:, 0], ref64.sum((1, 2)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[:, 1], (ref64 ** 2).sum((1, 2)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("norm", ["global", "affine", "none"])
@pytest.mark.parametrize("n,k,m,t", [(2, 24, 12, 77), (2, 256, 256, 500)])
def test_conv1x1_prologue(H, dev, norm, n, k, m, t):
    from puresound_amd import _abi
    x = _rand((n, k, t), 11) + 0.3
    w = _rand((m, k), 12, -0.2, 0.2)
    gamma, beta, slope = _rand((k,), 13, 0.5, 1.5), _rand((k,), 14, -0.2, 0.2), torch.tensor([0.2])
    if norm == "global":
        a = O.glob_ln(x, gamma, beta)
    elif norm == "affine":
        a = gamma.reshape(1, -1, 1) * x + beta.reshape(1, -1, 1)
    else:
        a = x
    a = O.prelu(a, slope)
    ref = torch.matmul(w, a)
    xd = H.pad_rows(x.to(dev))
    stats = None
    if norm == "global":
        # producer statistics: one part per utterance, exact fp64 sums
        stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
    kind = {"global": _abi.PS_NORM_GLOBAL, "affine": _abi.PS_NORM_AFFINE, "none": _abi.PS_NORM_NONE}[norm]
    # device copies stay referenced: the prologue carries raw pointers
    g_d, b_d, s_d = gamma.to(dev), beta.to(dev), slope.to(dev)
    pro = H.make_prologue(kind, True, stats, k * t, 1e-8, g_d, b_d, s_d)
    y, _ = H.conv1x1(xd, t, H.pack_wt(w.to(dev)), m, pro)
    torch.cuda.synchronize()
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5


@pytest.mark.parametrize("p,dil,causal", [(3, 1, False), (3, 2, False), (3, 4, False), (3, 8, False), (3, 128, False),
                                          (3, 3, False), (5, 9, False), (3, 1, True), (3, 2, True), (3, 16, True)])
@pytest.mark.parametrize("n,h,t", [(2, 12, 77), (1, 40, 1500), (2, 19, 4001)])
def test_dwconv(H, dev, p, dil, causal, n, h, t):
    from puresound_amd import _abi
    x = _rand((n, h, t), 15) + 0.1
    w = _rand((h, 1, p), 16)
    b = _rand((h,), 17)
    gamma, beta, slope = _rand((h,), 18, 0.5, 1.5), _rand((h,), 19, -0.2, 0.2), torch.tensor([0.3])
    a = O.prelu(O.glob_ln(x, gamma, beta), slope)
    left = (p - 1) * dil if causal else ((p - 1) // 2) * dil
    ref = O.dilated_conv(a, w, b, dil, left)
    ref = ref[..., :t] if causal else ref
    stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
    g_d, b_d, s_d = gamma.to(dev), beta.to(dev), slope.to(dev)
    pro = H.make_prologue(_abi.PS_NORM_GLOBAL, True, stats, h * t, 1e-8, g_d, b_d, s_d)
    # 0: the wave-private kernel where the shape allows it (P = 3, halo <= 256), bit 0: the workgroup-synchronised kernel
    for flags in (0, 1):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            y, st = H.dwconv(H.pad_rows(x.to(dev)), t, w.to(dev), b.to(dev), dil, left, pro, want_stats=True)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 1e-5, flags
        s = st.sum(1).cpu().numpy()
        np.testing.assert_allclose(s[:, 0], ref.double().sum((1, 2)).numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(s[:, 1], (ref.double() ** 2).sum((1, 2)).numpy(), rtol=1e-5)


@pytest.mark.parametrize("grid_cap", [1, 2, 3])
def test_conv1x1_long_runs_across_utterances(H, dev, grid_cap):
    """A persistent workgroup's run of tiles crosses utterance and m-tile boundaries (grid capped by the
    test hook): per-utterance norm tables, bias rows and the residual window must all follow."""
    from puresound_amd import _abi
    n, k, m, t = 5, 40, 300, 300
    x = _rand((n, k, t), 31) + 0.2
    w = _rand((m, k), 32, -0.2, 0.2)
    b, bn, res = _rand((m,), 33), _rand((n, m), 34), _rand((n, m, t), 35)
    gamma, beta, slope = _rand((k,), 36, 0.5, 1.5), _rand((k,), 37, -0.2, 0.2), torch.tensor([0.15])
    a = O.prelu(O.glob_ln(x, gamma, beta), slope)
    ref = torch.matmul(w, a) + b.reshape(1, -1, 1) + bn.reshape(n, m, 1)
    stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
    g_d, b_d, s_d = gamma.to(dev), beta.to(dev), slope.to(dev)
    pro = H.make_prologue(_abi.PS_NORM_GLOBAL, True, stats, k * t, 1e-8, g_d, b_d, s_d)
    xd, wd, bd, bnd, rd = H.pad_rows(x.to(dev)), H.pack_wt(w.to(dev)), b.to(dev), bn.to(dev), H.pad_rows(res.to(dev))
    old = H.lib().ps_debug_flags(grid_cap << 8)
    try:
        y, st = H.conv1x1(xd, t, wd, m, pro, bd, bnd, None, want_stats=True)
        y2, _ = H.conv1x1(xd, t, wd, m, pro, bd, bnd, rd)
        y3, _ = H.conv1x1(xd, t, wd, m, None, bd, bnd, rd)
        torch.cuda.synchronize()
    finally:
        H.lib().ps_debug_flags(old)
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5
    assert rel_max(y2[..., :t].cpu().numpy(), (ref + res).numpy()) < 2e-5
    ref3 = torch.matmul(w, x) + b.reshape(1, -1, 1) + bn.reshape(n, m, 1) + res
    assert rel_max(y3[..., :t].cpu().numpy(), ref3.numpy()) < 2e-5
    s = st.sum(1).cpu().numpy()
    np.testing.assert_allclose(s[:, 0], ref.double().sum((1, 2)).numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[:, 1], (ref.double() ** 2).sum((1, 2)).numpy(), rtol=1e-5)


def test_embed_bias(H, dev):
    dvec, w = _rand((3, 19), 20), _rand((11, 19), 21)
    for normalize in (False, True):
        d = dvec / dvec.norm(dim=1, keepdim=True) if normalize else dvec
        out = H.embed_bias(dvec.to(dev), w.to(dev), normalize)
        assert rel_max(out.cpu().numpy(), (d @ w.t()).numpy()) < 1e-5


def test_pad_unpad_roundtrip(H, dev):
    x = _rand((3, 7, 131), 22).to(dev)
    p = H.pad_rows(x)
    assert p.shape[-1] == 384 and torch.equal(p[..., :131], x) and float(p[..., 131:].abs().max()) == 0.0
    assert torch.equal(H.unpad_rows(p, 131), x)


def test_abi_rejects_bad_arguments(H, dev):
    x = torch.zeros(1, 16, 100, device=dev)  # ldt = 100 is not a multiple of 128
    with pytest.raises(RuntimeError, match="ldt"):
        H.conv1x1(x, 100, H.pack_wt(torch.zeros(8, 16, device=dev)), 8)
    with pytest.raises(RuntimeError, match="ROCm device only"):
        H.pad_rows(torch.zeros(2, 3, 5))


# ------------------------------------------------------------------------------------------------
# masker and wrapper against the reference's golden vectors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["ctn_embed", "ctn_dil3_k5", "ctn_gated", "ctn_gated_causal", "ctn_gated_causal_gln", "tcn_cln"])
def test_masker_matches_reference_golden(PA, dev, golden_dir, name):
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    dvec = torch.tensor(g["dvec"]).to(dev) if "dvec" in g else None
    y = model(torch.tensor(g["x"]).to(dev), dvec)
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", ["tiny_free", "tiny_free_relu_causal", "cfg2_short", "cfg2_full"])
def test_wrapper_inference_matches_reference_golden(PA, dev, golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    noisy = det_wave(c["seed"], c["B"], c["L"])
    wav = model.inference(noisy.to(dev))
    assert wav.shape == g["wav"].shape
    assert rel_max(wav.cpu().numpy(), g["wav"]) < TOL
    # and against the oracle on the same inputs (pre-clamp, so the clamp cannot hide an error)
    taps = {}
    O.inference(noisy, sd, cases.oracle_cfg(name), None, taps)
    feats, t = model.encoder.encode_padded(noisy.to(dev))
    mask = model.masker.forward_padded(feats, t)
    assert rel_max(feats[..., :t].cpu().numpy(), taps["feats"].numpy()) < 1e-5
    mact = model.mask_constraint.lower()
    assert rel_max(O.get_mask(mask[..., :t].cpu(), mact).numpy(), taps["mask"].numpy()) < TOL
    pre = model.encoder.decode_padded(feats, t, mask, mact, "none")
    assert rel_max(pre.cpu().numpy(), g["wav_preclamp"]) < TOL


# ------------------------------------------------------------------------------------------------
# speaker branch (BASELINE config 3): TCN x5 -> attentive statistics pooling -> 1x1 projection -> dvec
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,c,t", [(2, 24, 77), (3, 512, 249), (1, 40, 1500), (2, 16, 3999), (1, 8, 4096), (1, 8, 5000)])
def test_attn_stats_pool_kernel(H, dev, n, c, t):
    """rows of <= 4096 frames: the one-pass kernel (rows in registers); debug bit 0 / longer rows: the three-pass kernel."""
    from puresound_amd import _abi
    logits = _rand((n, c, t), 31, -3.0, 3.0)
    x = _rand((n, c, t), 32)
    a = torch.softmax(logits.double(), 2)
    mean = (a * x.double()).sum(2)
    std = torch.sqrt((a * (x.double() - mean.unsqueeze(2)) ** 2).sum(2).clamp(1e-12))
    ref = torch.cat((mean, std), 1).float().numpy()
    for flags in (0, 1):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            out = H.attn_stats_pool(H.pad_rows(logits.to(dev)), H.pad_rows(x.to(dev)), t)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert out.shape == (n, 2 * c)
        assert rel_max(out.cpu().numpy(), ref) < 2e-5, flags


@pytest.mark.parametrize("n,k,m,t", [(2, 24, 12, 77), (2, 128, 512, 249)])
def test_conv1x1_relu_affine_tanh_prologue(H, dev, n, k, m, t):
    """tanh(BN(ReLU(x))) on load: the attention branch of AttentiveStatisticsPooling (lobe/pooling.py:109-112)."""
    from puresound_amd import _abi
    x = _rand((n, k, t), 41, -2.0, 2.0)
    w, b = _rand((m, k), 42, -0.2, 0.2), _rand((m,), 43)
    scale, shift = _rand((k,), 44, 0.5, 1.5), _rand((k,), 45, -0.3, 0.3)
    ref = torch.matmul(w, torch.tanh(torch.relu(x) * scale.reshape(1, -1, 1) + shift.reshape(1, -1, 1)))
    ref = ref + b.reshape(1, -1, 1)
    sc_d, sh_d = scale.to(dev), shift.to(dev)
    pro = H.make_prologue(_abi.PS_NORM_AFFINE, False, None, 0.0, 0.0, sc_d, sh_d, None, pre_relu=True, post_tanh=True)
    y, _ = H.conv1x1(H.pad_rows(x.to(dev)), t, H.pack_wt(w.to(dev)), m, pro, b.to(dev))
    torch.cuda.synchronize()
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5


def test_attentive_stats_pooling_module_matches_oracle(PA, dev):
    from puresound_amd.nnet.lobe.pooling import AttentiveStatisticsPooling
    pool = AttentiveStatisticsPooling(48, 16).eval()
    sd = det_state_dict(pool)
    sd["tdnn.2.running_var"] = sd["tdnn.2.running_var"].abs() + 0.5
    pool.load_state_dict(sd)
    pool.to(dev)
    x = _rand((3, 48, 333), 51)
    ref = O.attentive_stats_pooling(x, sd, "")
    out = pool(x.to(dev))
    assert out.shape == ref.shape == (3, 96, 1)
    assert rel_max(out.cpu().numpy(), ref.numpy()) < 2e-5
    assert torch.equal(pool(x.to(dev), lengths=torch.ones(3, device=dev)), out)  # (lengths: tests/test_round2_gpu.py)
    # return_weight: the attention map itself (pooling.py:109-113), without and with relative lengths
    att = O.conv1x1(x, sd["tdnn.0.weight"], sd["tdnn.0.bias"])
    att = O.batch_norm_eval(torch.relu(att), sd, "tdnn.2.")
    att = O.conv1x1(torch.tanh(att), sd["conv.weight"], sd["conv.bias"])
    w = pool(x.to(dev), return_weight=True)
    assert w.shape == (3, 48, 333)
    assert rel_max(w.cpu().numpy(), torch.softmax(att, 2).numpy()) < 2e-5
    lens = torch.tensor([1.0, 0.5, 0.25])
    w2 = pool(x.to(dev), lengths=lens.to(dev), return_weight=True).cpu()
    for i, cnt in enumerate((333, 167, 84)):  # frame t takes part iff t < lens * L
        assert rel_max(w2[i, :, :cnt].numpy(), torch.softmax(att[i, :, :cnt], 1).numpy()) < 2e-5
        assert float(w2[i, :, cnt:].abs().max() if cnt < 333 else 0.0) == 0.0
    with pytest.raises(RuntimeError):
        pool(x.to(dev), lengths=torch.ones(3))  # a CPU tensor
    with pytest.raises(RuntimeError):
        pool.train()(x.to(dev))


def test_tse_wrapper_matches_reference_golden(PA, dev, golden_dir):
    name = "cfg3_short"
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    assert model.overall_parameters == cases.PARAM_COUNTS[name]
    noisy = det_wave(c["seed"], c["B"], c["L"])
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"])
    dvec = model.inference_tse_embedding(enroll.to(dev))
    assert dvec.shape == (c["B"], 192, 1)
    assert rel_max(dvec[..., 0].cpu().numpy(), g["dvec"]) < TOL
    wav = model.inference(noisy.to(dev), enroll.to(dev))
    assert wav.shape == g["wav"].shape
    assert rel_max(wav.cpu().numpy(), g["wav"]) < TOL
    # pre-clamp waveform and the oracle's intermediate taps
    taps = {}
    O.inference(noisy, sd, cases.oracle_cfg(name), enroll, taps)
    assert rel_max(dvec[..., 0].cpu().numpy(), taps["dvec"].numpy()) < TOL
    feats, t = model.encoder.encode_padded(noisy.to(dev))
    mask = model.masker.forward_padded(feats, t, dvec[..., 0])
    pre = model.encoder.decode_padded(feats, t, mask, "relu", "none")
    assert rel_max(pre.cpu().numpy(), g["wav_preclamp"]) < TOL
    # enrolment of a different length than the mixture, and a 4-utterance batch over two streams
    n4 = det_wave(5, 4, 4000)
    e4 = det_wave(6, 4, 2500)
    ref4 = O.inference(n4, sd, cases.oracle_cfg(name), e4)
    out4 = model.inference(n4.to(dev), e4.to(dev))
    assert rel_max(out4.cpu().numpy(), ref4.numpy()) < TOL
    model.hip_streams = 1
    assert torch.equal(model.inference(n4.to(dev), e4.to(dev)), out4)
    with pytest.raises(RuntimeError):
        model.inference(n4.to(dev))               # masker takes an embedding, none given
    with pytest.raises(RuntimeError):
        model.inference(n4.to(dev), e4[:2].to(dev))


def edge_ok(a, b, rtol=1e-3):
    return bool(np.all(np.abs(a - b) <= rtol * np.maximum(np.abs(b), 1.0)))


@pytest.mark.parametrize("name", ["tiny_stft", "tiny_stft_keepdc", "cfg1_short", "cfg1_full"])
def test_stft_wrapper_matches_reference_golden(PA, dev, golden_dir, name):
    """STFT encoder + complex masks + iSTFT decoder (BASELINE config 1 family).  The first/last 16 samples are
    divided by a window sum as small as 1.4e-9 and are compared element-relative."""
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    noisy = det_wave(c["seed"], c["B"], c["L"])
    wav = model.inference(noisy.to(dev)).cpu().numpy()
    assert wav.shape == g["wav"].shape
    assert rel_max(wav[:, 16:-16], g["wav"][:, 16:-16]) < TOL
    assert np.all(wav[:, 0] == 0)  # window sum 0 at sample 0: never divided
    enc = model.encoder.encoder
    feats, t = enc.encode_padded(noisy.to(dev), model.drop_first_bin)
    mask = model.masker.forward_padded(feats, t)
    pre = enc.decode_padded(H_complex(feats, mask), t, model.drop_first_bin, "none").cpu().numpy()
    assert rel_max(pre[:, 16:-16], g["wav_preclamp"][:, 16:-16]) < TOL
    assert edge_ok(pre, g["wav_preclamp"])
    if "feats_sub" in g:
        assert rel_max(feats[..., :t][:, ::7, ::5].cpu().numpy(), g["feats_sub"]) < 1e-5
        assert rel_max(mask[..., :t][:, ::7, ::5].cpu().numpy(), g["mask_sub"]) < TOL


def H_complex(feats, mask):
    from puresound_amd import hip
    return hip.complex_mask(feats, mask, "linear")


def test_stft_module_level_matches_reference_golden(PA, dev, golden_dir):
    name = "enc_stft"
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    wav = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    feats = model(wav)
    assert feats.shape == g["feats"].shape
    assert rel_max(feats.cpu().numpy(), g["feats"]) < 1e-5
    rec = model.inverse(torch.tensor(g["feats"]).to(dev)).cpu().numpy()
    assert rec.shape == g["rec"].shape
    assert rel_max(rec[:, 16:-16], g["rec"][:, 16:-16]) < TOL and edge_ok(rec, g["rec"])


def test_complex_mask_and_frame_kernels(H, dev):
    x, m = _rand((2, 12, 70), 41), _rand((2, 12, 70), 42)
    out = H.complex_mask(H.pad_rows(x.to(dev)), H.pad_rows(m.to(dev)), "relu")[..., :70].cpu()
    ref = O.apply_tf_masks(x, O.get_mask(m, "relu"), "complex", "complex")
    assert rel_max(out[:, :6].numpy(), ref[..., 0].numpy()) < 1e-6
    assert rel_max(out[:, 6:].numpy(), ref[..., 1].numpy()) < 1e-6
    wav = _rand((2, 333), 43)
    fr, t = H.frame(wav.to(dev), 20, 6)
    assert torch.equal(fr[..., :t].cpu(), O.frame(wav, 20, 6).transpose(1, 2))


def test_input_is_not_modified_and_result_is_deterministic(PA, dev):
    model = cases.build(PA.NS, "tiny_free").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    noisy = det_wave(3, 4, 2000).to(dev)
    keep = noisy.clone()
    a = model.inference(noisy)
    b = model.inference(noisy)
    assert torch.equal(noisy, keep)
    assert torch.equal(a, b)  # slab-reduced statistics: bitwise reproducible


@pytest.mark.parametrize("gemm", ["fp32", "bf16x3"])
def test_stream_split_is_bit_identical(PA, dev, gemm):
    """fp32 MFMA path: slab-reduced statistics make an utterance's result independent of its batch, bit for bit.  The
    split GEMM picks its kernel (and with it the grouping of the partial statistics) by the size of the launch, so
    there the results agree to fp32 rounding instead."""
    model = cases.build(PA.NS, "tiny_free").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision(gemm)
    same = torch.equal if gemm == "fp32" else (lambda a, b: bool(((a - b).abs().max() <= 2e-6 * b.abs().max()).item()))
    noisy = det_wave(4, 25, 2500).to(dev)  # (the split starts at 16 utterances)
    model.hip_streams = 1
    one = model.inference(noisy)
    for lanes in (2, 3):
        model.hip_streams = lanes
        assert same(model.inference(noisy), one)
    side = torch.cuda.Stream(dev)  # and from a caller-chosen stream
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        other = model.inference(noisy)
    side.synchronize()
    assert torch.equal(other, model.inference(noisy))


def test_plan_follows_weight_updates(PA, dev):
    model = cases.build(PA.NS, "tiny_free").eval().to(dev)
    noisy = det_wave(3, 2, 1500).to(dev)
    a = model.inference(noisy)
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    b = model.inference(noisy)
    assert not torch.equal(a, b)
    ref = O.inference(noisy.cpu(), sd, cases.oracle_cfg("tiny_free"))
    assert rel_max(b.cpu().numpy(), ref.numpy()) < TOL


# ------------------------------------------------------------------------------------------------
# BASELINE size (32 x 4 s): size-independent properties
# ------------------------------------------------------------------------------------------------
def test_full_batch_properties(PA, dev, golden_dir):
    """At B=32 x 64000 the oracle is too slow to run per test; use properties instead:
    (1) utterances are independent: row i of the batched run equals the B=1 run of that utterance bit for bit;
    (2) row 0 is the reference's golden utterance (cfg2_full);  (3) length law 64000 -> 64000; (4) |y| <= 1."""
    name = "cfg2_full"
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision("fp32")  # (bit-for-bit batch independence is the fp32 MFMA path's property)
    first = det_wave(cases.CASES[name]["seed"], 1, 64000)
    rest = det_wave(99, 31, 64000)
    batch = torch.cat([first, rest]).to(dev)
    out = model.inference(batch)
    assert out.shape == (32, 64000)
    assert float(out.abs().max()) <= 1.0
    assert rel_max(out[0:1].cpu().numpy(), g["wav"]) < TOL
    for i in (0, 7, 31):
        single = model.inference(batch[i:i + 1])
        assert torch.equal(single[0], out[i])
    assert torch.isfinite(out).all()


# ------------------------------------------------------------------------------------------------
# recurrent maskers: kernels, DPRNN / SkiM modules, wrapper (BASELINE config 4), streaming (config 5)
# ------------------------------------------------------------------------------------------------
from oracle import dualpath_oracle as DP  # noqa: E402

RNN_CASES = [n for n, c in cases.CASES.items() if c["kind"] == "rnn"]


def _lstm_sd(inp, hid, bi, seed):
    import torch.nn as nn
    m = nn.LSTM(inp, hid, num_layers=1, bidirectional=bi, batch_first=True)
    sd = {k: _rand(tuple(v.shape), seed + i, -0.4, 0.4) for i, (k, v) in enumerate(m.state_dict().items())}
    m.load_state_dict(sd)
    return m, sd


@pytest.mark.parametrize("hid,bi,mode", [(8, False, "intra"), (8, True, "inter"), (64, False, "inter"),
                                         (64, True, "intra"), (20, False, "intra"), (80, True, "inter"), (128, True, "inter"), (128, False, "intra"),
                                         (256, False, "intra")])
def test_lstm_kernel(H, dev, hid, bi, mode):
    from puresound_amd.nnet._plans import lstm_plan
    n, c, k, s = (2, 12, 5, 7) if hid not in (64, 128) or mode == "inter" else (2, 12, 8, 9)
    m, sd = _lstm_sd(c, hid, bi, 60)
    x = _rand((n, c, s * k), 61)
    d = 2 if bi else 1
    # oracle on the reference's batch-first sequences
    xs = x.transpose(1, 2).reshape(n, s, k, c)
    if mode == "intra":
        seqs, q, qs, steps, ss = xs.reshape(n * s, k, c), s, k, k, 1
    else:
        seqs, q, qs, steps, ss = xs.permute(0, 2, 1, 3).reshape(n * k, s, c), k, 1, s, k
    h0 = _rand((d, seqs.shape[0], hid), 62, -0.5, 0.5)
    c0 = _rand((d, seqs.shape[0], hid), 63, -0.5, 0.5)
    ref, (hn, cn) = DP.lstm(seqs, sd, "", bi, (h0, c0))
    p = lstm_plan(m.to(dev), torch.device(dev))
    t = s * k
    gx, _ = H.conv1x1(H.pad_rows(x.to(dev)), t, p["wih"], p["rows"], None, p["bias"])
    to_state = lambda v: H.pad_rows(v.reshape(d, n, q, hid).permute(1, 0, 3, 2).reshape(n, d * hid, q).to(dev))  # noqa: E731
    back = lambda v: v[..., :q].cpu().reshape(n, d, hid, q).permute(1, 0, 3, 2).reshape(d, n * q, hid)  # noqa: E731
    # debug flags pick the kernel variant for H = 64 / 128: 0 = by shape, 4 = 16 sequences per workgroup
    # (16x16x4 MFMA), 8 = 4 sequences per workgroup (4x4x1 MFMA), 2 = the generic VALU kernel
    from puresound_amd import _abi
    for flags in ((0, 4, 8, 2) if hid in (64, 128) else (0,)):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            hout, (hl, cl) = H.lstm(gx, p["whh_t"], hid, d, q, qs, steps, ss, to_state(h0), to_state(c0),
                                    want_state=True)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        got = hout[..., :t].cpu().transpose(1, 2).reshape(n, s, k, d * hid)
        got = got.reshape(n * s, k, -1) if mode == "intra" else got.permute(0, 2, 1, 3).reshape(n * k, s, -1)
        assert rel_max(got.numpy(), ref.numpy()) < 2e-5, flags
        assert rel_max(back(hl).numpy(), hn.numpy()) < 2e-5, flags
        assert rel_max(back(cl).numpy(), cn.numpy()) < 2e-5, flags


@pytest.mark.parametrize("bi,wscale", [(False, 1.0), (True, 1.0), (False, 1.5), (False, 1e-3)])
def test_lstm_whole_segment_kernel(H, dev, bi, wscale):
    """Intra pass at K = 20 (BASELINE config 4's shape): the whole-segment kernel (all 20 steps of 16 sequences held in
    registers, h' staged in LDS) against the oracle and against the 4-step-group kernel it replaces (debug bit 20); 18 sequences = one full and one ragged workgroup, both directions, initial and final states."""
    from puresound_amd.nnet._plans import lstm_plan
    from puresound_amd import _abi
    hid, n, c, k, s = 64, 2, 12, 20, 9
    m, sd = _lstm_sd(c, hid, bi, 160)
    if wscale != 1.0:   # saturating / vanishing recurrent weights: the fp16x2 kernel's per-matrix scale
        sd = {kk: (v * wscale if "weight_hh" in kk else v) for kk, v in sd.items()}
        m.load_state_dict(sd)
    x = _rand((n, c, s * k), 161)
    d = 2 if bi else 1
    seqs = x.transpose(1, 2).reshape(n * s, k, c)
    h0 = _rand((d, n * s, hid), 162, -0.5, 0.5)
    c0 = _rand((d, n * s, hid), 163, -0.5, 0.5)
    ref, (hn, cn) = DP.lstm(seqs, sd, "", bi, (h0, c0))
    p = lstm_plan(m.to(dev), torch.device(dev))
    t = s * k
    gx, _ = H.conv1x1(H.pad_rows(x.to(dev)), t, p["wih"], p["rows"], None, p["bias"])
    to_state = lambda v: H.pad_rows(v.reshape(d, n, s, hid).permute(1, 0, 3, 2).reshape(n, d * hid, s).to(dev))  # noqa: E731
    back = lambda v: v[..., :s].cpu().reshape(n, d, hid, s).permute(1, 0, 3, 2).reshape(d, n * s, hid)  # noqa: E731
    outs = []
    tol = 2e-5
    # (flags, f16x2): whole-segment fp32, the 4-step-group kernel, whole-segment with the fp16x2 recurrent product
    for flags, f16x2 in ((4, False), (4 | 1 << 20, False), (4, True)):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            hout, (hl, cl) = H.lstm(gx, p["whh_t"], hid, d, s, k, k, 1, to_state(h0), to_state(c0), want_state=True,
                                    f16x2=f16x2)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        got = hout[..., :t].cpu().transpose(1, 2).reshape(n * s, k, d * hid)
        e = (rel_max(got.numpy(), ref.numpy()), rel_max(back(hl).numpy(), hn.numpy()), rel_max(back(cl).numpy(), cn.numpy()))
        assert max(e) < tol, (flags, f16x2, e)
        outs.append(hout[..., :t].clone())
    e01 = rel_max(outs[0].cpu().numpy(), outs[1].cpu().numpy())
    e02 = rel_max(outs[0].cpu().numpy(), outs[2].cpu().numpy())
    assert e01 < tol / 2, e01         # same arithmetic, other data movement (4e-7; larger weights amplify)
    print("whole-segment LSTM: group-kernel vs segment", e01, "fp16x2 vs fp32", e02)
    assert e02 < tol / 2, e02         # fp16x2 product: fp32-class


@pytest.mark.parametrize("bi,wscale,s", [(False, 1.0, 50), (True, 1.0, 13), (False, 1e-3, 9), (False, 1.5, 9)])
def test_lstm_inter_pass_f16x2_kernel(H, dev, bi, wscale, s):
    """Inter-segment pass (strided steps, 4 sequences per workgroup): the fp16x2 recurrent product (v_mfma_f32_4x4x4_f16,
    three products) against the oracle and the fp32 4x4x1 kernel; ragged last workgroup (2 * 5 = 10 sequences), both
    directions, initial and final states, more steps than the pre-activation ring holds."""
    from puresound_amd.nnet._plans import lstm_plan
    hid, n, c, k = 64, 2, 12, 5
    m, sd = _lstm_sd(c, hid, bi, 180)
    if wscale != 1.0:
        sd = {kk: (v * wscale if "weight_hh" in kk else v) for kk, v in sd.items()}
        m.load_state_dict(sd)
    x = _rand((n, c, s * k), 181)
    d = 2 if bi else 1
    seqs = x.transpose(1, 2).reshape(n, s, k, c).permute(0, 2, 1, 3).reshape(n * k, s, c)
    h0 = _rand((d, n * k, hid), 182, -0.5, 0.5)
    c0 = _rand((d, n * k, hid), 183, -0.5, 0.5)
    ref, (hn, cn) = DP.lstm(seqs, sd, "", bi, (h0, c0))
    p = lstm_plan(m.to(dev), torch.device(dev))
    t = s * k
    gx, _ = H.conv1x1(H.pad_rows(x.to(dev)), t, p["wih"], p["rows"], None, p["bias"])
    to_state = lambda v: H.pad_rows(v.reshape(d, n, k, hid).permute(1, 0, 3, 2).reshape(n, d * hid, k).to(dev))  # noqa: E731
    back = lambda v: v[..., :k].cpu().reshape(n, d, hid, k).permute(1, 0, 3, 2).reshape(d, n * k, hid)  # noqa: E731
    outs = []
    tol = 2e-5 if s < 20 else 5e-5   # (50 dependent steps)
    for f16x2 in (False, True):
        hout, (hl, cl) = H.lstm(gx, p["whh_t"], hid, d, k, 1, s, k, to_state(h0), to_state(c0), want_state=True, f16x2=f16x2)
        torch.cuda.synchronize()
        got = hout[..., :t].cpu().transpose(1, 2).reshape(n, s, k, d * hid).permute(0, 2, 1, 3).reshape(n * k, s, -1)
        e = (rel_max(got.numpy(), ref.numpy()), rel_max(back(hl).numpy(), hn.numpy()), rel_max(back(cl).numpy(), cn.numpy()))
        assert max(e) < tol, (f16x2, e)
        outs.append(hout[..., :t].clone())
    assert rel_max(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < tol / 2


def test_lstm_f16x2_entry_is_the_fp32_kernel_off_its_shape(H, dev):
    """ps_lstm_f16x2_f32 documents: the two-term recurrent product only where a kernel for it exists (H = 64, 20 consecutive
    steps, or strided steps), exactly ps_lstm_f32 elsewhere -- here 8-step segments and H = 128 both ways: bit for bit."""
    from puresound_amd.nnet._plans import lstm_plan
    for hid, k, s, mode in ((64, 8, 9, "intra"), (128, 5, 7, "inter"), (128, 20, 4, "intra")):
        n, c = 2, 12
        m, _ = _lstm_sd(c, hid, False, 170)
        x = _rand((n, c, s * k), 171)
        p = lstm_plan(m.to(dev), torch.device(dev))
        gx, _ = H.conv1x1(H.pad_rows(x.to(dev)), s * k, p["wih"], p["rows"], None, p["bias"])
        q, qs, steps, ss = (s, k, k, 1) if mode == "intra" else (k, 1, s, k)
        a, _ = H.lstm(gx, p["whh_t"], hid, 1, q, qs, steps, ss)
        b, _ = H.lstm(gx, p["whh_t"], hid, 1, q, qs, steps, ss, f16x2=True)
        assert torch.equal(a[..., :s * k], b[..., :s * k]), (hid, k, s, mode)


@pytest.mark.parametrize("n,c,t", [(2, 16, 77), (1, 128, 300), (2, 512, 65), (2, 200, 130), (1, 70, 64)])
def test_chan_layernorm_kernel(H, dev, n, c, t):
    x, res, mul = _rand((n, c, t), 71, -2, 2), _rand((n, c, t), 72), _rand((n, c, t), 73)
    g, b, slope = _rand((c,), 74, 0.5, 1.5), _rand((c,), 75, -0.3, 0.3), torch.tensor([0.2])
    ln = DP.layer_norm(x.transpose(1, 2), g, b).transpose(1, 2)
    y = H.chan_layernorm(H.pad_rows(x.to(dev)), t, g.to(dev), b.to(dev), 1e-5, res=H.pad_rows(res.to(dev)))
    assert rel_max(y[..., :t].cpu().numpy(), (res + ln).numpy()) < 2e-5
    ref = torch.sigmoid(O.prelu(O.chan_ln(x, g, b), slope)) * mul
    y = H.chan_layernorm(H.pad_rows(x.to(dev)), t, g.to(dev), b.to(dev), 1e-8, slope=slope.to(dev), sigmoid=True,
                         mul=H.pad_rows(mul.to(dev)))
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5
    # C <= 256 runs with the thread's channels in registers (one pass over x); debug bit 23 keeps the three-pass kernel: the
    # same sums in the same order, bit for bit
    from puresound_amd import _abi
    old = _abi.lib().ps_debug_flags(1 << 23)
    try:
        y3 = H.chan_layernorm(H.pad_rows(x.to(dev)), t, g.to(dev), b.to(dev), 1e-8, slope=slope.to(dev), sigmoid=True,
                              mul=H.pad_rows(mul.to(dev)))
    finally:
        _abi.lib().ps_debug_flags(old)
    assert torch.equal(y3[..., :t], y[..., :t])


@pytest.mark.parametrize("name", RNN_CASES)
def test_recurrent_masker_matches_reference_golden(PA, dev, golden_dir, name):
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    embed = torch.tensor(g["embed"]).to(dev) if "embed" in g else None
    y = model(torch.tensor(g["x"]).to(dev), embed)
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < TOL


@pytest.mark.parametrize("t,k", [(25, 6), (24, 6), (27, 6), (53, 10), (7, 4)])
def test_segment_split_merge_kernels(H, dev, t, k):
    """SplitMerge.split / merge (lobe/trivial.py:178-241) incl. the reference's identity property
    (test/test_lobe.py:50-54)."""
    x = _rand((2, 5, t), 91)
    seg, rest = DP.split_overlap(x, k)                       # [N,S,K,C]
    xs, tp = H.segment_split(H.pad_rows(x.to(dev)), t, k)
    assert tp == seg.shape[1] * k and H.overlap_geometry(t, k)[0] == rest
    ref = seg.permute(0, 3, 1, 2).reshape(2, 5, tp)
    assert torch.equal(xs[..., :tp].cpu(), ref)
    back = H.segment_merge(xs, tp, t, k)
    assert torch.equal(back[..., :t].cpu(), x)
    y = _rand((2, 5, tp), 92)                                # merge of arbitrary segment data
    refm = DP.merge_overlap(y.reshape(2, 5, -1, k).permute(0, 2, 3, 1), rest)
    got = H.segment_merge(H.pad_rows(y.to(dev)), tp, t, k)
    assert torch.allclose(got[..., :t].cpu(), refm, atol=1e-7)


@pytest.mark.parametrize("name", ["cfg4_short", "cfg4_tse_short"])
def test_dprnn_wrapper_matches_reference_golden(PA, dev, golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    if name in cases.PARAM_COUNTS:
        assert model.overall_parameters == cases.PARAM_COUNTS[name]
    noisy = det_wave(c["seed"], c["B"], c["L"])
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]).to(dev) if "L_enroll" in c else None
    wav = model.inference(noisy.to(dev), enroll)
    assert wav.shape == g["wav"].shape
    assert rel_max(wav.cpu().numpy(), g["wav"]) < TOL
    # a batch of 5 over two HIP streams, ragged length (T % K != 0), against the oracle
    n5 = det_wave(9, 5, 4000 + 16 * 7)
    e5 = det_wave(10, 5, 2000) if enroll is not None else None
    ref = O.inference(n5, sd, cases.oracle_cfg(name), e5)
    out = model.inference(n5.to(dev), None if e5 is None else e5.to(dev))
    assert rel_max(out.cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("name", ["stream_tiny", "cfg5_demo"])
def test_streaming_skim_matches_reference_golden(PA, dev, golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    x, d = torch.tensor(g["x"]).to(dev), torch.tensor(g["embed"]).to(dev)
    k, frames = c["kw"]["seg_size"], c["frames"]
    y = model(x, d)
    assert rel_max(y.cpu().numpy(), g["y_offline"]) < TOL
    ys, seg_h, seg_c, mem_h, mem_c = [], None, None, None, None
    for i in range(frames // k):
        o, seg_h, mem_h, seg_c, mem_c = model.step_chunk(x[..., i * k:(i + 1) * k].permute(0, 2, 1), seg_h, mem_h,
                                                         seg_c, mem_c, d)
        ys.append(o)
    assert rel_max(torch.cat(ys, -1).cpu().numpy(), g["y_chunk"]) < TOL
    assert rel_max(torch.stack(seg_h).cpu().numpy(), g["chunk_seg_h"]) < TOL
    assert rel_max(torch.stack([torch.stack(p) for p in mem_h]).cpu().numpy(), g["chunk_mem_h"]) < TOL
    for use_graph in (False, True):
        model.init_status(streams=1, use_graph=use_graph)
        yf = torch.cat([model.step_frame(x[..., f].reshape(1, -1, 1), d) for f in range(frames)], -1)
        assert rel_max(yf.cpu().numpy(), g["y_frame"]) < TOL, use_graph
        assert rel_max(torch.stack(model.seg_lstm_h_states).cpu().numpy(), g["frame_seg_h"]) < TOL
        assert rel_max(torch.stack(model.seg_lstm_c_states).cpu().numpy(), g["frame_seg_c"]) < TOL
    # 3 concurrent streams: stream 0 is the fixture, the others run different audio / embeddings
    b = 3
    xs = torch.cat([x, x.flip(-1), x * 0.5], 0)
    ds = torch.cat([d, d * 0.5 + 0.1, d.flip(-1)], 0)
    model.init_status(streams=b)
    n_f = min(frames, 2 * k + 3)
    yb = torch.cat([model.step_frame(xs[..., f].reshape(b, -1, 1), ds) for f in range(n_f)], -1)
    assert rel_max(yb[0:1].cpu().numpy(), g["y_frame"][..., :n_f]) < TOL
    st = DP.SkimStream(det_state_dict(model), "", cases.rnn_args(c), streams=b)
    ref = torch.cat([st.step_frame(xs[..., f].cpu().reshape(b, 1, -1), ds.cpu()) for f in range(n_f)], -1)
    assert rel_max(yb.cpu().numpy(), ref.numpy()) < TOL


def test_demo_harness_matches_reference_golden(dev, golden_dir):
    """DemoTseNet.streaming_inference_chunk (egs/tse/demo/utils.py:78-128) on three 320-sample chunks."""
    from puresound_amd.streaming.demo import DemoTseNet
    name = "cfg5_demo"
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    h = c["harness"]
    net = DemoTseNet().eval()
    sd = det_state_dict(net)
    assert sum(v.numel() for v in sd.values()) == int(g["harness_n_params"])
    net.load_state_dict(sd)
    net.to(dev)
    wav = det_wave(c["seed"] + 200, 1, h["chunks"] * h["chunk"])
    d = torch.tensor(g["embed"]).to(dev)
    for use_graph in (False, True):
        net.init_streams(1, use_graph=use_graph)
        pre = None
        for i in range(h["chunks"]):
            pre = net.streaming_inference_chunk(wav[:, i * h["chunk"]:(i + 1) * h["chunk"]].to(dev), d[0], pre)
        assert pre.shape[-1] == g["harness_wav"].shape[-1]
        assert rel_max(pre[0].cpu().numpy(), g["harness_wav"]) < TOL, use_graph
    # 4 streams: stream 2 carries the fixture's audio and embedding
    b = 4
    wavs = det_wave(77, b, h["chunks"] * h["chunk"])
    wavs[2] = wav[0]
    ds = torch.rand(b, 192, generator=torch.Generator().manual_seed(5))
    ds[2] = torch.tensor(g["embed"])[0]
    net.init_streams(b)
    pre = None
    for i in range(h["chunks"]):
        pre = net.streaming_inference_chunk(wavs[:, i * h["chunk"]:(i + 1) * h["chunk"]].to(dev), ds.to(dev), pre)
    assert rel_max(pre[2].cpu().numpy(), g["harness_wav"]) < TOL


@pytest.mark.parametrize("norm,causal,film", [("gLN", False, True), ("cLN", True, False), ("bN1d", True, True),
                                               ("gLN", False, False)])
def test_gated_tcn_block_matches_oracle(PA, dev, norm, causal, film):
    """GatedTCN (conv_tasnet.py:93-215): concat / FiLM conditioning, gLN / cLN / bN1d, causal trim."""
    blk = PA.NS.GatedTCN(12, 10, 3, 2, emb_dim=5, causal=causal, tcn_norm=norm, use_film=film).eval()
    sd = det_state_dict(blk)
    blk.load_state_dict(sd)
    blk.to(dev)
    x, e = _rand((2, 12, 61), 81), _rand((2, 5), 82)
    ref = O.gated_tcn_block(x, sd, "", 3, 2, causal, norm, film, e)
    y = blk(x.to(dev), e.to(dev))
    assert rel_max(y.cpu().numpy(), ref.numpy()) < 2e-5
    ref0 = O.gated_tcn_block(x, sd, "", 3, 2, causal, norm, film, None) if film else None
    if film:  # FiLM blocks also run unconditioned (x_r = x)
        assert rel_max(blk(x.to(dev)).cpu().numpy(), ref0.numpy()) < 2e-5


def test_fused_streaming_step_kernels(H, dev):
    """ps_film_conv_f32 / ps_lstm_gates_cell_f32 / ps_proj_layernorm_f32 against their unfused compositions."""
    n, c, hid, t = 1, 12, 8, 37
    x = _rand((n, c, t), 101)
    ws, wb = _rand((c, c), 102, -0.3, 0.3), _rand((c, c), 103, -0.3, 0.3)
    rs, rb = _rand((n, c, t), 104), _rand((n, c, t), 105)
    ref = (torch.matmul(ws, x) + rs) * x + (torch.matmul(wb, x) + rb)
    pairs = torch.stack([ws, wb], 1).reshape(2 * c, c)
    res_pairs = torch.stack([rs, rb], 2).reshape(n, 2 * c, t)
    y = H.film_conv(H.pad_rows(x.to(dev)), t, H.pack_wt(pairs.to(dev)), H.pad_rows(res_pairs.to(dev)))
    assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5
    # gates + cell
    xh = _rand((n, c + hid, t), 106)
    w, bias = _rand((4 * hid, c + hid), 107, -0.3, 0.3), _rand((4 * hid,), 108)
    c0 = _rand((n, hid, t), 109)
    a = torch.matmul(w, xh) + bias.reshape(1, -1, 1)
    gi, gf, gg, go = [a[:, k * hid:(k + 1) * hid] for k in range(4)]
    c_ref = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gg)
    h_ref = torch.sigmoid(go) * torch.tanh(c_ref)
    order = (torch.arange(4).reshape(1, 4) * hid + torch.arange(hid).reshape(hid, 1)).reshape(-1)
    c_d = H.pad_rows(c0.to(dev))
    h_d = torch.zeros_like(c_d)
    H.lstm_gates_cell(H.pad_rows(xh.to(dev)), t, H.pack_wt(w[order].contiguous().to(dev)), bias[order].to(dev), c_d,
                      h_d, hid)
    assert rel_max(c_d[..., :t].cpu().numpy(), c_ref.numpy()) < 2e-5
    assert rel_max(h_d[..., :t].cpu().numpy(), h_ref.numpy()) < 2e-5
    # projection + LayerNorm + residual (+ second LayerNorm, + copy)
    for m, k in ((12, 8), (128, 256), (200, 20)):
        hx, res = _rand((n, k, t), 110), _rand((n, m, t), 111)
        wp, bp = _rand((m, k), 112, -0.3, 0.3), _rand((m,), 113)
        g1, b1, g2, b2 = _rand((m,), 114, 0.5, 1.5), _rand((m,), 115), _rand((m,), 116, 0.5, 1.5), _rand((m,), 117)
        p_ref = torch.matmul(wp, hx) + bp.reshape(1, -1, 1)
        y_ref = res + DP.layer_norm(p_ref.transpose(1, 2), g1, b1).transpose(1, 2)
        y2_ref = DP.layer_norm(y_ref.transpose(1, 2), g2, b2, 1e-5).transpose(1, 2)
        hx_d = H.pad_rows(hx.to(dev))
        cp = torch.zeros_like(hx_d)
        y, y2 = H.proj_layernorm(hx_d, t, H.pack_wt(wp.to(dev)), bp.to(dev), m, g1.to(dev), b1.to(dev), 1e-5,
                                 H.pad_rows(res.to(dev)), (g2.to(dev), b2.to(dev), 1e-5), x_copy=cp)
        assert rel_max(y[..., :t].cpu().numpy(), y_ref.numpy()) < 2e-5
        assert rel_max(y2[..., :t].cpu().numpy(), y2_ref.numpy()) < 2e-5
        assert torch.equal(cp[..., :t].cpu(), hx)


@pytest.mark.parametrize("n,k,m,t", [(3, 64, 128, 4000), (1, 64, 128, 1501), (5, 20, 100, 900), (2, 32, 256, 777),
                                     (1, 100, 200, 4100), (2, 128, 128, 1000), (1, 96, 256, 700)])
@pytest.mark.parametrize("res_inside", [False, True])
def test_proj_layernorm_on_long_rows(H, dev, n, k, m, t, res_inside):
    """ps_proj_layernorm_f32 on the offline paths' long rows: the row kernel (32 frames x all channels per wave,
    v_mfma_f32_32x32x2_f32) and, with ps_debug_flags bit 4, the 16-frame kernel of the streaming step -- both against
    res + LN(W x + b) / LN(W x + b + res) in fp64."""
    from puresound_amd import _abi
    hx, res = _rand((n, k, t), 210), _rand((n, m, t), 211)
    wp, bp = _rand((m, k), 212, -0.3, 0.3), _rand((m,), 213)
    g1, b1 = _rand((m,), 214, 0.5, 1.5), _rand((m,), 215)
    p = torch.matmul(wp.double(), hx.double()) + bp.double().reshape(1, -1, 1)
    if res_inside:
        p = p + res.double()
    mu = p.mean(1, keepdim=True)
    var = ((p - mu) ** 2).mean(1, keepdim=True)
    ref = (p - mu) / torch.sqrt(var + 1e-5) * g1.double().reshape(1, -1, 1) + b1.double().reshape(1, -1, 1)
    if not res_inside:
        ref = ref + res.double()
    outs = []
    # 0: by shape (K = 64, M = 128: the pipelined row kernel), bit 21: the unpipelined row kernel, 16: the 16-frame kernel
    for flags in (0, 1 << 21, 16):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            y, _ = H.proj_layernorm(H.pad_rows(hx.to(dev)), t, H.pack_wt(wp.to(dev)), bp.to(dev), m, g1.to(dev), b1.to(dev),
                                    1e-5, H.pad_rows(res.to(dev)), res_inside=res_inside)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert rel_max(y[..., :t].cpu().double().numpy(), ref.numpy()) < 2e-5, flags
        outs.append(y[..., :t].cpu())
    assert rel_max(outs[0].numpy(), outs[2].numpy()) < 1e-5
    assert rel_max(outs[0].numpy(), outs[1].numpy()) < 2e-6   # the two row kernels: the same arithmetic


# ------------------------------------------------------------------------------------------------
# bf16 matrix pipe: plain bf16 products (planes = 1) and the fp32-accurate 3-way split (planes = 3)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("planes,tol", [(3, 2e-6), (1, 2e-2)])
@pytest.mark.parametrize("n,k,m,t,mode", [(2, 24, 12, 77, "plain"), (2, 256, 256, 500, "norm_stats"),
                                          (1, 256, 512, 300, "norm_res"), (2, 512, 256, 129, "stats"),
                                          (1, 40, 300, 128, "affine"),
                                          # many tiles: the persistent ping-pong kernel, runs across utterances, odd tile counts
                                          (9, 64, 256, 3800, "norm_stats"), (5, 48, 512, 3700, "norm_res"),
                                          (9, 40, 200, 3800, "plain"),
                                          # deep K loops over many supertiles: the counted DMA waits of the ping-pong
                                          # kernel (a wait that is too weak shows as stale operands, first with planes = 1)
                                          (8, 512, 256, 3999, "stats"), (8, 256, 512, 3999, "norm_res"),
                                          (8, 256, 256, 3999, "norm_stats"),
                                          # per-utterance bias (the speaker embedding folded into in_conv) at that size
                                          (8, 512, 256, 3999, "stats_bias_n")])
def test_conv1x1_bf16_planes(H, dev, planes, tol, n, k, m, t, mode):
    from puresound_amd import _abi
    x = _rand((n, k, t), 121) + 0.2
    w, b = _rand((m, k), 122, -0.2, 0.2), _rand((m,), 123)
    gamma, beta, slope = _rand((k,), 124, 0.5, 1.5), _rand((k,), 125, -0.2, 0.2), torch.tensor([0.2])
    a, pro, keep = x.double(), None, None
    if mode.startswith("norm"):
        a = O.prelu(O.glob_ln(a, gamma.double(), beta.double()), slope.double())
        stats = torch.stack([x.double().sum((1, 2)), (x.double() ** 2).sum((1, 2))], -1).reshape(n, 1, 2).to(dev)
        keep = (stats, gamma.to(dev), beta.to(dev), slope.to(dev))
        pro = H.make_prologue(_abi.PS_NORM_GLOBAL, True, keep[0], k * t, 1e-8, keep[1], keep[2], keep[3])
    elif mode == "affine":
        a = O.prelu(gamma.double().reshape(1, -1, 1) * a + beta.double().reshape(1, -1, 1), slope.double())
        keep = (gamma.to(dev), beta.to(dev), slope.to(dev))
        pro = H.make_prologue(_abi.PS_NORM_AFFINE, True, None, 0.0, 0.0, keep[0], keep[1], keep[2])
    ref = torch.matmul(w.double(), a) + b.double().reshape(1, -1, 1)
    bias_n = _rand((n, m), 127) if mode.endswith("bias_n") else None
    if bias_n is not None:
        ref = ref + bias_n.double().reshape(n, m, 1)
    res = _rand((n, m, t), 126) if mode == "norm_res" else None
    want = mode in ("norm_stats", "stats", "stats_bias_n")
    if res is not None:
        ref = ref + res.double()
    # flag bit 27 = simple kernel only (small grids take its 256 x 32 tile), bit 29 = keep its 256 x 128 tile,
    # bit 28 = the persistent ping-pong kernel at any size
    # bit 30 = the single-wave-per-SIMD experiment (where the launch is large enough for it)
    # bit 5 (32) = the two-barrier ping-pong kernel instead of the interleaved one-barrier kernel
    for flags in (0, 32, 1 << 27, (1 << 27) | (1 << 29), 1 << 28, (1 << 28) | 32, 1 << 30):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            y, st = H.conv1x1_bf16(H.pad_rows(x.to(dev)), t, H.pack_wt_bf16(w.to(dev), planes), m, pro, b.to(dev),
                                   None if bias_n is None else bias_n.to(dev),
                                   None if res is None else H.pad_rows(res.to(dev)), want_stats=want)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert rel_max(y[..., :t].cpu().double().numpy(), ref.numpy()) < tol, flags
        # the two kernels round and accumulate identically (only the residual enters at the other end of the sum)
        if flags == 0:
            first = y[..., :t].cpu().double().numpy()
        else:
            assert rel_max(y[..., :t].cpu().double().numpy(), first) < 1e-5, flags
        if want:
            s = st.sum(1).cpu().numpy()
            got = y[..., :t].cpu().double()
            np.testing.assert_allclose(s[:, 0], got.sum((1, 2)).numpy(), rtol=1e-6, atol=1e-3)
            np.testing.assert_allclose(s[:, 1], (got ** 2).sum((1, 2)).numpy(), rtol=1e-6)


@pytest.mark.parametrize("precision,tol", [("bf16x3", TOL), ("bf16", 3e-2)])
def test_conv_tasnet_on_the_bf16_matrix_pipe(PA, dev, golden_dir, precision, tol):
    """Config 2 with its GEMMs on the bf16 pipe: the 3-way split stays inside the fp32 tolerance against the
    reference's golden vector; plain bf16 products meet the bf16 acceptance of SURVEY 8(d) (l2-rel <= 3e-2)."""
    name = "cfg2_full"
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision(precision)
    noisy = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    feats, t = model.encoder.encode_padded(noisy)
    mask = model.masker.forward_padded(feats, t)
    pre = model.encoder.decode_padded(feats, t, mask, "relu", "none").cpu().numpy()
    if precision == "bf16x3":
        assert rel_max(pre, g["wav_preclamp"]) < tol
        assert rel_max(model.inference(noisy).cpu().numpy(), g["wav"]) < tol
    else:
        l2 = np.linalg.norm(pre - g["wav_preclamp"]) / np.linalg.norm(g["wav_preclamp"])
        assert l2 < tol
    model.masker.set_gemm_precision("fp32")
    assert rel_max(model.inference(noisy).cpu().numpy(), g["wav"]) < TOL


@pytest.mark.parametrize("name,gemm", [("cfg3_short", "fp32"), ("cfg4_short", "fp32"), ("cfg4_tse_short", "fp32"),
                                       ("cfg3_short", "bf16x3")])
def test_full_size_properties_of_the_other_baseline_configs(PA, dev, name, gemm):
    """BASELINE configs 3 and 4 at their full size (32 x 4 s per GPU, + 4 s enrolment): one full-length utterance
    against the oracle, and the batch properties -- utterances independent (row i of the batch == its B=1 run bit
    for bit), length law, |y| <= 1, finite."""
    c = cases.CASES[name]
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    # fp32 MFMA operands: rows bit for bit independent of the batch; split GEMM: to fp32 rounding
    if hasattr(model.masker, "set_gemm_precision"):
        model.masker.set_gemm_precision(gemm)
    for m in (model.speaker_net if getattr(model, "speaker_net", None) is not None else []):
        if hasattr(m, "gemm_precision"):
            m.gemm_precision = gemm
    tse = "L_enroll" in c
    noisy = det_wave(301, 32, 64000)
    enroll = det_wave(302, 32, 64000) if tse else None
    out = model.inference(noisy.to(dev), None if enroll is None else enroll.to(dev))
    assert out.shape == (32, 64000)
    assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
    for i in (0, 15, 16, 31):
        single = model.inference(noisy[i:i + 1].to(dev), None if enroll is None else enroll[i:i + 1].to(dev))
        if gemm == "fp32" and not name.startswith("cfg4"):
            assert torch.equal(single[0], out[i]), i
        else:
            # split GEMMs: fp32 rounding; DPRNN: the recurrence kernel depends on how many sequences a launch has (16 per
            # workgroup for a full batch, 4 for one utterance): the same sums in another order
            assert float((single[0] - out[i]).abs().max()) <= 1e-5, i
    ref = O.inference(noisy[5:6], sd, cases.oracle_cfg(name), None if enroll is None else enroll[5:6])
    assert rel_max(out[5:6].cpu().numpy(), ref.numpy()) < TOL


def test_sixty_four_streams_match_single_streams(dev):
    """BASELINE config 5 at its full size: 64 concurrent streams through the demo harness for two chunks; stream i of
    the batched run equals the same audio / embedding run alone (tolerance: the short-row GEMM splits K the same way
    for every frame, so this is exact), and one stream matches the oracle's harness."""
    from puresound_amd.streaming.demo import DemoTseNet
    c = cases.CASES["cfg5_demo"]
    net = DemoTseNet().eval()
    sd = det_state_dict(net)
    net.load_state_dict(sd)
    net.to(dev)
    b, n_chunks = 64, 2
    wav = det_wave(401, b, 320 * n_chunks)
    emb = torch.rand(b, 192, generator=torch.Generator().manual_seed(402))
    net.init_streams(b)
    pre = None
    for i in range(n_chunks):
        pre = net.streaming_inference_chunk(wav[:, i * 320:(i + 1) * 320].to(dev), emb.to(dev), pre)
    assert pre.shape == (b, 16 * (20 * n_chunks - 1) + 16)
    assert torch.isfinite(pre).all()
    for i in (0, 37, 63):
        net.init_streams(1)
        one = None
        for j in range(n_chunks):
            one = net.streaming_inference_chunk(wav[i:i + 1, j * 320:(j + 1) * 320].to(dev), emb[i:i + 1].to(dev), one)
        assert torch.equal(one[0], pre[i]), i
    st = DP.DemoStream(sd, cases.rnn_args(c), 1, 32, 16)
    ref = None
    for j in range(n_chunks):
        ref = st.step_chunk(wav[37:38, j * 320:(j + 1) * 320], emb[37:38], ref)
    assert rel_max(pre[37:38].cpu().numpy(), ref.numpy()) < TOL


# ------------------------------------------------------------------------------------------------
# 2-D convolutional maskers: Unet, UnetTcn, DPCRN, the ns_dpcrn_v0_causal preset
# ------------------------------------------------------------------------------------------------
from oracle import unet_oracle as UO  # noqa: E402


@pytest.mark.parametrize("transposed", [False, True])
def test_unfold2d_kernel(H, dev, transposed):
    import torch.nn.functional as F
    n, c1, c2, f, t = 2, 3, 2, 11, 37
    x1, x2 = _rand((n, c1, f, t), 131), _rand((n, c2, f, t), 132)
    x = torch.cat([x1, x2], 1)
    kf, kt, sf, df, dt = 3, 2, 2, 1, 1
    pad = lambda v: H.pad_rows(v.reshape(n, -1, t).to(dev)).view(n, v.shape[1], f, -1)  # noqa: E731
    if not transposed:
        w = _rand((4, c1 + c2, kf, kt), 133)
        ref = F.conv2d(F.pad(x, (kt - 1, 0, kf // 2, kf // 2)), w, None, stride=(sf, 1))
        f_out = ref.shape[2]
        taps = H.unfold2d(pad(x1), pad(x2), t, f_out, kf, kt, sf, df, dt, kf // 2, kt - 1, False)
        w2 = w.reshape(4, -1)
    else:
        w = _rand((c1 + c2, 4, kf, kt), 133)
        op = sf - kf + 2 * (kf // 2)
        ref = F.conv_transpose2d(x, w, None, stride=(sf, 1), padding=(kf // 2, 0), output_padding=(op, 0))[..., (kt - 1):]
        f_out = ref.shape[2]
        taps = H.unfold2d(pad(x1), pad(x2), t, f_out, kf, kt, sf, df, dt, kf // 2, kt - 1, True)
        w2 = w.permute(1, 0, 2, 3).reshape(4, -1)
    ld = taps.shape[-1] // f_out
    got = torch.einsum("mk,nkx->nmx", w2, taps.cpu()).view(n, 4, f_out, ld)[..., :t]
    assert rel_max(got.numpy(), ref.numpy()) < 1e-6


@pytest.mark.parametrize("transposed,m,c1,c2", [(False, 4, 3, 2), (True, 40, 3, 2), (False, 100, 3, 2), (True, 2, 3, 2),
                                                (True, 2, 40, 24), (False, 40, 20, 12), (True, 100, 30, 34), (True, 32, 16, 16)])
def test_conv2d_implicit_gemm_kernel(H, dev, transposed, m, c1, c2):
    """ps_conv2d_f32 against torch's Conv2d / ConvTranspose2d: the LDS-staged kernel (weights through LDS, compacted tap
    table; several 32-k chunks at the larger channel counts), the <= 4-channel kernel (m = 2: the mask layer) and the round-3
    kernel (debug bit 23)."""
    import torch.nn.functional as F
    from puresound_amd import _abi
    n, f, t = 2, 11, 150
    x1, x2 = _rand((n, c1, f, t), 151), _rand((n, c2, f, t), 152)
    x = torch.cat([x1, x2], 1)
    kf, kt, sf = 3, 2, 2
    b, slope = _rand((m,), 153), torch.tensor([0.2])
    pad = lambda v: H.pad_rows(v.reshape(n, -1, t).to(dev)).view(n, v.shape[1], f, -1)  # noqa: E731
    if not transposed:
        w = _rand((m, c1 + c2, kf, kt), 154, -0.3, 0.3)
        ref = F.conv2d(F.pad(x, (kt - 1, 0, kf // 2, kf // 2)), w, b, stride=(sf, 1))
        w2, shift = w.reshape(m, -1), kt - 1
    else:
        w = _rand((c1 + c2, m, kf, kt), 154, -0.3, 0.3)
        op = sf - kf + 2 * (kf // 2)
        ref = F.conv_transpose2d(x, w, b, stride=(sf, 1), padding=(kf // 2, 0), output_padding=(op, 0))[..., (kt - 1):]
        w2, shift = w.permute(1, 0, 2, 3).reshape(m, -1), kt - 1
    ref = torch.where(ref >= 0, ref, 0.2 * ref)
    outs = []
    for flags in (0, 1 << 23):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            y = H.conv2d(pad(x1), pad(x2), H.pack_wt(w2.contiguous().to(dev)), b.to(dev), m, t, ref.shape[2], kf, kt, sf, 1, 1,
                         kf // 2, shift, transposed, "prelu", slope.to(dev))
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        assert rel_max(y[..., :t].cpu().numpy(), ref.numpy()) < 2e-5, flags
        assert float(y[..., t:].abs().max()) == 0.0
        outs.append(y)
    assert rel_max(outs[0].cpu().numpy(), outs[1].cpu().numpy()) < 1e-5


UNET_CASES = [n for n, c in cases.CASES.items() if c["kind"] == "unet"]


@pytest.mark.parametrize("name", UNET_CASES)
def test_unet_family_matches_reference_golden(PA, dev, golden_dir, name):
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    x = torch.tensor(g["x"]).to(dev)
    y = model(x, torch.tensor(g["embed"]).to(dev)) if "embed" in g else model(x)
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", ["ns_dpcrn_short", "ns_dparn_short"])
def test_ns_dpcrn_preset_matches_reference_golden(PA, dev, golden_dir, name):
    """egs/ns/model.py:40-171 (ns_dpcrn_v0_causal, ns_dparn_v0_causal) through the wrapper: conv-STFT, DPCRN / DPARN,
    complex mask, iSTFT."""
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    assert model.overall_parameters == int(g["n_params"])
    noisy = det_wave(c["seed"], c["B"], c["L"])
    wav = model.inference(noisy.to(dev))
    assert wav.shape == g["wav"].shape
    sl = slice(16, wav.shape[1] - 16)
    assert rel_max(wav.cpu().numpy()[:, sl], g["wav"][:, sl]) < TOL
    # a longer, ragged batch against the oracle
    n3 = det_wave(77, 3, 9000 + 37)
    ref = O.inference(n3, sd, cases.oracle_cfg(name))
    out = model.inference(n3.to(dev))
    sl = slice(16, ref.shape[1] - 16)
    assert rel_max(out.cpu().numpy()[:, sl], ref.numpy()[:, sl]) < TOL


def test_norm_lobes_on_their_own(dev):
    """GlobLN / ChanLN / InstantLN called directly (lobe/norm.py:20-68), 3-D and 4-D inputs, against the formulas."""
    from puresound_amd.nnet.lobe.norm import ChanLN, GlobLN, InstantLN
    x3, x4 = _rand((3, 10, 77), 201, -2.0, 3.0), _rand((2, 6, 5, 41), 202, -1.0, 4.0)

    def fill(m, seed):
        m.gamma.data = _rand((m.channel_size,), seed, 0.5, 1.5)
        m.beta.data = _rand((m.channel_size,), seed + 1, -0.3, 0.3)
        return m.to(dev)

    for x in (x3, x4):
        c = x.shape[1]
        view = [1, c] + [1] * (x.dim() - 2)
        m = fill(GlobLN(c), 203)
        dims = list(range(1, x.dim()))
        mean = x.double().mean(dims, keepdim=True)
        var = ((x.double() - mean) ** 2).mean(dims, keepdim=True)
        ref = (x.double() - mean) / (var + 1e-8).sqrt() * m.gamma.detach().cpu().double().view(view) + m.beta.detach().cpu().double().view(view)
        assert rel_max(m(x.to(dev)).cpu().numpy(), ref.numpy()) < 2e-6
        m = fill(ChanLN(c), 205)
        mean = x.double().mean(1, keepdim=True)
        var = x.double().var(1, keepdim=True, unbiased=False)
        ref = (x.double() - mean) / (var + 1e-8).sqrt() * m.gamma.detach().cpu().double().view(view) + m.beta.detach().cpu().double().view(view)
        assert rel_max(m(x.to(dev)).cpu().numpy(), ref.numpy()) < 2e-6
    n, ch, c, t = x4.shape
    m = fill(InstantLN(ch * c), 207)
    flat = x4.double().reshape(n, ch * c, t)
    mean, var = flat.mean(1, keepdim=True), flat.var(1, keepdim=True, unbiased=False)
    ref = ((flat - mean) / (var + 1e-8).sqrt() * m.gamma.detach().cpu().double().view(1, -1, 1)
           + m.beta.detach().cpu().double().view(1, -1, 1)).reshape(n, ch, c, t)
    assert rel_max(m(x4.to(dev)).cpu().numpy(), ref.numpy()) < 2e-6
    with pytest.raises(RuntimeError):
        GlobLN(10)(x3)  # CPU tensor: no fallback


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "lobe"])
def test_depthwise_separable_lobe_on_its_own(PA, dev, golden_dir, name):
    """DepthwiseSeparableConv1d.forward (lobe/cnn.py:84-106) with the hid_channels transform, the skip connection, every
    norm the lobe takes, causal and not: the stage-by-stage HIP path against the reference's golden vectors."""
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    y = model(torch.tensor(g["x"]).to(dev))
    assert y.shape == g["y"].shape
    assert rel_max(y.cpu().numpy(), g["y"]) < TOL
    with pytest.raises(RuntimeError):
        model(torch.tensor(g["x"]))  # CPU tensor: no fallback


@pytest.mark.parametrize("name", ["tse_unet_tcn_causal_short", "tse_unet_tcn_short", "tse_skim_causal_short",
                                  "tse_skim_fbank_short", "tse_skim_vad_short", "cfg3_causal_short",
                                  "tse_unet_tcn_v1_short", "tse_skim_v0_short", "tse_skim_v1_short", "tse_skim_v2_short"])
def test_more_tse_presets_match_reference_golden(PA, dev, golden_dir, name):
    """egs/tse presets verbatim: tse_unet_tcn_v0_causal (STFT + UnetTcn with causal gated bN1d TCN + speaker net
    Magnitude -> 5 x GatedTCN -> ASP -> 1x1, real mask on the STFT) and tse_skim_v0_causal (FreeEncDec + SkiM/FiLM +
    TCN speaker net)."""
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    model.to(dev)
    assert model.overall_parameters == int(g["n_params"])
    if name in cases.PARAM_COUNTS:
        assert model.overall_parameters == cases.PARAM_COUNTS[name]
    noisy = det_wave(c["seed"], c["B"], c["L"])
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"])
    torch.manual_seed(c["seed"])  # (tse_skim_v2_short: SpecAugment draws its mask from the global generator, as in the
    dvec = model.inference_tse_embedding(enroll.to(dev))   # reference -- the fixture was made behind the same seed)
    assert rel_max(dvec[..., 0].cpu().numpy(), g["dvec"]) < TOL
    torch.manual_seed(c["seed"])
    wav = model.inference(noisy.to(dev), enroll.to(dev))
    assert wav.shape == g["wav"].shape
    edge = 16 if c["enc"]["kind"] == "stft" else 0
    sl = slice(edge, wav.shape[1] - edge) if edge else slice(None)
    assert rel_max(wav.cpu().numpy()[:, sl], g["wav"][:, sl]) < TOL


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "fbank"])
def test_fbank_encoder_matches_reference_golden(PA, dev, golden_dir, name):
    """FbankEnc / ConvMelSpectrogram: conv-STFT -> power -> mel projection."""
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    y = model(det_wave(c["seed"], c["B"], c["L"]).to(dev))
    assert y.shape == g["feats"].shape
    assert rel_max(y.cpu().numpy(), g["feats"]) < TOL


@pytest.mark.parametrize("e,heads,f,t,flags", [(16, 4, 9, 13, 0), (64, 4, 9, 13, 0), (64, 4, 9, 13, 1 << 23), (64, 4, 9, 13, 1 << 21), (128, 8, 64, 37, 0), (128, 8, 64, 37, 1 << 21),
                                               (64, 2, 20, 9, 0), (64, 1, 30, 6, 0), (64, 1, 30, 6, 1 << 23)])
def test_self_attention_kernel(H, dev, e, heads, f, t, flags):
    """ps_self_attention_f32 / ps_add_position_f32 against the oracle's multi-head attention, both sequence layouts
    (positions contiguous in time; positions strided over frequency rows as in DPARN), with and without causal mask.
    Head dimensions 16 / 32 / 64 with at most 64 positions take the register-score kernel (debug bit 23: the general one);
    (128, 8, 64, 37) is the DPARN bottleneck's shape with a ragged last workgroup."""
    from puresound_amd import _abi
    old_flags = _abi.lib().ps_debug_flags(flags)
    try:
        _self_attention_case(H, dev, e, heads, f, t)
    finally:
        _abi.lib().ps_debug_flags(old_flags)


def _self_attention_case(H, dev, e, heads, f, t):
    n = 2
    x = _rand((n, e, f, t), 141)
    w_in, w_out = _rand((3 * e, e), 142, -0.4, 0.4), _rand((e, e), 143, -0.4, 0.4)
    xp = H.pad_rows(x.reshape(n, e * f, t).to(dev)).view(n, e, f, -1)
    ld = xp.shape[-1]
    rows = xp.view(n, e, f * ld)
    frames = (f - 1) * ld + t
    qkv, _ = H.conv1x1(rows, frames, H.pack_wt(w_in.to(dev)), 3 * e)
    for causal in (False, True):
        # DPARN layout: one sequence per frame t, positions = frequency rows
        att = H.self_attention(qkv, e, heads, t, 1, f, ld, causal)
        out, _ = H.conv1x1(att, frames, H.pack_wt(w_out.to(dev)), e)
        got = out.view(n, e, f, ld)[..., :t].cpu()                                  # [N, E, F, T]
        seq = x.permute(0, 3, 2, 1).reshape(n * t, f, e)                             # [N*T, F, E]
        ref = UO.multihead_attention(seq, w_in, w_out, heads, causal).reshape(n, t, f, e).permute(0, 3, 2, 1)
        assert rel_max(got.numpy(), ref.numpy()) < 2e-5, causal
    # time layout: one sequence per (n, f) row... expressed as Q = f rows, positions = frames
    att = H.self_attention(qkv, e, heads, f, ld, t, 1, False)
    out, _ = H.conv1x1(att, frames, H.pack_wt(w_out.to(dev)), e)
    got = out.view(n, e, f, ld)[..., :t].cpu()
    seq = x.permute(0, 2, 3, 1).reshape(n * f, t, e)
    ref = UO.multihead_attention(seq, w_in, w_out, heads, False).reshape(n, f, t, e).permute(0, 3, 1, 2)
    assert rel_max(got.numpy(), ref.numpy()) < 2e-5
    pe = UO.positional_table(f, e)
    y = H.add_position(rows, pe.to(dev), t, 1, f, ld).view(n, e, f, ld)[..., :t].cpu()
    assert torch.allclose(y, x + pe.t().reshape(1, e, f, 1), atol=1e-6)


@pytest.mark.parametrize("hid,t", [(64, 13), (128, 10), (64, 16)])
def test_lstm_kernel_row_sequences_with_partial_step_group(H, dev, hid, t):
    """DPCRN's inter pass: one sequence per frequency row (q_stride = ld), consecutive frames, a step count that is
    not a multiple of the 16-byte step group."""
    from puresound_amd import _abi
    from puresound_amd.nnet._plans import lstm_plan
    n, c, f = 2, 12, 5
    m, sd = _lstm_sd(c, hid, False, 160)
    x = _rand((n, c, f, t), 161)
    ref, _ = DP.lstm(x.permute(0, 2, 3, 1).reshape(n * f, t, c), sd, "", False)
    p = lstm_plan(m.to(dev), torch.device(dev))
    xp = H.pad_rows(x.reshape(n, c * f, t).to(dev)).view(n, c, f, -1)
    ld = xp.shape[-1]
    frames = (f - 1) * ld + t
    gx, _ = H.conv1x1(xp.view(n, c, f * ld), frames, p["wih"], p["rows"], None, p["bias"])
    for flags in (0, 4, 8, 2):
        old = _abi.lib().ps_debug_flags(flags)
        try:
            hout, _ = H.lstm(gx, p["whh_t"], hid, 1, f, ld, t, 1)
            torch.cuda.synchronize()
        finally:
            _abi.lib().ps_debug_flags(old)
        got = hout.view(n, hid, f, ld)[..., :t].cpu().permute(0, 2, 3, 1).reshape(n * f, t, hid)
        assert rel_max(got.numpy(), ref.numpy()) < 2e-5, flags


# ------------------------------------------------------------------------------------------------
# signal scores (ps_wave_moments_f64) and the multi-output wrapper
# ------------------------------------------------------------------------------------------------
DB_TOL = 2e-3  # dB, absolute


def test_wave_moments_kernel(H, dev):
    a, b = _rand((7, 20011), 301) + 0.1, _rand((7, 20011), 302) - 0.05
    m = H.wave_moments(a.to(dev), b.to(dev)).cpu().numpy()
    ad, bd = a.double().numpy(), b.double().numpy()
    want = np.stack([ad.sum(1), bd.sum(1), (ad * ad).sum(1), (bd * bd).sum(1), (ad * bd).sum(1)], 1)
    np.testing.assert_allclose(m, want, rtol=1e-12, atol=1e-9)
    # strided rows (views of a wider buffer), one short row
    wide = _rand((3, 9000), 303).to(dev)
    m2 = H.wave_moments(wide[:, 100:5100], wide[:, 3000:8000]).cpu().numpy()
    w = wide.cpu().double().numpy()
    np.testing.assert_allclose(m2[:, 4], (w[:, 100:5100] * w[:, 3000:8000]).sum(1), rtol=1e-12, atol=1e-9)
    with pytest.raises(RuntimeError):
        H.wave_moments(a.to(dev), b[:, :100].to(dev))


def test_sdr_scores_match_reference_on_hip(PA, dev, golden_dir):
    from puresound_amd.nnet.loss.sdr import SDRLoss, inactive_sdr_loss, l2_norm, si_snr
    g = _load(golden_dir, "loss_sdr_modes")
    est, ref, est3, ref3, labels = (x.to(dev) for x in cases.loss_inputs(cases.CASES["loss_sdr_modes"]))
    for mode in ("sisnr", "sdsdr", "sdr", "tsdr", "sasdr", "sasisnr", "satsdr"):
        a, b = (est3, ref3) if mode.startswith("sa") else (est, ref)
        np.testing.assert_allclose(SDRLoss.init_mode(mode, reduction=False)(a, b).cpu().numpy(), g[mode], atol=DB_TOL, rtol=0)
        np.testing.assert_allclose(SDRLoss.init_mode(mode, reduction=True)(a, b).cpu().numpy(), g[mode + "_mean"],
                                   atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(SDRLoss.init_mode("sisnr", reduction=False)(est, ref, labels).cpu().numpy(),
                               g["sisnr_inactive"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(SDRLoss.init_mode("sisnr", reduction=False, threshold=-20.0)(est, ref).cpu().numpy(),
                               g["sisnr_threshold"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(SDRLoss(scaled=True, zero_mean=False, reduction=False)(est, ref).cpu().numpy(),
                               g["raw_no_zero_mean"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(si_snr(est, ref, reduction=False).cpu().numpy(), g["si_snr"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(inactive_sdr_loss(est, ref, reduction=False).cpu().numpy(), g["inactive_sdr"],
                               atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(l2_norm(est, ref).cpu().numpy(), (est.cpu().double() * ref.cpu().double()).sum(-1, keepdim=True),
                               rtol=1e-6)
    with pytest.raises(NameError):
        SDRLoss.init_mode("snr")
    with pytest.raises(AssertionError):
        SDRLoss.init_mode("sasdr")(est, ref)  # source-aggregated modes need [N, M, L]
    with pytest.raises(RuntimeError):
        si_snr(est.cpu(), ref.cpu())  # no CPU fallback


@pytest.mark.parametrize("name", ["simo_free", "simo_stft"])
def test_simo_wrapper_matches_reference_on_hip(PA, dev, golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    noisy = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    wav = model.inference(noisy).cpu().numpy()
    assert wav.shape == g["wav"].shape
    sl = slice(16, -16) if c["enc"]["kind"] == "stft" else slice(None)
    assert rel_max(wav[..., sl], g["wav"][..., sl]) < TOL
    ref_clean = det_wave(c["seed"] + 1, c["B"] * c["heads"], c["L_ref"]).reshape(c["B"], c["heads"], c["L_ref"]).to(dev)
    labels = torch.zeros(c["B"], c["heads"], dtype=torch.bool, device=dev)
    labels[0, 1] = True
    np.testing.assert_allclose(model(noisy, ref_clean, labels).cpu().numpy(), g["loss"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(model(noisy, ref_clean, torch.zeros_like(labels)).cpu().numpy(), g["loss_all_active"],
                               atol=DB_TOL, rtol=0)


# ------------------------------------------------------------------------------------------------
# uninitialised memory: pad frames / scratch buffers come from torch.empty, i.e. from whatever the caching allocator
# last held.  Nothing read from there may reach a result (a 0 * NaN in a statistics mask once did).
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_free", "tiny_free_relu_causal", "tiny_stft", "cfg1_short", "cfg2_short", "cfg3_short",
                                  "cfg4_short", "cfg4_tse_short", "tse_unet_tcn_causal_short", "tse_unet_tcn_short",
                                  "ns_dpcrn_short", "ns_dparn_short", "tse_skim_causal_short", "tse_skim_fbank_short",
                                  "cfg3_causal_short"])
@pytest.mark.parametrize("gemm", ["fp32", "bf16x3", "fp16x2"])
def test_results_do_not_depend_on_uninitialised_memory(PA, dev, golden_dir, name, gemm):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    torch.cuda.empty_cache()
    junk = [torch.full((1 << 24,), float("nan"), device=dev) for _ in range(12)]  # 768 MiB of NaN back into the cache
    del junk
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    if hasattr(model.masker, "set_gemm_precision"):
        model.masker.set_gemm_precision(gemm)
    elif gemm != "fp32":
        pytest.skip("no GEMM arithmetic switch on this masker")
    noisy = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]).to(dev) if "L_enroll" in c else None
    for _ in range(3):  # (the first call may still find clean blocks)
        out = model.inference(noisy, enroll) if enroll is not None else model.inference(noisy)
        assert torch.isfinite(out).all()
        sl = slice(16, -16) if c["enc"]["kind"] == "stft" else slice(None)
        assert rel_max(out.cpu().numpy()[:, sl], g["wav"][:, sl]) < TOL


# ------------------------------------------------------------------------------------------------
# hipGraph replay of a whole inference call (small batches)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["tiny_free", "cfg2_short", "cfg3_short", "cfg4_short", "cfg3_causal_short"])
def test_graphed_inference_is_bit_identical(PA, dev, name):
    from puresound_amd.graphs import GraphedInference
    c = cases.CASES[name]
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    fast = GraphedInference(model)
    for seed, batch in ((c["seed"], c["B"]), (c["seed"] + 7, c["B"]), (c["seed"] + 9, 1)):  # replay, replay, new shape
        noisy = det_wave(seed, batch, c["L"]).to(dev)
        enroll = det_wave(seed + 1, batch, c["L_enroll"]).to(dev) if "L_enroll" in c else None
        eager = model.inference(noisy, enroll) if enroll is not None else model.inference(noisy)
        a = fast(noisy, enroll)
        b = fast(noisy, enroll)
        assert torch.equal(a, eager) and torch.equal(b, eager) and a.data_ptr() != b.data_ptr()
    assert len(fast._graphs) == 2
    sd = det_state_dict(model)
    key = next(k for k in sd if k.endswith("weight") and sd[k].dim() > 1)
    sd[key] = sd[key] * 1.5  # a weight update drops the graphs (same signature as the kernel plans)
    model.load_state_dict(sd)
    noisy = det_wave(c["seed"], c["B"], c["L"]).to(dev)
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]).to(dev) if "L_enroll" in c else None
    eager = model.inference(noisy, enroll) if enroll is not None else model.inference(noisy)
    assert torch.equal(fast(noisy, enroll), eager)
    with pytest.raises(RuntimeError):
        fast(noisy.cpu())
