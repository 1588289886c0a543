"""Pin the CPU oracle (oracle/separator_oracle.py) against vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import cases
from conftest import rel_max
from detweights import det_state_dict, det_wave
from oracle import separator_oracle as O
import puresound_amd.nnet as PA

TOL = 1e-4  # north-star tolerance (max-abs relative to max|ref|); the oracle lands far inside it


def edge_ok(a, b, rtol=1e-3):
    return bool(np.all(np.abs(a - b) <= rtol * np.maximum(np.abs(b), 1.0)))


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def _sd(name):
    return det_state_dict(cases.build(PA.NS, name), mode=cases.CASES[name].get("weights", "plain"))


WRAP = [n for n, c in cases.CASES.items() if c["kind"] == "wrap"]
MASK = [n for n, c in cases.CASES.items() if c["kind"] == "masker"]
ENC = [n for n, c in cases.CASES.items() if c["kind"] == "encdec"]


@pytest.mark.parametrize("name", WRAP)
def test_wrapper_inference_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    noisy = det_wave(c["seed"], c["B"], c["L"], c.get("amp", 0.5))
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]) if "L_enroll" in c else None
    taps = {}
    torch.manual_seed(c["seed"])  # (SpecAugment -- tse_skim_v2_short -- draws its mask from the global generator)
    wav = O.inference(noisy, sd, cases.oracle_cfg(name), enroll, taps)
    assert wav.shape == g["wav"].shape
    edge = 16 if c["enc"]["kind"] == "stft" else 0  # iSTFT edges are ill-conditioned
    sl = slice(edge, wav.shape[1] - edge) if edge else slice(None)
    assert rel_max(wav.numpy()[:, sl], g["wav"][:, sl]) < TOL
    assert rel_max(taps["wav_preclamp"].numpy()[:, sl], g["wav_preclamp"][:, sl]) < TOL
    if edge:
        # edge samples are divided by a window-sum as small as 1.4e-9: compare them element-relative
        assert edge_ok(taps["wav_preclamp"].numpy(), g["wav_preclamp"])
        assert np.all(wav.numpy()[:, 0] == 0)  # window-sum is 0 at sample 0 -> never divided
    if "dvec" in g:
        assert rel_max(taps["dvec"].numpy(), g["dvec"]) < TOL
    if "mask_sub" in g:
        assert rel_max(taps["mask"][:, ::7, ::5].numpy(), g["mask_sub"]) < TOL
        if "feats_sub" in g:
            assert rel_max(taps["feats"][:, ::7, ::5].numpy(), g["feats_sub"]) < TOL
        if "block0_sub" in g:
            assert rel_max(taps["block0"][:, ::7, ::5].numpy(), g["block0_sub"]) < TOL
    # fixtures must exercise the clamp but not be saturated
    if c["wrap"].get("output_constraint", "linear") == "linear":
        frac = float((np.abs(g["wav_preclamp"]) > 1).mean())
        assert frac < 0.9


@pytest.mark.parametrize("name", MASK)
def test_masker_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    sd = {"masker." + k: v for k, v in _sd(name).items()}
    args = cases.full_masker_args(cases.CASES[name]["masker"])
    dvec = torch.tensor(g["dvec"]) if "dvec" in g else None
    y = O.conv_tasnet(torch.tensor(g["x"]), sd, "masker.", args, dvec)
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", ENC)
def test_encdec_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = {"encoder." + k: v for k, v in _sd(name).items()}
    wav = det_wave(c["seed"], c["B"], c["L"])
    enc = c["enc"]
    if enc["kind"] == "free":
        feats = O.free_encode(wav, sd["encoder.encoder.weight"], enc["hop"], enc.get("relu", False))
        rec = O.free_decode(torch.tensor(g["feats"]), sd["encoder.decoder.weight"], enc["hop"])
        assert rel_max(rec.numpy(), g["rec"]) < TOL
    else:
        feats = O.stft_encode(wav, sd["encoder.encoder.wsin"], sd["encoder.encoder.wcos"], enc["hop"])
        rec = O.istft_decode(torch.tensor(g["feats"]), sd, "encoder.encoder.", enc["hop"])
        assert rel_max(rec.numpy()[:, 16:-16], g["rec"][:, 16:-16]) < TOL
        assert edge_ok(rec.numpy(), g["rec"])
    assert feats.shape == g["feats"].shape
    assert rel_max(feats.numpy(), g["feats"]) < TOL
    # length law: L_out = (T-1)*hop + win
    win = enc.get("win", enc.get("n_fft"))
    t = (c["L"] - win) // enc["hop"] + 1
    assert rec.shape[1] == (t - 1) * enc["hop"] + win


@pytest.mark.parametrize("name", sorted(cases.PARAM_COUNTS))
def test_param_counts_known_answers(golden_dir, name):
    g = _load(golden_dir, name)
    assert int(g["n_params"]) == cases.PARAM_COUNTS[name]
    model = cases.build(PA.NS, name)
    assert sum(p.numel() for p in model.parameters()) == cases.PARAM_COUNTS[name]


def test_overlap_add_variants_agree():
    x = torch.randn(2, 9, 20)
    assert torch.allclose(O.overlap_add_sum(x, 6), O.overlap_add_sum_fast(x, 6), atol=1e-6)


# ------------------------------------------------------------------------------------------------
# recurrent maskers (oracle/dualpath_oracle.py)
# ------------------------------------------------------------------------------------------------
from oracle import dualpath_oracle as DP  # noqa: E402

LOBE = [n for n, c in cases.CASES.items() if c["kind"] == "lobe"]


@pytest.mark.parametrize("name", LOBE)
def test_depthwise_separable_lobe_oracle_matches_reference(golden_dir, name):
    """DepthwiseSeparableConv1d on its own, with the hid_channels transform and the skip connection (lobe/cnn.py:84-106)."""
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = {k: v.double() for k, v in det_state_dict(cases.build(PA.NS, name)).items()}
    kw = c["kw"]
    y = O.ds_conv(torch.tensor(g["x"]).double(), sd, "", kw.get("kernel", 3), kw.get("dilation", 1),
                  kw.get("causal", False), kw["norm_cls"], kw.get("stride", 1))
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL


RNN = [n for n, c in cases.CASES.items() if c["kind"] == "rnn"]
STREAM = [n for n, c in cases.CASES.items() if c["kind"] == "stream"]


@pytest.mark.parametrize("name", RNN)
def test_recurrent_masker_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    fn = {"DPRNN": DP.dprnn, "SkiM": DP.skim}[c["cls"]]
    embed = torch.tensor(g["embed"]) if "embed" in g else None
    y = fn(torch.tensor(g["x"]), sd, "", cases.rnn_args(c), embed)
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", STREAM)
def test_streaming_skim_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    args = cases.rnn_args(c)
    x, d = torch.tensor(g["x"]), torch.tensor(g["embed"])
    k, frames = args["seg_size"], c["frames"]
    # offline
    y = DP.skim(x, sd, "", args, d)
    assert rel_max(y.numpy(), g["y_offline"]) < TOL
    # whole-segment chunks, states carried by the caller
    ys, seg_h, seg_c, mem_h, mem_c = [], None, None, None, None
    for i in range(frames // k):
        o, seg_h, mem_h, seg_c, mem_c = DP.skim_step_chunk(x[..., i * k:(i + 1) * k].transpose(1, 2), sd, "", args,
                                                           seg_h, mem_h, seg_c, mem_c, d)
        ys.append(o)
    y_chunk = torch.cat(ys, -1)
    assert rel_max(y_chunk.numpy(), g["y_chunk"]) < TOL
    assert rel_max(torch.stack(seg_h).numpy(), g["chunk_seg_h"]) < TOL
    assert rel_max(torch.stack([torch.stack(p) for p in mem_h]).numpy(), g["chunk_mem_h"]) < TOL
    # frame by frame with module-held state
    st = DP.SkimStream(sd, "", args, streams=1)
    y_frame = torch.cat([st.step_frame(x[..., f].reshape(1, 1, -1), d) for f in range(frames)], -1)
    assert rel_max(y_frame.numpy(), g["y_frame"]) < TOL
    assert rel_max(torch.stack(st.seg_h).numpy(), g["frame_seg_h"]) < TOL
    assert rel_max(torch.stack(st.seg_c).numpy(), g["frame_seg_c"]) < TOL
    # the reference's own property (test/test_streaming.py:61-116): streaming == offline on whole segments
    whole = frames // k * k
    assert float((y_chunk - y[..., :whole]).abs().mean()) < 1e-6
    assert float((y_frame[..., :whole] - y[..., :whole]).abs().mean()) < 1e-6
    # several streams at once == each stream alone
    st2 = DP.SkimStream(sd, "", args, streams=2)
    x2 = torch.cat([x, x.flip(-1)], 0)
    d2 = torch.cat([d, d * 0.5 + 0.1], 0)
    n_f = min(frames, 2 * k + 3)
    y2 = torch.cat([st2.step_frame(x2[..., f].reshape(2, 1, -1), d2) for f in range(n_f)], -1)
    assert rel_max(y2[0:1].numpy(), g["y_frame"][..., :n_f]) < TOL


def test_demo_harness_matches_reference(golden_dir):
    """DemoTseNet.streaming_inference_chunk on three 320-sample chunks (egs/tse/demo/utils.py:78-128)."""
    name = "cfg5_demo"
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    h = c["harness"]
    demo = cases.build_demo(PA.NS, c)
    sd = det_state_dict(demo)
    assert sum(v.numel() for k, v in sd.items()) == int(g["harness_n_params"])
    wav = det_wave(c["seed"] + 200, 1, h["chunks"] * h["chunk"])
    d = torch.tensor(g["embed"])
    st = DP.DemoStream(sd, cases.rnn_args(c), 1, h["win"], h["hop"])
    pre = None
    for i in range(h["chunks"]):
        pre = st.step_chunk(wav[:, i * h["chunk"]:(i + 1) * h["chunk"]], d, pre)
    assert pre.shape[-1] == g["harness_wav"].shape[-1]
    assert rel_max(pre[0].numpy(), g["harness_wav"]) < TOL


def test_split_merge_identity():
    """test/test_lobe.py:50-54: merge(split(x)) == x."""
    x = torch.rand(3, 7, 53)
    for k in (4, 6, 10):
        seg, rest = DP.split_overlap(x, k)
        assert torch.allclose(DP.merge_overlap(seg, rest), x, atol=1e-7)


# ------------------------------------------------------------------------------------------------
# 2-D convolutional maskers (oracle/unet_oracle.py)
# ------------------------------------------------------------------------------------------------
from oracle import unet_oracle as UO  # noqa: E402

UNET = [n for n, c in cases.CASES.items() if c["kind"] == "unet"]


@pytest.mark.parametrize("name", UNET)
def test_unet_family_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    args = cases.unet_args(c)
    x = torch.tensor(g["x"])
    if c["oracle"] == "unet_tcn":
        y = UO.unet_tcn(x, sd, "", args, torch.tensor(g["embed"]))
    else:
        y = {"unet": UO.unet, "dpcrn": UO.dpcrn, "dparn": UO.dparn}[c["oracle"]](x, sd, "", args)
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "fbank"])
def test_fbank_encoder_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    wav = det_wave(c["seed"], c["B"], c["L"])
    y = O.fbank_encode(wav, sd, "encoder.", c["kw"]["hop_length"], c["kw"]["trainable"])
    assert y.shape == g["feats"].shape
    assert rel_max(y.numpy(), g["feats"]) < TOL


# ------------------------------------------------------------------------------------------------
# signal scores (loss/sdr.py) and the multi-output wrapper (base_nn.py:780-939)
# ------------------------------------------------------------------------------------------------
from oracle import loss_oracle as LO  # noqa: E402

DB_TOL = 2e-3  # dB, absolute: the scores are logarithms of ratios of fp32 sums (the reference's own fp32 noise at 60 dB)


def test_sdr_scores_match_reference(golden_dir):
    g = _load(golden_dir, "loss_sdr_modes")
    est, ref, est3, ref3, labels = cases.loss_inputs(cases.CASES["loss_sdr_modes"])
    for mode in LO.MODES:
        f = LO.mode_flags(mode)
        a, b = (est3, ref3) if f["source_aggregated"] else (est, ref)
        np.testing.assert_allclose(LO.sdr_loss(a, b, reduction=False, **f).numpy(), g[mode], atol=DB_TOL, rtol=0)
        np.testing.assert_allclose(LO.sdr_loss(a, b, reduction=True, **f).numpy(), g[mode + "_mean"], atol=DB_TOL, rtol=0)
    f = LO.mode_flags("sisnr")
    np.testing.assert_allclose(LO.sdr_loss(est, ref, reduction=False, inactive_labels=labels, **f).numpy(),
                               g["sisnr_inactive"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(LO.sdr_loss(est, ref, reduction=False, threshold=-20.0, **f).numpy(),
                               g["sisnr_threshold"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(LO.sdr_loss(est, ref, scaled=True, zero_mean=False, reduction=False).numpy(),
                               g["raw_no_zero_mean"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(LO.si_snr(est, ref, reduction=False).numpy(), g["si_snr"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(LO.inactive_sdr_loss(est, ref, reduction=False).numpy(), g["inactive_sdr"], atol=DB_TOL,
                               rtol=0)
    with pytest.raises(NameError):
        LO.mode_flags("snr")


@pytest.mark.parametrize("name", ["simo_free", "simo_stft"])
def test_simo_wrapper_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    model = cases.build(PA.NS, name)
    sd = {k: v.float() for k, v in det_state_dict(model).items()}
    cfg = cases.oracle_cfg(name)
    noisy = det_wave(c["seed"], c["B"], c["L"])
    wav = LO.simo_inference(noisy, sd, cfg)
    sl = slice(16, -16) if c["enc"]["kind"] == "stft" else slice(None)
    assert rel_max(wav.numpy()[..., sl], g["wav"][..., sl]) < TOL
    ref_clean = det_wave(c["seed"] + 1, c["B"] * c["heads"], c["L_ref"]).reshape(c["B"], c["heads"], c["L_ref"])
    labels = torch.zeros(c["B"], c["heads"], dtype=torch.bool)
    labels[0, 1] = True
    np.testing.assert_allclose(LO.simo_forward(noisy, ref_clean, sd, cfg, labels).numpy(), g["loss"], atol=DB_TOL, rtol=0)
    np.testing.assert_allclose(LO.simo_forward(noisy, ref_clean, sd, cfg, torch.zeros_like(labels)).numpy(),
                               g["loss_all_active"], atol=DB_TOL, rtol=0)


def test_mask_functions_and_magphase_match_reference(golden_dir):
    """apply_tf_masks / get_mask / _apply_complex_mask_on_polar (base_nn.py:41-190) and the conv-STFT's "MagPhase"
    output (lobe/encoder.py:384-389): the oracle's restatements against the imported reference's values."""
    import torch.nn as nn
    import puresound_amd.nnet as PA
    c = cases.CASES["mask_functions"]
    g = _load(golden_dir, "mask_functions")
    tf_rep, mask, wav = cases.func_inputs(c)
    for con in ("linear", "relu", "sigmoid"):
        assert rel_max(O.get_mask(mask, con).numpy(), g["get_mask_" + con]) < TOL
    assert rel_max(O.apply_tf_masks(tf_rep, mask, "complex", "complex").numpy(), g["complex_complex"]) < TOL
    assert rel_max(O.apply_tf_masks(tf_rep, mask, "real", "real").numpy(), g["real_real"]) < TOL
    re, im = torch.chunk(tf_rep, 2, dim=1)
    mre, mim = torch.chunk(mask, 2, dim=1)
    got = O.apply_complex_mask_on_polar(torch.stack([re, im], -1), torch.stack([mre, mim], -1))
    assert rel_max(got.numpy(), g["polar"]) < TOL
    with pytest.raises(RuntimeError):
        O.apply_tf_masks(tf_rep, mask, "polar", "polar")
    for tr in (True, False):
        enc = PA.ConvEncDec(fft_length=c["n_fft"], win_type="hann", win_length=c["n_fft"], hop_length=c["hop"],
                            trainable=tr, output_format="MagPhase")
        sd = det_state_dict(enc)
        got = O.stft_magphase(wav, sd["encoder.wsin"], sd["encoder.wcos"], c["hop"], tr).numpy()
        want = g["magphase_trainable" if tr else "magphase_fixed"]
        assert rel_max(got[..., 0], want[..., 0]) < TOL
        # phases: compare on the unit circle (atan2 is discontinuous at +-pi) where the bin is not numerically empty
        big = want[..., 0] > 1e-3 * want[..., 0].max()
        assert np.abs(np.exp(1j * got[..., 1]) - np.exp(1j * want[..., 1]))[big].max() < 1e-3


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "atten"])
def test_mha_self_atten_layer_matches_reference(golden_dir, name):
    """MhaSelfAttenLayer on its own (lobe/attention.py:115-232), plain and "improved" (LSTM feed-forward), causal or not."""
    from oracle import unet_oracle as UO
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = {k: v.double() for k, v in det_state_dict(cases.build(PA.NS, name)).items()}
    kw = c["kw"]
    y = UO.mha_self_atten_layer(torch.tensor(g["x"]).double(), sd, "", c["args"][2], kw["position_encoding"], c["causal"],
                                kw["improved"], kw.get("bidirectional", False))
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL


@pytest.mark.parametrize("name", [n for n, c in cases.CASES.items() if c["kind"] == "single_rnn"])
def test_single_rnn_matches_reference(golden_dir, name):
    """SingleRNN on its own (lobe/rnn.py:9-55) with each cell type its constructor accepts: LSTM, GRU, RNN."""
    from oracle import unet_oracle as UO
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = {k: v.double() for k, v in det_state_dict(cases.build(PA.NS, name)).items()}
    y = UO.single_rnn(torch.tensor(g["x"]).double(), sd, "", c["kw"]["bidirectional"], c["args"][0].upper())
    assert y.shape == g["y"].shape
    assert rel_max(y.numpy(), g["y"]) < TOL
