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
    return det_state_dict(cases.build(PA.NS, name))


WRAP = [n for n, c in cases.CASES.items() if c["kind"] == "wrap"]
MASK = [n for n, c in cases.CASES.items() if c["kind"] == "masker"]
ENC = [n for n, c in cases.CASES.items() if c["kind"] == "encdec"]


@pytest.mark.parametrize("name", WRAP)
def test_wrapper_inference_matches_reference(golden_dir, name):
    c = cases.CASES[name]
    g = _load(golden_dir, name)
    sd = _sd(name)
    noisy = det_wave(c["seed"], c["B"], c["L"])
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]) if "L_enroll" in c else None
    taps = {}
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
        assert rel_max(taps["feats"][:, ::7, ::5].numpy(), g["feats_sub"]) < TOL
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
