"""Generate tests/golden/*.npz by importing the REAL reference (mcw519/PureSound) from /root/reference.

Run in the build container only (the reference does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every case in cases.py it builds the reference model, loads formula-generated weights
(detweights.det_state_dict), runs it on formula-generated inputs and stores inputs' seeds and the
reference's outputs (plus a few sub-sampled intermediate taps).  Fixtures are data only.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

# torchaudio is not installed; the reference imports it only for SpecAugment (lobe/trivial.py:7).
if "torchaudio" not in sys.modules:
    ta = types.ModuleType("torchaudio")
    taf = types.ModuleType("torchaudio.functional")

    def _mask_along_axis(specgram, mask_param, mask_value, axis, p=1.0):
        """torchaudio.functional.mask_along_axis as published (torchaudio 0.9 ... 2.x; requirements.txt:4 leaves the version
        open): one span per call, drawn from the global generator, for the whole batch."""
        if axis not in (1, 2):
            raise ValueError("Only Frequency and Time masking are supported")
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (specgram.size(axis) - value)
        mask_start = (min_value.long()).squeeze()
        mask_end = (min_value.long() + value.long()).squeeze()
        mask = torch.arange(0, specgram.shape[axis], device=specgram.device, dtype=specgram.dtype)
        mask = (mask >= mask_start) & (mask < mask_end)
        if axis == 1:
            mask = mask.unsqueeze(-1)
        return specgram.masked_fill(mask, mask_value)

    taf.mask_along_axis = _mask_along_axis
    ta.functional = taf
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = taf

from puresound.nnet.base_nn import SiMoTaskWrapModule, SoTaskWrapModule  # noqa: E402
from puresound.nnet.loss.sdr import SDRLoss, inactive_sdr_loss, si_snr  # noqa: E402
from puresound.nnet.conv_tasnet import TCN, ConvTasNet, GatedTCN  # noqa: E402
from puresound.nnet.lobe.encoder import ConvEncDec, FbankEnc, FreeEncDec  # noqa: E402
from puresound.nnet.lobe.pooling import AttentiveStatisticsPooling  # noqa: E402
from puresound.nnet.dprnn import DPRNN  # noqa: E402
from puresound.nnet.skim import SkiM  # noqa: E402
from puresound.streaming.skim_inference import StreamingSkiM  # noqa: E402
from puresound.nnet.unet import Unet, UnetTcn  # noqa: E402
from puresound.nnet.dpcrn import DPCRN  # noqa: E402
from puresound.nnet.dparn import DPARN, DPARN_Mout  # noqa: E402
from puresound.nnet.lobe.trivial import Magnitude, SpecAugment  # noqa: E402
from puresound.nnet.lobe.rnn import SingleRNN  # noqa: E402
from puresound.nnet.lobe.attention import MhaSelfAttenLayer  # noqa: E402
from puresound.nnet.lobe.cnn import DepthwiseSeparableConv1d  # noqa: E402

import cases  # noqa: E402
from detweights import det_state_dict, det_wave  # noqa: E402

REF = cases.namespace(SoTaskWrapModule=SoTaskWrapModule, SiMoTaskWrapModule=SiMoTaskWrapModule, SDRLoss=SDRLoss, TCN=TCN, ConvTasNet=ConvTasNet, GatedTCN=GatedTCN,
                      ConvEncDec=ConvEncDec, FreeEncDec=FreeEncDec,
                      AttentiveStatisticsPooling=AttentiveStatisticsPooling, DPRNN=DPRNN, SkiM=SkiM,
                      StreamingSkiM=StreamingSkiM, Unet=Unet, UnetTcn=UnetTcn, DPCRN=DPCRN, DPARN=DPARN, DPARN_Mout=DPARN_Mout,
                      Magnitude=Magnitude, SpecAugment=SpecAugment, FbankEnc=FbankEnc, SingleRNN=SingleRNN,
                      MhaSelfAttenLayer=MhaSelfAttenLayer,
                      DepthwiseSeparableConv1d=DepthwiseSeparableConv1d)


def sub(x: torch.Tensor, cs: int = 7, ts: int = 5) -> np.ndarray:
    """Sub-sample a [N,C,T(,2)] tap so fixtures stay small."""
    return x[:, ::cs, ::ts].contiguous().numpy()


@torch.no_grad()
def run_wrap(name, c):
    model = cases.build(REF, name).eval()
    sd = det_state_dict(model, mode=c.get("weights", "plain"))
    model.load_state_dict(sd)
    noisy = det_wave(c["seed"], c["B"], c["L"], c.get("amp", 0.5))
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]) if "L_enroll" in c else None
    out = {"n_params": np.int64(sum(p.numel() for p in model.parameters()))}
    torch.manual_seed(c["seed"])  # (SpecAugment draws from the global generator; nothing else on the path does)
    wav = model.inference(noisy.clone(), None if enroll is None else enroll.clone())
    torch.manual_seed(c["seed"])
    out["wav"] = wav.numpy()
    # step through the same reference methods to capture taps (base_nn.py:690-722)
    feats, enr = model._get_feature(noisy.clone(), None if enroll is None else enroll.clone())
    dvec = None
    if enr is not None and model.encoder_spk is not None and hasattr(model.encoder_spk, "n_banks"):
        pass  # _get_feature already ran the FbankEnc
    if enr is not None and model.embedding_free_tse:
        dvec = enr
    elif enr is not None:
        dvec = enr
        for layer in model.speaker_net:
            dvec = layer(dvec)
        dvec = dvec.squeeze(-1)
        out["dvec"] = dvec.numpy()
    mask = model.masker(feats, dvec) if dvec is not None else model.masker(feats)
    mask = model.get_mask(mask, model.mask_constraint)
    enh = model.apply_tf_masks(feats, mask, f_type=model.f_type, mask_type=model.mask_type)
    pre = model._get_waveform(enh)
    out["wav_preclamp"] = pre.numpy()
    small = c["L"] <= 4000
    if small and c["masker"].get("cls") in ("Unet", "UnetTcn", "DPCRN", "DPARN"):
        out["mask_sub"] = sub(mask)
    elif small and "cls" in c["masker"]:
        out["feats_sub"] = sub(feats)
        out["mask_sub"] = sub(mask)
    elif small:
        out["feats_sub"] = sub(feats)
        out["mask_sub"] = sub(mask)
        blk0 = model.masker.tcn_list[0][0]
        b0 = blk0(feats, torch.nn.functional.normalize(dvec, p=2, dim=1) if (dvec is not None and model.masker.embed_norm) else dvec) \
            if model.masker.tcn_with_embed[0] else blk0(feats)
        out["block0_sub"] = sub(b0)
    chk = model._wav_output_constrain(pre.clone(), mode=model.output_constraint)
    assert torch.equal(chk, wav), name
    return out


@torch.no_grad()
def run_masker(name, c):
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    m = c["masker"]
    g = np.random.Generator(np.random.Philox(key=c["seed"]))
    x = torch.tensor(g.uniform(-1, 1, (c["B"], m["input_dim"], c["T"])), dtype=torch.float32)
    out = {"x": x.numpy()}
    if m["embed_dim"] > 0:
        dvec = torch.tensor(g.uniform(-1, 1, (c["B"], m["embed_dim"])), dtype=torch.float32)
        out["dvec"] = dvec.numpy()
        y = model(x.clone(), dvec.clone())
    else:
        y = model(x.clone())
    out["y"] = y.numpy()
    return out


@torch.no_grad()
def run_encdec(name, c):
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    wav = det_wave(c["seed"], c["B"], c["L"])
    feats = model(wav.clone())
    rec = model.inverse(feats.clone())
    return {"feats": feats.numpy(), "rec": rec.numpy()}


@torch.no_grad()
def run_fbank(name, c):
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    wav = det_wave(c["seed"], c["B"], c["L"])
    return {"feats": model(wav.clone()).numpy()}


def _uniform(seed, shape, lo=-1.0, hi=1.0):
    g = np.random.Generator(np.random.Philox(key=seed))
    return torch.tensor(g.uniform(lo, hi, shape), dtype=torch.float32)


@torch.no_grad()
def run_lobe(name, c):
    """a lobe on its own: x [B, C, T] -> y"""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    x = _uniform(c["seed"], (c["B"], c["args"][0], c["T"]))
    return {"x": x.numpy(), "y": model(x.clone()).numpy()}


@torch.no_grad()
def run_atten(name, c):
    """MhaSelfAttenLayer on its own: x [B, C, T] -> y, forward(x, causal)"""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    x = _uniform(c["seed"], (c["B"], c["args"][0], c["T"]))
    return {"x": x.numpy(), "y": model(x.clone(), causal=c["causal"]).numpy()}


@torch.no_grad()
def run_single_rnn(name, c):
    """SingleRNN on its own: x [B, C, T] -> proj(rnn(x)) (lobe/rnn.py:37-55)"""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    x = _uniform(c["seed"], (c["B"], c["args"][1], c["T"]))
    return {"x": x.numpy(), "y": model(x.clone()).numpy()}


@torch.no_grad()
def run_rnn(name, c):
    """DPRNN / SkiM at module level: x [B,C,T] (+ embedding vector or enrolment features)."""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    x = _uniform(c["seed"], (c["B"], c["args"][0], c["T"]))
    out = {"x": x.numpy()}
    embed = None
    if c.get("embed") == "vec":
        embed = _uniform(c["seed"] + 100, (c["B"], c["kw"]["embed_dim"]))
    elif c.get("embed") == "feat":
        embed = _uniform(c["seed"] + 100, (c["B"], c["args"][0], c["Te"]))
    if embed is not None:
        out["embed"] = embed.numpy()
    y = model(x.clone(), None if embed is None else embed.clone())
    out["y"] = y.numpy()
    return out


@torch.no_grad()
def run_unet(name, c):
    """Unet / UnetTcn / DPCRN at module level: x [B, input_dim, T] (+ embedding)."""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    x = _uniform(c["seed"], (c["B"], c["kw"]["input_dim"], c["T"]))
    out = {"x": x.numpy()}
    if "embed" in c:
        e = _uniform(c["seed"] + 100, (c["B"], c["embed"]))
        out["embed"] = e.numpy()
        y = model(x.clone(), e.clone())
    else:
        y = model(x.clone())
    out["y"] = y.numpy()
    return out


@torch.no_grad()
def run_stream(name, c):
    """StreamingSkiM: offline forward, step_chunk over whole segments, step_frame over every frame; for the demo
    preset also the harness of egs/tse/demo/utils.py (DemoTseNet.streaming_inference_chunk) on three chunks."""
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    cin, e, k, frames = c["args"][0], c["kw"]["embed_dim"], c["kw"]["seg_size"], c["frames"]
    x = _uniform(c["seed"], (1, cin, frames), 0.0, 1.0)
    d = _uniform(c["seed"] + 100, (1, e), 0.0, 1.0)
    out = {"x": x.numpy(), "embed": d.numpy()}
    out["y_offline"] = model(x.clone(), d.clone()).numpy()
    ys, seg_h, seg_c, mem_h, mem_c = [], None, None, None, None
    for i in range(frames // k):
        o, seg_h, mem_h, seg_c, mem_c = model.step_chunk(x[..., i * k:(i + 1) * k].permute(0, 2, 1), seg_h, mem_h,
                                                         seg_c, mem_c, d)
        ys.append(o)
    out["y_chunk"] = torch.cat(ys, -1).numpy()
    out["chunk_seg_h"] = torch.stack(seg_h).numpy()
    out["chunk_mem_h"] = torch.stack([torch.stack(p) for p in mem_h]).numpy()
    model.init_status()
    yf = [model.step_frame(x[..., f].view(1, -1, 1), d) for f in range(frames)]
    out["y_frame"] = torch.cat(yf, -1).numpy()
    out["frame_seg_h"] = torch.stack(model.seg_lstm_h_states).numpy()
    out["frame_seg_c"] = torch.stack(model.seg_lstm_c_states).numpy()
    if "harness" in c:
        import importlib.util
        spec = importlib.util.spec_from_file_location("demo_utils", "/root/reference/egs/tse/demo/utils.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        net = mod.DemoTseNet()
        net.eval()
        sd = det_state_dict(net)
        net.load_state_dict(sd)
        net.masker.init_status()
        h = c["harness"]
        wav = det_wave(c["seed"] + 200, 1, h["chunks"] * h["chunk"])
        pre = None
        for i in range(h["chunks"]):
            pre = net.streaming_inference_chunk(wav[:, i * h["chunk"]:(i + 1) * h["chunk"]], d[0], pre)
        out["harness_wav"] = pre.numpy()
        out["harness_n_params"] = np.int64(sum(p.numel() for p in net.parameters()))
    return out


@torch.no_grad()
def run_loss(name, c):
    """Every SDRLoss alias (per-row values), the inactive-label path, a thresholded run, si_snr, inactive_sdr_loss."""
    est, ref, est3, ref3, labels = cases.loss_inputs(c)
    out = {}
    for mode in ("sisnr", "sdsdr", "sdr", "tsdr", "sasdr", "sasisnr", "satsdr"):
        agg = mode.startswith("sa")
        a, b = (est3, ref3) if agg else (est, ref)
        out[mode] = SDRLoss.init_mode(mode, reduction=False)(a.clone(), b.clone()).numpy()
        out[mode + "_mean"] = SDRLoss.init_mode(mode, reduction=True)(a.clone(), b.clone()).numpy()
    out["sisnr_inactive"] = SDRLoss.init_mode("sisnr", reduction=False)(est.clone(), ref.clone(), labels).numpy()
    out["sisnr_threshold"] = SDRLoss.init_mode("sisnr", reduction=False, threshold=-20.0)(est.clone(), ref.clone()).numpy()
    out["raw_no_zero_mean"] = SDRLoss(scaled=True, zero_mean=False, reduction=False)(est.clone(), ref.clone()).numpy()
    out["si_snr"] = si_snr(est.clone(), ref.clone(), reduction=False).numpy()
    out["inactive_sdr"] = inactive_sdr_loss(est.clone(), ref.clone(), reduction=False).numpy()
    return out


@torch.no_grad()
def run_simo(name, c):
    model = cases.build(REF, name).eval()
    model.load_state_dict(det_state_dict(model))
    noisy = det_wave(c["seed"], c["B"], c["L"])
    wav = model.inference(noisy.clone())
    ref_clean = det_wave(c["seed"] + 1, c["B"] * c["heads"], c["L_ref"]).reshape(c["B"], c["heads"], c["L_ref"])
    labels = torch.zeros(c["B"], c["heads"], dtype=torch.bool)
    labels[0, 1] = True
    loss = model(noisy.clone(), ref_clean.clone(), labels)
    loss_all_active = model(noisy.clone(), ref_clean.clone(), torch.zeros_like(labels))
    return {"wav": wav.numpy(), "loss": loss.numpy(), "loss_all_active": loss_all_active.numpy()}


@torch.no_grad()
def run_func(name, c):
    """apply_tf_masks / get_mask / the _apply_* helpers called directly on the reference's base class, and
    ConvEncDec(output_format="MagPhase").forward."""
    from puresound.nnet.base_nn import EncDecMaskerBaseModel
    m = EncDecMaskerBaseModel()
    tf_rep, mask, wav = cases.func_inputs(c)
    out = {}
    for con in ("linear", "relu", "sigmoid"):
        out["get_mask_" + con] = m.get_mask(mask.clone(), con).numpy()
    out["complex_complex"] = m.apply_tf_masks(tf_rep.clone(), mask.clone(), "complex", "complex").numpy()   # [N,C,T,2]
    out["real_real"] = m.apply_tf_masks(tf_rep.clone(), mask.clone(), "real", "real").numpy()
    re, im = torch.chunk(tf_rep, 2, dim=1)
    mre, mim = torch.chunk(mask, 2, dim=1)
    out["polar"] = m._apply_complex_mask_on_polar(torch.stack([re, im], -1), torch.stack([mre, mim], -1)).numpy()
    for tr in (True, False):
        enc = ConvEncDec(fft_length=c["n_fft"], win_type="hann", win_length=c["n_fft"], hop_length=c["hop"],
                         trainable=tr, output_format="MagPhase").eval()
        enc.load_state_dict(det_state_dict(enc))
        out["magphase_trainable" if tr else "magphase_fixed"] = enc(wav.clone()).numpy()
    return out


def dump_state_dict_keys():
    """Key -> shape of every reference state_dict the mirror modules must reproduce (drop-in checkpoints)."""
    import json
    out = {}
    for name, c in cases.CASES.items():
        if c["kind"] in ("loss", "func"):  # functions of tensors, no parameters
            continue
        model = cases.build(REF, name)
        out[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict_keys.json:", sum(len(v) for v in out.values()), "keys")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    dump_state_dict_keys()
    for name, c in cases.CASES.items():
        fn = {"wrap": run_wrap, "masker": run_masker, "encdec": run_encdec, "rnn": run_rnn,
              "lobe": run_lobe, "atten": run_atten, "single_rnn": run_single_rnn, "stream": run_stream, "unet": run_unet, "fbank": run_fbank, "loss": run_loss, "simo": run_simo, "func": run_func}[c["kind"]]
        if only and name not in only:
            continue
        out = fn(name, c)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:28s} -> {os.path.getsize(path) / 1024:8.1f} KiB  keys={sorted(out)}")


if __name__ == "__main__":
    main()
