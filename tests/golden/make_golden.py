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
    taf.mask_along_axis = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    ta.functional = taf
    sys.modules["torchaudio"] = ta
    sys.modules["torchaudio.functional"] = taf

from puresound.nnet.base_nn import SoTaskWrapModule  # noqa: E402
from puresound.nnet.conv_tasnet import TCN, ConvTasNet, GatedTCN  # noqa: E402
from puresound.nnet.lobe.encoder import ConvEncDec, FreeEncDec  # noqa: E402
from puresound.nnet.lobe.pooling import AttentiveStatisticsPooling  # noqa: E402

import cases  # noqa: E402
from detweights import det_state_dict, det_wave  # noqa: E402

REF = cases.namespace(SoTaskWrapModule=SoTaskWrapModule, TCN=TCN, ConvTasNet=ConvTasNet, GatedTCN=GatedTCN,
                      ConvEncDec=ConvEncDec, FreeEncDec=FreeEncDec,
                      AttentiveStatisticsPooling=AttentiveStatisticsPooling)


def sub(x: torch.Tensor, cs: int = 7, ts: int = 5) -> np.ndarray:
    """Sub-sample a [N,C,T(,2)] tap so fixtures stay small."""
    return x[:, ::cs, ::ts].contiguous().numpy()


@torch.no_grad()
def run_wrap(name, c):
    model = cases.build(REF, name).eval()
    sd = det_state_dict(model)
    model.load_state_dict(sd)
    noisy = det_wave(c["seed"], c["B"], c["L"])
    enroll = det_wave(c["seed"] + 1, c["B"], c["L_enroll"]) if "L_enroll" in c else None
    out = {"n_params": np.int64(sum(p.numel() for p in model.parameters()))}
    wav = model.inference(noisy.clone(), None if enroll is None else enroll.clone())
    out["wav"] = wav.numpy()
    # step through the same reference methods to capture taps (base_nn.py:690-722)
    feats, enr = model._get_feature(noisy.clone(), None if enroll is None else enroll.clone())
    dvec = None
    if enr is not None:
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
    if small:
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


def dump_state_dict_keys():
    """Key -> shape of every reference state_dict the mirror modules must reproduce (drop-in checkpoints)."""
    import json
    out = {}
    for name in cases.CASES:
        model = cases.build(REF, name)
        out[name] = {k: list(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("state_dict_keys.json:", sum(len(v) for v in out.values()), "keys")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    dump_state_dict_keys()
    for name, c in cases.CASES.items():
        fn = {"wrap": run_wrap, "masker": run_masker, "encdec": run_encdec}[c["kind"]]
        out = fn(name, c)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name:28s} -> {os.path.getsize(path) / 1024:8.1f} KiB  keys={sorted(out)}")


if __name__ == "__main__":
    main()
