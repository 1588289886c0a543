"""Case table shared by make_golden.py (reference side) and the parity tests (oracle / HIP side).

A case names a constructor recipe that is valid BOTH for the reference's classes and for this repo's
mirror classes (same names, same positional order), the oracle config that describes the same model,
and the synthetic inputs.  `build(ns, name)` instantiates it from whichever namespace is handed in.
"""
from types import SimpleNamespace

CTN_FULL = dict(tcn_kernel=3, tcn_dim=256, repeat_tcn=3, tcn_dilated_basic=2, per_tcn_stack=8,
                tcn_norm="gLN", dconv_norm="gGN", causal=False, tcn_layer="normal")


def masker_args(input_dim, embed_dim, embed_norm, tcn_with_embed, **kw):
    a = dict(input_dim=input_dim, embed_dim=embed_dim, embed_norm=embed_norm, tcn_with_embed=list(tcn_with_embed))
    a.update(kw)
    return a


# name -> spec.  "wrap" cases go through SoTaskWrapModule.inference; others are module-level.
CASES = {
    # ---- end-to-end wrapper cases (BASELINE configs) --------------------------------------
    "cfg2_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                       masker=masker_args(512, 0, False, [0] * 8, **CTN_FULL),
                       wrap=dict(mask_constraint="ReLU"), B=2, L=4000, seed=1234),
    "cfg2_full": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                      masker=masker_args(512, 0, False, [0] * 8, **CTN_FULL),
                      wrap=dict(mask_constraint="ReLU"), B=1, L=64000, seed=1234),
    "cfg1_short": dict(kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
                       masker=masker_args(512, 0, False, [0] * 8, **CTN_FULL),
                       wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                 drop_first_bin=True), B=2, L=4000, seed=1234),
    "cfg1_full": dict(kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
                      masker=masker_args(512, 0, False, [0] * 8, **CTN_FULL),
                      wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                drop_first_bin=True), B=1, L=64000, seed=1234),
    "cfg3_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                       masker=masker_args(512, 192, True, [1, 0, 0, 0, 0, 0, 0, 0], **CTN_FULL),
                       speaker_net=dict(n_tcn=5, C=512, H=256, att=128, E=192),
                       wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=4000, seed=1234),
    # ---- reduced wrapper cases: odd sizes, ragged tails, sigmoid/linear constraints -----------
    "tiny_free": dict(kind="wrap", enc=dict(kind="free", win=16, hop=8, C=24),
                      masker=masker_args(24, 0, False, [0, 0, 0], tcn_kernel=3, tcn_dim=12, repeat_tcn=2,
                                         tcn_dilated_basic=2, per_tcn_stack=3, tcn_norm="gLN",
                                         dconv_norm="gGN", causal=False, tcn_layer="normal"),
                      wrap=dict(mask_constraint="sigmoid", output_constraint="sigmoid"), B=3, L=1003, seed=7),
    "tiny_free_relu_causal": dict(kind="wrap", enc=dict(kind="free", win=16, hop=8, C=24, relu=True),
                                  masker=masker_args(24, 0, False, [0, 0, 0], tcn_kernel=3, tcn_dim=12,
                                                     repeat_tcn=2, tcn_dilated_basic=2, per_tcn_stack=3,
                                                     tcn_norm="bN1d", dconv_norm="bN1d", causal=True,
                                                     tcn_layer="normal"),
                                  wrap=dict(mask_constraint="linear"), B=2, L=777, seed=8),
    "tiny_stft": dict(kind="wrap", enc=dict(kind="stft", n_fft=32, hop=8, drop_first_bin=True),
                      masker=masker_args(32, 0, False, [0, 0, 0], tcn_kernel=3, tcn_dim=12, repeat_tcn=2,
                                         tcn_dilated_basic=2, per_tcn_stack=3, tcn_norm="gLN",
                                         dconv_norm="gGN", causal=False, tcn_layer="normal"),
                      wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                drop_first_bin=True), B=2, L=515, seed=9),
    "tiny_stft_keepdc": dict(kind="wrap", enc=dict(kind="stft", n_fft=32, hop=8, drop_first_bin=False),
                             masker=masker_args(34, 0, False, [0, 0], tcn_kernel=3, tcn_dim=12, repeat_tcn=1,
                                                tcn_dilated_basic=2, per_tcn_stack=2, tcn_norm="gLN",
                                                dconv_norm="gGN", causal=False, tcn_layer="normal"),
                             wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                       drop_first_bin=False), B=1, L=300, seed=10),
    # ---- module-level cases -------------------------------------------------------------------
    "ctn_embed": dict(kind="masker", masker=masker_args(16, 6, True, [1, 0, 1], tcn_kernel=3, tcn_dim=8,
                                                         repeat_tcn=2, tcn_dilated_basic=2, per_tcn_stack=3,
                                                         tcn_norm="gLN", dconv_norm="gGN", causal=False,
                                                         tcn_layer="normal"),
                      B=3, T=61, seed=11),
    "ctn_dil3_k5": dict(kind="masker", masker=masker_args(20, 0, False, [0, 0, 0], tcn_kernel=5, tcn_dim=10,
                                                           repeat_tcn=1, tcn_dilated_basic=3, per_tcn_stack=3,
                                                           tcn_norm="gLN", dconv_norm="gGN", causal=False,
                                                           tcn_layer="normal"),
                        B=2, T=97, seed=12),
    "ctn_gated": dict(kind="masker", masker=masker_args(16, 6, False, [1, 0], tcn_kernel=3, tcn_dim=8,
                                                         repeat_tcn=2, tcn_dilated_basic=2, per_tcn_stack=2,
                                                         tcn_norm="gLN", causal=False, tcn_layer="gated"),
                      B=2, T=45, seed=13),
    "ctn_gated_causal": dict(kind="masker", masker=masker_args(16, 0, False, [0, 0], tcn_kernel=3, tcn_dim=8,
                                                                repeat_tcn=1, tcn_dilated_basic=2,
                                                                per_tcn_stack=2, tcn_norm="bN1d", causal=True,
                                                                tcn_layer="gated"),
                             B=2, T=45, seed=14),
    "tcn_cln": dict(kind="masker", masker=masker_args(16, 0, False, [0, 0], tcn_kernel=3, tcn_dim=8,
                                                       repeat_tcn=1, tcn_dilated_basic=2, per_tcn_stack=2,
                                                       tcn_norm="cLN", dconv_norm="cLN", causal=True,
                                                       tcn_layer="normal"),
                    B=2, T=33, seed=15),
    "enc_free": dict(kind="encdec", enc=dict(kind="free", win=32, hop=16, C=20), B=3, L=500, seed=16),
    "enc_free_relu_ragged": dict(kind="encdec", enc=dict(kind="free", win=20, hop=6, C=9, relu=True),
                                 B=2, L=211, seed=17),
    "enc_stft": dict(kind="encdec", enc=dict(kind="stft", n_fft=64, hop=16, drop_first_bin=False),
                     B=2, L=400, seed=18),
}

# parameter counts the reference documents / the survey measured (known answers)
PARAM_COUNTS = {"cfg2_short": 7977032, "cfg1_short": 8207432, "cfg3_short": 10108119}


def build_encoder(ns, enc):
    if enc["kind"] == "free":
        return ns.FreeEncDec(win_length=enc["win"], hop_length=enc["hop"], laten_length=enc["C"],
                             output_active=enc.get("relu", False))
    return ns.ConvEncDec(fft_length=enc["n_fft"], win_type="hann", win_length=enc["n_fft"],
                         hop_length=enc["hop"], trainable=True, output_format="Complex")


def build_masker(ns, m):
    kw = {k: v for k, v in m.items() if k not in ("input_dim", "embed_dim", "embed_norm")}
    return ns.ConvTasNet(m["input_dim"], m["embed_dim"], m["embed_norm"], **kw)


def build_speaker_net(ns, s):
    import torch.nn as nn
    return nn.ModuleList(
        [ns.TCN(s["C"], s["H"], 3, dilation=2 ** i, causal=False, tcn_norm="gLN", dconv_norm="gGN")
         for i in range(s["n_tcn"])]
        + [ns.AttentiveStatisticsPooling(s["C"], s["att"]), nn.Conv1d(s["C"] * 2, s["E"], 1, bias=False)])


def build(ns, name):
    """Instantiate case `name` from namespace `ns` (the reference's classes or this repo's)."""
    c = CASES[name]
    if c["kind"] == "wrap":
        kw = dict(c["wrap"])
        if "speaker_net" in c:
            kw["speaker_net"] = build_speaker_net(ns, c["speaker_net"])
        return ns.SoTaskWrapModule(encoder=build_encoder(ns, c["enc"]), masker=build_masker(ns, c["masker"]),
                                   verbose=False, **kw)
    if c["kind"] == "masker":
        return build_masker(ns, c["masker"])
    if c["kind"] == "encdec":
        return build_encoder(ns, c["enc"])
    raise KeyError(c["kind"])


def oracle_cfg(name):
    """The oracle's description of a wrapper case."""
    c = CASES[name]
    enc = dict(c["enc"])
    cfg = dict(encoder=enc, masker=full_masker_args(c["masker"]))
    cfg.update({k: v for k, v in c["wrap"].items() if k != "drop_first_bin"})
    if "speaker_net" in c:
        cfg["speaker_net"] = dict(n_tcn=c["speaker_net"]["n_tcn"])
    return cfg


def full_masker_args(m):
    a = dict(embed_norm=False, tcn_layer="normal", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2,
             per_tcn_stack=5, repeat_tcn=4, tcn_norm="gLN", dconv_norm="gGN", causal=False)
    a.update(m)
    return a


def namespace(**classes):
    return SimpleNamespace(**classes)
