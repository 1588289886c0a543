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
    # configs 2 and 3 with the weights a trained checkpoint may have (detweights mode "wild": norm gains over 2^-4 .. 2^4,
    # biases in +-2, PReLU slopes up to 3, weight rows spanning 2^18): the fp16x2 range logic end to end
    "cfg2_wild_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                            masker=masker_args(512, 0, False, [0] * 8, **CTN_FULL),
                            wrap=dict(mask_constraint="ReLU"), B=2, L=4000, seed=1234, weights="wild", amp=1e-3),
    "cfg3_wild_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                            masker=masker_args(512, 192, True, [1, 0, 0, 0, 0, 0, 0, 0], **CTN_FULL),
                            speaker_net=dict(n_tcn=5, C=512, H=256, att=128, E=192),
                            wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=4000, seed=1234, weights="wild",
                            amp=1e-3),
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
    # td_tse_conv_tasnet_v0_causal verbatim (egs/tse/model.py:142-183): causal bN1d masker, the non-causal gLN speaker net
    "cfg3_causal_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=512),
                              masker=masker_args(512, 192, True, [1, 0, 0, 0, 0, 0, 0, 0],
                                                 **dict(CTN_FULL, tcn_norm="bN1d", dconv_norm="bN1d", causal=True)),
                              speaker_net=dict(n_tcn=5, C=512, H=256, att=128, E=192),
                              wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=4000, seed=1234),
    # BASELINE config 4: DPRNN hyper-parameters of veve_dprnn_v0_causal (egs/tse/model.py:614-630) as a plain
    # separator, and the preset verbatim (embedding-free TSE: the enrolment pass seeds the inter-LSTM states)
    "cfg4_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
                       masker=dict(cls="DPRNN", args=(128, 64, 128),
                                   kw=dict(n_blocks=6, seg_size=20, seg_overlap=False, causal=True)),
                       wrap=dict(mask_constraint="ReLU"), B=2, L=4000, seed=1234),
    "cfg4_tse_short": dict(kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
                           masker=dict(cls="DPRNN", args=(128, 64, 128),
                                       kw=dict(n_blocks=6, seg_size=20, seg_overlap=False, causal=True, embed_dim=0,
                                               embed_norm=False, block_with_embed=(False,) * 6,
                                               embedding_free_tse=True)),
                           wrap=dict(mask_constraint="ReLU", embedding_free_tse=True), B=2, L=4000, L_enroll=3000,
                           seed=1234),
    # the real egs/ns model: ns_dpcrn_v0_causal verbatim (egs/ns/model.py:40-82), 1 380 043 parameters
    "ns_dpcrn_short": dict(kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
                           masker=dict(cls="DPCRN", args=(), oracle="dpcrn",
                                       kw=dict(input_type="RI", input_dim=512, activation_type="PReLU", norm_type="bN2d",
                                               dropout=0.1, channels=(1, 32, 32, 32, 64, 128), transpose_t_size=2,
                                               transpose_delay=False, skip_conv=False, kernel_t=(2, 2, 2, 2, 2),
                                               kernel_f=(5, 3, 3, 3, 3), stride_t=(1, 1, 1, 1, 1),
                                               stride_f=(2, 2, 1, 1, 1), dilation_t=(1, 1, 1, 1, 1),
                                               dilation_f=(1, 1, 1, 1, 1), delay=(0, 0, 0, 0, 0), rnn_hidden=128)),
                           wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                     drop_first_bin=True), B=2, L=4000, seed=1234),
    # two more TSE presets verbatim: tse_unet_tcn_v0_causal (egs/tse/model.py:246-306: STFT + UnetTcn(bN2d, gated bN1d
    # TCN, causal) + speaker net Magnitude -> 5 x GatedTCN(gLN) -> ASP -> 1x1) and tse_skim_v0_causal (:418-463)
    "tse_unet_tcn_causal_short": dict(
        kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
        masker=dict(cls="UnetTcn", args=(), oracle="unet_tcn",
                    kw=dict(embed_dim=192, embed_norm=True, input_type="RI", input_dim=512, activation_type="PReLU",
                            norm_type="bN2d", channels=(1, 32, 64, 128, 128, 128, 128), transpose_t_size=2,
                            transpose_delay=True, skip_conv=False, kernel_t=(2,) * 6, kernel_f=(5,) * 6,
                            stride_t=(1,) * 6, stride_f=(2,) * 6, dilation_t=(1,) * 6, dilation_f=(1,) * 6,
                            delay=(0,) * 6, tcn_layer="gated", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2,
                            per_tcn_stack=5, repeat_tcn=3, tcn_with_embed=[1, 0, 0, 0, 0], tcn_norm="bN1d",
                            dconv_norm="bN1d", causal=True)),
        speaker_net=dict(n_tcn=5, C=256, H=128, att=128, E=192, block="gated", magnitude=True),
        wrap=dict(mask_constraint="linear", drop_first_bin=True), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_unet_tcn_short": dict(   # tse_unet_tcn_v0 (egs/tse/model.py:184-244): gLN as the 2-D convolution norm, non-causal
        kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
        masker=dict(cls="UnetTcn", args=(), oracle="unet_tcn",
                    kw=dict(embed_dim=192, embed_norm=True, input_type="RI", input_dim=512, activation_type="PReLU",
                            norm_type="gLN", channels=(1, 32, 64, 128, 128, 128, 128), transpose_t_size=2,
                            transpose_delay=True, skip_conv=False, kernel_t=(2,) * 6, kernel_f=(5,) * 6,
                            stride_t=(1,) * 6, stride_f=(2,) * 6, dilation_t=(1,) * 6, dilation_f=(1,) * 6,
                            delay=(0,) * 6, tcn_layer="gated", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2,
                            per_tcn_stack=5, repeat_tcn=3, tcn_with_embed=[1, 0, 0, 0, 0], tcn_norm="gLN",
                            dconv_norm="gGN", causal=False)),
        speaker_net=dict(n_tcn=5, C=256, H=128, att=128, E=192, block="gated", magnitude=True),
        wrap=dict(mask_constraint="linear", drop_first_bin=True), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_unet_tcn_v1_short": dict(   # tse_unet_tcn_v1 (egs/tse/model.py:308-369): v0 with FiLM conditioning in the gated TCN
        kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
        masker=dict(cls="UnetTcn", args=(), oracle="unet_tcn",
                    kw=dict(embed_dim=192, embed_norm=True, input_type="RI", input_dim=512, activation_type="PReLU",
                            norm_type="gLN", channels=(1, 32, 64, 128, 128, 128, 128), transpose_t_size=2,
                            transpose_delay=True, skip_conv=False, kernel_t=(2,) * 6, kernel_f=(5,) * 6,
                            stride_t=(1,) * 6, stride_f=(2,) * 6, dilation_t=(1,) * 6, dilation_f=(1,) * 6,
                            delay=(0,) * 6, tcn_layer="gated", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2,
                            per_tcn_stack=5, repeat_tcn=3, tcn_with_embed=[1, 0, 0, 0, 0], tcn_norm="gLN",
                            dconv_norm="gGN", causal=False, tcn_use_film=True)),
        speaker_net=dict(n_tcn=5, C=256, H=128, att=128, E=192, block="gated", magnitude=True),
        wrap=dict(mask_constraint="linear", drop_first_bin=True), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_skim_v1_short": dict(   # tse_skim_v1_causal (egs/tse/model.py:465-507): a bidirectional SingleRNN as the speaker net
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        masker=dict(cls="SkiM", args=(128, 256, 128),
                    kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")),
        speaker_net=dict(block="rnn", C=128, H=192, att=128, E=192, bidirectional=True),
        wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_skim_v0_short": dict(   # tse_skim_v0 (egs/tse/model.py:371-416): the non-causal (bidirectional) SkiM preset
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        masker=dict(cls="SkiM", args=(128, 256, 128),
                    kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=False, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")),
        speaker_net=dict(n_tcn=5, C=128, H=256, att=128, E=192),
        wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=3000, seed=1234),
    # tse_skim_v2_causal (egs/tse/model.py:509-558) WITHOUT its SpecAugment layer, which draws random masks even in
    # eval mode (lobe/trivial.py:326-335) and so has no reproducible output: FbankEnc enrolment encoder + TCN speaker net
    "tse_skim_fbank_short": dict(
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        enc_spk=dict(kw=dict(trainable=False, output_format="Magnitude", n_banks=80)),
        masker=dict(cls="SkiM", args=(128, 256, 128),
                    kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")),
        speaker_net=dict(n_tcn=5, C=80, H=256, att=128, E=192),
        wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=3000, seed=1234),
    # tse_skim_v2_causal verbatim (egs/tse/model.py:509-558): SpecAugment(10, 0, 0.0) in front of the speaker net.  The layer
    # draws its mask from torch's global generator; generator and tests seed it with the case's seed before every pass.
    "tse_skim_v2_short": dict(
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        enc_spk=dict(kw=dict(trainable=False, output_format="Magnitude", n_banks=80)),
        masker=dict(cls="SkiM", args=(128, 256, 128),
                    kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")),
        speaker_net=dict(n_tcn=5, C=80, H=256, att=128, E=192, specaug=(10, 0, 0.0)),
        wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_skim_vad_short": dict(   # tse_skim_v0_causal_vad (egs/tse/model.py:560-606): sigmoid output, H = 64, 2 blocks
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        masker=dict(cls="SkiM", args=(128, 64, 128),
                    kw=dict(n_blocks=2, seg_size=150, seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1], embed_fusion="FiLM")),
        speaker_net=dict(n_tcn=5, C=128, H=256, att=128, E=192),
        wrap=dict(mask_constraint="ReLU", output_constraint="Sigmoid"), B=2, L=4000, L_enroll=3000, seed=1234),
    "tse_skim_causal_short": dict(
        kind="wrap", enc=dict(kind="free", win=32, hop=16, C=128, relu=True),
        masker=dict(cls="SkiM", args=(128, 256, 128),
                    kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=True, embed_dim=192, embed_norm=True,
                            block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM")),
        speaker_net=dict(n_tcn=5, C=128, H=256, att=128, E=192),
        wrap=dict(mask_constraint="ReLU"), B=2, L=4000, L_enroll=3000, seed=1234),
    # ns_dparn_v0_causal verbatim (egs/ns/model.py:128-171)
    "ns_dparn_short": dict(kind="wrap", enc=dict(kind="stft", n_fft=512, hop=128, drop_first_bin=True),
                           masker=dict(cls="DPARN", args=(), oracle="dparn",
                                       kw=dict(input_type="RI", input_dim=512, activation_type="PReLU", norm_type="bN2d",
                                               dropout=0.1, channels=(1, 32, 32, 32, 64, 128), transpose_t_size=2,
                                               transpose_delay=False, skip_conv=False, kernel_t=(2, 2, 2, 2, 2),
                                               kernel_f=(5, 3, 3, 3, 3), stride_t=(1, 1, 1, 1, 1),
                                               stride_f=(2, 2, 1, 1, 1), dilation_t=(1, 1, 1, 1, 1),
                                               dilation_f=(1, 1, 1, 1, 1), delay=(0, 0, 0, 0, 0), rnn_hidden=128,
                                               nhead=8)),
                           wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex",
                                     drop_first_bin=True), B=2, L=4000, seed=1234),
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
    # causal with the constructor's default gLN: the reference's gated block normalises over the T + padding frames it
    # trims only after out_conv (conv_tasnet.py:203-211); T + padding crosses a row-length boundary here
    "ctn_gated_causal_gln": dict(kind="masker", masker=masker_args(16, 6, True, [1, 0], tcn_kernel=3, tcn_dim=8,
                                                                    repeat_tcn=2, tcn_dilated_basic=2,
                                                                    per_tcn_stack=2, tcn_norm="gLN", causal=True,
                                                                    tcn_layer="gated"),
                                 B=2, T=127, seed=15),
    "tcn_cln": dict(kind="masker", masker=masker_args(16, 0, False, [0, 0], tcn_kernel=3, tcn_dim=8,
                                                       repeat_tcn=1, tcn_dilated_basic=2, per_tcn_stack=2,
                                                       tcn_norm="cLN", dconv_norm="cLN", causal=True,
                                                       tcn_layer="normal"),
                    B=2, T=33, seed=15),
    # ---- DepthwiseSeparableConv1d on its own (lobe/cnn.py:9-106): the hid_channels transform and the skip connection
    # that no Conv-TasNet preset uses, every norm the lobe accepts, causal and not
    # MhaSelfAttenLayer on its own (lobe/attention.py:115-232): the plain transformer block and the "improved" one (LSTM
    # in place of the first feed-forward Linear), forward(x, causal)
    "mha_plain": dict(kind="atten", cls="MhaSelfAttenLayer", args=(32, 48, 4),
                      kw=dict(dropout=0.0, improved=False, position_encoding=True), causal=False, B=3, T=37, seed=81),
    "mha_improved_bi": dict(kind="atten", cls="MhaSelfAttenLayer", args=(32, 24, 4),
                            kw=dict(dropout=0.0, improved=True, bidirectional=True, position_encoding=False),
                            causal=False, B=3, T=37, seed=82),
    "mha_improved_causal": dict(kind="atten", cls="MhaSelfAttenLayer", args=(24, 40, 2),
                                kw=dict(dropout=0.0, improved=True, bidirectional=False, position_encoding=False),
                                causal=True, B=2, T=50, seed=83),
    "dsc_plain_ggn": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(12, 20),
                          kw=dict(norm_cls="gGN", kernel=5, dilation=3), B=2, T=61, seed=61),
    "dsc_transform_skip_gln": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(12, 20),
                                   kw=dict(hid_channels=18, norm_cls="gLN", kernel=3, dilation=2, skip=True),
                                   B=2, T=70, seed=62),
    "dsc_causal_bn": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(10, 10),
                          kw=dict(hid_channels=16, norm_cls="bN1d", kernel=3, dilation=4, skip=True, causal=True),
                          B=3, T=50, seed=63, bn_stats=True),
    "dsc_causal_cln": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(10, 14),
                           kw=dict(norm_cls="cLN", kernel=2, dilation=1, skip=True, causal=True), B=2, T=33, seed=64),
    "dsc_stride2_gln": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(12, 20),
                            kw=dict(hid_channels=18, norm_cls="gLN", kernel=3, dilation=2, stride=2), B=2, T=71, seed=65),
    "dsc_stride3_cln": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(10, 14),
                            kw=dict(norm_cls="cLN", kernel=5, dilation=1, stride=3), B=2, T=50, seed=66),
    # causal with a stride: both-sided padding, every s-th frame, the last `padding` frames cut (cnn.py:62-71, 100-101)
    "dsc_stride2_causal_bn": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(10, 12),
                                  kw=dict(hid_channels=16, norm_cls="bN1d", kernel=3, dilation=2, stride=2, causal=True),
                                  B=2, T=53, seed=67, bn_stats=True),
    "dsc_stride3_causal_cln": dict(kind="lobe", cls="DepthwiseSeparableConv1d", args=(10, 14),
                                   kw=dict(norm_cls="cLN", kernel=2, dilation=3, stride=3, causal=True), B=2, T=40, seed=68),
    # ---- recurrent maskers, module level: T mod K in {0, 1, K-1}, causal / bidirectional, FiLM / Gate
    # conditioning, embedding-free TSE (embed = enrolment features), overlapped segments
    "dprnn_causal_r0": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16), kw=dict(n_blocks=2, seg_size=5, causal=True),
                            B=2, T=20, seed=21),
    "dprnn_causal_r1": dict(kind="rnn", cls="DPRNN", args=(16, 8, 12), kw=dict(n_blocks=2, seg_size=5, causal=True),
                            B=2, T=21, seed=22),
    "dprnn_bi_rk1": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16), kw=dict(n_blocks=2, seg_size=5, causal=False),
                         B=3, T=24, seed=23),
    "dprnn_film": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16),
                       kw=dict(n_blocks=2, seg_size=5, causal=True, embed_dim=6, embed_norm=True,
                               block_with_embed=[1, 0]), B=2, T=23, seed=24, embed="vec"),
    "dprnn_embfree": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16),
                          kw=dict(n_blocks=2, seg_size=5, causal=True, block_with_embed=[0, 0],
                                  embedding_free_tse=True),
                          B=2, T=22, seed=25, embed="feat", Te=17),
    "dprnn_embfree_bi": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16),
                             kw=dict(n_blocks=2, seg_size=4, causal=False, block_with_embed=[0, 0],
                                     embedding_free_tse=True),
                             B=2, T=19, seed=26, embed="feat", Te=30),
    "dprnn_overlap": dict(kind="rnn", cls="DPRNN", args=(16, 8, 16),
                          kw=dict(n_blocks=2, seg_size=6, seg_overlap=True, causal=False), B=2, T=25, seed=27),
    "skim_causal": dict(kind="rnn", cls="SkiM", args=(16, 12, 16), kw=dict(n_blocks=3, seg_size=7, causal=True),
                        B=2, T=30, seed=31),
    "skim_bi_film": dict(kind="rnn", cls="SkiM", args=(16, 12, 10),
                         kw=dict(n_blocks=3, seg_size=7, causal=False, embed_dim=6, embed_norm=True,
                                 embed_fusion="FiLM", block_with_embed=[1, 0, 1]), B=2, T=29, seed=32, embed="vec"),
    "skim_gate": dict(kind="rnn", cls="SkiM", args=(16, 12, 16),
                      kw=dict(n_blocks=2, seg_size=8, causal=True, embed_dim=6, embed_norm=False,
                              embed_fusion="Gate", block_with_embed=[1, 1]), B=2, T=24, seed=33, embed="vec"),
    "skim_overlap": dict(kind="rnn", cls="SkiM", args=(16, 12, 16),
                         kw=dict(n_blocks=2, seg_size=6, seg_overlap=True, causal=True), B=2, T=25, seed=34),
    # ---- streaming: offline forward vs step_chunk vs step_frame (the reference's own property,
    # test/test_streaming.py:61-116) on its tiny model and on the demo preset (BASELINE config 5,
    # egs/tse/demo/utils.py:51-66) across two Mem-LSTM updates; plus three 320-sample chunks through the demo
    # harness (sliding window, encoder, mask, decoder, averaging overlap-add; utils.py:78-128)
    "stream_tiny": dict(kind="stream", args=(5, 20, 5),
                        kw=dict(seg_size=10, seg_overlap=False, causal=True, n_blocks=4, embed_dim=10,
                                embed_norm=True, embed_fusion="FiLM", block_with_embed=[1, 1, 1, 1]),
                        frames=47, seed=41),
    "cfg5_demo": dict(kind="stream", args=(128, 256, 128),
                      kw=dict(n_blocks=4, seg_size=150, seg_overlap=False, causal=True, embed_dim=192,
                              embed_norm=True, block_with_embed=[1, 1, 1, 1], embed_fusion="FiLM"),
                      frames=307, seed=42, harness=dict(win=32, hop=16, C=128, chunks=3, chunk=320)),
    # ---- 2-D convolutional maskers: Unet, UnetTcn (normal / gated + FiLM), DPCRN
    "unet_small": dict(kind="unet", cls="Unet", oracle="unet",
                       kw=dict(input_type="RI", input_dim=32, channels=(1, 4, 6), kernel_t=(2, 3), kernel_f=(3, 5),
                               stride_t=(1, 1), stride_f=(2, 2), dilation_t=(1, 1), dilation_f=(1, 1), delay=(0, 1),
                               transpose_t_size=2, dropout=0.0), B=2, T=19, seed=51),
    "unet_real_skipconv": dict(kind="unet", cls="Unet", oracle="unet",
                               kw=dict(input_type="Real", input_dim=20, activation_type="ReLU", channels=(1, 3, 5),
                                       kernel_t=(1, 2), kernel_f=(5, 3), stride_t=(1, 1), stride_f=(4, 1),
                                       dilation_t=(1, 1), dilation_f=(1, 1), delay=(0, 0), transpose_t_size=3,
                                       skip_conv=True, multi_output=2, dropout=0.0), B=2, T=17, seed=52),
    "unettcn_small": dict(kind="unet", cls="UnetTcn", oracle="unet_tcn",
                          kw=dict(embed_dim=6, embed_norm=True, input_type="RI", input_dim=64, channels=(1, 4, 6),
                                  kernel_t=(2, 2), kernel_f=(5, 3), stride_t=(1, 1), stride_f=(2, 2), dilation_t=(1, 1),
                                  dilation_f=(1, 1), delay=(0, 0), tcn_dim=8, per_tcn_stack=2, repeat_tcn=2,
                                  tcn_with_embed=[1, 0], dropout=0.0), B=2, T=23, seed=53, embed=6),
    "unettcn_gated_film": dict(kind="unet", cls="UnetTcn", oracle="unet_tcn",
                               kw=dict(embed_dim=6, embed_norm=True, input_type="RI", input_dim=64, channels=(1, 4, 6),
                                       transpose_delay=True, kernel_t=(2, 2), kernel_f=(5, 5), stride_t=(1, 1),
                                       stride_f=(2, 2), dilation_t=(1, 1), dilation_f=(1, 1), delay=(0, 0),
                                       tcn_layer="gated", tcn_use_film=True, tcn_dim=8, per_tcn_stack=2, repeat_tcn=1,
                                       tcn_with_embed=[1, 0], dropout=0.0), B=2, T=21, seed=54, embed=6),
    "unettcn_gln2d": dict(kind="unet", cls="UnetTcn", oracle="unet_tcn",
                          kw=dict(embed_dim=6, embed_norm=True, input_type="RI", input_dim=64, norm_type="gLN",
                                  channels=(1, 4, 6), transpose_delay=True, kernel_t=(2, 2), kernel_f=(5, 5),
                                  stride_t=(1, 1), stride_f=(2, 2), dilation_t=(1, 1), dilation_f=(1, 1), delay=(0, 0),
                                  tcn_layer="gated", tcn_dim=8, per_tcn_stack=2, repeat_tcn=1, tcn_with_embed=[1, 0],
                                  tcn_norm="gLN", dropout=0.0), B=2, T=21, seed=56, embed=6),
    "dpcrn_small": dict(kind="unet", cls="DPCRN", oracle="dpcrn",
                        kw=dict(input_type="RI", input_dim=32, channels=(1, 4, 6, 8), transpose_delay=True,
                                kernel_t=(2, 2, 2), kernel_f=(5, 3, 3), stride_t=(1, 1, 1), stride_f=(2, 2, 1),
                                dilation_t=(1, 1, 1), dilation_f=(1, 1, 1), delay=(0, 0, 0), rnn_hidden=8, dropout=0.0),
                        B=2, T=18, seed=55),
    "dparn_small": dict(kind="unet", cls="DPARN", oracle="dparn",
                        kw=dict(input_type="RI", input_dim=32, channels=(1, 4, 6, 8), transpose_delay=False,
                                kernel_t=(2, 2, 2), kernel_f=(5, 3, 3), stride_t=(1, 1, 1), stride_f=(2, 2, 1),
                                dilation_t=(1, 1, 1), dilation_f=(1, 1, 1), delay=(0, 0, 0), rnn_hidden=12, nhead=2,
                                dropout=0.0), B=2, T=18, seed=57),
    # dparn.py:249-401: two masks out of the last transposed convolution, frames trimmed at the front
    "dparn_mout_small": dict(kind="unet", cls="DPARN_Mout", oracle="dparn",
                             kw=dict(input_type="RI", input_dim=32, channels=(1, 4, 6, 8), transpose_delay=True,
                                     kernel_t=(2, 2, 2), kernel_f=(5, 3, 3), stride_t=(1, 1, 1), stride_f=(2, 2, 1),
                                     dilation_t=(1, 1, 1), dilation_f=(1, 1, 1), delay=(0, 0, 0), multi_output=3,
                                     rnn_hidden=10, nhead=1, dropout=0.0), B=2, T=19, seed=58),
    # lobe/rnn.py:9-55 on its own: the three cell types the constructor accepts
    "srnn_lstm_bi": dict(kind="single_rnn", args=("LSTM", 12, 20), kw=dict(bidirectional=True), B=3, T=37, seed=71),
    "srnn_gru": dict(kind="single_rnn", args=("GRU", 12, 20), kw=dict(bidirectional=False), B=5, T=37, seed=72),
    "srnn_gru_bi": dict(kind="single_rnn", args=("gru", 10, 64), kw=dict(bidirectional=True), B=2, T=50, seed=73),
    "srnn_rnn_bi": dict(kind="single_rnn", args=("RNN", 12, 33), kw=dict(bidirectional=True), B=3, T=21, seed=74),
    "enc_free": dict(kind="encdec", enc=dict(kind="free", win=32, hop=16, C=20), B=3, L=500, seed=16),
    "enc_free_relu_ragged": dict(kind="encdec", enc=dict(kind="free", win=20, hop=6, C=9, relu=True),
                                 B=2, L=211, seed=17),
    "enc_fbank": dict(kind="fbank", kw=dict(fft_length=64, win_length=64, hop_length=16, trainable=False,
                                            output_format="Magnitude", n_banks=12), B=2, L=500, seed=19),
    "enc_fbank_trainable": dict(kind="fbank", kw=dict(fft_length=128, win_length=128, hop_length=32, trainable=True,
                                                      output_format="Magnitude", n_banks=20), B=2, L=700, seed=20),
    "enc_stft": dict(kind="encdec", enc=dict(kind="stft", n_fft=64, hop=16, drop_first_bin=False),
                     B=2, L=400, seed=18),
    # ---- signal scores (loss/sdr.py) and the multi-output wrapper (base_nn.py:780-939), SURVEY 8(f) row 4 ----
    "loss_sdr_modes": dict(kind="loss", B=5, M=3, L=4000, seed=71),
    # mask application functions (base_nn.py:41-190) on their own, and the conv-STFT's "MagPhase" output
    # (lobe/encoder.py:384-389) with trainable (sqrt) and fixed (power) kernels
    "mask_functions": dict(kind="func", B=2, C=7, T=41, n_fft=64, hop=16, L=400, seed=81),
    "simo_free": dict(kind="simo", enc=dict(kind="free", win=16, hop=8, C=24), heads=2,
                      masker=masker_args(24, 0, False, [0, 0], tcn_kernel=3, tcn_dim=12, repeat_tcn=2,
                                         tcn_dilated_basic=2, per_tcn_stack=2),
                      wrap=dict(mask_constraint="ReLU"), B=2, L=1500, L_ref=1400, seed=73),
    "simo_stft": dict(kind="simo", enc=dict(kind="stft", n_fft=32, hop=8, drop_first_bin=True), heads=2,
                      masker=masker_args(32, 0, False, [0, 0], tcn_kernel=3, tcn_dim=12, repeat_tcn=1,
                                         tcn_dilated_basic=2, per_tcn_stack=2),
                      wrap=dict(mask_constraint="linear", f_type="Complex", mask_type="Complex", drop_first_bin=True),
                      B=2, L=1200, L_ref=1100, seed=74),
}

# parameter counts measured on the reference itself by make_golden.py (its docstrings quote 13 372 725 for
# tse_unet_tcn_v0_causal and 6 375 442 for tse_skim_v0_causal; the models it builds have the numbers below)
PARAM_COUNTS = {"cfg2_short": 7977032, "cfg1_short": 8207432, "cfg3_short": 10108119, "cfg4_tse_short": 723585,
                "ns_dpcrn_short": 1380043, "tse_unet_tcn_causal_short": 13324533,
                "tse_skim_causal_short": 6375440}


def build_encoder(ns, enc):
    if enc["kind"] == "free":
        return ns.FreeEncDec(win_length=enc["win"], hop_length=enc["hop"], laten_length=enc["C"],
                             output_active=enc.get("relu", False))
    return ns.ConvEncDec(fft_length=enc["n_fft"], win_type="hann", win_length=enc["n_fft"],
                         hop_length=enc["hop"], trainable=True, output_format="Complex")


def build_masker(ns, m):
    if "cls" in m:
        return getattr(ns, m["cls"])(*m["args"], **m["kw"])
    kw = {k: v for k, v in m.items() if k not in ("input_dim", "embed_dim", "embed_norm")}
    return ns.ConvTasNet(m["input_dim"], m["embed_dim"], m["embed_norm"], **kw)


def build_simo_masker(ns, c):
    """A separation masker with the contract SiMoTaskWrapModule expects ([N, C, T] -> [N, M, C, T]): M independent
    Conv-TasNet heads on the same features (keys `masker.heads.{m}.*`)."""
    import torch
    import torch.nn as nn

    class Heads(nn.Module):
        def __init__(self):
            super().__init__()
            self.heads = nn.ModuleList([build_masker(ns, c["masker"]) for _ in range(c["heads"])])

        def forward(self, x):
            return torch.stack([h(x) for h in self.heads], dim=1)

    return Heads()


def build_speaker_net(ns, s):
    import torch.nn as nn
    if s.get("block") == "rnn":
        return nn.ModuleList([ns.SingleRNN(rnn_type="LSTM", input_size=s["C"], hidden_size=s["H"],
                                           bidirectional=s.get("bidirectional", True), dropout=0.05),
                              ns.AttentiveStatisticsPooling(s["C"], s["att"]),
                              nn.Conv1d(s["C"] * 2, s["E"], 1, bias=False)])
    if s.get("block") == "gated":
        return nn.ModuleList(
            ([ns.Magnitude(drop_first=False)] if s.get("magnitude") else [])
            + [ns.GatedTCN(s["C"], s["H"], 3, dilation=2 ** i, causal=False, tcn_norm="gLN") for i in range(s["n_tcn"])]
            + [ns.AttentiveStatisticsPooling(s["C"], s["att"]), nn.Conv1d(s["C"] * 2, s["E"], 1, bias=False)])
    return nn.ModuleList(
        ([ns.SpecAugment(*s["specaug"])] if s.get("specaug") else [])
        + [ns.TCN(s["C"], s["H"], 3, dilation=2 ** i, causal=False, tcn_norm="gLN", dconv_norm="gGN")
           for i in range(s["n_tcn"])]
        + [ns.AttentiveStatisticsPooling(s["C"], s["att"]), nn.Conv1d(s["C"] * 2, s["E"], 1, bias=False)])


def func_inputs(c):
    """tf_rep / mask as [N, 2C, T] channel halves (the layout apply_tf_masks takes) and a waveform."""
    from detweights import det_wave
    n, ch, t = c["B"], c["C"], c["T"]
    tf_rep = det_wave(c["seed"], n * 2 * ch, t).reshape(n, 2 * ch, t) * 2.0
    mask = det_wave(c["seed"] + 1, n * 2 * ch, t).reshape(n, 2 * ch, t) * 3.0
    wav = det_wave(c["seed"] + 2, n, c["L"])
    return tf_rep, mask, wav


def build(ns, name):
    """Instantiate case `name` from namespace `ns` (the reference's classes or this repo's)."""
    c = CASES[name]
    if c["kind"] == "wrap":
        kw = dict(c["wrap"])
        if "speaker_net" in c:
            kw["speaker_net"] = build_speaker_net(ns, c["speaker_net"])
        if "enc_spk" in c:
            kw["encoder_spk"] = ns.FbankEnc(**c["enc_spk"]["kw"])
        return ns.SoTaskWrapModule(encoder=build_encoder(ns, c["enc"]), masker=build_masker(ns, c["masker"]),
                                   verbose=False, **kw)
    if c["kind"] == "simo":
        return ns.SiMoTaskWrapModule(encoder=build_encoder(ns, c["enc"]), masker=build_simo_masker(ns, c),
                                     loss_func_wav=ns.SDRLoss.init_mode("sisnr", reduction=False), verbose=False,
                                     **c["wrap"])
    if c["kind"] == "masker":
        return build_masker(ns, c["masker"])
    if c["kind"] == "encdec":
        return build_encoder(ns, c["enc"])
    if c["kind"] in ("rnn", "lobe", "atten"):
        return getattr(ns, c["cls"])(*c["args"], **c["kw"])
    if c["kind"] == "single_rnn":
        return ns.SingleRNN(*c["args"], **c["kw"])
    if c["kind"] == "unet":
        return getattr(ns, c["cls"])(**c["kw"])
    if c["kind"] == "fbank":
        return ns.FbankEnc(**c["kw"])
    if c["kind"] == "stream":
        return ns.StreamingSkiM(*c["args"], **c["kw"])
    raise KeyError(c["kind"])


def unet_args(spec):
    """The oracle's description of a Unet / UnetTcn / DPCRN constructor call (defaults of unet.py:35-53, 307-340,
    dpcrn.py:85-104)."""
    a = dict(input_type="RI", input_dim=512, activation_type="PReLU", norm_type="bN2d", dropout=0.05,
             transpose_t_size=2, transpose_delay=False, skip_conv=False, multi_output=1, embed_dim=0, embed_norm=False,
             tcn_layer="normal", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2, per_tcn_stack=5, repeat_tcn=4,
             tcn_with_embed=[1, 0, 0, 0, 0], tcn_use_film=False, tcn_norm="gLN", dconv_norm="gGN", causal=False,
             spectral_compress=False)
    if spec["cls"] in ("DPCRN", "DPARN", "DPARN_Mout"):
        a["nhead"] = 1
        if spec["cls"] == "DPARN_Mout":
            a["multi_output"] = 2
        a.update(channels=(1, 32, 32, 32, 64, 128), kernel_t=(2, 2, 2, 2, 2), stride_t=(1, 1, 1, 1, 1),
                 dilation_t=(1, 1, 1, 1, 1), kernel_f=(5, 3, 3, 3, 3), stride_f=(2, 2, 1, 1, 1),
                 dilation_f=(1, 1, 1, 1, 1), delay=(0, 0, 0, 0, 0))
    else:
        a.update(channels=(1, 1, 8, 8, 16, 16), kernel_t=(5, 1, 9, 1, 1), stride_t=(1, 1, 1, 1, 1),
                 dilation_t=(1, 1, 1, 1, 1), kernel_f=(1, 5, 1, 5, 1), stride_f=(1, 4, 1, 4, 1),
                 dilation_f=(1, 1, 1, 1, 1), delay=(0, 0, 1, 0, 0))
    a.update(spec["kw"])
    return a


def rnn_args(spec):
    """The oracle's description of a DPRNN / SkiM constructor call (defaults of dprnn.py:27-40, skim.py:280-294)."""
    a = dict(n_blocks=2, seg_size=20, seg_overlap=False, causal=True, embed_dim=0, embed_norm=False,
             block_with_embed=None, embedding_free_tse=False, embed_fusion=None)
    a.update(spec["kw"])
    a["input_size"], a["hidden_size"], a["output_size"] = spec["args"]
    return a


def oracle_cfg(name):
    """The oracle's description of a wrapper case."""
    c = CASES[name]
    enc = dict(c["enc"])
    if c["kind"] == "simo":
        cfg = dict(encoder=enc, masker=full_masker_args(c["masker"]), heads=c["heads"])
        cfg.update({k: v for k, v in c["wrap"].items() if k != "drop_first_bin"})
        return cfg
    if c["masker"].get("cls") in ("Unet", "UnetTcn", "DPCRN", "DPARN"):
        cfg = dict(encoder=enc, masker=unet_args(c["masker"]), masker_kind=c["masker"]["oracle"])
    elif "cls" in c["masker"]:
        cfg = dict(encoder=enc, masker=rnn_args(c["masker"]), masker_kind=c["masker"]["cls"].lower())
    else:
        cfg = dict(encoder=enc, masker=full_masker_args(c["masker"]))
    cfg.update({k: v for k, v in c["wrap"].items() if k not in ("drop_first_bin", "embedding_free_tse")})
    if "enc_spk" in c:
        cfg["encoder_spk"] = dict(hop=c["enc_spk"]["kw"].get("hop_length", 128),
                                  trainable=c["enc_spk"]["kw"].get("trainable", True))
    if "speaker_net" in c:
        cfg["speaker_net"] = {k: v for k, v in c["speaker_net"].items()
                              if k in ("n_tcn", "block", "magnitude", "bidirectional", "specaug")}
    return cfg


def full_masker_args(m):
    a = dict(embed_norm=False, tcn_layer="normal", tcn_kernel=3, tcn_dim=256, tcn_dilated_basic=2,
             per_tcn_stack=5, repeat_tcn=4, tcn_norm="gLN", dconv_norm="gGN", causal=False)
    a.update(m)
    return a


def namespace(**classes):
    return SimpleNamespace(**classes)


def build_demo(ns, c):
    """The demo harness' model (egs/tse/demo/utils.py:47-72): encoder + streaming masker under the keys
    `encoder.*` / `masker.*`."""
    import torch.nn as nn
    h = c["harness"]
    net = nn.Module()
    net.encoder = ns.FreeEncDec(win_length=h["win"], hop_length=h["hop"], laten_length=h["C"], output_active=True)
    net.masker = ns.StreamingSkiM(*c["args"], **c["kw"])
    return net


def loss_inputs(c):
    """Deterministic (estimate, reference) pairs of the loss case: reference = uniform noise with a per-row offset,
    estimate = gain * reference + noise at per-row levels from -5 dB to +60 dB (the last row nearly identical), and
    the [B, M, L] tensors of the source-aggregated modes."""
    import numpy as np
    import torch
    g = np.random.Generator(np.random.Philox(key=c["seed"]))
    b, m, length = c["B"], c["M"], c["L"]
    ref3 = g.uniform(-0.5, 0.5, (b, m, length)) + g.uniform(-0.05, 0.05, (b, m, 1))
    noise = g.uniform(-0.5, 0.5, (b, m, length))
    snr_db = np.linspace(-5.0, 60.0, b * m).reshape(b, m, 1)
    gain = g.uniform(0.5, 1.5, (b, m, 1))
    est3 = gain * ref3 + noise * 10 ** (-snr_db / 20) + g.uniform(-0.02, 0.02, (b, m, 1))
    est3, ref3 = torch.tensor(est3, dtype=torch.float32), torch.tensor(ref3, dtype=torch.float32)
    labels = torch.tensor([False, True, False, False, True][:b] + [False] * max(0, b - 5))
    return est3[:, 0].contiguous(), ref3[:, 0].contiguous(), est3, ref3, labels
