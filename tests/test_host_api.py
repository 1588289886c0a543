"""CPU-side checks of the drop-in boundary: the mirror modules expose the reference's constructor
signatures, state_dict keys/shapes and exception types; the C-ABI library loads and exports every symbol
include/puresound_hip.h declares; the product path refuses CPU tensors instead of falling back."""
import ctypes
import inspect
import json
import os
import re

import pytest
import torch

import cases
import puresound_amd.nnet as PA
from puresound_amd import _abi
from puresound_amd.nnet.lobe.cnn import DepthwiseSeparableConv1d
from puresound_amd.nnet.lobe.encoder import ConvSTFT
from puresound_amd.nnet.lobe.norm import get_norm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------------
# state_dict layout == reference (fixture dumped from the imported reference by make_golden.py)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(n for n, c in cases.CASES.items() if c["kind"] not in ("loss", "func")))
def test_state_dict_keys_and_shapes_match_reference(golden_dir, name):
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))[name]
    mine = {k: list(v.shape) for k, v in cases.build(PA.NS, name).state_dict().items()}
    assert list(mine) == list(ref) or sorted(mine) == sorted(ref)
    assert mine == ref


def test_load_state_dict_strict_roundtrip():
    from detweights import det_state_dict
    m = cases.build(PA.NS, "cfg3_short")
    sd = det_state_dict(m)
    missing, unexpected = m.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_constructor_signatures_match_reference_order():
    """Positional order matters: the recipes call e.g. ConvTasNet(512, 192, True, tcn_kernel=...) (egs/tse/model.py:103)."""
    def names(f):
        return [p for p in inspect.signature(f).parameters if p != "self"]
    assert names(PA.ConvTasNet.__init__) == [
        "input_dim", "embed_dim", "embed_norm", "tcn_layer", "tcn_kernel", "tcn_dim", "tcn_dilated_basic",
        "per_tcn_stack", "repeat_tcn", "tcn_with_embed", "tcn_norm", "dconv_norm", "causal"]
    assert names(PA.TCN.__init__) == ["in_channels", "hid_channels", "kernel", "dilation", "dropout", "emb_dim",
                                      "causal", "tcn_norm", "dconv_norm"]
    assert names(PA.GatedTCN.__init__) == ["in_channels", "hid_channels", "kernel", "dilation", "dropout", "emb_dim",
                                           "causal", "tcn_norm", "use_film"]
    assert names(PA.FreeEncDec.__init__) == ["win_length", "laten_length", "hop_length", "output_active"]
    assert names(PA.ConvEncDec.__init__) == ["fft_length", "win_type", "win_length", "freq_bins", "hop_length",
                                             "freq_scale", "iSTFT", "fmin", "fmax", "sr", "trainable", "output_format"]
    assert names(PA.SoTaskWrapModule.__init__) == [
        "encoder", "masker", "embedding_free_tse", "encoder_spk", "speaker_net", "loss_func_wav", "loss_func_spk",
        "loss_func_others", "f_type", "mask_type", "mask_constraint", "output_constraint", "drop_first_bin", "verbose"]


def test_get_args_roundtrip():
    m = PA.ConvTasNet(32, 8, True, tcn_dim=16, per_tcn_stack=2, repeat_tcn=1, tcn_with_embed=[1, 0])
    args = m.get_args
    assert set(args) == {"input_dim", "embed_dim", "embed_norm", "tcn_norm", "dconv_norm", "tcn_layer", "tcn_dim",
                         "tcn_kernel", "tcn_dilated_basic", "repeat_tcn", "per_tcn_stack", "tcn_with_embed", "causal"}
    m2 = PA.ConvTasNet(**args)
    assert list(m2.state_dict()) == list(m.state_dict())


# ---------------------------------------------------------------------------------------------------
# exception types of the reference
# ---------------------------------------------------------------------------------------------------
def test_error_conventions():
    with pytest.raises(NameError):  # conv_tasnet.py:275
        PA.ConvTasNet(16, 0, tcn_layer="bogus", per_tcn_stack=1, tcn_with_embed=[0])
    with pytest.raises(AssertionError):  # conv_tasnet.py:277
        PA.ConvTasNet(16, 0, per_tcn_stack=3, tcn_with_embed=[0, 0])
    with pytest.raises(NameError):  # norm.py:102
        get_norm("LayerNorm")
    with pytest.raises(AssertionError):  # cnn.py:40-44
        DepthwiseSeparableConv1d(8, 8, norm_cls="gGN", causal=True)
    with pytest.raises(AssertionError):
        PA.TCN(8, 8, 3, 1, causal=True, tcn_norm="bN1d", dconv_norm="gLN")
    with pytest.raises(TypeError):  # encoder.py:339-340
        ConvSTFT(torch.hann_window(16), n_fft=32)
    with pytest.raises(NotImplementedError):  # encoder.py:155
        PA.ConvEncDec(fft_length=32, win_type="hamming", win_length=32, hop_length=8)
    with pytest.raises(NameError):  # encoder.py:406-411
        ConvSTFT(torch.hann_window(32), n_fft=32, iSTFT=False).inverse(torch.zeros(1, 17, 4, 2))


def _wrapper(**kw):
    enc = PA.FreeEncDec(16, 8, 8)
    msk = PA.ConvTasNet(8, 0, tcn_dim=4, per_tcn_stack=1, repeat_tcn=1, tcn_with_embed=[0])
    return PA.SoTaskWrapModule(enc, msk, verbose=False, **kw)


def test_wrapper_argument_errors_precede_device_work():
    m = _wrapper()
    assert m.task == 0
    assert m.check_mask_constraint("ReLU") == "relu"
    with pytest.raises(NotImplementedError):  # base_nn.py:95
        m.check_mask_constraint("softmax")
    with pytest.raises(NameError):  # base_nn.py:79
        m.check_mask_pairing("complex", "real")
    with pytest.raises(UnboundLocalError):  # base_nn.py:127 (the reference's own bug, reproduced as its exception)
        m.check_mask_pairing("real", "complex")
    with pytest.raises(NotImplementedError):
        m.forward(noisy=torch.zeros(1, 100))


def test_verbose_keeps_the_train_mode_quirk(capsys):
    m = PA.SoTaskWrapModule(PA.FreeEncDec(16, 8, 8),
                            PA.ConvTasNet(8, 0, tcn_dim=4, per_tcn_stack=1, repeat_tcn=1, tcn_with_embed=[0]))
    assert m.training  # base_nn.py:775 leaves train() on
    assert "Total params" in capsys.readouterr().out


def test_task_labels():
    import torch.nn as nn
    enc, msk = PA.FreeEncDec(16, 8, 8), PA.ConvTasNet(8, 0, tcn_dim=4, per_tcn_stack=1, repeat_tcn=1, tcn_with_embed=[0])
    assert PA.SoTaskWrapModule(enc, msk, verbose=False).task == 0
    assert PA.SoTaskWrapModule(enc, msk, embedding_free_tse=True, verbose=False).task == 4
    spk = nn.ModuleList([nn.Conv1d(8, 4, 1)])
    assert PA.SoTaskWrapModule(enc, msk, speaker_net=spk, verbose=False).task is None
    assert PA.SoTaskWrapModule(enc, msk, speaker_net=spk, loss_func_wav=nn.Identity(), verbose=False).task == 1
    assert PA.SoTaskWrapModule(enc, msk, speaker_net=spk, loss_func_spk=nn.Identity(), verbose=False).task == 2
    assert PA.SoTaskWrapModule(enc, msk, speaker_net=spk, loss_func_wav=nn.Identity(), loss_func_spk=nn.Identity(),
                               loss_func_others=nn.Identity(), verbose=False).task == 3


# ---------------------------------------------------------------------------------------------------
# CPU tensors: the operators that have a CPU registration compute (stock ATen, tests/test_cpu_path.py); the HIP entry
# points themselves never fall back, and the operators without a CPU registration say so
# ---------------------------------------------------------------------------------------------------
def test_hip_entry_points_refuse_cpu_tensors_and_uncovered_operators_say_so():
    from puresound_amd import hip
    m = _wrapper().eval()
    assert m.inference(torch.zeros(1, 400)).shape == (1, 400)      # FreeEncDec + ConvTasNet: CPU registrations
    with pytest.raises(RuntimeError, match="no CPU fallback"):      # the kernels' launchers: ROCm tensors only
        hip.free_encode(torch.zeros(1, 400), m.encoder.encoder.weight.detach(), m.encoder.hop_length, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.encoder.encode_padded(torch.zeros(1, 400))
    with pytest.raises(RuntimeError, match="ROCm device only"):
        _abi.require_device(torch.zeros(2), "x")
    with pytest.raises(RuntimeError, match="no CPU path"):          # a recurrent masker has no CPU registration
        PA.DPRNN(8, 4, 8, n_blocks=1, seg_size=4).eval()(torch.zeros(1, 8, 20))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "puresound_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "separator_oracle" not in src, f


def test_modules_deepcopy_and_pickle_without_device_pointers():
    import copy
    import pickle
    m = cases.build(PA.NS, "tiny_free")
    m2 = copy.deepcopy(m)
    assert list(m2.state_dict()) == list(m.state_dict())
    pickle.loads(pickle.dumps(m))


# ---------------------------------------------------------------------------------------------------
# C ABI
# ---------------------------------------------------------------------------------------------------
def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "puresound_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(ps_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_abi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    handle = ctypes.CDLL(_abi.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 15
    for sym in declared:
        assert hasattr(handle, sym), f"{sym} declared in include/puresound_hip.h but not exported"
    assert sorted(_abi.SIGNATURES) == declared, "ctypes binding and header disagree"


def test_abi_host_side_helpers_need_no_gpu():
    lib = _abi.lib()
    assert lib.ps_abi_version() == _abi.ABI_VERSION
    # padded rows: odd multiples of 128 frames (no power-of-two row stride)
    for t, want in ((1, 128), (128, 128), (129, 384), (256, 384), (3999, 4224), (497, 640)):
        assert lib.ps_padded_frames(t) == want == _abi.padded_frames(t)
    assert lib.ps_conv1x1_stats_parts(256, 3999) == 1 * 32 * 4
    assert lib.ps_dwconv_stats_parts(256, 3999) == 4 * 16
    assert lib.ps_stats_parts(256, 3999) == 128
    assert lib.ps_conv_tasnet_workspace_bytes(32, 512, 256, 3999) > 3 * 32 * 256 * 4224 * 4
    # argument validation happens before any launch
    rc = lib.ps_conv1x1_f32(None, None, None, 1, 16, 16, 10, 128, None, None, None, None, None, None)
    assert rc == -1 and b"null pointer" in lib.ps_last_error()


def test_wrapper_level_gemm_precision_switch():
    """SoTaskWrapModule.set_gemm_precision reaches every module of masker and speaker branch that has a choice; the default of
    the recurrent / attention maskers is "fp16x2" like the TCN blocks'; unknown names raise."""
    import cases
    import puresound_amd.nnet as PA
    from puresound_amd.nnet._plans import PlanCache
    model = cases.build(PA.NS, "tse_skim_v1_short").eval()
    mods = [m for root in (model.masker, model.speaker_net) for m in root.modules() if hasattr(m, "gemm_precision")]
    assert len(mods) > 5 and all(m.gemm_precision == "fp16x2" for m in mods)
    assert model.set_gemm_precision("fp32") is model
    assert all(m.gemm_precision == "fp32" for m in mods)
    assert any(isinstance(m, PlanCache) for m in model.speaker_net.modules())
    with pytest.raises(ValueError, match="gemm precision"):
        model.set_gemm_precision("fp8")
    tcn = cases.build(PA.NS, "cfg3_short").eval()
    tcn.set_gemm_precision("bf16x3")
    assert all(m.gemm_precision == "bf16x3" for root in (tcn.masker, tcn.speaker_net) for m in root.modules()
               if hasattr(m, "gemm_precision"))


def test_align_rules_of_the_two_wrappers():
    """base_nn.py:398-412 (single output: a longer reference is cut) and :874-888 (multi output: the estimate is sliced
    to the longer reference's length, i.e. nothing happens); both left-pad a shorter reference."""
    import torch
    from puresound_amd import hip
    from puresound_amd.nnet.base_nn import _align_waveform_simo
    from oracle import loss_oracle as LO
    enh, short, long_ = torch.arange(10.).reshape(1, 10), torch.ones(1, 7), torch.ones(1, 13)
    assert hip.align_reference(short, 10).tolist() == [[0., 0., 0.] + [1.] * 7]
    assert hip.align_reference(long_, 10).shape == (1, 10)
    for ref in (short, long_, enh):
        e, r = LO.align_waveform_single(enh, ref)
        assert torch.equal(r, hip.align_reference(ref, 10)) and e is enh
        e2, r2 = _align_waveform_simo(enh, ref)
        e3, r3 = LO.align_waveform(enh, ref)
        assert torch.equal(e2, e3) and torch.equal(r2, r3)
    assert _align_waveform_simo(enh, long_)[1].shape == (1, 13)
