"""The PyTorch custom-op boundary (puresound_amd/ops.py): every reference-API forward / inverse of the mirror modules is
a torch.ops.puresound_amd.* call with HIP, Meta and (raising) CPU registrations, so that the recipe's export action --
torch.jit.trace of Sequential(encoder, *speaker_net), encoder, encoder.decoder and masker
(egs/tse/main.py:406-443 of the reference) -- records operators instead of constants.

CPU part: registration, shape propagation on the meta device and under FakeTensorMode, tracing on meta tensors, the
`puresound` import alias.  GPU part: the four traces of the export action on the td_tse_conv_tasnet_v0 preset
(BASELINE config 3) replayed against the eager modules, through a save / load round trip."""
import io
import os
import subprocess
import sys

import pytest
import torch

import cases
import puresound_amd.nnet as PA
from conftest import rel_max
from detweights import det_state_dict, det_wave
from puresound_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _export_parts(model):
    """The four modules the export action traces (egs/tse/main.py:414-443)."""
    head = model.encoder_spk if model.encoder_spk is not None else model.encoder
    spk_net = torch.nn.Sequential(*([head] + [*model.speaker_net]))
    return spk_net, model.encoder, model.encoder.decoder, model.masker


def test_every_operator_has_hip_meta_and_cpu_registrations():
    expected = {"free_encode", "free_decode", "stft_encode", "istft_decode", "conv_tasnet_fwd", "tcn_block_fwd",
                "gated_tcn_fwd", "attn_stats_pool_fwd", "magnitude_fwd", "dprnn_fwd", "skim_fwd", "unet_fwd",
                "unet_tcn_fwd", "dpcrn_fwd", "dparn_fwd", "lstm_seq_fwd", "skim_step", "skim_chunk"}
    assert expected <= set(ops.OP_NAMES)
    for name in ops.OP_NAMES:
        for key in ("CUDA", "Meta", "CPU"):
            assert torch._C._dispatch_has_kernel_for_dispatch_key(f"puresound_amd::{name}", key), (name, key)


def test_cpu_dispatch_computes_where_registered_and_raises_elsewhere():
    enc = PA.FreeEncDec(32, 64, 16)
    feats = enc(torch.zeros(2, 4000))                      # CPU registration: a stock ATen composition
    assert feats.shape == (2, 64, (4000 - 32) // 16 + 1) and not feats.requires_grad
    assert torch.ops.puresound_amd.free_decode(torch.zeros(2, 64, 10), enc.decoder.weight, 16).shape == (2, 9 * 16 + 32)
    with pytest.raises(RuntimeError, match="no CPU path"):  # an operator without one
        torch.ops.puresound_amd.lstm_seq_fwd(torch.zeros(1, 2, 8), torch.zeros(8, 2), None, None)


@pytest.mark.parametrize("name", ["cfg3_short", "cfg2_short", "cfg1_short", "cfg4_short"])
def test_module_forwards_propagate_shapes_on_the_meta_device(name):
    c = cases.CASES[name]
    model = cases.build(PA.NS, name).eval().to("meta")
    wav = torch.empty(2, c["L"], device="meta")
    feats = model.encoder(wav)
    if feats.dim() == 4:  # STFT encoder: [N, F, T, 2]
        assert model.encoder.inverse(feats).shape[0] == 2
        return
    n, ch, t = feats.shape
    hop, win = model.encoder.hop_length, model.encoder.win_length
    assert t == (c["L"] - win) // hop + 1
    assert model.encoder.inverse(feats).shape == (2, (t - 1) * hop + win)
    assert model.encoder.decoder(feats).shape == (2, 1, (t - 1) * hop + win)
    dvec = None
    if getattr(model, "speaker_net", None) is not None:
        spk_net, _, _, _ = _export_parts(model)
        dvec = spk_net(torch.empty(2, c.get("L_enroll", c["L"]), device="meta"))
        assert dvec.shape[0] == 2 and dvec.shape[-1] == 1
        dvec = dvec.squeeze(-1)
    out = model.masker(feats, dvec) if dvec is not None else model.masker(feats)
    assert out.shape[0] == 2 and out.shape[-1] == t


def test_sequence_and_streaming_operators_propagate_shapes_on_the_meta_device():
    """lstm_seq_fwd and the streaming step (skim_step: the frame API, skim_chunk: the chunk API) are operators
    with Meta kernels, so a tracer or a shape check sees them."""
    gx, whh = torch.empty(3, 50, 4 * 64, device="meta"), torch.empty(4 * 64, 64, device="meta")
    y, h, c = torch.ops.puresound_amd.lstm_seq_fwd(gx, whh, None, torch.empty(3, 64, device="meta"))
    assert y.shape == (3, 50, 64) and h.shape == (3, 64) and c.shape == (3, 64)
    with pytest.raises(RuntimeError, match="4H"):
        torch.ops.puresound_amd.lstm_seq_fwd(gx, torch.empty(100, 64, device="meta"), None, None)
    from puresound_amd.streaming.skim_inference import StreamingSkiM
    m = StreamingSkiM(16, 8, 24, n_blocks=3, seg_size=5, causal=True, embed_dim=4, embed_norm=True, embed_fusion="FiLM",
                      block_with_embed=[1, 1, 1]).to("meta")
    x, e = torch.empty(2, 5, 16, device="meta"), torch.empty(2, 4, device="meta")
    out, seg_h, mem_h, seg_c, mem_c = m.step_chunk(x, None, None, None, None, e)
    assert out.shape == (2, 24, 5) and len(seg_h) == len(seg_c) == len(mem_h) == len(mem_c) == 2
    assert all(t.shape == (1, 2, 8) for t in seg_h + seg_c) and all(t.shape == (1, 2, 8) for p in mem_h + mem_c for t in p)
    # the frame step: states are (mutable) arguments of the operator
    state = [torch.empty(1, 8, 128, device="meta") for _ in range(3 + 3 + 4 + 4)]
    params, cfg = ops.call_args(m, "skim_step")
    yf = torch.ops.puresound_amd.skim_step(torch.empty(2, 16, device="meta"), e, state, 0, params, cfg)
    assert yf.shape == (2, 24, 1)
    with pytest.raises(RuntimeError, match="no CPU path"):
        torch.ops.puresound_amd.lstm_seq_fwd(torch.zeros(1, 2, 8), torch.zeros(8, 2), None, None)


def test_routed_forwards_keep_the_reference_call_forms():
    """The operator wrapper keeps the forward's own parameter names (forward(x, embed=...), forward(x, dvec=...)), refuses
    to hand a detached result to a caller that expects gradients, and reports the U-Net's multi-output shape."""
    masker = PA.ConvTasNet(16, 8, True, tcn_kernel=3, tcn_dim=8, repeat_tcn=1, tcn_dilated_basic=2, per_tcn_stack=2,
                           tcn_with_embed=[1, 0], tcn_norm="gLN", dconv_norm="gGN", causal=False).eval().to("meta")
    x, d = torch.empty(2, 16, 50, device="meta"), torch.empty(2, 8, device="meta")
    name = [p for p in __import__("inspect").signature(masker._hip_forward).parameters][1]
    assert masker(x, d).shape == masker(x, **{name: d}).shape == (2, 16, 50)
    with pytest.raises(TypeError):
        masker(x, nonsense=d)
    masker.train()
    with pytest.raises(RuntimeError, match="inference only"):
        masker(x, d)
    with torch.no_grad():
        assert masker(x, d).shape == (2, 16, 50)
    # the parameter list of a call follows a parameter swap (load_state_dict(assign=True) replaces the tensors)
    from puresound_amd.ops import _call_cache
    masker.eval()
    p1, c1 = _call_cache(masker, "conv_tasnet_fwd")
    masker.load_state_dict({k: v.clone() for k, v in masker.state_dict().items()}, assign=True)
    p2, c2 = _call_cache(masker, "conv_tasnet_fwd")
    assert c1 is c2 and all(a is b for a, b in zip(p2, ops.module_tensors(masker))) and not any(a is b for a, b in zip(p1, p2))


def test_operators_work_under_fake_tensor_mode():
    from torch._subclasses.fake_tensor import FakeTensorMode
    enc = PA.FreeEncDec(32, 64, 16)
    with FakeTensorMode(allow_non_fake_inputs=True) as mode:
        wav = mode.from_tensor(torch.empty(3, 8000, device="meta"))
        w = mode.from_tensor(enc.encoder.weight.detach().to("meta"))
        feats = torch.ops.puresound_amd.free_encode(wav, w, 16, False)
        assert feats.shape == (3, 64, 499)
        assert torch.ops.puresound_amd.free_decode(feats, w, 16).shape == (3, 8000)


def test_export_action_traces_record_the_operators_on_meta_tensors():
    model = cases.build(PA.NS, "cfg3_short").eval().to("meta")
    spk_net, encoder, decoder, masker = _export_parts(model)
    wav = torch.empty(1, 8000, device="meta")
    # (the speaker net's last layer is the recipe's own stock nn.Conv1d: ATen's meta kernel for it does not survive
    #  torch.jit.trace on meta tensors, so that layer is traced in the GPU test only)
    g_spk = torch.jit.trace(spk_net[:-1], wav, check_trace=False)
    g_enc = torch.jit.trace(encoder, wav, check_trace=False)
    x = encoder(wav)
    dvec = torch.empty(1, 192, 1, device="meta")
    g_dec = torch.jit.trace(decoder, x, check_trace=False)
    g_mask = torch.jit.trace(masker, (x, dvec.squeeze(-1)), check_trace=False)
    assert "puresound_amd::free_encode" in str(g_enc.graph)
    assert "puresound_amd::free_decode" in str(g_dec.graph)
    assert "puresound_amd::conv_tasnet_fwd" in str(g_mask.graph)
    s = str(g_spk.inlined_graph)
    assert "puresound_amd::tcn_block_fwd" in s and "puresound_amd::attn_stats_pool_fwd" in s
    # parameters are inputs of the recorded calls, not baked constants: the traced masker still owns them
    assert len(list(g_mask.parameters())) == len(list(masker.parameters()))


def test_puresound_import_alias_resolves_to_the_hip_modules():
    code = ("import puresound.nnet.conv_tasnet as a, puresound_amd.nnet.conv_tasnet as b; "
            "from puresound.nnet.base_nn import SoTaskWrapModule; from puresound.nnet.lobe.encoder import FreeEncDec; "
            "from puresound.streaming.skim_inference import StreamingSkiM; "
            "assert a.ConvTasNet is b.ConvTasNet; print('ok')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "compat"), ROOT]))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_export_action_traces_replay_to_the_eager_outputs():
    """egs/tse/main.py:406-443 on the mirror of td_tse_conv_tasnet_v0: trace the four modules, save and load them,
    and replay: same outputs as the eager modules (the operators run the same kernels), and the chained traced
    modules reproduce SoTaskWrapModule.inference."""
    dev = torch.device("cuda:0")
    name = "cfg3_short"
    c = cases.CASES[name]
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    spk_net, encoder, decoder, masker = _export_parts(model)
    dummy = torch.rand(1, 16000, device=dev)
    with torch.no_grad():
        traced = {}
        traced["spk"] = torch.jit.trace(spk_net, dummy)
        traced["enc"] = torch.jit.trace(encoder, dummy)
        dummy_x = traced["enc"](dummy)
        dummy_dvec = traced["spk"](dummy)
        traced["dec"] = torch.jit.trace(decoder, dummy_x)
        traced["mask"] = torch.jit.trace(masker, (dummy_x, dummy_dvec.squeeze(-1)))
        loaded = {}
        for k, m in traced.items():
            buf = io.BytesIO()
            torch.jit.save(m, buf)
            buf.seek(0)
            loaded[k] = torch.jit.load(buf, map_location=dev)
        noisy = det_wave(7, 2, c["L"]).to(dev)
        enroll = det_wave(8, 2, c["L_enroll"]).to(dev)
        for tm in (traced, loaded):
            x = tm["enc"](noisy)
            assert torch.equal(x, encoder(noisy))
            dvec = tm["spk"](enroll)
            # (the speaker net ends in the recipe's stock nn.Conv1d, an ATen / MIOpen kernel whose algorithm choice is
            #  not pinned between calls: compare to rounding, everything that is ours bit for bit)
            assert torch.allclose(dvec, spk_net(enroll), rtol=1e-5, atol=1e-6)
            assert torch.equal(tm["spk"][:-1](enroll) if isinstance(tm["spk"], torch.nn.Sequential) else
                               spk_net[:-1](enroll), spk_net[:-1](enroll))
            mask = tm["mask"](x, dvec.squeeze(-1))
            assert torch.equal(mask, masker(x, dvec.squeeze(-1)))
            wav = tm["dec"](x * torch.relu(mask))
            assert torch.equal(wav, decoder(x * torch.relu(mask)))
            ref = model.inference(noisy, enroll)
            got = wav.squeeze(1).clamp(-1, 1)
            assert got.shape == ref.shape
            assert float((got - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.gpu
def test_traced_module_runs_without_the_module_that_recorded_it():
    """A loaded trace carries parameters + constructor JSON: the operator rebuilds its module from them."""
    dev = torch.device("cuda:0")
    m = PA.ConvTasNet(64, 8, True, tcn_dim=32, per_tcn_stack=2, repeat_tcn=2, tcn_with_embed=[1, 0]).eval().to(dev)
    x, d = torch.rand(2, 64, 300, device=dev), torch.rand(2, 8, device=dev)
    with torch.no_grad():
        want = m(x, d).clone()
        t = torch.jit.trace(m, (x, d))
        buf = io.BytesIO()
        torch.jit.save(t, buf)
        del m, t
        ops._LIVE.clear()
        buf.seek(0)
        t2 = torch.jit.load(buf, map_location=dev)
        assert torch.equal(t2(x, d), want)


@pytest.mark.gpu
def test_lstm_seq_fwd_matches_nn_lstm_on_the_device():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    for n, t, i, h in ((3, 37, 20, 64), (2, 150, 16, 128), (5, 9, 12, 24)):
        lstm = torch.nn.LSTM(i, h, batch_first=True)
        x = torch.randn(n, t, i)
        h0, c0 = torch.randn(1, n, h) * 0.3, torch.randn(1, n, h) * 0.3
        ref, (hn, cn) = lstm(x, (h0, c0))
        gx = x @ lstm.weight_ih_l0.t() + lstm.bias_ih_l0 + lstm.bias_hh_l0
        y, hT, cT = torch.ops.puresound_amd.lstm_seq_fwd(gx.detach().to(dev), lstm.weight_hh_l0.detach().to(dev),
                                                         h0[0].to(dev), c0[0].to(dev))
        assert rel_max(y.cpu().numpy(), ref.detach().numpy()) < 1e-5
        assert rel_max(hT.cpu().numpy(), hn[0].detach().numpy()) < 1e-5
        assert rel_max(cT.cpu().numpy(), cn[0].detach().numpy()) < 1e-5
        y0, _, _ = torch.ops.puresound_amd.lstm_seq_fwd(gx.detach().to(dev), lstm.weight_hh_l0.detach().to(dev), None, None)
        ref0, _ = lstm(x)
        assert rel_max(y0.cpu().numpy(), ref0.detach().numpy()) < 1e-5


@pytest.mark.gpu
def test_streaming_frame_step_is_a_traceable_operator():
    """StreamingSkiM.step_frame goes through torch.ops.puresound_amd.skim_step: a trace of it records the operator with
    the state tensors as (mutated) inputs, and replaying the trace walks the stream exactly as eager calls do --
    across a Mem-LSTM update."""
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm device")
    from puresound_amd.streaming.skim_inference import StreamingSkiM
    dev = torch.device("cuda:0")
    kw = dict(n_blocks=3, seg_size=5, causal=True, embed_dim=4, embed_norm=True, embed_fusion="FiLM",
              block_with_embed=[1, 1, 1])
    m = StreamingSkiM(16, 8, 16, **kw).eval()
    m.load_state_dict(det_state_dict(m))
    m.to(dev)
    b = 2
    x = det_wave(71, b, 12 * 16, 1.0).reshape(b, 12, 16).to(dev)
    e = det_wave(72, b, 4, 1.0).to(dev)
    m.init_status(streams=b, use_graph=False)
    eager = [m.step_frame(x[:, i], e).clone() for i in range(12)]
    # the operator called directly on a fresh state set equals the method
    m.init_status(streams=b, use_graph=False)
    params, cfg = ops.call_args(m, "skim_step")
    state = [t.clone() for t in m.state_tensors()]
    direct = []
    for i in range(12):
        direct.append(torch.ops.puresound_amd.skim_step(x[:, i], e, state, i % 5, params, cfg).clone())
    for a, c in zip(eager, direct):
        assert torch.equal(a, c)
    # chunk API through its operator == the frames of one segment
    m2 = StreamingSkiM(16, 8, 16, **kw).eval()
    m2.load_state_dict(det_state_dict(m2))
    m2.to(dev)
    out, *_ = m2.step_chunk(x[:, :5].contiguous(), None, None, None, None, e)
    assert rel_max(out.cpu().numpy(), torch.cat(eager[:5], dim=2).cpu().numpy()) < 1e-5
    # a trace records the operator
    class Step(torch.nn.Module):
        def __init__(self, net):
            super().__init__()
            self.net = net

        def forward(self, xx, ee):
            return self.net.step_frame(xx, ee)

    m.init_status(streams=b, use_graph=False)
    traced = torch.jit.trace(Step(m), (x[:, 0], e), check_trace=False)
    assert "puresound_amd::skim_step" in str(traced.inlined_graph)
    # replay: the trace holds the state tensors (the module's live ones, updated in place) and the frame counter it was
    # recorded with, so right after recording -- which consumed frame 0 -- it walks frames 1..3 of the segment exactly
    # as the eager calls did (the Mem-LSTM update of frame 4 depends on the counter, which a trace freezes)
    for i in range(1, 4):
        assert torch.equal(traced(x[:, i], e), eager[i]), i
