"""Task wrapper on the HIP path (mirror of the inference side of puresound/nnet/base_nn.py:11-777).

`SoTaskWrapModule.inference` is the hot path: encoder -> [speaker_net] -> masker -> get_mask ->
apply_tf_masks -> encoder.inverse -> output constraint.  Here the whole chain stays in the padded
device layout: the encoder kernel writes [N,C,ldt], the masker updates it in place and the decoder
kernel fuses mask activation, mask application, overlap-add and the clamp.  The training-loss
forwards (`_forward*`, base_nn.py:426-672) are out of scope and raise.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import hip
from .._abi import TcnBlock
from .conv_tasnet import TCN, ConvTasNet, GatedTCN
from .lobe.trivial import Magnitude, SpecAugment
from .dprnn import DPRNN
from .skim import SkiM
from .unet import Unet
from .lobe.encoder import ConvEncDec, FbankEnc, FreeEncDec
from .lobe.pooling import AttentiveStatisticsPooling
from .lobe.rnn import SingleRNN

_MASK_ACTS = ("linear", "relu", "sigmoid")
_STREAMS = {}


def _side_streams(dev: torch.device, n: int):
    """Per-device pool of side streams for sub-batch overlap (created once, reused)."""
    pool = _STREAMS.setdefault((dev.type, dev.index), [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(dev))
    return pool[:n]


class BaseModel(nn.Module):
    """base_nn.py:11-32."""

    def __init__(self) -> None:
        super().__init__()

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    @property
    def overall_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters())

    @property
    def overall_trainable_parameters(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def get_state_dict(self):
        return self.state_dict()


class EncDecMaskerBaseModel(BaseModel):
    """Argument checking of get_mask / apply_tf_masks with the reference's exception types
    (base_nn.py:41-95); the arithmetic itself is fused into the decoder kernel."""

    def check_mask_constraint(self, mask_constraint: str) -> str:
        c = mask_constraint.lower()
        if c not in _MASK_ACTS:
            raise NotImplementedError  # base_nn.py:94-95
        return c

    def check_mask_pairing(self, mask_type: str, f_type: str) -> str:
        mt, ft = mask_type.lower(), f_type.lower()
        if (mt, ft) == ("real", "real"):
            return "real"
        if (mt, ft) == ("complex", "complex"):
            return "complex"
        if (mt, ft) == ("polar", "polar"):
            return "polar"
        if (mt, ft) == ("real", "complex"):
            # the reference reads `mask` before assignment here (base_nn.py:127)
            raise UnboundLocalError("local variable 'mask' referenced before assignment")
        raise NameError  # base_nn.py:78-79


    # -- the reference's own functions (base_nn.py:41-190) on device tensors -----------------------------------------
    def get_mask(self, mask: torch.Tensor, mask_constraint: str = "linear") -> torch.Tensor:
        """base_nn.py:81-95, [N, C, T] -> [N, C, T]."""
        c = self.check_mask_constraint(mask_constraint)
        hip.require_device(mask, "get_mask")
        if c == "linear":
            return mask
        t = mask.shape[-1]
        ones = torch.ones_like(mask)  # act(mask) = 1 * act(mask): the mask kernel applies the constraint
        return hip.unpad_rows(hip.real_mask(hip.pad_rows(ones), hip.pad_rows(mask), c), t)

    def apply_tf_masks(self, tf_rep: torch.Tensor, est_masks: torch.Tensor, mask_type: str, f_type: str) -> torch.Tensor:
        """base_nn.py:41-79 on [N, C, T] / [N, 2C, T] device tensors.  (complex, complex) -> [N, C, T, 2];
        (real, real) -> [N, C, T]; (real, complex) fails as the reference does (it reads `mask` before assignment,
        :127); (polar, polar) fails as the reference does too: it stacks the mask halves on dim 1 (:74) and the
        broadcast in _apply_complex_mask_on_polar then raises RuntimeError for every shape."""
        pairing = self.check_mask_pairing(mask_type, f_type)
        hip.require_device(tf_rep, "apply_tf_masks")
        t = tf_rep.shape[-1]
        if pairing == "real":
            return self._apply_mag_mask_on_mag(tf_rep, est_masks)
        if pairing == "polar":
            raise RuntimeError("The size of tensor a must match the size of tensor b: apply_tf_masks(polar, polar) "
                               "stacks the mask on dim 1 in the reference (base_nn.py:74) and cannot broadcast; call "
                               "_apply_complex_mask_on_polar with [N, C, T, 2] operands")
        y = hip.unpad_rows(hip.complex_mask(hip.pad_rows(tf_rep), hip.pad_rows(est_masks), "linear"), t)
        half = y.shape[1] // 2
        return torch.stack((y[:, :half], y[:, half:]), dim=-1)

    def _apply_mag_mask_on_mag(self, tf_rep: torch.Tensor, est_masks: torch.Tensor) -> torch.Tensor:
        """base_nn.py:146-159: tf_rep * est_masks on [N, C, T]."""
        hip.require_device(tf_rep, "_apply_mag_mask_on_mag")
        t = tf_rep.shape[-1]
        return hip.unpad_rows(hip.real_mask(hip.pad_rows(tf_rep), hip.pad_rows(est_masks), "linear"), t)

    def _apply_complex_mask_on_reim(self, tf_rep: torch.Tensor, est_masks: torch.Tensor) -> torch.Tensor:
        """base_nn.py:131-144 (_mul_c, :97-112): [N, C, T, 2] x [N, C, T, 2] -> [N, C, T, 2]."""
        return self._pairwise4(tf_rep, est_masks, lambda a, b: hip.complex_mask(a, b, "linear"))

    def _apply_complex_mask_on_polar(self, tf_rep: torch.Tensor, est_mask: torch.Tensor) -> torch.Tensor:
        """base_nn.py:161-190: [N, C, T, 2] x [N, C, T, 2] -> [N, C, T, 2]."""
        return self._pairwise4(tf_rep, est_mask, hip.polar_mask)

    @staticmethod
    def _pairwise4(a4: torch.Tensor, b4: torch.Tensor, fn) -> torch.Tensor:
        hip.require_device(a4, "mask application")
        if a4.dim() != 4 or a4.shape[-1] != 2 or a4.shape != b4.shape:
            raise RuntimeError("expected two [N, C, T, 2] tensors")
        t = a4.shape[2]
        a = torch.cat((a4[..., 0], a4[..., 1]), dim=1).contiguous()
        b = torch.cat((b4[..., 0], b4[..., 1]), dim=1).contiguous()
        y = hip.unpad_rows(fn(hip.pad_rows(a), hip.pad_rows(b)), t)
        half = y.shape[1] // 2
        return torch.stack((y[:, :half], y[:, half:]), dim=-1)


class SoTaskWrapModule(EncDecMaskerBaseModel):
    """Single-output task wrapper (speech enhancement / target speech extraction), base_nn.py:193-777."""

    def __init__(self, encoder: nn.Module, masker: nn.Module, embedding_free_tse: bool = False,
                 encoder_spk: Optional[nn.Module] = None, speaker_net: Optional[nn.Module] = None,
                 loss_func_wav: Optional[nn.Module] = None, loss_func_spk: Optional[nn.Module] = None,
                 loss_func_others: Optional[nn.Module] = None, f_type: str = "real", mask_type: str = "real",
                 mask_constraint: str = "linear", output_constraint: str = "linear", drop_first_bin: bool = False,
                 verbose: bool = True) -> None:
        super().__init__()
        self.f_type = f_type
        self.mask_type = mask_type
        self.encoder = encoder
        self.masker = masker
        self.embedding_free_tse = embedding_free_tse
        self.encoder_spk = encoder_spk
        self.speaker_net = speaker_net
        self.loss_func_wav = loss_func_wav
        self.loss_func_spk = loss_func_spk
        self.loss_func_others = loss_func_others
        self.mask_constraint = mask_constraint
        self.output_constraint = output_constraint
        self.drop_first_bin = drop_first_bin
        self.task = self.check_task()
        print(f"Current task label: {self.task}")
        if verbose:
            self._verbose()

    def check_task(self):
        """Task label, same decision table as base_nn.py:263-317."""
        if self.speaker_net is None:
            if not self.embedding_free_tse:
                label = 0
                print("Initialized a SE or BSS model, ignored the Encoder-spk." if self.encoder_spk is not None
                      else "Initialized a SE or BSS model.")
            else:
                label = 4
                print("Initialized a TSE model which dont need speaker net.")
            return label
        label = 1
        print("Initialized a multi-task model, including two separate speech encoder." if self.encoder_spk is not None
              else "Initialized a multi-task model, sharing a same speech encoder.")
        if self.loss_func_spk is not None:
            if self.loss_func_wav is None:
                label = 2
                print("Contrastive learning via speaker loss function.")
            elif self.loss_func_others is None:
                label = 1
                print("Multi-task training has two loss function.")
            else:
                label = 3
                print("Multi-task training has three loss function.")
        elif self.loss_func_wav is None:
            label = None
            print("Inference mode.")
        else:
            label = 1
            print("Multi-task training has only one loss function.")
        return label

    def forward(self, **kwargs):
        raise NotImplementedError(
            "SoTaskWrapModule.forward computes training losses (base_nn.py:426-688); puresound_amd covers the "
            "inference path only -- use .inference()")

    # -- the hot path -----------------------------------------------------------------------------
    @torch.no_grad()
    def set_gemm_precision(self, name: str):
        """One switch for the whole model (not in the reference: the HIP path's arithmetic): "fp16x2" | "fp32" | "bf16x3" |
        "bf16" for every module of the masker and the speaker branch that has a choice (TCN stacks, recurrent maskers,
        attention layers, SingleRNN)."""
        from ._plans import PlanCache, _PLANES
        if name not in _PLANES:
            raise ValueError(f"gemm precision must be one of {sorted(_PLANES)}")
        for root in (self.masker, self.speaker_net):
            if root is None:
                continue
            for m in root.modules():
                if isinstance(m, PlanCache):
                    m.gemm_precision = name
                    m._plan = None
                elif hasattr(m, "gemm_precision"):
                    m.gemm_precision = name
            if hasattr(root, "set_gemm_precision") and not isinstance(root, PlanCache):
                root.set_gemm_precision(name)
        return self

    def inference(self, noisy: torch.Tensor, enroll: Optional[torch.Tensor] = None) -> torch.Tensor:
        """noisy [N,L] (+ enroll [N,L']) -> enhanced waveform [N,L_out] (base_nn.py:690-722)."""
        if noisy.device.type == "cpu":  # BASELINE configs[0] / the recipes' --backend cpu: stock ATen (nnet/cpu_path.py)
            from . import cpu_path
            return cpu_path.wrapper_inference(self, noisy, enroll)
        return self._inference(noisy, enroll, None)

    @torch.no_grad()
    def inference_scored(self, noisy: torch.Tensor, ref_clean: torch.Tensor, enroll: Optional[torch.Tensor] = None,
                         loss_func: Optional[nn.Module] = None, inactive_labels: Optional[torch.Tensor] = None):
        """inference() and the signal score of its result against ref_clean [N, L_ref] in one go -> (enhanced [N, L_out],
        score).  Not a method of the reference: it is what its evaluation loops do after inference (`_align_waveform`,
        base_nn.py:398-412, then SDRLoss / si_snr, loss/sdr.py:104-299), with the score's five moments gathered by the
        decoder kernel while it writes the waveform instead of by further passes over it.  loss_func: an
        SDRLoss (default self.loss_func_wav, else SDRLoss.init_mode("sisnr")); the score is its forward value."""
        from .loss.sdr import SDRLoss
        loss = loss_func if loss_func is not None else getattr(self, "loss_func_wav", None)
        if loss is None:
            loss = SDRLoss.init_mode("sisnr")
        if not isinstance(loss, SDRLoss) or loss.source_aggregated:
            raise NotImplementedError("inference_scored: an SDRLoss on [N, L] signals (not a source-aggregated mode)")
        if ref_clean.dim() != 2 or ref_clean.shape[0] != noisy.shape[0]:
            raise RuntimeError("inference_scored: ref_clean must be [N, L_ref] with the batch of `noisy`")
        if noisy.device.type == "cpu":  # (the recipes' --backend cpu: stock ATen forward, the moments as five fp64 sums)
            enh = self.inference(noisy, enroll)
            a, b = enh.double(), hip.align_reference(ref_clean, enh.shape[-1]).double()
            moments = torch.stack([a.sum(-1), b.sum(-1), (a * a).sum(-1), (b * b).sum(-1), (a * b).sum(-1)], -1)
            return enh, loss.from_moments(moments, enh.shape[-1], inactive_labels)
        hip.require_device(ref_clean, "SoTaskWrapModule.inference_scored")
        enh, moments = self._inference(noisy, enroll, ref_clean.float())
        return enh, loss.from_moments(moments, enh.shape[-1], inactive_labels)

    def _inference(self, noisy: torch.Tensor, enroll: Optional[torch.Tensor], ref: Optional[torch.Tensor]):
        """The HIP path of inference(); with `ref` also the moments [N, 5] of (estimate, aligned ref)."""
        hip.require_device(noisy, "SoTaskWrapModule.inference")
        mask_act = self.check_mask_constraint(self.mask_constraint)
        pairing = self.check_mask_pairing(self.mask_type, self.f_type)
        out_mode = self.output_constraint.lower()
        if out_mode not in ("linear", "sigmoid"):
            raise NameError("Non support type.")  # base_nn.py:421-422
        stft = isinstance(self.encoder, ConvEncDec)
        if pairing == "polar":
            # the reference's apply_tf_masks stacks the polar mask on dim 1 (base_nn.py:74) and then fails to broadcast
            raise RuntimeError("The size of tensor a must match the size of tensor b (apply_tf_masks with polar masks, "
                               "as in the reference: base_nn.py:70-76)")
        if stft:
            if pairing not in ("complex", "real"):
                raise NotImplementedError("HIP inference path with an STFT encoder: (complex, complex) or (real, real)")
        elif not isinstance(self.encoder, FreeEncDec) or pairing != "real":
            raise NotImplementedError("HIP inference path: FreeEncDec encoder with (real, real) masks")
        if not isinstance(self.masker, (ConvTasNet, DPRNN, SkiM, Unet)):
            raise NotImplementedError(f"HIP inference path: ConvTasNet / DPRNN / SkiM / Unet-family masker "
                                      f"(got {type(self.masker).__name__})")
        recurrent = isinstance(self.masker, (DPRNN, SkiM))
        if recurrent and stft:
            raise NotImplementedError("HIP inference path: recurrent maskers run behind the FreeEncDec encoder")
        if self.embedding_free_tse and not isinstance(self.masker, DPRNN):
            raise NotImplementedError("embedding_free_tse needs a DPRNN masker (dprnn.py:120-125)")
        if enroll is not None:
            hip.require_device(enroll, "SoTaskWrapModule.inference")
            if self.speaker_net is None and not self.embedding_free_tse:
                raise NotImplementedError("HIP speaker branch: an enrolment needs a speaker_net (or embedding_free_tse)")
        # the dual-path maskers pad the frame axis to whole segments: make the encoder leave room for it
        need = self.masker.padded_frames_needed if recurrent else None

        if stft:
            # _get_feature (STFT branch, base_nn.py:337-345) -> masker -> get_mask + complex apply_tf_masks
            # (:56-61) -> _get_waveform (:380-395, zero DC re-inserted = DC columns left out of the synthesis
            # weight) -> ConvSTFT.inverse -> output constraint
            enc = self.encoder.encoder
            dvec = None if enroll is None else self._speaker_embedding(enroll.contiguous())
            feats, t = enc.encode_padded(noisy.contiguous(), self.drop_first_bin)
            mask = self.masker.forward_padded(feats, t, dvec)
            enh = hip.complex_mask(feats, mask, mask_act) if pairing == "complex" else hip.real_mask(feats, mask, mask_act)
            wav = enc.decode_padded(enh, t, self.drop_first_bin, out_mode)
            if ref is None:
                return wav
            return wav, hip.wave_moments(wav, hip.align_reference(ref, wav.shape[-1]))  # (the iSTFT has no epilogue form)

        def run(part: torch.Tensor, lane: int, out: Optional[torch.Tensor],
                part_enroll: Optional[torch.Tensor] = None, part_ref: Optional[torch.Tensor] = None):
            dvec, kw = None, {}
            if part_enroll is not None and self.embedding_free_tse:    # base_nn.py:706-707: the masker gets
                dvec, te = self.encoder.encode_padded(part_enroll, need)  # the enrolment FEATURES
                kw = dict(embed_frames=te)
            elif part_enroll is not None:                              # base_nn.py:347-350, 697-705
                dvec = self._speaker_embedding(part_enroll, lane)
            feats, t = self.encoder.encode_padded(part, need)          # _get_feature, base_nn.py:319-345
            if getattr(self.masker, "takes_input_range", False) and hasattr(self.encoder, "feature_bound"):
                kw["x_amax"] = self.encoder.feature_bound(part)        # the features' range without a pass over them
            mask = self.masker.forward_padded(feats, t, dvec, lane=lane, **kw)  # base_nn.py:709-714
            # get_mask + apply_tf_masks + _get_waveform + _wav_output_constrain, base_nn.py:716-721
            if part_ref is not None:
                return self.encoder.decode_scored_padded(feats, t, part_ref, mask, mask_act, out_mode, out)
            return self.encoder.decode_padded(feats, t, mask, mask_act, out_mode, out)

        # Utterances are independent, so a batch CAN be split over `hip_streams` HIP streams (the tail of one half's kernels
        # overlapping the other half's GEMMs; results bit-identical to the single-stream run).  Default 1 since round 4: with
        # the fp16x2 arithmetic nothing is left for a second lane to hide, and halving the batch costs -- config 3 14.2 ms
        # against 12.5, config 4 3.2 against 2.1 (its recurrences fall off the 16-sequence kernels), the benchmark's step
        # nothing; every benchmark of rounds 2-3 already set 1 by hand.  model.hip_streams = 2 restores the two lanes.
        n = noisy.shape[0]
        # (below 16 utterances a launch no longer fills the chip and halving it costs more than the overlap returns:
        #  tools/batch_sweep.py)
        lanes = min(int(getattr(self, "hip_streams", 1)), n // 8) if n >= 16 else 1
        if isinstance(self.masker, SkiM) and self.masker.causal:
            # the reference's causal Mem-LSTM hand-over leaks the last segment state of utterance n-1 into
            # utterance n (skim.py:102-109): keep the batch in one piece so the result stays identical to it
            lanes = 1
        if enroll is not None and enroll.shape[0] != n:
            raise RuntimeError("inference: noisy and enroll must have the same batch size")
        if lanes <= 1:
            return run(noisy.contiguous(), 0, None, None if enroll is None else enroll.contiguous(), ref)
        noisy = noisy.contiguous()
        enroll = None if enroll is None else enroll.contiguous()
        dev = noisy.device
        win, hop = self.encoder.win_length, self.encoder.hop_length
        t_frames = (noisy.shape[1] - win) // hop + 1
        out = torch.empty(n, (t_frames - 1) * hop + win, dtype=torch.float32, device=dev)
        moments = None if ref is None else torch.empty(n, 5, dtype=torch.float64, device=dev)
        cur = torch.cuda.current_stream(dev)
        pool = _side_streams(dev, lanes)
        bounds = [n * i // lanes for i in range(lanes + 1)]
        # Plans (packed weights, prologue tables) are built by torch ops on whatever stream is current: build or
        # re-validate them HERE, on the caller's stream, which every lane waits for -- built lazily inside lane 0 they
        # would be read by lane 1 with no ordering.  What is still created lazily inside a lane (per-precision weight
        # planes, identity tables) is created once: the first forked call runs its lanes one after the other.
        self._prepare_plans(dev)
        first = not getattr(self, "_lanes_warm", False)
        object.__setattr__(self, "_lanes_warm", True)
        # (the cooperative LSTM's workgroups spin on each other and must all be resident at once: two lanes launching it side
        #  by side could each hold CUs the other is waiting for until both give up -- the lanes take the streamed kernel)
        coop, hip.COOP_LSTM = hip.COOP_LSTM, False
        try:
            for i, s in enumerate(pool):
                s.wait_stream(cur)
                if first and i > 0:
                    s.wait_stream(pool[i - 1])
                with torch.cuda.stream(s):
                    got = run(noisy[bounds[i]:bounds[i + 1]], i, out[bounds[i]:bounds[i + 1]],
                              None if enroll is None else enroll[bounds[i]:bounds[i + 1]],
                              None if ref is None else ref[bounds[i]:bounds[i + 1]])
                    if ref is not None:
                        moments[bounds[i]:bounds[i + 1]] = got[1]
        finally:
            hip.COOP_LSTM = coop
        for s in pool:
            cur.wait_stream(s)
        return out if ref is None else (out, moments)

    def _prepare_plans(self, dev: torch.device) -> None:
        """Build / re-validate every kernel-side plan of the masker and the speaker branch on the current stream."""
        from ._plans import PlanCache
        for m in self.modules():
            if isinstance(m, ConvTasNet):
                if m.tcn_layer.lower() == "normal":
                    m.block_array(dev)
            elif isinstance(m, (TCN, GatedTCN)):
                m.plan(dev)
            elif isinstance(m, AttentiveStatisticsPooling):
                m._get_plan(dev)
            elif isinstance(m, PlanCache) and hasattr(m, "_build"):
                m._plan_get(dev, m._build)

    def _align_waveform(self, enh_wav: torch.Tensor, ref_wav: torch.Tensor):
        """base_nn.py:398-412: a shorter reference is left-padded with zeros ("align from last"), a longer one is cut."""
        return enh_wav, hip.align_reference(ref_wav, enh_wav.shape[-1])

    # -- speaker branch (base_nn.py:697-705, 724-738) ---------------------------------------------------
    def _speaker_embedding_from_feats(self, x: torch.Tensor, t: int, x_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """speaker_net layer by layer on padded enrolment features (base_nn.py:697-705): Magnitude, TCN (consecutive
        plain blocks go through the fused driver together), GatedTCN, AttentiveStatisticsPooling, then the k=1
        projection conv(s) on the pooled vector -> dvec [N, E]."""
        layers = list(self.speaker_net) if isinstance(self.speaker_net, (nn.ModuleList, nn.Sequential)) else None
        if layers is None:
            raise NotImplementedError("HIP speaker branch: speaker_net must be a ModuleList / Sequential")
        pooled = None
        i = 0
        while i < len(layers):
            lay = layers[i]
            if pooled is not None:
                if not (isinstance(lay, nn.Conv1d) and lay.kernel_size == (1,) and lay.bias is None):
                    raise NotImplementedError("HIP speaker branch: after the pooling only Conv1d(k=1, bias=False)")
                pooled = hip.embed_bias(pooled, lay.weight.detach()[:, :, 0].float().contiguous(), False)
                i += 1
            elif isinstance(lay, Magnitude):
                x = lay.forward_padded(x, t)
                x_amax = None
                i += 1
            elif isinstance(lay, SpecAugment):   # tse_skim_v2_causal: masks even in eval mode, as the reference does
                x = lay.forward_padded(x, t)
                i += 1
            elif isinstance(lay, TCN):
                run = []
                while i < len(layers) and isinstance(layers[i], TCN):
                    if layers[i].emb_dim != 0:
                        raise NotImplementedError("HIP speaker branch: TCN blocks with their own embedding input")
                    run.append(layers[i])
                    i += 1
                plans = [m.plan(x.device) for m in run]
                if all(p["fused"] for p in plans):
                    blocks = (TcnBlock * len(plans))(*[p["block"] for p in plans])
                    # (x_amax: a bound on the features for fp16x2 blocks -- without one the driver measures them: 93 us)
                    # the driver's scratch (three hidden maps: 415 MB at 32 x 4 s) is kept, not re-allocated and
                    # zero-filled per call as hip.conv_tasnet does for callers without one
                    need = hip.lib().ps_conv_tasnet_workspace_bytes(x.shape[0], run[0].in_channels, run[0].hid_channels, t)
                    ws = self.__dict__.get("_spk_workspace")
                    if ws is None or ws.numel() < need or ws.device != x.device:
                        ws = torch.zeros(need, dtype=torch.uint8, device=x.device)
                        object.__setattr__(self, "_spk_workspace", ws)
                    x = hip.conv_tasnet(blocks, len(plans), x, t, run[0].in_channels, run[0].hid_channels, None, False,
                                        workspace=ws, x_amax=x_amax, bf16_rows=all(p["rows_bf16"] for p in plans))
                else:
                    for m in run:
                        x = m.forward_padded_staged(x, t, None)
            elif isinstance(lay, GatedTCN):
                x = lay.forward_padded(x, t, None)
                i += 1
            elif isinstance(lay, SingleRNN):                                    # tse_skim_v1_causal
                x = lay.forward_padded(x, t)
                i += 1
            elif isinstance(lay, AttentiveStatisticsPooling):
                pooled = lay.forward_padded(x, t)                                # [N, 2C]
                i += 1
            else:
                raise NotImplementedError(f"HIP speaker branch: no kernel path for {type(lay).__name__}")
        if pooled is None:
            raise NotImplementedError("HIP speaker branch: speaker_net must end in pooling + projection")
        return pooled

    def _speaker_embedding(self, enroll: torch.Tensor, lane: int = 0) -> torch.Tensor:
        """enroll [N,L'] -> dvec [N,E]; the enrolment goes through encoder_spk when there is one, else through the
        shared encoder (base_nn.py:347-375)."""
        enc = self.encoder_spk if self.encoder_spk is not None else self.encoder
        if isinstance(enc, ConvEncDec):
            x, t = enc.encoder.encode_padded(enroll, self.drop_first_bin)
        elif isinstance(enc, (FreeEncDec, FbankEnc)):
            x, t = enc.encode_padded(enroll)
        else:
            raise NotImplementedError(f"HIP speaker branch: no kernel path for a {type(enc).__name__} enrolment encoder")
        bound = enc.feature_bound(enroll) if isinstance(enc, FreeEncDec) else None
        return self._speaker_embedding_from_feats(x, t, bound)

    @torch.no_grad()
    def inference_tse_embedding(self, enroll: Optional[torch.Tensor] = None) -> torch.Tensor:
        """enroll [N,L'] -> [N,E,1], as the reference returns it (base_nn.py:724-738: not squeezed)."""
        hip.require_device(enroll, "SoTaskWrapModule.inference_tse_embedding")
        return self._speaker_embedding(enroll.contiguous()).unsqueeze(2)

    @torch.no_grad()
    def probe_lookahead_receptive_field(self, device=None):
        """The reference's NaN-propagation probe (base_nn.py:746-772) on the HIP path: a 10 s input whose second
        (first) half is +inf; the first (last) NaN of the output gives the look-ahead (receptive field) in samples.
        Returns (lookahead, receptive_field), each an int or the string "infinite" exactly as the reference prints."""
        import numpy as np
        dev = torch.device(device) if device is not None else next(self.parameters()).device
        hip.require_device(torch.empty(0, device=dev), "probe_lookahead_receptive_field")
        g = torch.Generator().manual_seed(0)
        x_spk = torch.rand(1, 10 * 16000, generator=g).to(dev)

        def run(x):
            try:
                return self.inference(x.to(dev), x_spk).cpu().numpy()
            except (RuntimeError, NotImplementedError, TypeError):
                return self.inference(x.to(dev)).cpu().numpy()

        x = torch.rand(1, 10 * 16000, generator=g)
        x[..., 5 * 16000:] = float("inf")
        first = int(np.where(np.isnan(run(x)))[-1][0])
        lookahead = "infinite" if first == 0 else 80000 - first
        x = torch.rand(1, 10 * 16000, generator=g)
        x[..., :-5 * 16000] = float("inf")
        last = int(np.where(np.isnan(run(x)))[-1][-1])
        receptive = "infinite" if last - (80000 - 1) == 80000 else last - (80000 - 1)
        return lookahead, receptive

    def _verbose(self):
        """The reference probes look-ahead / receptive field with two 10-s inferences and leaves the module in train()
        mode (base_nn.py:740-777).  The probe needs the parameters on a ROCm device (there is no CPU path): at
        construction time they are on the CPU, so the numbers are printed only for a model built under a device
        context; `probe_lookahead_receptive_field()` runs it later.  The train() quirk is kept."""
        print("---------------Verbose logging---------------")
        self.eval()
        print(f"Current training mode is: {self.training}")
        print(f"Total params: {self.overall_parameters}")
        p = next(self.parameters(), None)
        if p is not None and p.is_cuda:
            lookahead, receptive = self.probe_lookahead_receptive_field()
            print(f"Lookahead(samples): {lookahead}")
            print(f"Receptive Fields(samples): {receptive}")
        else:
            print("Lookahead / receptive-field probe skipped: parameters are not on a ROCm device yet (no CPU "
                  "fallback); call probe_lookahead_receptive_field() after .to(device)")
        self.train()
        print(f"Current training mode is: {self.training}")
        print("---------------Verbose logging---------------")



def _align_waveform_simo(enh_wav: torch.Tensor, ref_wav: torch.Tensor):
    """base_nn.py:874-888, the multi-output wrapper's variant: a shorter reference is left-padded with zeros; for a longer
    one the reference slices the ESTIMATE to the reference's length -- a no-op, so the lengths stay different and its loss
    fails on them.  Mirrored as is (the single-output wrapper's variant, :398-412, cuts the reference instead)."""
    enh_l, ref_l = enh_wav.shape[-1], ref_wav.shape[-1]
    if ref_l < enh_l:
        ref_wav = torch.nn.functional.pad(ref_wav, (enh_l - ref_l, 0))
    elif ref_l > enh_l:
        enh_wav = enh_wav[..., :ref_l]
    return enh_wav, ref_wav


class SiMoTaskWrapModule(EncDecMaskerBaseModel):
    """Single input, multi output (speech separation) wrapper, inference side of base_nn.py:780-939: the masker returns
    M masks [N, M, C, T] for one mixture; every (utterance, source) pair is masked and decoded like a row of a batch of
    N*M utterances, so mask constraint, mask multiply, decoder GEMV + overlap-add and the output constraint run in the
    same fused decoder launch as the single-output path.  `forward(noisy, ref_clean, inactive_labels)` returns the
    signal loss of the estimate (forward only; the HIP path has no autograd)."""

    def __init__(self, encoder: nn.Module, masker: nn.Module, loss_func_wav: Optional[nn.Module] = None,
                 f_type: str = "real", mask_type: str = "real", mask_constraint: str = "linear",
                 output_constraint: str = "linear", drop_first_bin: bool = False, verbose: bool = True) -> None:
        super().__init__()
        self.f_type = f_type
        self.mask_type = mask_type
        self.encoder = encoder
        self.masker = masker
        self.loss_func_wav = loss_func_wav
        self.mask_constraint = mask_constraint
        self.output_constraint = output_constraint
        self.drop_first_bin = drop_first_bin
        if verbose:
            print(f"Total params: {self.overall_parameters}")

    def _align_waveform(self, enh_wav: torch.Tensor, ref_wav: torch.Tensor):
        return _align_waveform_simo(enh_wav, ref_wav)

    @torch.no_grad()
    def inference(self, noisy: torch.Tensor) -> torch.Tensor:
        """noisy [N, L] -> separated waveforms [N, M, L_out] (base_nn.py:922-939)."""
        return self._inference(noisy, None)

    def _inference(self, noisy: torch.Tensor, ref: Optional[torch.Tensor]):
        """inference(); with ref [N, M, L_ref] also the moments [N*M, 5] of (estimate, aligned ref) per (utterance, source),
        gathered by the decoder launch where the encoder is the learned filterbank."""
        hip.require_device(noisy, "SiMoTaskWrapModule.inference")
        mask_act = self.check_mask_constraint(self.mask_constraint)
        pairing = self.check_mask_pairing(self.mask_type, self.f_type)
        out_mode = self.output_constraint.lower()
        if out_mode not in ("linear", "sigmoid"):
            raise NameError("Non support type.")
        stft = isinstance(self.encoder, ConvEncDec)
        if stft and pairing not in ("complex", "real"):
            raise NotImplementedError("HIP inference path with an STFT encoder: (complex, complex) or (real, real)")
        if not stft and (not isinstance(self.encoder, FreeEncDec) or pairing != "real"):
            raise NotImplementedError("HIP inference path: FreeEncDec encoder with (real, real) masks")
        noisy = noisy.contiguous()
        if stft:
            enc = self.encoder.encoder
            feats, t = enc.encode_padded(noisy, self.drop_first_bin)
        else:
            feats, t = self.encoder.encode_padded(noisy)
        mask = self.masker(feats[..., :t])  # any masker with the reference contract: [N, C, T] -> [N, M, C, T]
        assert mask.dim() == 4, "SIMO task needed 4D tensor has shape [N, Ch, C, T]"
        batch, chout, fdim, tdim = mask.shape
        if fdim != feats.shape[1] or tdim != t:
            raise RuntimeError(f"SiMo masker returned {tuple(mask.shape)} for features [{batch}, {feats.shape[1]}, {t}]")
        mask_pad = hip.pad_rows(mask.reshape(batch * chout, fdim, tdim).float())
        if mask_pad.shape[-1] != feats.shape[-1]:
            raise RuntimeError("SiMo: padded mask rows do not match the feature rows")
        feats_rep = feats.repeat_interleave(chout, dim=0)  # noisy.unsqueeze(1).repeat(1, chout, 1, 1), base_nn.py:929
        if stft:
            if pairing == "complex":
                enh = hip.complex_mask(feats_rep, mask_pad, mask_act)
                # base_nn.py:934 reshapes the complex product [N*M, F, T, 2] to [N, M, 2F, T]: the interleaved
                # (re, im) pairs are re-read as 2F rows of T frames.  That is the arithmetic a model trained with the
                # reference has seen, so it is reproduced (a gather on the small masked spectrum, off the hot path)
                half = fdim // 2
                pairs = torch.stack([enh[:, :half, :t], enh[:, half:, :t]], dim=-1)
                enh = hip.pad_rows(pairs.reshape(batch * chout, fdim, t).contiguous())
            else:
                enh = hip.real_mask(feats_rep, mask_pad, mask_act)
            wav = enc.decode_padded(enh, t, self.drop_first_bin, out_mode)
        elif ref is not None:
            if ref.dim() != 3 or tuple(ref.shape[:2]) != (batch, chout):
                raise RuntimeError(f"SiMo: ref_clean must be [N, M, L] = [{batch}, {chout}, L], got {tuple(ref.shape)}")
            hip.require_device(ref, "SiMoTaskWrapModule.forward")
            lout = (t - 1) * self.encoder.hop_length + self.encoder.win_length
            if ref.shape[-1] > lout:  # base_nn.py:885-887 leaves the lengths different and the loss fails on them
                raise RuntimeError(f"The size of tensor a ({lout}) must match the size of tensor b ({ref.shape[-1]}) at "
                                   f"non-singleton dimension 1 (SiMo _align_waveform does not cut a longer reference)")
            wav, moments = self.encoder.decode_scored_padded(
                feats_rep, t, ref.float().reshape(batch * chout, -1), mask_pad, mask_act, out_mode)
            return wav.reshape(batch, chout, -1), moments
        else:
            wav = self.encoder.decode_padded(feats_rep, t, mask_pad, mask_act, out_mode)
        wav = wav.reshape(batch, chout, -1)
        if ref is None:
            return wav
        enh, aligned = _align_waveform_simo(wav, ref.float())
        return wav, hip.wave_moments(enh.reshape(batch * chout, -1), aligned.reshape(batch * chout, -1))

    @torch.no_grad()
    def forward(self, noisy: torch.Tensor, ref_clean: torch.Tensor,
                inactive_labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """base_nn.py:899-920 without autograd: separated estimate -> align -> loss_func_wav([N*M, L], [N*M, L])."""
        if self.loss_func_wav is None:
            raise RuntimeError("SiMoTaskWrapModule.forward needs loss_func_wav")
        labels = None if inactive_labels is None else inactive_labels.reshape(-1)
        from .loss.sdr import SDRLoss
        if isinstance(self.loss_func_wav, SDRLoss) and not self.loss_func_wav.source_aggregated:
            # the score is algebra on five moments per row: the decoder launch leaves them behind
            enh, moments = self._inference(noisy, ref_clean)
            return self.loss_func_wav.from_moments(moments, enh.shape[-1], labels)
        enh = self.inference(noisy)
        batch, chout = enh.shape[:2]
        enh, ref_clean = self._align_waveform(enh, ref_clean)
        return self.loss_func_wav(enh.reshape(batch * chout, -1), ref_clean.reshape(batch * chout, -1), labels)
