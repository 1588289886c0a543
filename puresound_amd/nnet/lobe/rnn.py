"""Recurrent lobes (mirror of puresound/nnet/lobe/rnn.py:9-55): SingleRNN holds an nn.LSTM (or nn.GRU / nn.RNN) and its projection under
the reference's keys (`rnn.*`, `proj.*`).  The dual-path blocks of DPCRN / DPARN drive its LSTM through ps_lstm_f32
with their own strided walks; as a layer of its own (the speaker net of tse_skim_v1_causal, egs/tse/model.py:487-495)
it runs one sequence per utterance over the frame axis: input projection GEMM, ps_lstm_f32, output projection GEMM."""
import torch
import torch.nn as nn

from ... import hip
from ...ops import op_module, same_shape
from .._plans import PlanCache, linear_plan, lstm_plan, rnn_plan


@op_module("single_rnn_fwd", same_shape)
class SingleRNN(PlanCache, nn.Module):
    def __init__(self, rnn_type: str, input_size: int, hidden_size: int, bidirectional: bool = False,
                 dropout: float = 0.0):
        super().__init__()
        rnn_type = rnn_type.upper()
        assert rnn_type in ["RNN", "LSTM", "GRU"], f"Only support 'RNN', 'LSTM' and 'GRU', current type: {rnn_type}"
        self.rnn_type = rnn_type
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.num_direction = int(bidirectional) + 1
        self.rnn = getattr(nn, rnn_type)(input_size, hidden_size, 1, batch_first=True, bidirectional=bidirectional)
        self.drop = nn.Dropout(p=dropout)
        self.proj = nn.Linear(hidden_size * self.num_direction, input_size)

    def _build(self, device):
        if self.training and self.drop.p > 0:
            raise RuntimeError("SingleRNN: dropout is active; the HIP path is inference only -- call .eval()")
        if self.rnn_type != "LSTM":   # GRU / RNN: the plain kernel (ps_rnn_f32): no recipe builds them
            return dict(rnn=rnn_plan(self.rnn, device), proj=linear_plan(self.proj, device))
        return dict(rnn=lstm_plan(self.rnn, device, self.gemm_precision), proj=linear_plan(self.proj, device))

    def forward_padded(self, x: torch.Tensor, t: int) -> torch.Tensor:
        """padded [N, C, ldt] -> [N, C, ldt]: proj(LSTM(x)) over the t frames of each utterance (lobe/rnn.py:37-55)."""
        p = self._plan_get(x.device, self._build)
        rnn, proj = p["rnn"], p["proj"]
        n, _, ldt = x.shape
        if self.rnn_type != "LSTM":
            gx, _ = hip.conv1x1(x, t, rnn["wih"], rnn["rows"], None, rnn["bias"],
                                out=torch.empty(n, rnn["rows"], ldt, dtype=torch.float32, device=x.device))
            hseq = hip.rnn(gx, rnn["whh_t"], rnn["kind"], rnn["H"], rnn["D"], 1, ldt, t, 1, rnn["bhn"])
            y, _ = hip.conv1x1(hseq, t, proj["wt"], proj["M"], None, proj["bias"],
                               out=torch.empty(n, proj["M"], ldt, dtype=torch.float32, device=x.device))
            return y
        if (rnn["planes"] == 2 and rnn["I"] >= 64 and rnn["H"] in (256, 192)
                and hip.lstm_fmajor_h256_ok(n, ldt, rnn["D"], 1, ldt, t, 1)
                and hip.conv1x1_f16x2_fmajor_ok(n, rnn["I"], rnn["rows"], t, ldt)):
            # fp16x2 arithmetic: frame-major pre-activations + the recurrence that streams W_hh (ps_lstm_fmajor_h256_f16x2_f32;
            # the bidirectional 192-unit LSTM over the enrolment of tse_skim_v1: 4000 dependent steps), projection in fp16x2
            if "wih_f16x2" not in rnn:
                rnn["wih_f16x2"] = hip.pack_wt_f16x2(rnn["wih_rows"])
                rnn["whh_h256"] = hip.pack_whh_h256(rnn["whh_t"])
                proj["f16x2"] = hip.pack_wt_f16x2(proj["w_rows"])
            gx_fm = hip.conv1x1_f16x2_fmajor(x, t, rnn["wih_f16x2"][0], rnn["wih_f16x2"][1], rnn["rows"], rnn["bias"],
                                             x_amax=hip.absmax(x, t))
            hseq, _ = hip.lstm_fmajor_h256(gx_fm, rnn["whh_h256"][0], rnn["whh_h256"][1], rnn["D"], 1, ldt, t, 1)
            y, _, _ = hip.conv1x1_f16x2(hseq, t, proj["f16x2"][0], proj["f16x2"][1], proj["M"], None, proj["bias"], x_bound=1.0,
                                        out=torch.empty(n, proj["M"], ldt, dtype=torch.float32, device=x.device))
            return y
        gx = torch.empty(n, rnn["rows"], ldt, dtype=torch.float32, device=x.device)
        if rnn["planes"] and rnn["I"] >= 64:
            planes = 3 if rnn["planes"] == 2 else rnn["planes"]   # (fp16x2 needs a range pass: the fp32-class bf16 split here)
            if planes not in rnn["wih_planes"]:
                rnn["wih_planes"][planes] = hip.pack_wt_bf16(rnn["wih_rows"], planes)
            hip.conv1x1_bf16(x, t, rnn["wih_planes"][planes], rnn["rows"], None, rnn["bias"], out=gx)
        else:
            hip.conv1x1(x, t, rnn["wih"], rnn["rows"], None, rnn["bias"], out=gx)
        # one sequence per utterance: q = 1, the steps walk the frame axis
        hseq, _ = hip.lstm(gx, rnn["whh_t"], rnn["H"], rnn["D"], 1, ldt, t, 1)
        y, _ = hip.conv1x1(hseq, t, proj["wt"], proj["M"], None, proj["bias"],
                           out=torch.empty(n, proj["M"], ldt, dtype=torch.float32, device=x.device))
        return y

    def forward(self, x: torch.Tensor):
        """x [N, C, T] -> [N, C, T]."""
        hip.require_device(x, "SingleRNN.forward")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t), t)
