"""Recurrent lobes (mirror of puresound/nnet/lobe/rnn.py:9-55): SingleRNN holds an nn.LSTM and its projection under
the reference's keys (`rnn.*`, `proj.*`); the dual-path blocks drive it through ps_lstm_f32."""
import torch
import torch.nn as nn


class SingleRNN(nn.Module):
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

    def forward(self, x: torch.Tensor):
        raise NotImplementedError("SingleRNN runs inside DPRNNblock2D on the HIP path (ps_lstm_f32); LSTM cells only")
