"""Attentive statistics pooling (mirror of puresound/nnet/lobe/pooling.py:58-126): parameters only for
now -- the speaker branch (BASELINE config 3) is a later row of the hot-path table."""
import torch
import torch.nn as nn


class AttentiveStatisticsPooling(nn.Module):
    def __init__(self, channels, attention_channels=128):
        super().__init__()
        self.eps = 1e-12
        self.tdnn = nn.Sequential(
            nn.Conv1d(in_channels=channels, out_channels=attention_channels, kernel_size=1, dilation=1),
            nn.ReLU(), nn.BatchNorm1d(attention_channels))
        self.tanh = nn.Tanh()
        self.conv = nn.Conv1d(in_channels=attention_channels, out_channels=channels, kernel_size=1)

    def forward(self, x: torch.Tensor, lengths=None, return_weight: bool = False):
        raise NotImplementedError("AttentiveStatisticsPooling has no HIP kernel yet (speaker branch, config 3)")
