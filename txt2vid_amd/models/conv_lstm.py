"""ConvLSTM of the TGANv2 frame-seed generator — same parameters / state_dict as
txt2vid/models/conv_lstm.py:6-97; the recurrence runs in `functional.ConvLSTMFn` (gate convolutions
on the MFMA conv kernel with padding-only taps skipped, fused gate math, explicit BPTT)."""
import torch.nn as nn

from .. import functional as TF
from .layers import Conv2d


class ConvLSTMCell(nn.Module):
    def __init__(self, input_channels, hidden_channels, kernel_size):
        super().__init__()
        assert hidden_channels % 2 == 0
        self.input_channels, self.hidden_channels, self.kernel_size = input_channels, hidden_channels, kernel_size
        self.num_features = 4
        self.padding = int((kernel_size - 1) / 2)
        a = (self.kernel_size, 1, self.padding)
        self.Wxi = Conv2d(input_channels, hidden_channels, *a, bias=True)
        self.Whi = Conv2d(hidden_channels, hidden_channels, *a, bias=False)
        self.Wxf = Conv2d(input_channels, hidden_channels, *a, bias=True)
        self.Whf = Conv2d(hidden_channels, hidden_channels, *a, bias=False)
        self.Wxc = Conv2d(input_channels, hidden_channels, *a, bias=True)
        self.Whc = Conv2d(hidden_channels, hidden_channels, *a, bias=False)
        self.Wxo = Conv2d(input_channels, hidden_channels, *a, bias=True)
        self.Who = Conv2d(hidden_channels, hidden_channels, *a, bias=False)
        # the reference's peephole terms Wci/Wcf/Wco are constant zeros (conv_lstm.py:47-49): dropped.
        # Master copies TAP-MAJOR in memory ([kh][kw][Cout][Cin]; shape, values and state_dict stay [Cout,Cin,kh,kw]): on the
        # generator's 1x1 state only the centre tap is ever used or receives a gradient, so with this layout the live part of
        # each 37.7 MB weight — and of its gradient and Adam moments — is ONE contiguous 4.2 MB slice: the optimiser, the
        # re-packing after it and the data-parallel exchange touch 33.5 MB instead of 302 MB per step.
        for conv in (self.Wxi, self.Whi, self.Wxf, self.Whf, self.Wxc, self.Whc, self.Wxo, self.Who):
            conv.weight = nn.Parameter(TF.tap_major(conv.weight.data))

    def gate_params(self):
        wx = [self.Wxi.weight, self.Wxf.weight, self.Wxc.weight, self.Wxo.weight]
        bx = [self.Wxi.bias, self.Wxf.bias, self.Wxc.bias, self.Wxo.bias]
        wh = [self.Whi.weight, self.Whf.weight, self.Whc.weight, self.Who.weight]
        return wx, bx, wh


class ConvLSTM(nn.Module):
    """Single-layer use only (the generator's `hidden_channels=[fm_channels]`)."""

    def __init__(self, input_channels, hidden_channels, kernel_size, step=1, effective_step=[1]):
        super().__init__()
        if len(hidden_channels) != 1 or input_channels != hidden_channels[0]:
            raise NotImplementedError('hot path: one cell with input_channels == hidden_channels')
        self.input_channels = [input_channels] + list(hidden_channels)
        self.hidden_channels = hidden_channels
        self.kernel_size = kernel_size
        self.num_layers = 1
        self.step = step
        self.effective_step = effective_step
        self.cell0 = ConvLSTMCell(input_channels, hidden_channels[0], kernel_size)

    def forward(self, input):
        wx, bx, wh = self.cell0.gate_params()
        hs = TF.conv_lstm(input, self.step, wx, bx, wh)           # [steps,B,C,h,w]
        outputs = [hs[t] for t in range(self.step) if t in self.effective_step]
        return outputs, (hs[self.step - 1], None)

    def forward_stacked(self, input):
        """[steps,B,C,h,w] without unpacking (the generator consumes all 16 steps)."""
        wx, bx, wh = self.cell0.gate_params()
        return TF.conv_lstm(input, self.step, wx, bx, wh)
