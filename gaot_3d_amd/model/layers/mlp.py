"""MLP building blocks with the reference's names and state_dict layout (src/model/layers/mlp.py:
LinearChannelMLP :308-335, ChannelMLP :227-305); every affine map runs on the HIP GEMM."""
import torch
import torch.nn as nn

from ... import functional as GF

# per-node single linears (lifting / recovery / geoembed) stay in exact-fp32 MFMA in every mode: they are HBM-bound, so
# bf16 operands would buy nothing.  The two-layer projection is fused (Mlp2Fn) in bf16 mode.
_FP32 = 0


class LinearChannelMLP(nn.Module):
    """Stack of nn.Linear parameters [layers[j] -> layers[j+1]], GELU(erf) between, none after the last."""

    def __init__(self, layers, non_linearity="gelu", dropout=0.0):
        super().__init__()
        self.n_layers = len(layers) - 1
        assert self.n_layers >= 1
        if dropout > 0.0:
            raise NotImplementedError("dropout > 0 is not supported by the HIP path (reference default is 0)")
        self.non_linearity = non_linearity
        self.fcs = nn.ModuleList([nn.Linear(layers[j], layers[j + 1]) for j in range(self.n_layers)])

    def forward(self, x):
        if GF.Mlp2Fn.eligible(x, self.fcs, self.non_linearity):   # bf16 mode, C -> {64,128,256} -> <=4: fused, no [N, hidden] in HBM
            return GF.Mlp2Fn.apply(x, self.fcs[0].weight, self.fcs[0].bias, self.fcs[1].weight, self.fcs[1].bias)
        for i, fc in enumerate(self.fcs):
            x = GF.linear(x, fc.weight, fc.bias, act=self.non_linearity if i < self.n_layers - 1 else None,
                          precision=_FP32)
        return x


class ChannelMLP(nn.Module):
    """Conv1d(k=1) parameter storage ([out,in,1]) of mlp_type='channel'.  The reference applies it to
    [C, N] tensors (callers transpose around it, magno.py:545,575,775,796-797); numerically it is the
    same per-node affine map, so ``forward`` here takes and returns the row-major [N, C] layout and
    callers skip the two transposes."""

    def __init__(self, in_channels, out_channels=None, hidden_channels=None, n_layers=2, n_dim=2,
                 non_linearity="gelu", dropout=0.0, **kwargs):
        super().__init__()
        self.n_layers = n_layers
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.hidden_channels = in_channels if hidden_channels is None else hidden_channels
        if dropout > 0.0:
            raise NotImplementedError("dropout > 0 is not supported by the HIP path (reference default is 0)")
        self.non_linearity = non_linearity
        self.fcs = nn.ModuleList()
        for i in range(n_layers):
            cin = self.in_channels if i == 0 else self.hidden_channels
            cout = self.out_channels if i == n_layers - 1 else self.hidden_channels
            self.fcs.append(nn.Conv1d(cin, cout, 1))

    def forward(self, x):
        if GF.Mlp2Fn.eligible(x, self.fcs, self.non_linearity):   # bf16 mode, C -> {64,128,256} -> <=4: fused, no [N, hidden] in HBM
            return GF.Mlp2Fn.apply(x, self.fcs[0].weight, self.fcs[0].bias, self.fcs[1].weight, self.fcs[1].bias)
        for i, fc in enumerate(self.fcs):
            x = GF.linear(x, fc.weight, fc.bias, act=self.non_linearity if i < self.n_layers - 1 else None,
                          precision=_FP32)
        return x
