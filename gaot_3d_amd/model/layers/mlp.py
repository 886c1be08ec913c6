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


class MLP(nn.Module):
    """reference src/model/layers/mlp.py:44-72 (the small MLP of the time-conditioned norm): ``layers`` ModuleList of
    nn.Linear, activation between them, none after the last; num_layers <= 2 collapses to ONE Linear(input, output)"""

    def __init__(self, input_size: int, output_size: int, hidden_size: int, num_layers: int = 3, activation: str = "swish"):
        super().__init__()
        if num_layers <= 2:
            self.layers = nn.ModuleList([nn.Linear(input_size, output_size)])
        else:
            self.layers = nn.ModuleList([nn.Linear(input_size, hidden_size)])
            for _ in range(num_layers - 2):
                self.layers.append(nn.Linear(hidden_size, hidden_size))
            self.layers.append(nn.Linear(hidden_size, output_size))
        if activation not in ("none", "swish", "silu", "gelu", "relu"):
            raise ValueError(f"Activation function {activation} not found")
        self.activation = {"swish": "silu"}.get(activation, activation)

    def forward(self, x):
        act = None if self.activation == "none" else self.activation
        for layer in self.layers[:-1]:
            x = GF.linear(x, layer.weight, layer.bias, act=act, precision=_FP32)
        return GF.linear(x, self.layers[-1].weight, self.layers[-1].bias, precision=_FP32)


class ConditionedNorm(nn.Module):
    """time-conditioned normalisation (reference mlp.py:74-128): scale = 1 + c * mlp_scale(c), bias = c * mlp_bias(c),
    out = x * scale[:, None, :] + bias[:, None, :] with c [B, 1] and x [B, S, C]"""

    def __init__(self, input_size: int, output_size: int, hidden_size: int):
        super().__init__()
        self.mlp_scale = MLP(input_size, output_size, hidden_size, num_layers=2, activation="none")
        self.mlp_bias = MLP(input_size, output_size, hidden_size, num_layers=2, activation="none")
        for layer in list(self.mlp_scale.layers) + list(self.mlp_bias.layers):
            nn.init.normal_(layer.weight, std=0.01)

    def forward(self, c, x):
        from ... import edgeops as EO
        if c is None or not torch.is_tensor(c):
            raise TypeError("ConditionedNorm needs the condition as a [batch, 1] tensor (reference mlp.py:112-126)")
        b = x.shape[0]
        c = c.to(x.device, torch.float32).reshape(b, -1)
        if c.shape[1] != 1:
            raise ValueError("ConditionedNorm on the HIP path takes one conditioning scalar per batch element")
        cs = c.reshape(b).contiguous()
        sm1 = EO.RowScaleFn.apply(self.mlp_scale(c), cs)      # c * mlp_scale(c)   (scale - 1)
        bias = EO.RowScaleFn.apply(self.mlp_bias(c), cs)      # c * mlp_bias(c)
        outs = [EO.AffineColsFn.apply(x[i].reshape(-1, x.shape[-1]), sm1[i], bias[i]) for i in range(b)]
        y = outs[0].unsqueeze(0) if b == 1 else torch.stack(outs)
        return y.view_as(x)
