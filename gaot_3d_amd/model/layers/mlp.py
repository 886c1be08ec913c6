"""MLP building blocks with the reference's names and state_dict layout (src/model/layers/mlp.py:
LinearChannelMLP :308-335, ChannelMLP :227-305); every affine map runs on the HIP GEMM."""
import torch
import torch.nn as nn

from ... import functional as GF

# per-node single linears (lifting / recovery / geoembed) stay in exact-fp32 MFMA in every mode: they are HBM-bound, so
# bf16 operands would buy nothing.  The two-layer projection is fused (Mlp2Fn) in bf16 mode.
_FP32 = 0


def activation_name(fn) -> str:
    """The reference passes activations as callables (``non_linearity=F.gelu``, mlp.py:227-335,
    ``channel_mlp_non_linearity=F.gelu``, integral_transform.py:35) or builds them from a name with ``activation_fn``
    (mlp.py:27-35: "none", "swish", or any ``F.<name>``); the HIP kernels take an activation id (csrc/common.h GAOT_ACT_*).
    Accepts the callable (F.gelu, torch.tanh, nn.ELU() ... with torch's default parameters) or its name.  The fused GNO /
    projection kernels are built for erf-GELU; every other activation runs on the general per-edge / per-node path."""
    import torch.nn.functional as F
    from ...ops import ACT
    if fn is None:
        return "none"
    if isinstance(fn, str):
        name = {"swish": "silu", "identity": "none"}.get(fn.lower(), fn.lower())
    elif isinstance(fn, nn.GELU):
        name = "gelu" if getattr(fn, "approximate", "none") == "none" else "gelu_tanh"
    elif isinstance(fn, nn.Identity):
        name = "none"
    elif isinstance(fn, nn.Module):
        table = {nn.ReLU: "relu", nn.SiLU: "silu", nn.Tanh: "tanh", nn.LeakyReLU: "leaky_relu", nn.ELU: "elu",
                 nn.Sigmoid: "sigmoid", nn.Softplus: "softplus", nn.SELU: "selu", nn.ReLU6: "relu6",
                 nn.Hardswish: "hardswish", nn.Mish: "mish"}
        name = table.get(type(fn))
        defaults = {"leaky_relu": ("negative_slope", 0.01), "elu": ("alpha", 1.0)}
        if name in defaults and getattr(fn, defaults[name][0]) != defaults[name][1]:
            raise NotImplementedError(f"{fn!r}: only torch's default parameters have a HIP kernel")
        if name == "softplus" and (fn.beta != 1.0 or fn.threshold != 20.0):
            raise NotImplementedError(f"{fn!r}: only torch's default parameters have a HIP kernel")
    else:
        table = {F.gelu: "gelu", F.relu: "relu", torch.relu: "relu", F.silu: "silu", F.tanh: "tanh", torch.tanh: "tanh",
                 F.leaky_relu: "leaky_relu", F.elu: "elu", F.sigmoid: "sigmoid", torch.sigmoid: "sigmoid",
                 F.softplus: "softplus", F.selu: "selu", F.relu6: "relu6", F.hardswish: "hardswish", F.mish: "mish"}
        name = table.get(fn)
    if name is None or name not in ACT:
        raise NotImplementedError(f"activation {fn!r} has no HIP kernel (supported: {sorted(k for k in ACT if k)})")
    return name


class LinearChannelMLP(nn.Module):
    """Stack of nn.Linear parameters [layers[j] -> layers[j+1]], activation (default erf-GELU) between, none after the
    last, nn.Dropout after every layer when ``dropout`` > 0 (reference mlp.py:308-335)."""

    def __init__(self, layers, non_linearity="gelu", dropout=0.0):
        super().__init__()
        self.n_layers = len(layers) - 1
        assert self.n_layers >= 1
        self.non_linearity = activation_name(non_linearity)
        self.dropout_p = float(dropout)
        self.fcs = nn.ModuleList([nn.Linear(layers[j], layers[j + 1]) for j in range(self.n_layers)])

    def forward(self, x):
        p = self.dropout_p if self.training else 0.0
        if p == 0.0 and GF.Mlp2Fn.eligible(x, self.fcs, self.non_linearity):   # bf16 mode, C -> {64,128,256} -> <=4: fused
            return GF.Mlp2Fn.apply(x, self.fcs[0].weight, self.fcs[0].bias, self.fcs[1].weight, self.fcs[1].bias)
        for i, fc in enumerate(self.fcs):
            x = GF.linear(x, fc.weight, fc.bias, act=self.non_linearity if i < self.n_layers - 1 else None,
                          precision=_FP32)
            x = GF.dropout(x, p, self.training)
        return x

    forward_rows = forward      # [N, C] rows in, rows out (what the encoder / decoder hand over)


class ChannelMLP(nn.Module):
    """Conv1d(k=1) parameter storage ([out,in,1]) of mlp_type='channel' (reference mlp.py:227-305).  ``forward`` takes the
    reference's channels-first layout -- [C, N], [B, C, N] or [B, C, x1, x2, ...] -- and returns the same layout;
    numerically a Conv1d(k=1) is the per-node affine map, so the kernels run on row-major [N, C] and ``forward_rows`` is
    the entry point the encoder / decoder use (they hold [N, C] and the reference only transposes around this module,
    magno.py:545,575,775,796-797)."""

    def __init__(self, in_channels, out_channels=None, hidden_channels=None, n_layers=2, n_dim=2,
                 non_linearity="gelu", dropout=0.0, **kwargs):
        super().__init__()
        self.n_layers = n_layers
        self.in_channels = in_channels
        self.out_channels = in_channels if out_channels is None else out_channels
        self.hidden_channels = in_channels if hidden_channels is None else hidden_channels
        self.non_linearity = activation_name(non_linearity)
        self.dropout_p = float(dropout)
        self.fcs = nn.ModuleList()
        for i in range(n_layers):
            cin = self.in_channels if i == 0 else self.hidden_channels
            cout = self.out_channels if i == n_layers - 1 else self.hidden_channels
            self.fcs.append(nn.Conv1d(cin, cout, 1))

    def forward_rows(self, x):
        p = self.dropout_p if self.training else 0.0
        if p == 0.0 and GF.Mlp2Fn.eligible(x, self.fcs, self.non_linearity):   # bf16 mode, C -> {64,128,256} -> <=4: fused
            return GF.Mlp2Fn.apply(x, self.fcs[0].weight, self.fcs[0].bias, self.fcs[1].weight, self.fcs[1].bias)
        for i, fc in enumerate(self.fcs):
            x = GF.linear(x, fc.weight, fc.bias, act=self.non_linearity if i < self.n_layers - 1 else None,
                          precision=_FP32)
            x = GF.dropout(x, p, self.training)
        return x

    def forward(self, x):
        size = list(x.shape)
        if x.dim() == 2:                                   # unbatched Conv1d input [C, N]
            return self.forward_rows(x.transpose(0, 1)).transpose(0, 1)
        if x.dim() < 2:
            raise ValueError(f"ChannelMLP expects [C, N] or [B, C, ...], got {tuple(size)}")
        b, c = size[0], size[1]
        rows = x.reshape(b, c, -1).permute(0, 2, 1).reshape(-1, c)      # [B * N, C]
        y = self.forward_rows(rows)
        return y.view(b, -1, self.out_channels).permute(0, 2, 1).reshape(b, self.out_channels, *size[2:])


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
