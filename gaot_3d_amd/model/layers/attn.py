"""Latent U-ViT Transformer with the reference's class names, config dataclasses and state_dict layout
(src/model/layers/attn.py: configs :15-44, GroupQueryFlashAttention :51-135, FFN :137-165, RMSNorm
:167-178, TransformerBlock :180-244, Transformer :246-325), computing through the HIP kernels:
RMSNorm row kernel, MFMA GEMMs (q|k|v and w1|w3 written into one fused buffer each), RoPE, flash
attention fwd/bwd, SwiGLU.  Time-conditional norm (use_conditional_norm, reference mlp.py:74-128) rescales the input
of attention / FFN per batch element (needs ``condition`` as a [batch, 1] tensor)."""
import os
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn

from ... import functional as GF
from ...utils.dataclass import shallow_asdict
from .mlp import ConditionedNorm


@dataclass
class AttentionConfig:
    hidden_size: int = 256
    num_heads: int = 8
    num_kv_heads: int = 8
    use_conditional_norm: bool = False
    cond_norm_hidden_size: int = 4
    atten_dropout: float = 0.1
    positional_embedding: str = "absolute"
    H: Optional[int] = None
    W: Optional[int] = None


@dataclass
class FFNConfig:
    hidden_size: int = 1024
    use_conditional_norm: bool = False
    cond_norm_hidden_size: int = 4


@dataclass
class TransformerConfig:
    patch_size: int = 8
    hidden_size: int = 256
    use_attn_norm: bool = True
    use_ffn_norm: bool = True
    norm_eps: float = 1e-6
    num_layers: int = 3
    positional_embedding: str = "absolute"
    use_long_range_skip: bool = True
    attn_config: AttentionConfig = field(default_factory=AttentionConfig)
    ffn_config: FFNConfig = field(default_factory=FFNConfig)


SKIP_TAPS = {"on": os.environ.get("GAOT_SKIP_TAPS", "1") != "0"}      # A/B switch of Transformer._forward's skip taps (tests)
SEED_BLOCK = {"on": os.environ.get("GAOT_SEED_BLOCK", "1") != "0"}    # A/B switch: the blocks' attention seeds by one launch


class RotaryEmbedding(nn.Module):
    """Parameter container matching ``rotary_embedding_torch.RotaryEmbedding(dim)`` as the reference uses it
    (attn.py:86-87): a non-trainable parameter ``freqs`` = 1/theta^(arange(0,dim,2)/dim).  The rotation itself
    (interleaved pairs, position = flattened token index) is fused into the attention Function (csrc/rowops.hip
    k_rope).  rope: third-party, unpinned."""

    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)


class GroupQueryFlashAttention(nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_size: int = 128, num_heads: int = 8,
                 num_kv_heads: int = 4, use_conditional_norm: bool = False, cond_norm_hidden_size: int = 4,
                 atten_dropout: float = 0.0, H: int = 64, W: int = 64, positional_embedding: str = "absolute"):
        super().__init__()
        assert hidden_size % num_heads == 0, f"hidden_size {hidden_size} must be divisible by num_heads {num_heads}"
        assert num_heads % num_kv_heads == 0, f"num_heads {num_heads} must be divisible by num_kv_heads {num_kv_heads}"
        self.num_heads = num_heads
        self.num_kv_heads = num_kv_heads
        self.num_repeat = num_heads // num_kv_heads
        self.head_dim = hidden_size // num_heads   # 32: flash kernels; anything else: the unfused general path
        self.atten_dropout = atten_dropout
        kv_hidden = self.head_dim * num_kv_heads
        self.q_proj = nn.Linear(input_size, hidden_size, bias=False)
        self.k_proj = nn.Linear(input_size, kv_hidden, bias=False)
        self.v_proj = nn.Linear(input_size, kv_hidden, bias=False)
        self.o_proj = nn.Linear(hidden_size, output_size, bias=False)
        self.correction = ConditionedNorm(1, output_size, cond_norm_hidden_size) if use_conditional_norm else None
        if positional_embedding == "rope":
            self.rotary_emb = RotaryEmbedding(dim=self.head_dim)

    def forward(self, x, condition: Optional[float] = None, relative_positions: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, project: bool = True, qkv: Optional[torch.Tensor] = None):
        """``residual`` (extension): added to the output inside the o_proj GEMM epilogue (the block's `x + attn(...)`).
        ``project=False`` (extension, TransformerBlock): return the heads' outputs [B*S, heads * head_dim] WITHOUT o_proj -- the caller
        applies it (functional.BlockTailFn: o_proj, the residual, ffn_norm and the FFN in one launch)."""
        dp = float(self.atten_dropout) if self.training else 0.0     # reference attn.py:122-126
        if qkv is not None:
            # (extension, TransformerBlock) the q | k | v projection already exists as the kernels' image (functional.NormQKVFn): x is only
            # consulted for its shape; unsharded, head_dim 32
            b, s, _ = x.shape
            freqs = self.rotary_emb.freqs if (relative_positions is not None and hasattr(self, "rotary_emb")) else None
            o = GF.AttentionFn.apply(qkv, freqs, b, s, self.num_heads, self.num_kv_heads, dp, None)
            if not project:
                o._gaot_attn_dims = (b, s, self.num_heads, self.num_kv_heads)
                return o
            y = GF.linear(o, self.o_proj.weight, None, residual=None if residual is None else residual.reshape(b * s, -1))
            return y.view(b, s, -1)
        if self.correction is not None:                               # attn.py:101-102
            x = self.correction(c=condition, x=x)
        b, s, _ = x.shape
        GF.colocate([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight])   # no-op once done
        freqs = self.rotary_emb.freqs if (relative_positions is not None and hasattr(self, "rotary_emb")) else None
        seq_group = getattr(self, "_seq_group", None)
        # unsharded bf16 path: the projection is written straight as the attention kernels' bf16 image (RoPE and the q scale
        # in the GEMM epilogue), see MultiLinearFn
        spec = None
        if self.head_dim == 32 and seq_group is None and getattr(self, "_head_group", None) is None:
            spec = (freqs, b, s, self.num_heads, self.num_kv_heads, 1.0 / (32 ** 0.5))
        if seq_group is not None and self.head_dim == 32:
            from ...sharding import SeqAttnFn, seq_attn_eligible
            if seq_attn_eligible(x, self):
                # sequence-parallel, bf16 mode: projection, both all-to-alls and the flash kernels as one autograd node; every
                # exchanged tensor is bf16 and is written / read in the exchange layout by the producing / consuming kernel
                assert b == 1, "sequence-parallel attention splits ONE sample"
                o = SeqAttnFn.apply(x, seq_group, self.num_heads, self.num_kv_heads, freqs, dp, self.q_proj.weight,
                                    self.k_proj.weight, self.v_proj.weight)
                if not project:
                    return o.reshape(b * s, -1)
                y = GF.linear(o.reshape(b * s, -1), self.o_proj.weight, None,
                              residual=None if residual is None else residual.reshape(b * s, -1))
                return y.view(b, s, -1)
        qkv = GF.multi_linear(x, [self.q_proj.weight, self.k_proj.weight, self.v_proj.weight], image_spec=spec)  # [B*S, (h+2hkv)*32]
        if self.head_dim != 32:
            # any other head size: the general path (functional.attention_general); sharded, the same exchanges around it
            head_group, hd = getattr(self, "_head_group", None), self.head_dim
            if seq_group is not None:
                import torch.distributed as dist
                from ...sharding import HeadsToSeqFn, SeqToHeadsFn
                g = dist.get_world_size(seq_group)
                assert b == 1, "sequence-parallel attention splits ONE sample"
                ql = SeqToHeadsFn.apply(qkv, seq_group, self.num_heads, self.num_kv_heads, hd)       # all rows, my heads
                o = GF.attention_general(ql, freqs, b, s * g, self.num_heads // g, self.num_kv_heads // g, hd, dp,
                                         dist.get_rank(seq_group) * (self.num_heads // g), self.num_heads)
                o = HeadsToSeqFn.apply(o, seq_group)
            elif head_group is not None:
                import torch.distributed as dist
                from ...sharding import GatherHeadsFn, LocalHeadsFn
                g = dist.get_world_size(head_group)
                if self.num_heads % g or self.num_kv_heads % g:
                    raise ValueError(f"head-parallel attention: {self.num_heads} / {self.num_kv_heads} heads do not divide over {g} ranks")
                ql = LocalHeadsFn.apply(qkv, head_group, self.num_heads, self.num_kv_heads, hd)
                o = GatherHeadsFn.apply(GF.attention_general(ql, freqs, b, s, self.num_heads // g, self.num_kv_heads // g, hd, dp,
                                                             dist.get_rank(head_group) * (self.num_heads // g), self.num_heads),
                                        head_group)
            else:
                o = GF.attention_general(qkv, freqs, b, s, self.num_heads, self.num_kv_heads, hd, dp)
        elif seq_group is not None:
            # sequence-parallel (gaot_3d_amd/sharding.py): x holds this rank's token rows; one all-to-all hands every rank
            # ALL rows of ITS heads, the kernels run unchanged on them, a second all-to-all brings the rows back
            import torch.distributed as dist
            from ...sharding import HeadsToSeqFn, SeqToHeadsFn
            g, r = dist.get_world_size(seq_group), dist.get_rank(seq_group)
            assert b == 1, "sequence-parallel attention splits ONE sample"
            qkv = SeqToHeadsFn.apply(qkv, seq_group, self.num_heads, self.num_kv_heads)
            o = GF.AttentionFn.apply(qkv, freqs, b, s * g, self.num_heads // g, self.num_kv_heads // g, dp, None,
                                     r * (self.num_heads // g), self.num_heads)
            o = HeadsToSeqFn.apply(o, seq_group)
        else:
            o = GF.AttentionFn.apply(qkv, freqs, b, s, self.num_heads, self.num_kv_heads, dp,
                                     getattr(self, "_head_group", None))
        if not project:
            if (self.head_dim == 32 and seq_group is None and getattr(self, "_head_group", None) is None and o.dim() == 2
                    and o.shape[0] == b * s):
                # straight out of AttentionFn: its backward can take dO as the kernels' bf16 image (functional.BlockTailFn)
                o._gaot_attn_dims = (b, s, self.num_heads, self.num_kv_heads)
                return o
            return o.reshape(b * s, -1)
        y = GF.linear(o, self.o_proj.weight, None, residual=None if residual is None else residual.reshape(b * s, -1))
        return y.view(b, s, -1)

    @classmethod
    def from_config(cls, input_size: int, output_size: int, config: AttentionConfig):
        return cls(input_size, output_size, **shallow_asdict(config))


class FFN(nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_size: int = 256, use_conditional_norm: bool = False,
                 cond_norm_hidden_size: int = 4):
        super().__init__()
        self.w1 = nn.Linear(input_size, hidden_size, bias=False)
        self.w2 = nn.Linear(hidden_size, output_size, bias=False)
        self.w3 = nn.Linear(input_size, hidden_size, bias=False)
        self.correction = ConditionedNorm(1, output_size, cond_norm_hidden_size) if use_conditional_norm else None
        self.hidden = hidden_size

    def forward(self, x, condition: Optional[float] = None, residual: Optional[torch.Tensor] = None):
        """``residual`` (extension): added inside the w2 GEMM epilogue (the block's `h + ffn(h)`)."""
        if self.correction is not None:     # attn.py:155-159: the correction acts on the FFN's OUTPUT, before the residual
            y = self.correction(c=condition, x=self._forward_ffn(x, None))
            return y if residual is None else GF.add(y, residual.view_as(y))
        return self._forward_ffn(x, residual)

    def _forward_ffn(self, x, residual):
        shp = x.shape
        GF.colocate([self.w1.weight, self.w3.weight])
        if GF.FFNFn.eligible(x, self.w1.weight, self.w3.weight, self.w2.weight):   # bf16 path: bf16 intermediates
            same = residual is x
            return GF.FFNFn.apply(x, self.w1.weight, self.w3.weight, self.w2.weight, None if same else residual, same)
        ag = GF.multi_linear(x, [self.w1.weight, self.w3.weight])   # [rows, 2F] = [w1 x | w3 x]
        u = GF.SwiGLUFn.apply(ag, self.hidden)
        res = None if residual is None else residual.reshape(-1, residual.shape[-1])
        return GF.linear(u, self.w2.weight, None, residual=res).view(*shp[:-1], -1)

    @classmethod
    def from_config(cls, input_size: int, output_size: int, config: FFNConfig):
        return cls(input_size, output_size, **shallow_asdict(config))


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        y, yb = GF.RMSNormFn.apply(x, self.weight, self.eps)
        if yb.numel():
            y._gaot_bf16 = yb      # picked up by the GEMM that consumes y (functional.bf16_copy_of)
        return y

    def forward_with_residual(self, x, tap: bool = False):
        """(norm(x), x) with x routed through the same autograd node: see functional.RMSNormResFn; ``tap``: (norm(x), x, x) -- a
        third alias of x for the U-ViT's long-range skip"""
        out = GF.RMSNormResFn.apply(x, self.weight, self.eps, bool(tap))
        y, xres, yb = out[:3]
        if yb.numel():
            y._gaot_bf16 = yb
        return (y, xres, out[3]) if tap else (y, xres)


class TransformerBlock(nn.Module):
    def __init__(self, input_size: int, output_size: int, use_attn_norm: bool = True, use_ffn_norm: bool = True,
                 norm_eps: float = 1e-6, attn_config: AttentionConfig = None, ffn_config: FFNConfig = None,
                 skip_connection: bool = False):
        super().__init__()
        attn_config = attn_config if attn_config is not None else AttentionConfig()
        ffn_config = ffn_config if ffn_config is not None else FFNConfig()
        self.attn = GroupQueryFlashAttention.from_config(input_size, attn_config.hidden_size, config=attn_config)
        self.ffn = FFN.from_config(attn_config.hidden_size, output_size, config=ffn_config)
        self.attn_norm = RMSNorm(input_size, eps=norm_eps) if use_attn_norm else None
        self.ffn_norm = RMSNorm(attn_config.hidden_size, eps=norm_eps) if use_ffn_norm else None
        self.skip_connection = skip_connection
        if self.skip_connection:
            self.skip_proj = nn.Linear(input_size + output_size, input_size)

    def forward(self, x, condition: Optional[float] = None, relative_positions: Optional[torch.Tensor] = None,
                skip: Optional[torch.Tensor] = None, want_input_tap: bool = False):
        """``want_input_tap`` (extension, used by Transformer._forward): also return an alias of the block's INPUT whose gradient the
        attention norm's backward kernel adds itself -- the caller hands it to the mirrored decoder block as its long-range skip
        (reference attn.py:282-288) instead of the input tensor, which then keeps a single consumer.  None when the block cannot
        provide one (no attention norm, or a skip projection in front of it)."""
        tap, asked = None, want_input_tap
        qkv = None
        a = self.attn
        head_ok = (self.attn_norm is not None and x.is_cuda and x.shape[-1] == 256 and torch.is_grad_enabled() and a.correction is None
                   and a.head_dim == 32 and getattr(a, "_seq_group", None) is None and getattr(a, "_head_group", None) is None)
        if head_ok:
            GF.colocate([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight])   # (no-op once done)
            b_, s_, _d = x.shape
            freqs = a.rotary_emb.freqs if (relative_positions is not None and hasattr(a, "rotary_emb")) else None
            spec = (freqs, b_, s_, a.num_heads, a.num_kv_heads, 1.0 / (32 ** 0.5))
            wts = (a.q_proj.weight, a.k_proj.weight, a.v_proj.weight)
        if self.skip_connection and skip is not None:
            b, s, d = x.shape
            want_input_tap = False
            if head_ok and skip.shape == x.shape and GF.CatNormQKVFn.eligible(x, skip, self.skip_proj.weight, self.attn_norm.weight, wts, spec):
                # the skip projection, attn_norm, the three projections and RoPE in ONE launch
                qkv, xres = GF.CatNormQKVFn.apply(x, skip, self.skip_proj.weight, self.skip_proj.bias, self.attn_norm.weight, self.attn_norm.eps,
                                                  spec, *wts)
                x = h = xres
            else:
                x = GF.cat_linear([x.reshape(b * s, d), skip.reshape(b * s, -1)], self.skip_proj.weight,
                                  self.skip_proj.bias).view(b, s, -1)
        if head_ok and qkv is None:
            if GF.NormQKVFn.eligible(x, self.attn_norm.weight, wts, spec):
                # attn_norm, the three projections and RoPE in ONE launch, written as the attention kernels' image
                outs = GF.NormQKVFn.apply(x, self.attn_norm.weight, self.attn_norm.eps, bool(want_input_tap), spec, *wts)
                qkv, xres = outs[0], outs[1]
                tap = outs[2] if want_input_tap else None
                h = x      # (shape only: the attention module reads the image)
        if qkv is not None:
            pass
        elif self.attn_norm is None:
            h, xres = x, x
        elif want_input_tap and x.is_cuda:
            h, xres, tap = self.attn_norm.forward_with_residual(x, tap=True)
        else:
            h, xres = self.attn_norm.forward_with_residual(x)
        f = self.ffn
        if (self.ffn_norm is not None and f.correction is None and x.is_cuda and x.shape[-1] == 256 and torch.is_grad_enabled()
                and GF.BlockTailFn.enabled()):
            GF.colocate([f.w1.weight, f.w3.weight])     # (no-op once done)
            if GF.BlockTailFn.eligible(xres, self.attn.o_proj.weight, self.ffn_norm.weight, f.w1.weight, f.w3.weight, f.w2.weight):
                # o_proj, the first residual, ffn_norm, the FFN and the second residual in ONE launch (bf16 mode, d_model 256)
                o = self.attn(h, condition=condition, relative_positions=relative_positions, project=False, qkv=qkv)
                b, s, d = x.shape
                out = GF.BlockTailFn.apply(o, xres.reshape(b * s, d), self.attn.o_proj.weight, self.ffn_norm.weight, self.ffn_norm.eps,
                                           f.w1.weight, f.w3.weight, f.w2.weight, getattr(o, "_gaot_attn_dims", None)).view(b, s, d)
                return (out, tap) if asked else out
        h = self.attn(h, condition=condition, relative_positions=relative_positions, residual=xres, qkv=qkv)   # x + attn(norm(x))
        if self.ffn_norm is not None and f.correction is None and h.is_cuda and h.shape[-1] == 256 and torch.is_grad_enabled():
            GF.colocate([f.w1.weight, f.w3.weight])     # (no-op once done)
        if (self.ffn_norm is not None and f.correction is None
                and GF.NormFFNFn.eligible(h, self.ffn_norm.weight, f.w1.weight, f.w3.weight, f.w2.weight)):
            # ffn_norm, the FFN and the residual in one launch (bf16 mode, d_model 256)
            out = GF.NormFFNFn.apply(h, self.ffn_norm.weight, self.ffn_norm.eps, f.w1.weight, f.w3.weight, f.w2.weight)
            return (out, tap) if asked else out
        h = h if self.ffn_norm is None else self.ffn_norm(h)
        # NB: the second residual adds the *normalised* h (reference attn.py:226-229)
        out = self.ffn(h, condition=condition, residual=h)                                           # h + ffn(h)
        return (out, tap) if asked else out

    @classmethod
    def from_config(cls, input_size: int, output_size: int, skip_connection: bool = False,
                    config: TransformerConfig = None):
        config = config if config is not None else TransformerConfig()
        config.attn_config.positional_embedding = config.positional_embedding
        kwargs = shallow_asdict(config)
        for k in ("num_layers", "hidden_size", "positional_embedding", "use_long_range_skip", "patch_size"):
            kwargs.pop(k)
        return cls(input_size, output_size, skip_connection=skip_connection, **kwargs)


class Transformer(nn.Module):
    def __init__(self, input_size: int, output_size: int, config: TransformerConfig = None):
        super().__init__()
        config = config if config is not None else TransformerConfig()
        hidden = config.hidden_size
        n = config.num_layers
        self.use_long_range_skip = config.use_long_range_skip
        self.input_proj = nn.Linear(input_size, hidden) if input_size != hidden else nn.Identity()
        self.output_proj = nn.Linear(hidden, output_size) if hidden != output_size else nn.Identity()
        self.encoder_layers = nn.ModuleList([TransformerBlock.from_config(hidden, hidden, False, config)
                                             for _ in range(n // 2)])
        self.middle_layer = TransformerBlock.from_config(hidden, hidden, False, config) if n % 2 == 1 else None
        self.decoder_layers = nn.ModuleList([TransformerBlock.from_config(hidden, hidden, True, config)
                                             for _ in range(n // 2)])

    def _matrices(self):
        """every weight matrix a block hands to a bf16 GEMM, in the fused form the operators use"""
        mats = []
        blocks = list(self.encoder_layers) + ([self.middle_layer] if self.middle_layer is not None else []) + list(self.decoder_layers)
        for blk in blocks:
            a, f = blk.attn, blk.ffn
            GF.colocate([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight])
            GF.colocate([f.w1.weight, f.w3.weight])
            mats += [GF.fused_view([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight]), a.o_proj.weight,
                     GF.fused_view([f.w1.weight, f.w3.weight]), f.w2.weight]
            if getattr(blk, "skip_proj", None) is not None:
                mats.append(blk.skip_proj.weight)
        for proj in (self.input_proj, self.output_proj):
            if isinstance(proj, nn.Linear):
                mats.append(proj.weight)
        return mats

    def _ffn_matrices(self):
        """the FFN weights whose bf16 transposes the backward wants (FFNFn.backward)"""
        blocks = list(self.encoder_layers) + ([self.middle_layer] if self.middle_layer is not None else []) + list(self.decoder_layers)
        mats = []
        for blk in blocks:
            f = blk.ffn
            mats += [GF.fused_view([f.w1.weight, f.w3.weight]), f.w2.weight]
        return mats

    def forward(self, x, condition: Optional[float] = None, relative_positions: Optional[torch.Tensor] = None):
        if x.is_cuda:    # bf16 mode: ONE launch rounds all weight matrices of the blocks (46 separate casts otherwise)
            mats = self._matrices()   # co-locates w1|w3 first
            ffn = self._ffn_matrices()
            GF.precast_weights(mats, ffn if self.training else ())
            # fragment-ordered images for the fused FFN kernels, one launch (with the backward's images when it will run)
            blocks = list(self.encoder_layers) + ([self.middle_layer] if self.middle_layer is not None else []) + list(self.decoder_layers)
            GF.prepack_ffn(zip(ffn[0::2], ffn[1::2]), self.training and torch.is_grad_enabled(), wos=[blk.attn.o_proj.weight for blk in blocks])
            GF.prepack_qkv([GF.fused_view([blk.attn.q_proj.weight, blk.attn.k_proj.weight, blk.attn.v_proj.weight]) for blk in blocks]
                           if torch.is_grad_enabled() else [])
            GF.prepack_skip([blk.skip_proj.weight for blk in blocks if blk.skip_connection] if torch.is_grad_enabled() else [])
            if self.training and getattr(self, "_seq_group", None) is None and SEED_BLOCK["on"]:
                # the attention seeds of all blocks with ONE launch (each block draws its own otherwise)
                n_drop = sum(1 for blk in blocks if float(blk.attn.atten_dropout) > 0.0 and blk.attn.head_dim == 32)
                GF.reserve_dropout_seeds(x.device, n_drop)
        try:
            return self._forward(x, condition, relative_positions)
        finally:
            GF.release_precast()
            if x.is_cuda:
                GF.release_dropout_seeds(x.device)

    def _forward(self, x, condition, relative_positions):
        if isinstance(self.input_proj, nn.Linear):
            x = GF.linear(x, self.input_proj.weight, self.input_proj.bias)
        # long-range skips: the output of encoder block i feeds block i + 1 AND decoder block n - 1 - i.  The skip handed to the decoder
        # is an alias produced by the NEXT block's attention norm (RMSNormResFn tap): both gradients of x_i then meet inside that
        # norm's backward kernel and the autograd engine runs no accumulation pass over [S, d] per skip (4 ATen adds per step at L = 10)
        skips = []
        taps = self.use_long_range_skip and torch.is_grad_enabled() and SKIP_TAPS["on"]
        for li, layer in enumerate(self.encoder_layers):
            if taps and li > 0:
                x, tap = layer(x, condition=condition, relative_positions=relative_positions, want_input_tap=True)
                if tap is not None:
                    skips[-1] = tap
            else:
                x = layer(x, condition=condition, relative_positions=relative_positions)
            skips.append(x)
        if self.middle_layer is not None:
            if taps and skips:
                x, tap = self.middle_layer(x, condition=condition, relative_positions=relative_positions, want_input_tap=True)
                if tap is not None:
                    skips[-1] = tap
            else:
                x = self.middle_layer(x, condition=condition, relative_positions=relative_positions)
        for layer in self.decoder_layers:
            skip = skips.pop() if self.use_long_range_skip else None
            x = layer(x, condition=condition, relative_positions=relative_positions, skip=skip)
        if isinstance(self.output_proj, nn.Linear):
            x = GF.linear(x, self.output_proj.weight, self.output_proj.bias)
        return x
