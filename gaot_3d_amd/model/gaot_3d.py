"""GAOT3D: encode (MAGNO) -> process (patchify, U-ViT) -> decode (MAGNO), with the reference's
constructor, buffers, parameter names and forward signature (src/model/gaot_3d.py:19-51, 248-332) so
that it drops into the reference-style trainer (`model(batch=batch, tokens_pos=...)`,
src/trainer/stat.py:544-547) and loads its checkpoints with strict=True."""
from typing import Optional

import torch
import torch.nn as nn

from .. import functional as GF
from .layers.attn import Transformer, TransformerConfig
from .layers.magno import MAGNOConfig, MAGNODecoder, MAGNOEncoder


class GAOT3D(nn.Module):
    def __init__(self, input_size: int, output_size: int, magno_config: MAGNOConfig = None,
                 attn_config: TransformerConfig = None, latent_tokens: tuple = (32, 32, 32),
                 norm_domin: list = [(-1, -1, -1), (1, 1, 1)]):
        super().__init__()
        magno_config = magno_config if magno_config is not None else MAGNOConfig()
        attn_config = attn_config if attn_config is not None else TransformerConfig()
        self.input_size = input_size
        self.output_size = output_size
        self.node_latent_size = magno_config.lifting_channels
        self.patch_size = attn_config.patch_size
        self.D, self.H, self.W = latent_tokens
        self.num_latent_tokens = self.D * self.H * self.W
        self.coord_dim = magno_config.gno_coord_dim
        lo, hi = norm_domin
        mg = torch.meshgrid(torch.linspace(lo[0], hi[0], self.D), torch.linspace(lo[1], hi[1], self.H),
                            torch.linspace(lo[2], hi[2], self.W), indexing="ij")
        self.register_buffer("latent_tokens", torch.stack(mg, dim=-1).reshape(-1, self.coord_dim))
        self.encoder = MAGNOEncoder(in_channels=input_size, out_channels=self.node_latent_size, gno_config=magno_config)
        self.processor = self.init_processor(self.node_latent_size, attn_config)
        self.decoder = MAGNODecoder(in_channels=self.node_latent_size, out_channels=output_size, gno_config=magno_config)
        # precompute_edges=False: the graphs are built on the device against the regular D x H x W token grid
        self.encoder.latent_dims = self.decoder.latent_dims = (self.D, self.H, self.W)

    def init_processor(self, node_latent_size, config):
        tok = self.patch_size ** 3 * node_latent_size
        self.patch_linear = nn.Linear(tok, tok)
        self.positional_embedding_name = config.positional_embedding
        for k, v in (("D", self.D), ("H", self.H), ("W", self.W)):
            setattr(config.attn_config, k, v)
        self._pe_cache = {}
        return Transformer(input_size=tok, output_size=tok, config=config)

    # -- constant tables -----------------------------------------------------------------------------
    def _patch_positions(self):
        p = self.patch_size
        return torch.stack(torch.meshgrid(torch.arange(self.D // p, dtype=torch.float32),
                                          torch.arange(self.H // p, dtype=torch.float32),
                                          torch.arange(self.W // p, dtype=torch.float32), indexing="ij"),
                           dim=-1).reshape(-1, 3)

    def _absolute_pe(self, device):
        """sum over the 3 patch coordinates of sin (even cols) / cos (odd cols), omega_k = 10000^(-2k/dim)
        (reference gaot_3d.py:102-144) -- a constant of the model, built once on the host."""
        key = str(device)
        if key not in self._pe_cache:
            dim = self.patch_size ** 3 * self.node_latent_size
            pos = self._patch_positions()
            freq = 1 / 10000 ** (2 * torch.arange(0, dim // 2, dtype=torch.float32) / dim)
            ang = pos[:, :, None] * freq[None, None, :]
            pe = torch.zeros(pos.shape[0], dim)
            pe[:, 0::2] = torch.sin(ang).sum(dim=1)
            pe[:, 1::2] = torch.cos(ang).sum(dim=1)
            self._pe_cache[key] = pe.to(device)
        return self._pe_cache[key]

    # -- stages ----------------------------------------------------------------------------------------
    def process(self, rndata: Optional[torch.Tensor] = None, condition: Optional[float] = None) -> torch.Tensor:
        b, m, c = rndata.shape
        d, h, w, p = self.D, self.H, self.W, self.patch_size
        assert m == d * h * w, f"n_regional_nodes ({m}) is not equal to D*H*W ({d * h * w})"
        assert d % p == 0 and h % p == 0 and w % p == 0, "Dimensions must be divisible by patch size"
        s = (d // p) * (h // p) * (w // p)
        tok = p * p * p * c
        x = GF.PatchifyFn.apply(rndata, b, d, h, w, p, c, True).view(b * s, tok)
        seq_group = getattr(self, "_seq_group", None)
        s_loc, pe = s, None
        if self.positional_embedding_name == "absolute":
            pe = self._absolute_pe(x.device)
        if seq_group is not None:
            # sequence-parallel Transformer of a point-sharded sample (gaot_3d_amd/sharding.py): every row-wise operator
            # from here to the un-patchify runs on this rank's S/G token rows
            import torch.distributed as dist
            from ..sharding import AllGatherRowsFn, SliceRowsFn
            assert b == 1, "sequence-parallel processing splits ONE sample"
            x = SliceRowsFn.apply(x, seq_group)
            s_loc = x.shape[0]
            if pe is not None:
                r0 = dist.get_rank(seq_group) * s_loc
                pe = pe[r0:r0 + s_loc].contiguous()
        x = GF.linear(x, self.patch_linear.weight, self.patch_linear.bias)
        relative_positions = None
        if pe is not None:
            x = GF.AddFn.apply(x, pe, s_loc * tok)
        elif self.positional_embedding_name == "rope":
            relative_positions = True  # the reference passes the 3-D positions only as an on/off flag
        x = self.processor(x.view(b, s_loc, tok), condition=condition, relative_positions=relative_positions)
        x = x.reshape(b * s_loc, tok)
        if seq_group is not None:
            x = AllGatherRowsFn.apply(x, seq_group)
        x = GF.PatchifyFn.apply(x, b, d, h, w, p, c, False)
        return x.view(b, d * h * w, c)

    def forward(self, batch, tokens_pos: Optional[torch.Tensor] = None, tokens_batch_idx: Optional[torch.Tensor] = None,
                query_coord_pos: Optional[torch.Tensor] = None, query_coord_batch_idx: Optional[torch.Tensor] = None,
                condition: Optional[float] = None) -> torch.Tensor:
        num_graphs = batch.num_graphs
        device = batch.pos.device
        if tokens_pos is None:
            assert tokens_batch_idx is None, "tokens_batch_idx should be None if tokens_pos is None"
            tokens_pos = self.latent_tokens
        if tokens_batch_idx is None:
            lat = tokens_pos.to(device)
            lat = lat if num_graphs == 1 else lat.repeat(num_graphs, 1)
            # a per-(batch size, device) constant: built once (arange + repeat_interleave are ATen launches on the replayed step otherwise)
            cache = self.__dict__.setdefault("_lat_bidx_cache", {})
            lat_bidx = cache.get((num_graphs, device))
            if lat_bidx is None:
                lat_bidx = torch.arange(num_graphs, device=device).repeat_interleave(self.num_latent_tokens)
                cache[(num_graphs, device)] = lat_bidx
        else:
            assert tokens_pos.shape[0] == tokens_batch_idx.shape[0], "tokens_pos and tokens_batch_idx must have same length"
            lat, lat_bidx = tokens_pos.to(device), tokens_batch_idx.to(device)
        if query_coord_pos is None:
            q_pos, q_bidx = batch.pos, batch.batch
        else:
            assert query_coord_batch_idx is not None, "query_coord_batch_idx is required if query_coord_pos is provided"
            assert query_coord_pos.shape[0] == query_coord_batch_idx.shape[0]
            q_pos, q_bidx = query_coord_pos.to(device), query_coord_batch_idx.to(device)
        lat = lat.contiguous()
        rndata = self.encoder(batch=batch, latent_tokens_pos=lat, latent_tokens_batch_idx=lat_bidx)
        rndata = self.process(rndata=rndata, condition=condition)
        flat = rndata.view(-1, self.node_latent_size)
        shard_group = getattr(self, "_shard_group", None)
        if shard_group is not None and getattr(self, "_seq_group", None) is None:
            # decoder runs on this rank's points only: its latent gradient is a partial sum (sequence-parallel: the
            # reduce-scatter inside process() already sums it)
            from ..sharding import AllReduceGradFn
            flat = AllReduceGradFn.apply(flat, shard_group)
        return self.decoder(rndata_flat=flat, phys_pos_query=q_pos, batch_idx_phys_query=q_bidx, latent_tokens_pos=lat,
                            latent_tokens_batch_idx=lat_bidx, batch=batch)
