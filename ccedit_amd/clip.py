"""CLIP text encoder (the network inside the reference's FrozenCLIPEmbedder) on the HIP kernels.

Reference call site: sgm/modules/encoders/modules.py:358-420 — `CLIPTextModel.from_pretrained(
"openai/clip-vit-large-patch14")`, `forward(text)` -> tokens -> `outputs.last_hidden_state` (layer="last") =
`crossattn` (B, 77, 768).  The architecture is HF transformers' `CLIPTextTransformer` (transformers==4.19.1 in the
reference's requirements.txt:34): token + learned position embeddings, 12 pre-LN blocks [LayerNorm, causal
12-head self-attention (d = 64), residual, LayerNorm, fc1 768->3072, quick_gelu, fc2, residual], final LayerNorm.

Parameter names are the checkpoint's (`conditioner.embedders.0.transformer.text_model.encoder.layers.3.mlp.fc1.weight`,
...), held fp32; `.pack(device)` builds the bf16 kernel operands (q|k|v fused into one GEMM).  Everything runs
through ccedit_amd.ops: embedding lookup kernel, layernorm, tap_gemm (LINEAR, quick-gelu epilogue), flash attention
with the causal flag.  It runs once per clip, outside the denoising loop.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .hip import ACT_QUICK_GELU
from .layers import Linear, Norm, pack_tree
from .packing import pack_concat

LN_EPS = 1e-5


class _Embeddings(nn.Module):
    def __init__(self, vocab: int, hidden: int, max_len: int):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, hidden)
        self.position_embedding = nn.Embedding(max_len, hidden)
        for p in self.parameters():
            p.requires_grad = False
        # persistent in transformers 4.19.1, hence present in CCEdit checkpoints
        self.register_buffer("position_ids", torch.arange(max_len).expand((1, -1)))


class _SelfAttention(nn.Module):
    def __init__(self, hidden: int, heads: int):
        super().__init__()
        self.heads, self.d = heads, hidden // heads
        self.k_proj, self.v_proj = Linear(hidden, hidden), Linear(hidden, hidden)
        self.q_proj, self.out_proj = Linear(hidden, hidden), Linear(hidden, hidden)
        self._qkv = None

    def post_pack(self, device):
        self._qkv = pack_concat([self.q_proj.weight, self.k_proj.weight, self.v_proj.weight],
                                [self.q_proj.bias, self.k_proj.bias, self.v_proj.bias], device=device)

    def run(self, h, x, b: int, l: int):
        c = self.heads * self.d
        qkv = ops.linear(h, self._qkv)                                  # [b*l, 3c]
        o = ops.attention(qkv[:, :c], qkv[:, c:2 * c], qkv[:, 2 * c:], self.heads, self.d, batches=b, lq=l, lk=l, causal=True)
        return ops.linear(o, self.out_proj.pw, res1=x)


class _MLP(nn.Module):
    def __init__(self, hidden: int, inter: int):
        super().__init__()
        self.fc1, self.fc2 = Linear(hidden, inter), Linear(inter, hidden)


class _Layer(nn.Module):
    def __init__(self, hidden: int, inter: int, heads: int):
        super().__init__()
        self.self_attn = _SelfAttention(hidden, heads)
        self.layer_norm1 = Norm(hidden, LN_EPS)
        self.mlp = _MLP(hidden, inter)
        self.layer_norm2 = Norm(hidden, LN_EPS)

    def run(self, x, b, l):
        h = ops.layernorm(x, self.layer_norm1.g, self.layer_norm1.b, LN_EPS)
        x = self.self_attn.run(h, x, b, l)
        h = ops.layernorm(x, self.layer_norm2.g, self.layer_norm2.b, LN_EPS)
        h = ops.linear(h, self.mlp.fc1.pw, act=ACT_QUICK_GELU)
        return ops.linear(h, self.mlp.fc2.pw, res1=x)


class _Encoder(nn.Module):
    def __init__(self, n: int, hidden: int, inter: int, heads: int):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(hidden, inter, heads) for _ in range(n)])


class CLIPTextTransformer(nn.Module):
    def __init__(self, vocab=49408, hidden=768, inter=3072, layers=12, heads=12, max_len=77):
        super().__init__()
        self.embeddings = _Embeddings(vocab, hidden, max_len)
        self.encoder = _Encoder(layers, hidden, inter, heads)
        self.final_layer_norm = Norm(hidden, LN_EPS)
        self.hidden, self.max_len = hidden, max_len


class CLIPTextModel(nn.Module):
    """`transformer` of FrozenCLIPEmbedder: keys `text_model.*`; `forward(input_ids)` -> last_hidden_state fp32."""

    def __init__(self, **cfg):
        super().__init__()
        self.text_model = CLIPTextTransformer(**cfg)
        self._packed = False
        self._tok = self._pos = None

    def pack(self, device=None):
        device = torch.device("cuda") if device is None else torch.device(device)
        pack_tree(self, device)
        e = self.text_model.embeddings
        self._tok = e.token_embedding.weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self._pos = e.position_embedding.weight.detach().to(device=device, dtype=torch.float32).contiguous()
        self._packed = True
        return self

    @torch.no_grad()
    def forward(self, input_ids: torch.Tensor) -> torch.Tensor:
        if not self._packed:
            raise RuntimeError("call .pack() after loading weights")
        tm = self.text_model
        b, l = input_ids.shape
        if l > tm.max_len:
            raise ValueError(f"sequence length {l} > {tm.max_len}")
        ids = input_ids.to(device=self._tok.device, dtype=torch.int64).contiguous()
        x = ops.embedding_lookup(ids, self._tok, self._pos[:l])
        for layer in tm.encoder.layers:
            x = layer.run(x, b, l)
        x = ops.layernorm(x, tm.final_layer_norm.g, tm.final_layer_norm.b, LN_EPS)
        return x.float().view(b, l, tm.hidden)
