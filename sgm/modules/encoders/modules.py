"""Conditioner (reference: sgm/modules/encoders/modules.py:84-204, 982-1023).

`GeneralConditioner` keeps the reference's routing exactly — embedders instantiated from their `target`
strings, `input_key` -> output key rules (`cond_img` -> `cond_feat`, `control_hint` -> `control_hint`, otherwise by
rank: 2 `vector`, 3 `crossattn`, 4/5 `concat`), concatenation of repeated keys, `force_zero_embeddings`, and
`get_unconditional_conditioning`.

Embedders:
  * `VAEEmbedder` (TVI2V `cond_img` -> `cond_feat`) is real: it runs the engine's first-stage encoder on the HIP
    kernels.
  * `FrozenCLIPEmbedder` (`txt` -> `crossattn`) is real: the CLIP ViT-L/14 text transformer runs on the HIP kernels
    (ccedit_amd/clip.py) with the checkpoint's parameter names.  It takes token ids (B,77) int64,
    raw strings when the HF tokenizer files are available locally, or an already computed (B,77,768) embedding.
  * `DepthMidasEncoder`, `DepthZoeEncoder` need an un-vendored third-party repo and weights that do not exist
    offline; they run once per clip outside the denoising loop.  Here they are *pass-through*
    embedders: they accept the already computed depth hint (B,3,T,H,W) in [-1,1] and raise with a clear message when
    handed raw video.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import os

import torch
import torch.nn as nn

from ccedit_amd.config import instantiate_from_config


class AbstractEmbModel(nn.Module):
    """encoders/modules.py:44-81: carries is_trainable / ucg_rate / input_key set by the conditioner."""

    def __init__(self):
        super().__init__()
        self.is_trainable = False
        self.ucg_rate = 0.0
        self.input_key: Optional[str] = None
        self.legacy_ucg_val = None


class _Precomputed(AbstractEmbModel):
    """An embedder whose network is outside this build: passes a precomputed tensor through."""
    what = "embedding"
    rank = 3

    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        if torch.is_tensor(x) and x.dim() == self.rank:
            return x
        raise NotImplementedError(
            f"{self.__class__.__name__}: its network/weights are not part of this build (they are not available "
            f"offline and run once per clip outside the denoising loop); pass the precomputed {self.what} tensor "
            f"as batch[{self.input_key!r}]")

    def encode(self, x):
        return self(x)


class FrozenCLIPEmbedder(AbstractEmbModel):
    """encoders/modules.py:358-420: HF CLIP text encoder, layer="last" -> last_hidden_state (B, 77, 768).
    `forward` accepts what the reference accepts (a list of strings; needs the tokenizer files of `version` on local
    disk — there is no network here) and, in addition, token ids (B, L<=77) int64 or a precomputed embedding."""

    LAYERS = ["last", "pooled", "hidden"]

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77, freeze=True, layer="last",
                 layer_idx=None, always_return_pooled=False, tokenizer_path=None):
        super().__init__()
        assert layer in self.LAYERS
        if layer != "last" or always_return_pooled:
            raise NotImplementedError("FrozenCLIPEmbedder: the CCEdit configs use layer='last' without the pooled output")
        from ccedit_amd.clip import CLIPTextModel
        self.version, self.max_length, self.layer = version, max_length, layer
        # where the tokenizer files live: an explicit directory (ctor / CCEDIT_CLIP_TOKENIZER / the scripts' --tokenizer_path),
        # else `version` as the reference passes it to CLIPTokenizer.from_pretrained (a hub id needs a populated local cache)
        self.tokenizer_path = tokenizer_path or os.environ.get("CCEDIT_CLIP_TOKENIZER") or version
        self.transformer = CLIPTextModel()
        self._tokenizer = None

    _expected_vocab = (49408, 49406, 49407)      # (entries, <|startoftext|>, <|endoftext|>) of openai/clip-vit-large-patch14

    def freeze(self):
        self.transformer = self.transformer.eval()
        for p in self.parameters():
            p.requires_grad = False

    def pack(self, device=None):
        self.transformer.pack(device)
        return self

    def tokenize(self, text) -> torch.Tensor:
        if self._tokenizer is None:
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.tokenizer_path, local_files_only=True)
                # The ids index the 49408-row CLIP ViT-L/14 embedding table: only THAT vocabulary is accepted (recent transformers
                # build an empty tokenizer when the files are missing; a different or truncated BPE table would condition on the
                # wrong rows or run past the table).  Tests swap the expectation through _expected_vocab.
                want = self._expected_vocab
                if len(tok) < 16:
                    raise FileNotFoundError(f"vocabulary of {self.tokenizer_path} not found locally ({len(tok)} entries)")
                if want is not None and (len(tok) != want[0] or tok.bos_token_id != want[1] or tok.eos_token_id != want[2]):
                    raise FileNotFoundError(f"{self.tokenizer_path}: {len(tok)} entries, bos {tok.bos_token_id}, eos {tok.eos_token_id} — "
                                            f"not the CLIP ViT-L/14 vocabulary (expected {want})")
                self._tokenizer = tok
            except Exception as e:          # no vocabulary files offline
                raise NotImplementedError(
                    f"FrozenCLIPEmbedder: the tokenizer files of {self.tokenizer_path!r} are not on this machine "
                    f"({type(e).__name__}); point tokenizer_path / CCEDIT_CLIP_TOKENIZER / --tokenizer_path at them, or pass token ids (B,{self.max_length}) int64 or a precomputed (B,{self.max_length},768) embedding as batch['txt']")
        enc = self._tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                              return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    def forward(self, text):
        if torch.is_tensor(text) and text.is_floating_point():
            if text.dim() != 3:
                raise ValueError(f"precomputed text embedding must be (B, L, C), got {tuple(text.shape)}")
            return text
        tokens = text if torch.is_tensor(text) else self.tokenize(text)
        return self.transformer(tokens)

    def encode(self, text):
        return self(text)


class DepthMidasEncoder(_Precomputed):
    """encoders/modules.py:1346-1392: MiDaS depth of every keyframe, min-max normalised to [-1,1], 3 channels.
    The MiDaS network itself (un-vendored ControlNet-v1-1 annotator) is outside this build; the encoder takes either
    the finished hint (B,3,T,H,W) or the network's RAW depth (B,1,T,H,W) and applies the reference's normalisation."""
    what = "depth hint (B,3,T,H,W) in [-1,1], or raw MiDaS depth (B,1,T,H,W)"
    rank = 5

    @staticmethod
    def normalize(depth: torch.Tensor) -> torch.Tensor:
        """:1376-1386 — global min / max over the whole batch of frames, sign flipped (near = bright), 3 channels."""
        d = depth.float().clone()
        d -= torch.min(d)
        d /= torch.max(d)
        d = -(torch.clamp(d, 0, 1) * 2 - 1)
        return d.repeat(1, 3, 1, 1, 1).to(depth.dtype)

    def forward(self, x):
        if torch.is_tensor(x) and x.dim() == 5 and x.shape[1] == 1:
            return self.normalize(x)
        if torch.is_tensor(x) and x.dim() == 5 and x.shape[1] == 3:
            # The reference feeds RGB keyframes here and the depth network turns them into a hint whose three channels
            # are copies of one map (:1386 `repeat(1, 3, ...)`).  A finished hint therefore has identical channels; RGB
            # frames do not — refuse them instead of silently feeding colour to the ControlNet as "depth".
            if not (torch.equal(x[:, 0], x[:, 1]) and torch.equal(x[:, 1], x[:, 2])):
                raise NotImplementedError(
                    f"{self.__class__.__name__}: got a 3-channel tensor whose channels differ — these look like RGB "
                    f"keyframes.  The depth network is not part of this build (un-vendored annotator + weights); compute "
                    f"the depth outside and pass either the raw depth (B,1,T,H,W) or the finished hint (B,3,T,H,W: one "
                    f"map in [-1,1] replicated over the channels) as batch[{self.input_key!r}]")
            return x
        return super().forward(x)


class DepthZoeEncoder(DepthMidasEncoder):
    """encoders/modules.py:1289-1342 (ZoeDepth): per-clip 2nd / 85th percentile normalisation via kthvalue."""
    what = "depth hint (B,3,T,H,W) in [-1,1], or raw ZoeDepth output (B,1,T,H,W)"

    @staticmethod
    def normalize(depth: torch.Tensor) -> torch.Tensor:
        """:1324-1336 — vmin / vmax = kthvalue at int(0.02 n) / int(0.85 n) over each clip's C*T*H*W values."""
        d = depth.float().clone()
        flat = d.view(d.shape[0], -1)
        vmin = torch.kthvalue(flat, int(0.02 * d[0].numel()), dim=1).values
        vmax = torch.kthvalue(flat, int(0.85 * d[0].numel()), dim=1).values
        d -= vmin[:, None, None, None, None]
        d /= (vmax - vmin)[:, None, None, None, None]
        d = torch.clamp(d, 0, 1) * 2 - 1
        return d.repeat(1, 3, 1, 1, 1).to(depth.dtype)


class VAEEmbedder(AbstractEmbModel):
    """encoders/modules.py:982-1023: cond_feat = scale_factor * first_stage_model.encode(cond_img).  The engine
    attaches first_stage_model / scale_factor (diffusion.py:375-385, `setup_vaeembedder`)."""

    def __init__(self, down_blur_factor=1, *args, **kwargs):
        super().__init__()
        assert down_blur_factor >= 1, "down_blur_factor must be >= 1"
        if down_blur_factor != 1:
            raise NotImplementedError("VAEEmbedder.down_blur_factor > 1 (bilinear blur) is not used by the shipped configs")
        self.down_blur_factor = down_blur_factor

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        assert "first_stage_model" in self.__dict__, "first_stage_model not defined"
        assert hasattr(self, "scale_factor"), "scale_factor not defined"
        from ccedit_amd import ops
        z = self.__dict__["first_stage_model"].encode(x)
        return ops.axpby(z, z, float(self.scale_factor), 0.0)

    def encode(self, x):
        return self(x)


class GeneralConditioner(nn.Module):
    OUTPUT_DIM2KEYS = {2: "vector", 3: "crossattn", 4: "concat", 5: "concat"}
    KEY2CATDIM = {"vector": 1, "crossattn": 2, "concat": 1}
    _KEYED = {"cond_img": "cond_feat", "interpolate_first": "interpolate_first", "interpolate_last": "interpolate_last",
              "interpolate_first_last": "interpolate_first_last", "control_hint": "control_hint"}

    def __init__(self, emb_models: Optional[List[dict]] = None):
        super().__init__()
        embedders = []
        for embconfig in (emb_models or []):
            embedder = instantiate_from_config(embconfig)
            assert isinstance(embedder, AbstractEmbModel), \
                f"embedder model {embedder.__class__.__name__} has to inherit from AbstractEmbModel"
            embedder.is_trainable = embconfig.get("is_trainable", False)
            embedder.ucg_rate = embconfig.get("ucg_rate", 0.0)
            if "input_key" in embconfig:
                embedder.input_key = embconfig["input_key"]
            elif "input_keys" in embconfig:
                embedder.input_keys = embconfig["input_keys"]
            else:
                raise KeyError(f"need either 'input_key' or 'input_keys' for embedder {embedder.__class__.__name__}")
            # encoders/modules.py:117-131: stored like the reference.  It only acts together with ucg_rate > 0
            # (training-time conditioning dropout, :135-143); get_unconditional_conditioning zeroes ucg_rate, so at
            # inference it is inert — the shipped yamls set `legacy_ucg_value: ""` on the CLIP embedder.
            embedder.legacy_ucg_val = embconfig.get("legacy_ucg_value", None)
            embedders.append(embedder.eval())
        self.embedders = nn.ModuleList(embedders)

    @torch.no_grad()
    def forward(self, batch: Dict, force_zero_embeddings: Optional[List] = None) -> Dict[str, torch.Tensor]:
        output: Dict[str, torch.Tensor] = {}
        force_zero_embeddings = force_zero_embeddings or []
        for embedder in self.embedders:
            if getattr(embedder, "input_key", None) is not None:
                emb_out = embedder(batch[embedder.input_key])
            else:
                emb_out = embedder(*[batch[k] for k in embedder.input_keys])
            assert isinstance(emb_out, (torch.Tensor, list, tuple)), \
                f"encoder outputs must be tensors or a sequence, but got {type(emb_out)}"
            if not isinstance(emb_out, (list, tuple)):
                emb_out = [emb_out]
            for emb in emb_out:
                key = getattr(embedder, "input_key", None)
                out_key = self._KEYED.get(key) or self.OUTPUT_DIM2KEYS[emb.dim()]
                if embedder.ucg_rate > 0.0:
                    raise NotImplementedError("ucg_rate > 0 is training-time conditioning dropout")
                if key in force_zero_embeddings:
                    emb = torch.zeros_like(emb)
                if out_key in output:
                    output[out_key] = torch.cat((output[out_key], emb), self.KEY2CATDIM[out_key])
                else:
                    output[out_key] = emb
        return output

    def get_unconditional_conditioning(self, batch_c, batch_uc=None, force_uc_zero_embeddings=None) -> Tuple[Dict, Dict]:
        force_uc_zero_embeddings = force_uc_zero_embeddings or []
        rates = [e.ucg_rate for e in self.embedders]
        for e in self.embedders:
            e.ucg_rate = 0.0
        c = self(batch_c)
        uc = self(batch_c if batch_uc is None else batch_uc, force_uc_zero_embeddings)
        for e, r in zip(self.embedders, rates):
            e.ucg_rate = r
        return c, uc
