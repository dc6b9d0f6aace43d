"""DINOv2 vision transformer with the optional guidance patch-embed of Amodal-Depth-Anything
(reference DA2/dinov2.py:44-448; the unguided twin is RAW/dinov2.py -- here one class serves both,
``guide_type=None`` giving the raw model's parameter set).

The module tree exists for the parameter schema (state_dict keys identical to the reference) and for
standalone use of the pieces; the whole-model fast path is ``hip_ext.engine.DepthEngine``.
"""
import math
from functools import partial
from typing import Sequence, Union

import torch
import torch.nn as nn
from torch.nn.init import trunc_normal_

from .dinov2_layers import MemEffAttention, Mlp, PatchEmbed, SwiGLUFFNFused
from .dinov2_layers import NestedTensorBlock as Block

# guidance channels per guide_type (reference dinov2.py:109-125)
GUIDE_IN_CHANS = {"image+mask+observation": 5, "image+mask": 4, "image+observation": 4, "mask+observation": 2,
                  "mask": 1, "observation": 1}


class DinoVisionTransformer(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, ffn_bias=True, proj_bias=True, drop_path_rate=0.0, drop_path_uniform=False, init_values=None,
                 embed_layer=PatchEmbed, act_layer=nn.GELU, block_fn=Block, ffn_layer="mlp", block_chunks=1,
                 num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1, guide_type=None):
        super().__init__()
        if block_chunks > 0 or num_register_tokens > 0 or drop_path_rate > 0.0:
            raise NotImplementedError("chunked blocks / register tokens / drop-path are unused by Depth-Anything-V2 inference")
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens, self.n_blocks, self.num_heads, self.patch_size = 1, depth, num_heads, patch_size
        self.num_register_tokens, self.register_tokens = 0, None
        self.interpolate_antialias, self.interpolate_offset = interpolate_antialias, interpolate_offset

        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        self.guide_type = guide_type
        if guide_type is not None and guide_type != "none":  # None = the raw (unguided) model
            if guide_type not in GUIDE_IN_CHANS:
                raise NotImplementedError
            self.patch_embed_guidance = embed_layer(img_size=img_size, patch_size=patch_size,
                                                    in_chans=GUIDE_IN_CHANS[guide_type], embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + self.num_tokens, embed_dim))

        if ffn_layer == "mlp":
            ffn = Mlp
        elif ffn_layer in ("swiglufused", "swiglu"):
            ffn = SwiGLUFFNFused
        else:
            raise NotImplementedError
        self.ffn_kind = "mlp" if ffn is Mlp else "swiglu"
        self.blocks = nn.ModuleList([
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, proj_bias=proj_bias,
                     ffn_bias=ffn_bias, drop_path=0.0, norm_layer=norm_layer, act_layer=act_layer, ffn_layer=ffn,
                     init_values=init_values)
            for _ in range(depth)])
        self.chunked_blocks = False
        self.norm = norm_layer(embed_dim)
        self.head = nn.Identity()
        self.mask_token = nn.Parameter(torch.zeros(1, embed_dim))  # unused at inference; kept for strict loading
        self.init_weights()

    @property
    def has_guidance(self):
        return hasattr(self, "patch_embed_guidance")

    def init_weights(self):
        trunc_normal_(self.pos_embed, std=0.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():  # timm ViT init (reference dinov2.py:359-364)
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def interpolate_pos_encoding(self, x, w, h):
        """Position table for an input of w x h pixels (reference dinov2.py:199-230): identity on the native
        square grid, bicubic resample with the +0.1 offset otherwise."""
        npatch, n = x.shape[1] - 1, self.pos_embed.shape[1] - 1
        if npatch == n and w == h:
            return self.pos_embed
        pos = self.pos_embed.float()
        w0, h0 = w // self.patch_size + self.interpolate_offset, h // self.patch_size + self.interpolate_offset
        sq = math.sqrt(n)
        if self.interpolate_antialias:
            raise NotImplementedError("antialiased position-table resampling is not used by any shipped configuration")
        import hip_ext
        out = torch.empty(1 + int(w0) * int(h0), pos.shape[-1], dtype=torch.float32, device=pos.device)
        hip_ext.pos_embed_resize(pos[0].contiguous(), int(sq), pos.shape[-1], int(w0), int(h0), float(w0) / sq, float(h0) / sq, out)
        return out[None].to(x.dtype)

    def prepare_tokens_with_masks(self, x, masks=None, guidance_mask=None):
        if masks is not None:
            raise NotImplementedError("iBOT token masking is a training feature")
        _, _, w, h = x.shape
        t = self.patch_embed(x)
        if self.has_guidance:
            t = t + self.patch_embed_guidance(guidance_mask)
        t = torch.cat((self.cls_token.expand(t.shape[0], -1, -1), t), dim=1)
        return t + self.interpolate_pos_encoding(t, w, h)

    def get_intermediate_layers(self, x, n: Union[int, Sequence] = 1, reshape=False, return_class_token=False, norm=True,
                                guidance_mask=None):
        """Module-by-module path (reference dinov2.py:298-349).  DepthAnythingV2.forward does not come through here --
        it runs the fused engine -- but the results agree and tests compare the two."""
        from hip_ext import functional as HF
        t = self.prepare_tokens_with_masks(x, guidance_mask=guidance_mask)
        take = range(len(self.blocks) - n, len(self.blocks)) if isinstance(n, int) else n
        outs = []
        for i, blk in enumerate(self.blocks):
            t = blk(t)
            if i in take:
                outs.append(t)
        assert len(outs) == len(take), f"only {len(outs)} / {len(take)} blocks found"
        if norm:
            outs = [HF.layer_norm(o, self.norm.weight, self.norm.bias, self.norm.eps) for o in outs]
        cls = [o[:, 0] for o in outs]
        outs = [o[:, 1:] for o in outs]
        if reshape:
            B, _, w, h = x.shape
            outs = [o.reshape(B, w // self.patch_size, h // self.patch_size, -1).permute(0, 3, 1, 2).contiguous() for o in outs]
        return tuple(zip(outs, cls)) if return_class_token else tuple(outs)

    def forward_features(self, x, masks=None):
        from hip_ext import functional as HF
        t = self.prepare_tokens_with_masks(x, masks)
        for blk in self.blocks:
            t = blk(t)
        tn = HF.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps)
        return {"x_norm_clstoken": tn[:, 0], "x_norm_regtokens": tn[:, 1:1], "x_norm_patchtokens": tn[:, 1:],
                "x_prenorm": t, "masks": masks}

    def forward(self, *args, is_training=False, **kwargs):
        ret = self.forward_features(*args, **kwargs)
        return ret if is_training else self.head(ret["x_norm_clstoken"])


def _vit(embed_dim, depth, num_heads, patch_size=16, num_register_tokens=0, guide_type=None, **kwargs):
    return DinoVisionTransformer(patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads, mlp_ratio=4,
                                 block_fn=partial(Block, attn_class=MemEffAttention), num_register_tokens=num_register_tokens,
                                 guide_type=guide_type, **kwargs)


vit_small = partial(_vit, 384, 12, 6)
vit_base = partial(_vit, 768, 12, 12)
vit_large = partial(_vit, 1024, 24, 16)
vit_giant2 = partial(_vit, 1536, 40, 24)  # embed-dim per head stays 64


def DINOv2(model_name, guide_type=None):
    zoo = {"vits": vit_small, "vitb": vit_base, "vitl": vit_large, "vitg": vit_giant2}
    return zoo[model_name](img_size=518, patch_size=14, init_values=1.0,
                           ffn_layer="mlp" if model_name != "vitg" else "swiglufused", block_chunks=0,
                           num_register_tokens=0, interpolate_antialias=False, interpolate_offset=0.1, guide_type=guide_type)
