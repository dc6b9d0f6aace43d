"""PatchEmbed (reference DA2/dinov2_layers/patch_embed.py:26-89): (B,C,H,W) -> (B,N,D) through a
patch x patch / stride-patch convolution.  Here: a patchify (im2col) kernel + one MFMA GEMM."""
from torch import nn


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten_embedding=True):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.patches_resolution = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans, self.embed_dim, self.flatten_embedding = in_chans, embed_dim, flatten_embedding
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)  # parameter container
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        from hip_ext import functional as HF
        _, _, H, W = x.shape
        pH, pW = self.patch_size
        assert H % pH == 0, f"Input image height {H} is not a multiple of patch height {pH}"
        assert W % pW == 0, f"Input image width {W} is not a multiple of patch width: {pW}"
        y = self.norm(HF.patch_embed(x, self.proj.weight, self.proj.bias))
        if not self.flatten_embedding:
            y = y.reshape(-1, H // pH, W // pW, self.embed_dim)
        return y
