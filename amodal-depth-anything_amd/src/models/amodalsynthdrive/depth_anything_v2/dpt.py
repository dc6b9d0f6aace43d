"""Guided Depth-Anything-V2: DINOv2 encoder + DPT head with input_projection and Sigmoid tail
(reference DA2/dpt.py).  ``DepthAnythingV2.forward(x, guidance_mask) -> [B,1,H,W]``.

``forward`` hands the whole pass to ``hip_ext.engine.DepthEngine`` (one fixed launch sequence of HIP
kernels); the DPTHead / LayerNorm modules below also work standalone through ``hip_ext.functional``.
"""
import torch
import torch.nn as nn

from .dinov2 import DINOv2
from .util.blocks import FeatureFusionBlock, _make_scratch

INTERMEDIATE_LAYER_IDX = {"vits": [2, 5, 8, 11], "vitb": [2, 5, 8, 11], "vitl": [4, 11, 17, 23], "vitg": [9, 19, 29, 39]}


def _make_fusion_block(features, use_bn, size=None):
    return FeatureFusionBlock(features, nn.ReLU(False), deconv=False, bn=use_bn, expand=False, align_corners=True, size=size)


class LayerNorm(nn.Module):
    """Channels-first / channels-last LayerNorm (reference DA2/dpt.py:37-61); biased variance, eps in the sqrt."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_first"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps, self.data_format, self.normalized_shape = eps, data_format, (normalized_shape,)

    def forward(self, x):
        from hip_ext import functional as HF
        if self.data_format == "channels_last":
            return HF.layer_norm(x, self.weight, self.bias, self.eps)
        y = HF.layer_norm(x.permute(0, 2, 3, 1).contiguous(), self.weight, self.bias, self.eps)
        return y.permute(0, 3, 1, 2).contiguous()


class DPTHead(nn.Module):
    """final_act: 'sigmoid' (amodal default), 'none' ('ssi' losses), 'relu' (raw Depth-Anything-V2)."""

    def __init__(self, in_channels, features=256, use_bn=False, out_channels=(256, 512, 1024, 1024), use_clstoken=False,
                 loss_stategy=None, with_input_projection=True):
        super().__init__()
        oc = list(out_channels)
        self.use_clstoken = use_clstoken
        self.projects = nn.ModuleList([nn.Conv2d(in_channels, c, kernel_size=1) for c in oc])
        if use_clstoken:   # read-out of the class token into every patch token (reference DA2/dpt.py:110-117, RAW/dpt.py:83-90)
            self.readout_projects = nn.ModuleList(
                [nn.Sequential(nn.Linear(2 * in_channels, in_channels), nn.GELU()) for _ in oc])
        self.resize_layers = nn.ModuleList([
            nn.ConvTranspose2d(oc[0], oc[0], kernel_size=4, stride=4, padding=0),
            nn.ConvTranspose2d(oc[1], oc[1], kernel_size=2, stride=2, padding=0),
            nn.Identity(),
            nn.Conv2d(oc[3], oc[3], kernel_size=3, stride=2, padding=1)])
        self.scratch = _make_scratch(oc, features, groups=1, expand=False)
        self.scratch.stem_transpose = None
        for k in (1, 2, 3, 4):
            setattr(self.scratch, f"refinenet{k}", _make_fusion_block(features, use_bn))
        self.scratch.output_conv1 = nn.Conv2d(features, features // 2, kernel_size=3, stride=1, padding=1)
        tail = [nn.Conv2d(features // 2, 32, kernel_size=3, stride=1, padding=1), nn.ReLU(True),
                nn.Conv2d(32, 1, kernel_size=1, stride=1, padding=0)]
        if not with_input_projection:          # raw model: ReLU, Identity (RAW/dpt.py:109-115)
            tail += [nn.ReLU(True), nn.Identity()]
            self.final_act = "relu"
        elif "ssi" in (loss_stategy or ""):    # DA2/dpt.py:138-144
            self.final_act = "none"
        else:                                   # DA2/dpt.py:145-151
            tail += [nn.Sigmoid()]
            self.final_act = "sigmoid"
        self.scratch.output_conv2 = nn.Sequential(*tail)
        if with_input_projection:               # DA2/dpt.py:153-159
            self.input_projection = nn.ModuleList(
                [nn.Sequential(nn.Conv2d(c, c, kernel_size=3, padding=1), LayerNorm([c]), nn.ReLU()) for c in oc])

    def forward(self, out_features, patch_h, patch_w):
        """Module-level path (reference DA2/dpt.py:161-197) on NCHW fp32 tensors."""
        from hip_ext import functional as HF
        layers = []
        for i, feat in enumerate(out_features):
            x = feat[0]
            if self.use_clstoken:   # DA2/dpt.py:164-167
                lin = self.readout_projects[i][0]
                x = HF.linear(torch.cat((x, feat[1].unsqueeze(1).expand_as(x)), -1), lin.weight, lin.bias, gelu=True)
            x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], patch_h, patch_w)
            x = HF.conv2d(x, self.projects[i].weight, self.projects[i].bias)
            r = self.resize_layers[i]
            if isinstance(r, nn.ConvTranspose2d):
                x = HF.conv_transpose2d(x, r.weight, r.bias, r.stride[0])
            elif isinstance(r, nn.Conv2d):
                x = HF.conv2d(x, r.weight, r.bias, stride=2, padding=1)
            layers.append(x)
        if hasattr(self, "input_projection"):
            for i in range(4):
                conv, ln, _ = self.input_projection[i]
                layers[i] = torch.relu(ln(HF.conv2d(layers[i], conv.weight, conv.bias, padding=1)))
        s = self.scratch
        rn = [HF.conv2d(layers[i], getattr(s, f"layer{i + 1}_rn").weight, None, padding=1) for i in range(4)]
        path = s.refinenet4(rn[3], size=rn[2].shape[2:])
        path = s.refinenet3(path, rn[2], size=rn[1].shape[2:])
        path = s.refinenet2(path, rn[1], size=rn[0].shape[2:])
        path = s.refinenet1(path, rn[0])
        out = HF.conv2d(path, s.output_conv1.weight, s.output_conv1.bias, padding=1)
        out = HF.interpolate_bilinear_ac(out, (int(patch_h * 14), int(patch_w * 14)))
        c0, c2 = s.output_conv2[0], s.output_conv2[2]
        return HF.conv_tail(out, c0.weight, c0.bias, c2.weight, c2.bias, self.final_act)


import os as _os

# Head precision policy for "auto" (measured on MI355X against the reference goldens, profiles/r03_e_head_split_sweep.txt):
#   sigmoid heads of ViT-B / ViT-L (the benchmarked models)  -> single precision but the 1x1 out_convs of refinenet2-4 (round 4) and, on ViT-L, the
#                                                               projects' weights ("projw", round 4)
#   ViT-S (64-feature head) and every 'ssi' head             -> every head contraction in split precision (1.8e-3 -> 6.9e-4; cheap models)
#   raw (ReLU) ViT-G, features 384                           -> only the contractions whose operand rounding shows in the output and that are
#                                                               cheap: the tail conv, the 1x1 out_convs and projects, the three coarse
#                                                               layer_rn convs, resize_layers 1 and 3.  8 x 1022^2: 9.1e-4 at 37.4 images/s
#                                                               (everything split: 8.7e-4 at 28.4; nothing: 1.3e-3 at 41.3).  Splitting the
#                                                               ResidualConvUnit convs makes ViT-G parity WORSE (1.13e-3 -> 1.31e-3).
# round 4: + "oc1" (with the encoder's early blocks in split precision -- _UNBOUNDED_ENC_SPLIT_BLOCKS -- the head's share shows again:
# profiles/r04_e_raw_vitg_precision.txt)
_SIGMOID_SPLIT = ("out1", "out2", "out3")
# Round 5: the first rung's four 1x1 projects run as a FULL split product with fp8 correction terms (hip_ext.engine F8_CORR), read from the taps the
# ladder keeps [hi | lo8 | hi8] anyway -- instead of ViT-L's weight-only split (same time: 2x the projects' MACs) / nothing on ViT-B (+0.9 % of a bs=8
# step).  30 sigmoid ViT-B / ViT-L reference fixtures: rel-L1 -2.5 % in the geometric mean, worst 8.93e-4 -> 8.57e-4, the benchmarked batch 5.97e-4 -> 5.74e-4,
# headline unchanged (52.02 / 52.11 / 52.10 / 52.02 ms alternating): profiles/r05_t_first_rung_projects_fp8.txt.  ADA_RUNG1_PROJ_F8=0: the round-4 policy.
_RUNG1_PROJ_F8 = _os.environ.get("ADA_RUNG1_PROJ_F8", "1") == "1"


def _rung1_proj_f8():
    """... where the build offers the fp8 terms at all (fp16 operands, not masked by ADA_F8_CORR); otherwise the round-4 policy stands"""
    if not _RUNG1_PROJ_F8:
        return False
    import torch as _torch
    from hip_ext import engine as _E, operand_dtype as _opdt
    return _E.F8_HEAD and _opdt() == _torch.float16
_RAW_VITG_SPLIT = ("oc1", "oc2", "out", "rn1", "rn2", "rn3", "proj", "rs1", "rs3")


def _head_split_policy(mode, encoder, final_act, proj_f8=None):
    """Which contractions of the DPT head (hip_ext.engine.HEAD_GROUPS / HEAD_ALIASES) run in split precision.  ``mode``: "auto" |
    "split" (all) | "single" (none) | a comma-separated string / iterable of names.  ADA_HEAD_SPLIT overrides "auto" (experiments).
    ``proj_f8``: whether the sigmoid ViT-B / ViT-L first rung runs its projects as a full split product with fp8 correction terms (None: where
    the build offers them, _rung1_proj_f8) -- False when the caller's ``f8_terms`` keeps the head off the fp8 pipe: the round-4 policy then."""
    from hip_ext.engine import HEAD_GROUPS
    if mode == "auto" and _os.environ.get("ADA_HEAD_SPLIT") is not None:
        mode = _os.environ["ADA_HEAD_SPLIT"]
    if mode == "split":
        return frozenset(HEAD_GROUPS)
    if mode in ("single", "", "none"):
        return frozenset()
    if mode == "auto":
        if final_act == "sigmoid" and encoder != "vits":
            # round 4: the 1x1 out_convs of refinenet2-4 in split precision -- the largest single group of the head's operand noise in the
            # ViT-B / ViT-L oracle studies and all but free (K = features): worst fixture 9.25e-4 -> 6.9e-4, throughput unchanged
            # (profiles/r04_h_sigmoid_head_out_conv_split.txt).  refinenet1's out_conv is part of output_conv1's tap maps (engine OC1_COMMUTE)
            # ViT-L also runs its four 1x1 projects against [w_hi | w_lo] weights ("projw", weight-only split: 2x their 0.25 TFLOP per bs=32 step,
            # 0.4 % of it): the heavy-tailed 714 x 1022 stress fixture 8.9e-4 -> 6.9e-4, the benchmarked batch 6.17e-4 -> 5.94e-4; on ViT-B the
            # projects are not where the noise sits (9.0e-4 -> 8.8e-4 on its stress fixture, profiles/r04_r_*), so it keeps the three out_convs only
            if _rung1_proj_f8() if proj_f8 is None else proj_f8:
                return frozenset(_SIGMOID_SPLIT + ("proj",))
            return frozenset(_SIGMOID_SPLIT + (("projw",) if encoder == "vitl" else ()))
        if final_act == "relu" and encoder == "vitg":
            return frozenset(_RAW_VITG_SPLIT)
        return frozenset(HEAD_GROUPS)
    if isinstance(mode, str):
        mode = [g for g in mode.split(",") if g]
    return frozenset(mode)


# Round 4: the encoder's own operand noise.  For the unbounded-output ViT-G model the encoder alone reaches 0.7e-3 ... 1.05e-3 of the 1e-3 budget
# depending on the weight draw (tests/golden/raw_vitg_224_w1: 1.05e-3 with the WHOLE head in split precision), and half of that is injected by
# the first quarter of the blocks (later blocks amplify it): their linear layers run in split precision (PackedWeights.enc_split_blocks).
# (round 5: ViT-S every block -- a raw ViT-S draw on a plain noise input read 1.2e-3 with 4 of its 12 blocks split, 3.4e-4 with all of them,
#  profiles/r05_aa_*; the model is cheap)
_UNBOUNDED_ENC_SPLIT_BLOCKS = {"vits": 12, "vitb": 4, "vitl": 8, "vitg": 8}


def _encoder_split_policy(mode, encoder, final_act):
    """Number of leading transformer blocks whose linear layers run in split precision.  ``mode``: "auto" | int.  ADA_ENC_SPLIT overrides "auto"."""
    if mode == "auto" and _os.environ.get("ADA_ENC_SPLIT") is not None:
        mode = int(_os.environ["ADA_ENC_SPLIT"])
    if mode == "auto":
        # Unbounded heads -- the raw model's ReLU, the 'ssi' heads' logits -- have no sigmoid to compress what the encoder's fp16 operand rounding
        # leaves in the logits, and the leading blocks inject most of it (every later block amplifies it).  Reference fixtures with the head already
        # in split precision and every block in single precision: raw ViT-G 1.18e-3 (r04_e); 'ssi' on ViT-L 1.09e-3 (r04_q); heavy-tailed weights:
        # raw ViT-S / ViT-B 1.23e-3 / 1.23e-3 / 1.39e-3, 'ssi' on ViT-S 1.01e-3, raw ViT-L 0.97e-3 (r04_q).  With the first 4 (ViT-S / B) or 8
        # (ViT-L / G) blocks' linear layers in split precision every one of them is <= 6.1e-4 (ViT-G <= 7.8e-4).  The sigmoid models -- the
        # benchmarked ones -- keep every block in single precision.
        if final_act in ("relu", "none"):
            return _UNBOUNDED_ENC_SPLIT_BLOCKS[encoder]
        return 0
    return int(mode)


# Precision ladder of the sigmoid heads ("auto" policy only; hip_ext.engine.DepthEngine._escalate).  The single-precision head leaves a mean absolute
# LOGIT error of ~1.2e-3 (ViT-L) that the sigmoid compresses by r = sum s (1 - s) / sum s in the north-star metric mean|a - b| / mean|b|: r is 0.3-0.5
# for depth maps that span (0, 1) -- every centred reference fixture, the benchmarked batch -- and tends to 1 for maps concentrated near 0, where the
# default policy measured 1.17e-3 (ViT-L, mean 0.10) ... 2.3e-3 (ViT-B, all-zero image): profiles/r04_s_sigmoid_operating_point.txt.  Images whose
# first-rung output has r above the threshold get their HEAD re-run in split precision from the (always [hi | lo]) taps.  Threshold per encoder from the
# reference fixtures of round 5 (profiles/r05_h_precision_ladder.txt): first-rung rel-L1 / r is a property of the model -- ViT-B 1.6e-3 ... 2.3e-3 (image-like
# inputs at the top), ViT-L 1.5e-3 ... 2.1e-3 (heavy-tailed weights at the top) -- over centred, low-mean and structured inputs alike, so the rung
# holds ~9e-4 up to r = 0.42 / 0.45; the benchmarked batch has r <= 0.43, the centred noise fixtures r <= 0.41.  Second trigger: token diversity of the last
# tap below _LADDER_DIV (constant / checkerboard inputs: 0.02; everything else >= 0.23), where rounding errors add coherently over positions.
# ADA_LADDER_R overrides the threshold ("0" / "off" disables the ladder).
# Where the two correction terms of a split-precision product run on the fp8 matrix pipe instead of the fp16 one (hip_ext.engine F8_CORR: 2x the
# MACs' time instead of 3x; residual operand noise ~2^-14 instead of ~2^-22).  Measured on the reference fixtures (profiles/r05_q_fp8_terms_ab.txt):
# +1.8 % (encoder blocks) / +0.7 % (head) on the fixtures' rel-L1 in the geometric mean.  Used where a split product is a large share of the step --
# the raw (ReLU) models: BASELINE config 5 is raw ViT-G, 33.9 -> 36.1 images/s at <= 7.7e-4 -- and on the ladder's second rung (_LADDER_F8); the
# 'ssi' heads (unbounded logits, the thinnest margins: vitl_ssi_518_heavy 8.6e-4) and the sigmoid models' first rung (whose split groups are 1x1
# convolutions with K <= 256: nothing to gain) keep all three terms on the fp16 pipe.
def _f8_policy(final_act):
    return "both" if final_act == "relu" else "none"


_LADDER_F8 = "head"
# Round 6: the thresholds of the ladder are CALIBRATED per checkpoint on the device (hip_ext.engine.DepthEngine.calibrate, _EngineMixin._calibrate_ladder):
# round 5 carried them as constants fitted to synthetic weights -- r 0.42 (ViT-B) / 0.45 (ViT-L), third rung 0.75, tap diversity 0.10 -- which a real
# checkpoint has no reason to share.  The error budget is what is stated instead: a rung is left where its sensitivity-normalised logit error times the
# image's r reaches _LADDER_BUDGET of the 1e-3 bar.  What is left below are the UNCALIBRATED fallbacks, used only where the calibration cannot run (the
# first forward arrives inside a caller's stream capture) or is switched off (ADA_LADDER_CALIBRATE=0): one conservative set for every encoder.
_LADDER_BUDGET = float(_os.environ.get("ADA_LADDER_BUDGET", "9e-4"))
_LADDER_SAFETY = float(_os.environ.get("ADA_LADDER_SAFETY", "1.1"))          # on the error the calibration images show
_LADDER_RULE = _os.environ.get("ADA_LADDER_RULE", "cross")                   # cross | global (hip_ext.engine.DepthEngine.calibrate)
_LADDER_FALLBACK = dict(r=0.40, r3=0.75, div=0.10)
_LADDER_CALIBRATE = _os.environ.get("ADA_LADDER_CALIBRATE", "1") != "0"
_LADDER_CAL_SIZE = tuple(int(v) for v in _os.environ.get("ADA_LADDER_CAL_SIZE", "266x322").split("x"))
_LADDER_DIV_IN = 1e-4      # variance of the patchified input over the image's patches / their mean square: 0 for constant images and pixel checkerboards, ~0.25-1 otherwise
# What the second rung re-runs in split precision: the whole head -- on ViT-L without the ResidualConvUnit convolutions of the two finest levels (the four
# 148^2 and four 74^2 convs: 23 % of the rung's MACs): on the five low-mean / constant ViT-L fixtures that subset is as good or better (5.8-6.5e-4 against
# 5.4-7.5e-4 with everything split, profiles/r05_h_ladder_second_rung_subsets.txt); ViT-B needs them (8.5e-4 against 9.9e-4 on its worst fixture).
_LADDER_SKIP = {"vitl": ("rcu0", "rcu1")}


def _ladder_threshold(module, encoder, final_act, mode):
    """r threshold of the second rung, or None when the ladder does not apply (non-sigmoid heads, ViT-S -- whose whole head is in split precision
    already -- or an explicit head precision: the caller has chosen)."""
    env = _os.environ.get("ADA_LADDER_R")
    val = getattr(module, "precision_ladder", None)
    if val is None and env is not None:
        val = env
    if isinstance(val, str):
        val = 0.0 if val.lower() in ("off", "0", "false", "") else float(val)
    if val is False:
        val = 0.0
    if final_act != "sigmoid" or encoder == "vits" or mode != "auto" or _os.environ.get("ADA_HEAD_SPLIT") is not None:
        return None
    if val is None or val is True:
        val = _LADDER_FALLBACK["r"]       # provisional: replaced by the calibrated threshold (_calibrate_ladder) unless the caller named one
    return float(val) if val else None


def _flat_input_rung(module, final_act, mode):
    """The models WITHOUT a second rung (raw ReLU, 'ssi', and -- round 6 -- sigmoid ViT-S, whose whole head is in split precision already) keep one rung of the ladder: an image whose patch tokens are all alike (constant / checkerboard input: every
    patch of the patchified input the same: _LADDER_DIV_IN) is run again with every encoder block and the whole head in split precision -- its rounding errors add
    coherently over positions and nothing compresses them: raw ViT-B on an all-zero 518 x 518 image 1.15e-3 -> 4.7e-4, 'ssi' ViT-B 1.13e-3 -> 4.4e-4, raw ViT-L on
    a checkerboard 1.14e-3 -> 3.9e-4 (profiles/r05_aa_*).  Off with precision_ladder = False / ADA_LADDER_R = off, or when the caller chose a precision."""
    if mode != "auto" or _os.environ.get("ADA_HEAD_SPLIT") is not None or getattr(module, "encoder_precision", "auto") != "auto":
        return False
    val = getattr(module, "precision_ladder", None)
    if val is None:
        val = _os.environ.get("ADA_LADDER_R")
    if isinstance(val, str):
        return val.lower() not in ("off", "0", "false", "")
    return val is not False and val != 0.0


class _EngineMixin:
    """Lazily builds / refreshes the packed weights + launch plan whenever a parameter changes."""

    def _param_tensors(self, rescan=False):
        """The parameter / buffer tensors whose (storage address, version counter) pairs stamp the packed weights.  Walking ``state_dict()`` on every
        call cost 0.7-1.0 ms for ViT-L's 425 keys -- 18 % of the single-image latency; the list is cached together with the SLOT every tensor sits in
        (the owning sub-module's ``_parameters`` / ``_buffers`` dict and its key) and each call checks that every slot still holds the very object that
        was stamped: a Parameter replaced by attribute assignment on a sub-module, ``load_state_dict(..., assign=True)`` on a child, parametrize / pruning
        utilities all show up at once (round 5 re-walked the tree only every 32 calls: up to 31 forwards on the old weights).  ~20 us for ViT-L."""
        lst = self.__dict__.get("_engine_plist")
        if lst is not None and not rescan:
            for (d, k), t in zip(self.__dict__["_engine_pslots"], lst):
                if d.get(k) is not t:
                    lst = None
                    break
        if lst is None or rescan:
            slots, names = [], []
            for mname, mod in self.named_modules():      # the traversal order of state_dict(): a module's parameters, its persistent buffers, then its children
                for d, skip in ((mod._parameters, ()), (mod._buffers, mod._non_persistent_buffers_set)):
                    for k, v in d.items():
                        if v is not None and k not in skip:
                            slots.append((d, k))
                            names.append(f"{mname}.{k}" if mname else k)
            if names != list(self.state_dict(keep_vars=True).keys()):     # a module with a custom state_dict: fall back to stamping its state_dict (no slots)
                items = list(self.state_dict(keep_vars=True).items())
                names, slots = [k for k, _ in items], [({i: v}, i) for i, (_, v) in enumerate(items)]
            lst = [d[k] for d, k in slots]
            object.__setattr__(self, "_engine_pnames", names)
            object.__setattr__(self, "_engine_pslots", slots)
            object.__setattr__(self, "_engine_plist", lst)
        return lst

    def invalidate_engine(self):
        """Forget the cached parameter list (and with it the stamp): the next forward re-walks the module tree and re-packs if anything changed."""
        object.__setattr__(self, "_engine_plist", None)

    def _apply(self, fn, *a, **kw):     # .cuda() / .to() / .float(): tensors may be replaced
        out = super()._apply(fn, *a, **kw)
        self.invalidate_engine()
        return out

    def _engine(self):
        from hip_ext.engine import HEAD_GROUPS, DepthEngine, PackedWeights
        plist = self._param_tensors()
        hp = getattr(self, "head_precision", "auto")
        ladder_r = _ladder_threshold(self, self.encoder, self.depth_head.final_act, hp)
        f8 = getattr(self, "f8_terms", None) or _f8_policy(self.depth_head.final_act)     # module.f8_terms: "both" | "enc" | "head" | "none" overrides the policy
        stamp = tuple((v.data_ptr(), v._version) for v in plist) + (hp if isinstance(hp, str) else tuple(sorted(hp)),
                                                                      getattr(self, "encoder_precision", "auto"), ladder_r, f8,
                                                                      _flat_input_rung(self, self.depth_head.final_act, hp if isinstance(hp, str) else "groups"))
        if getattr(self, "_engine_stamp", None) != stamp:
            sd = {k: v.detach() for k, v in zip(self._engine_pnames, plist)}
            # Head precision policy ("auto"): the DPT head runs in split precision (3x its MACs) where its fp16 operand rounding
            # is what limits parity with the fp32 reference -- the unbounded-output models (raw ReLU head, 'ssi' heads: no sigmoid
            # compresses the logit noise) and ViT-S (64-feature head: few terms per output to average the rounding over).  The
            # sigmoid ViT-B/L models -- the benchmarked configurations -- keep the single-precision head as the FIRST RUNG of a precision
            # ladder (DepthEngine._escalate): images whose depth map sits where the sigmoid does not compress the logit error get the head re-run
            # in split precision.
            mode = getattr(self, "head_precision", "auto")
            # ONE decision for everything that touches the fp8 form on the sigmoid ViT-B / ViT-L models (with module.f8_terms = "none" / "enc"
            # the first rung's "proj" split, the taps' [hi | lo8 | hi8] form and the fp16-packed weights used to be derived separately and met in _kdup
            # as "three-term fp16 weights against a [hi | lo8 | hi8] operand"): the caller's f8_terms, when given, rules the first rung's projects, the
            # second rung's weights, the form of the taps between them and the third rung alike.
            f8_user = getattr(self, "f8_terms", None)
            head_f8_ok = f8_user is None or f8_user in ("head", "both")
            rung1_f8 = (mode == "auto" and self.depth_head.final_act == "sigmoid" and self.encoder != "vits" and _rung1_proj_f8() and head_f8_ok)
            ladder_f8 = _LADDER_F8 if f8_user is None else ("head" if head_f8_ok else "none")
            split = _head_split_policy(mode, self.encoder, self.depth_head.final_act, proj_f8=rung1_f8)
            guided, amodal_head = self.pretrained.has_guidance, hasattr(self.depth_head, "input_projection")
            enc_split = _encoder_split_policy(getattr(self, "encoder_precision", "auto"), self.encoder, self.depth_head.final_act)
            f8_only = None
            # the first rung of the sigmoid ViT-B / ViT-L heads (ladder on or off; an explicit head_precision or f8_terms is the caller's choice)
            if f8_user is None and rung1_f8 and "proj" in split:
                f8, f8_only = "head", ("proj",)
            pw = PackedWeights(sd, self.encoder, guided=guided, amodal_head=amodal_head,
                               split_head=split,
                               enc_split_blocks=enc_split, tap_split=ladder_r is not None, f8=f8, f8_only=f8_only,
                               tap_f8=(ladder_f8 in ("both", "head")) if ladder_r is not None else None)
            ladder = None
            if ladder_r is not None:
                encoder = self.encoder
                groups = tuple(g for g in HEAD_GROUPS if g not in _LADDER_SKIP.get(encoder, ()) and g != "projw")
                final_act, norm_in, depth = self.depth_head.final_act, bool(getattr(self, "normalise_input", False)), len(self.pretrained.blocks)
                every = tuple(g for g in HEAD_GROUPS if g != "projw")

                def third():
                    return DepthEngine(PackedWeights(sd, encoder, guided=guided, amodal_head=amodal_head, split_head=every, enc_split_blocks=depth, f8=f8_user or "both"), final_act, norm_in)
                ladder = dict(r=ladder_r, div=_LADDER_FALLBACK["div"], div_in=_LADDER_DIV_IN, make=lambda: PackedWeights(sd, encoder, guided=guided, amodal_head=amodal_head, split_head=groups, head_only=True, f8=ladder_f8),
                              r3=max(_LADDER_FALLBACK["r3"], ladder_r), make3=third)
            if ladder is None and _flat_input_rung(self, self.depth_head.final_act, mode):
                final_act, norm_in, depth, encoder, f8_ = self.depth_head.final_act, bool(getattr(self, "normalise_input", False)), len(self.pretrained.blocks), self.encoder, f8

                def everything():
                    return DepthEngine(PackedWeights(sd, encoder, guided=guided, amodal_head=amodal_head, split_head=tuple(g for g in HEAD_GROUPS if g != "projw"),
                                                     enc_split_blocks=depth, f8=f8_), final_act, norm_in)
                # (input-side trigger only: the raw models' last tap does not tell -- synthetic raw ViT-G reads a token diversity of 0.09 on a small NOISE image)
                # round 6: + the r trigger (sum of the metric's weights / sum |out|, hip_ext.engine._escalate) with a CALIBRATED threshold -- the un-centred raw maps of
                # profiles/r05_ac_* (mostly clipped, the rest just above the kink) read up to 1.1e-3 under the default policy and 6.6e-4 with everything split
                ladder = dict(div_in=_LADDER_DIV_IN, make3=everything)
            eng = DepthEngine(pw, self.depth_head.final_act, bool(getattr(self, "normalise_input", False)), ladder=ladder)
            object.__setattr__(self, "_engine_obj", eng)
            object.__setattr__(self, "_engine_stamp", stamp)
            if ladder is not None:
                self._calibrate_ladder(eng, plist, explicit_r=getattr(self, "precision_ladder", None) not in (None, True) or _os.environ.get("ADA_LADDER_R") is not None)
        return self._engine_obj

    def _calibrate_ladder(self, eng, plist, explicit_r=False):
        """Installs the calibrated thresholds (hip_ext.engine.DepthEngine.calibrate) into the engine's ladder and exposes the numbers as
        ``module.ladder_calibration``.  Calibration set: four noise and four image-like synthetic inputs (src/util/synth_weights.make_inputs; the guide tensor takes
        the trailing channels of [rgb | mask | observation], whatever the guide_type) plus one all-zero image that places the tap-diversity threshold, at
        _LADDER_CAL_SIZE.  Cached by the stamp of every parameter BUT the final 1x1 bias (the logit error does not depend on it -- the bias is added in fp32 -- and
        callers that sweep the operating point rewrite exactly that tensor).  A threshold the caller named (module.precision_ladder = <float> / ADA_LADDER_R)
        is kept; the third rung and the diversity trigger are calibrated all the same."""
        import torch as _torch
        lad = eng.ladder
        if not _LADDER_CALIBRATE or "make3" not in lad or not plist[0].is_cuda:     # (CPU parameters: the forward itself refuses)
            object.__setattr__(self, "ladder_calibration", None)
            return
        # (no capture case to handle: packing the weights synchronises too -- an engine cannot be BUILT inside a caller's stream capture; warm up first, as for any graph)
        skip = "depth_head.scratch.output_conv2.2.bias"
        key = tuple((v.data_ptr(), v._version) for n, v in zip(self._engine_pnames, plist) if n != skip) + (getattr(self, "f8_terms", None), _LADDER_BUDGET, _LADDER_SAFETY, _LADDER_RULE, _LADDER_CAL_SIZE)
        cal = self.__dict__.get("_ladder_cal")
        if cal is None or cal[0] != key:
            from src.util.synth_weights import make_inputs
            dev = plist[0].device
            H, W = _LADDER_CAL_SIZE
            parts = [make_inputs(4, H, W, seed=9001, device=dev, style="noise"), make_inputs(4, H, W, seed=9002, device=dev, style="structured"),
                     make_inputs(1, H, W, seed=0, device=dev, style="zeros")]
            x = _torch.cat([p[0] for p in parts], 0)
            guide = None
            cg = eng.w.guide_channels if eng.w.guided else 0
            if cg:
                guide = _torch.cat([_torch.cat([p[1], p[2], p[3]], 1) for p in parts], 0)[:, 5 - cg:].contiguous()
            cal = (key, eng.calibrate(x, guide, budget=_LADDER_BUDGET, safety=_LADDER_SAFETY, flat_index=x.shape[0] - 1, rule=_LADDER_RULE))
            object.__setattr__(self, "_ladder_cal", cal)
        res = dict(cal[1])
        if "r" in lad and not explicit_r and "r" in res:
            lad["r"] = res["r"]
        if "r" in lad:
            lad["r3"] = max(res["r3"], lad["r"]) if "eps2" in res else lad["r3"]
        else:
            lad["r3"] = res["r3"]
        if "div" in lad and "div" in res:
            lad["div"] = res["div"]
        res.update(r_installed=lad.get("r"), r3_installed=lad.get("r3"), div_installed=lad.get("div"))
        object.__setattr__(self, "ladder_calibration", res)

    def _run(self, x, guide, normalise=None):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # inference-only build: autograd through the HIP kernels is not provided (training is out of scope)
            pass
        eng = self._engine()
        B = x.shape[0]
        step = eng.max_batch(x.shape[-2], x.shape[-1])
        if B <= step:
            return eng.forward(x, guide, normalise)
        outs = [eng.forward(x[i:i + step], None if guide is None else guide[i:i + step], normalise).clone() for i in range(0, B, step)]
        return torch.cat(outs, dim=0)


class DepthAnythingV2(_EngineMixin, nn.Module):
    def __init__(self, encoder="vitl", features=256, out_channels=(256, 512, 1024, 1024), use_bn=False, use_clstoken=False,
                 guide_type=None, loss_stategy=None):
        super().__init__()
        self.intermediate_layer_idx = INTERMEDIATE_LAYER_IDX
        self.encoder = encoder
        if guide_type is None:
            raise NotImplementedError  # reference DA2/dinov2.py:124-125: guided model needs an explicit guide_type
        self.pretrained = DINOv2(model_name=encoder, guide_type=guide_type)
        self.depth_head = DPTHead(self.pretrained.embed_dim, features, use_bn, out_channels=out_channels,
                                  use_clstoken=use_clstoken, loss_stategy=loss_stategy or "")
        self.normalise_input = False

    def forward(self, x, guidance_mask):
        return self._run(x, guidance_mask)

    def forward_modular(self, x, guidance_mask):
        """Module-by-module evaluation (reference DA2/dpt.py:225-231) -- same kernels, unfused; used by tests."""
        if self.normalise_input:
            mean = torch.tensor([0.485, 0.456, 0.406], device=x.device).view(-1, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225], device=x.device).view(-1, 1, 1)
            x = (x - mean) / std
        ph, pw = x.shape[-2] // 14, x.shape[-1] // 14
        feats = self.pretrained.get_intermediate_layers(x, self.intermediate_layer_idx[self.encoder], return_class_token=True,
                                                        guidance_mask=guidance_mask)
        return self.depth_head(feats, ph, pw)
