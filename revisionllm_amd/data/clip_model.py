"""CLIP towers (ViT image encoder + causal text transformer) on the HIP kernels - SURVEY 8 f-4.

What the reference's feature extractors run upstream of the grounding path (``clip.load("ViT-L/14")`` ->
``encode_image`` / ``encode_text``; data/feature_extraction/clip_extractor.py:13-54, clip/model.py:223-242, 339-352).
Every matrix product, LayerNorm and attention goes through the C ABI (``rv_gemm`` with fused bias / QuickGELU / residual
epilogues, ``rv_layernorm``, ``rv_attention`` with 64-wide heads); torch only moves rows around (patch unfold, class
token concat, embedding gather).  Residual stream f32, GEMM inputs bf16, weights fragment-packed bf16 - as in the engine.
State-dict names are the checkpoint's own (``visual.conv1.weight``, ``transformer.resblocks.3.mlp.c_fc.weight``, ...).
"""
import torch

from .. import hip, ops
from ..utils import hashinit, synth


def _pad_k(t, mult=128):
    """Zero-pad the last dim to a multiple of ``mult`` (the patch embedding has K = 3 * 14 * 14 = 588)."""
    k = t.shape[-1]
    kp = (k + mult - 1) // mult * mult
    return t if kp == k else torch.nn.functional.pad(t, (0, kp - k))


class ClipTowers:
    def __init__(self, embed_dim=768, image_res=224, patch=14, v_width=1024, v_layers=24, ctx=77, vocab=49408, t_width=768,
                 t_layers=12, t_heads=None, device="cuda:0", op_dtype=None):
        self.cfg = dict(embed_dim=embed_dim, image_res=image_res, patch=patch, v_width=v_width, v_layers=v_layers, ctx=ctx,
                        vocab=vocab, t_width=t_width, t_layers=t_layers)
        self.v_heads = v_width // 64                      # clip/model.py:268
        self.t_heads = t_heads or t_width // 64           # build_model: transformer_heads = transformer_width // 64
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise hip.HipLibraryError("ClipTowers runs on the HIP kernels only (no CPU path)")
        self.op_dtype = hip.op_dtype(op_dtype)      # the library flavour the towers run on (fp16 by default)
        hip.lib(self.op_dtype)
        self.w = {}

    # ---- weights ---------------------------------------------------------------------------------------------------
    def _bind(self, name, t):
        dev = self.device
        t = t.to(dev)
        if name.endswith((".in_proj_weight", ".out_proj.weight", ".c_fc.weight", ".c_proj.weight")):
            self.w[name] = ops.pack_fragments(t.to(self.op_dtype).contiguous())
        elif name == "visual.conv1.weight":                # [W,3,p,p] -> [W, 3*p*p] padded in K
            self.w[name] = ops.pack_fragments(_pad_k(t.reshape(t.shape[0], -1)).to(self.op_dtype).contiguous())
        elif name in ("visual.proj", "text_projection"):   # x @ P  ->  rows of P^T
            self.w[name] = ops.pack_fragments(t.t().to(self.op_dtype).contiguous())
        elif name == "token_embedding.weight":
            self.w[name] = t.to(self.op_dtype).contiguous()
        else:                                              # biases, LayerNorm affine, embeddings
            self.w[name] = t.float().contiguous()

    def load_state_dict(self, sd):
        want = {n for n, *_ in synth.clip_towers_spec(**self.cfg)}
        missing = want - set(sd)
        if missing:
            raise KeyError(f"CLIP checkpoint lacks {sorted(missing)[:4]} ...")
        for n in want:
            self._bind(n, sd[n])
        return self

    def init_synthetic(self, seed=0, prefix="clip."):
        """Hash-initialised weights of the configured shapes (benches / parity tests; the oracle rebuilds the same values)."""
        for n, shp, a, base in synth.clip_towers_spec(**self.cfg):
            t = torch.empty(shp, dtype=torch.float32, device=self.device)
            ops.init_hash_(t, prefix + n, seed, a, base)
            self._bind(n, t)
        return self

    # ---- shared transformer block (clip/model.py:167-190) -------------------------------------------------------------
    def _block(self, x, p, n, L, heads, causal):
        w, W = self.w, x.shape[1]
        _, xn, _ = ops.layernorm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], want=("op16",), op_dtype=self.op_dtype)
        qkv = ops.gemm(xn, w[p + "attn.in_proj_weight"], bias=w[p + "attn.in_proj_bias"], w_packed=True).view(n, L, 3, heads, W // heads)
        a = ops.attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=causal).view(n * L, W)
        x = ops.gemm(a, w[p + "attn.out_proj.weight"], bias=w[p + "attn.out_proj.bias"], residual=x, out_dtype=torch.float32, w_packed=True)
        _, xn, _ = ops.layernorm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], want=("op16",), op_dtype=self.op_dtype)
        h = ops.gemm(xn, w[p + "mlp.c_fc.weight"], bias=w[p + "mlp.c_fc.bias"], act=hip.RV_ACT_QUICK_GELU, w_packed=True)
        return ops.gemm(h, w[p + "mlp.c_proj.weight"], bias=w[p + "mlp.c_proj.bias"], residual=x, out_dtype=torch.float32, w_packed=True)

    # ---- towers --------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def encode_image(self, img):
        """img [n,3,R,R] (normalised, any float dtype) -> f32 [n, embed_dim]   (VisualTransformer.forward)."""
        c, w = self.cfg, self.w
        n, p, W = img.shape[0], c["patch"], c["v_width"]
        g = c["image_res"] // p
        if tuple(img.shape[1:]) != (3, c["image_res"], c["image_res"]):
            raise ValueError(f"expected [n,3,{c['image_res']},{c['image_res']}] frames, got {tuple(img.shape)}")
        x = img.to(self.device, torch.float32)
        # stride = kernel convolution == GEMM over unfolded patches: rows (frame, gy, gx), columns (channel, py, px)
        patches = x.view(n, 3, g, p, g, p).permute(0, 2, 4, 1, 3, 5).reshape(n * g * g, 3 * p * p)
        tokens = ops.gemm(_pad_k(patches).to(self.op_dtype), w["visual.conv1.weight"], out_dtype=torch.float32, w_packed=True)
        L = g * g + 1
        x = torch.cat([w["visual.class_embedding"].expand(n, 1, W), tokens.view(n, g * g, W)], 1) + w["visual.positional_embedding"]
        x, _, _ = ops.layernorm(x.view(n * L, W).contiguous(), w["visual.ln_pre.weight"], w["visual.ln_pre.bias"], want=("f32",), op_dtype=self.op_dtype)
        for l in range(c["v_layers"]):
            x = self._block(x, f"visual.transformer.resblocks.{l}.", n, L, self.v_heads, False)
        cls = x.view(n, L, W)[:, 0].contiguous()
        _, cls16, _ = ops.layernorm(cls, w["visual.ln_post.weight"], w["visual.ln_post.bias"], want=("op16",), op_dtype=self.op_dtype)
        return ops.gemm(cls16, w["visual.proj"], out_dtype=torch.float32, w_packed=True)

    @torch.no_grad()
    def encode_text(self, tokens):
        """tokens int64 [n,ctx] -> dict(last_hidden_state f32 [n,ctx,W], pooler_output f32 [n,E])   (CLIP.encode_text)."""
        c, w = self.cfg, self.w
        n, L, W = tokens.shape[0], tokens.shape[1], c["t_width"]
        if L != c["ctx"]:
            raise ValueError(f"expected context length {c['ctx']}, got {L}")
        tok = ops.h2d(tokens, self.device, torch.long)
        x = (w["token_embedding.weight"][tok].float() + w["positional_embedding"]).view(n * L, W).contiguous()
        for l in range(c["t_layers"]):
            x = self._block(x, f"transformer.resblocks.{l}.", n, L, self.t_heads, True)
        hid, hid16, _ = ops.layernorm(x, w["ln_final.weight"], w["ln_final.bias"], want=("f32", "op16"), op_dtype=self.op_dtype)
        eot = hid16.view(n, L, W)[torch.arange(n, device=self.device), tok.argmax(-1)].contiguous()
        return dict(last_hidden_state=hid.view(n, L, W), pooler_output=ops.gemm(eot, w["text_projection"], out_dtype=torch.float32, w_packed=True))
