"""CLIPSeg adapter + decoder — functional CPU restatement.  Test infrastructure only.

Adapter: /root/reference/src/third_party/openai_clip/clipseg_adapter.py
  extract_vit_features :42-71 (hidden states after resblocks `extract_layers`), forward :73-98
  (decoder(hidden_states, conditional_embeddings=encode_text(ids)) → view(B,-1,H,W) → cat(-l, l)).
Decoder: transformers `CLIPSegDecoder` (third party; the reference instantiates it from
  "CIDAS/clipseg-rd64-refined", clipseg_adapter.py:30-37) — restated from the installed 5.15 source
  (modeling_clipseg.py CLIPSegDecoder / CLIPSegDecoderLayer / CLIPSegAttention / CLIPSegMLP), SURVEY Appendix A.3:
    out = reduce_i(act_i) (+ out);  FiLM at conditional_layer;  POST-LN layer (eps 1e-5, ReLU MLP);
    tokens[1:] → [B,C,h,w] → Conv3x3 → ReLU → ConvT(k4,s4) → ReLU → ConvT(k4,s4).
  PARITY UNPINNED BY THE REFERENCE; pinned by tests/golden/clipseg_*.npz captured from the reference adapter driving the
  installed decoder.
"""
import math

import torch
import torch.nn.functional as F

from . import text_ref, vit_ref
from .vit_ref import _sub


def decoder_layer(x, P, heads, eps=1e-5):
    """CLIPSegDecoderLayer: x = LN1(x + attn(x)); x = LN2(x + fc2(relu(fc1 x))).  P keys relative to 'layers.{i}.'."""
    B, N, D = x.shape
    dh = D // heads
    q = F.linear(x, P["self_attn.q_proj.weight"], P["self_attn.q_proj.bias"]).view(B, N, heads, dh).transpose(1, 2)
    k = F.linear(x, P["self_attn.k_proj.weight"], P["self_attn.k_proj.bias"]).view(B, N, heads, dh).transpose(1, 2)
    v = F.linear(x, P["self_attn.v_proj.weight"], P["self_attn.v_proj.bias"]).view(B, N, heads, dh).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v
    a = F.linear(a.transpose(1, 2).reshape(B, N, D), P["self_attn.out_proj.weight"], P["self_attn.out_proj.bias"])
    x = F.layer_norm(x + a, (D,), P["layer_norm1.weight"], P["layer_norm1.bias"], eps)
    h = F.linear(F.relu(F.linear(x, P["mlp.fc1.weight"], P["mlp.fc1.bias"])), P["mlp.fc2.weight"], P["mlp.fc2.bias"])
    return F.layer_norm(x + h, (D,), P["layer_norm2.weight"], P["layer_norm2.bias"], eps)


def decoder_forward(hidden_states, cond, P, heads=4, conditional_layer=0):
    """hidden_states: tuple of [B,N,Dv] in extraction order; cond [B,E]; P keys relative to 'decoder.'.  → logits [B,H,W]."""
    acts = hidden_states[::-1]
    out = None
    for i, act in enumerate(acts):
        r = F.linear(act, P[f"reduces.{i}.weight"], P[f"reduces.{i}.bias"])
        out = r if out is None else r + out
        if i == conditional_layer:
            mul = F.linear(cond, P["film_mul.weight"], P["film_mul.bias"])
            add = F.linear(cond, P["film_add.weight"], P["film_add.bias"])
            out = mul[:, None, :] * out + add[:, None, :]
        out = decoder_layer(out, _sub(P, f"layers.{i}."), heads)
    B, N, C = out.shape
    g = int(math.sqrt(N - 1))
    x = out[:, 1:, :].transpose(1, 2).reshape(B, C, g, g)
    x = F.relu(F.conv2d(x, P["transposed_convolution.0.weight"], P["transposed_convolution.0.bias"], padding=1))
    k1 = P["transposed_convolution.2.weight"].shape[-1]
    x = F.relu(F.conv_transpose2d(x, P["transposed_convolution.2.weight"], P["transposed_convolution.2.bias"], stride=k1))
    k2 = P["transposed_convolution.4.weight"].shape[-1]
    x = F.conv_transpose2d(x, P["transposed_convolution.4.weight"], P["transposed_convolution.4.bias"], stride=k2)
    return x.squeeze(1)


def adapter_forward(images, ids, P, vit_heads, text_heads, extract_layers=(3, 6, 9), dec_heads=4):
    """CLIPSegAdapter.forward (:73-98).  P: flat dict with 'clip_model.*' (OpenAI CLIP names) and 'decoder.*'."""
    clipP = _sub(P, "clip_model.")
    _, taps = vit_ref.openai_vit_forward(images, clipP, heads=vit_heads, taps=tuple(extract_layers))
    cond = text_ref.openai_text_forward(ids, clipP, heads=text_heads)
    logits = decoder_forward(tuple(taps), cond, _sub(P, "decoder."), heads=dec_heads)
    B, _, H, W = images.shape
    logits = logits.view(B, -1, H, W)
    return torch.cat([-logits, logits], dim=1) if logits.shape[1] == 1 else logits
