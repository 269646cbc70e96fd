"""LoRA linear / LoRA multi-head attention — functional CPU restatement.  Test infrastructure only.

Follows /root/reference/src/adapters/lora.py:
  LoRALayer scaling alpha/sqrt(r)                 :20-21
  merge_BA  (B @ A, [out,in])                     :46-51
  LinearLoRA.forward                              :78-90   (dense BA form, exactly as the reference)
  PlainMultiheadAttentionLoRA.forward             :155-199
"""
import math
import torch
import torch.nn.functional as F


def scaling(r, alpha):
    return alpha / math.sqrt(r)


def linear_lora(x, W, b, A, Bm, r, alpha, keep_mask=None, p_drop=0.0):
    """y = x Wᵀ + b + s · drop(x) (B A)ᵀ.   A: [r,in], Bm: [out,r].  lora.py:78-90.

    keep_mask ({0,1}, shape of x) replaces the training-mode dropout of :82-83 deterministically.
    """
    y = F.linear(x, W, b)
    if r > 0:
        xd = x if keep_mask is None else x * keep_mask / (1.0 - p_drop)
        BA = Bm @ A                                   # :46-51 (materialised, as the reference does)
        y = y + torch.matmul(xd, BA.T) * scaling(r, alpha)
    return y


def mha_lora(x_lbd, P, num_heads, r, alpha, attn_mask=None):
    """Sequence-first self-attention with LoRA on q,k,v,o (lora.py:155-199).

    P keys: {q_proj,k_proj,v_proj,proj}.{weight,bias,w_lora_A,w_lora_B}.  Returns [L,B,D].
    """
    L, B, D = x_lbd.shape
    dh = D // num_heads

    def lin(name, t):
        return linear_lora(t, P[f"{name}.weight"], P.get(f"{name}.bias"), P[f"{name}.w_lora_A"], P[f"{name}.w_lora_B"], r, alpha)

    q, k, v = lin("q_proj", x_lbd), lin("k_proj", x_lbd), lin("v_proj", x_lbd)
    # :178-184  [L, B*H, dh] -> [B,H,L,dh]
    q = q.view(L, B * num_heads, dh).transpose(0, 1).reshape(B, num_heads, L, dh)
    k = k.view(L, B * num_heads, dh).transpose(0, 1).reshape(B, num_heads, L, dh)
    v = v.view(L, B * num_heads, dh).transpose(0, 1).reshape(B, num_heads, L, dh)
    s = q @ k.transpose(-1, -2) / math.sqrt(dh)
    if attn_mask is not None:
        s = s + attn_mask
    o = torch.softmax(s, dim=-1) @ v                                   # SDPA :188
    o = o.permute(2, 0, 1, 3).reshape(L * B, D)                        # :193
    o = lin("proj", o)
    return o.view(L, B, D)


def kaiming_uniform_a5(shape, generator=None):
    """nn.init.kaiming_uniform_(a=sqrt(5)) as used for lora_A at lora.py:43."""
    fan_in = shape[1]
    bound = math.sqrt(6.0 / ((1 + 5.0) * fan_in))
    return (torch.rand(shape, generator=generator) * 2 - 1) * bound
