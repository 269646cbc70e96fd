"""Text towers — functional CPU restatement.  Test infrastructure only.

(1) bert_text_forward: BiomedCLIP text tower = open_clip 3.2.0 ``HFTextEncoder`` around HF
    ``BertModel`` (third-party; call site /root/reference/src/models/biomedclip/finetune.py:277;
    LoRA attribute paths src/adapters/lora.py:319-365).  Published arithmetic per SURVEY Appendix A.2:
    word+position+token_type(0) embeddings -> LN(1e-12); 12 post-LN layers with key-padding mask;
    CLS pool; proj = Linear(768,640,no bias) -> GELU -> Linear(640,512,no bias).
    PARITY UNPINNED BY THE REFERENCE — cross-checked against the installed transformers BertModel.
(2) openai_text_forward: in-tree CLIP.encode_text,
    /root/reference/src/third_party/openai_clip/model.py:361-374 with the causal mask of :346-352.
    PINNED by golden vectors from the imported reference.
"""
import torch
import torch.nn.functional as F

from .vit_ref import _attention, _sub, openai_block


def bert_layer(x, P, heads, key_mask_add, eps=1e-12, lora=None):
    """HF BertLayer, post-LN.  P keys relative to 'encoder.layer.{i}.'.  lora = dict(r, alpha): query/key/value/attention.output.dense
    carry LinearLoRA factors `<name>.w_lora_A/B` where present (reference src/adapters/lora.py:317-367, --tune_text_encoder)."""
    from .vit_ref import _maybe_lora_linear
    D = x.shape[-1]
    q = _maybe_lora_linear(x, P, "attention.self.query", lora)
    k = _maybe_lora_linear(x, P, "attention.self.key", lora)
    v = _maybe_lora_linear(x, P, "attention.self.value", lora)
    a = _attention(q, k, v, heads, key_mask_add)
    a = _maybe_lora_linear(a, P, "attention.output.dense", lora)
    x = F.layer_norm(x + a, (D,), P["attention.output.LayerNorm.weight"], P["attention.output.LayerNorm.bias"], eps)
    h = F.gelu(F.linear(x, P["intermediate.dense.weight"], P["intermediate.dense.bias"]))
    h = F.linear(h, P["output.dense.weight"], P["output.dense.bias"])
    return F.layer_norm(x + h, (D,), P["output.LayerNorm.weight"], P["output.LayerNorm.bias"], eps)


def bert_hidden(ids, P, heads=12, prefix="text.transformer.", pad_id=0, lora=None):
    B, L = ids.shape
    e = F.embedding(ids, P[prefix + "embeddings.word_embeddings.weight"], padding_idx=pad_id)   # HF: nn.Embedding(..., padding_idx=pad_token_id)
    e = e + P[prefix + "embeddings.position_embeddings.weight"][:L][None]
    e = e + P[prefix + "embeddings.token_type_embeddings.weight"][0][None, None]
    D = e.shape[-1]
    x = F.layer_norm(e, (D,), P[prefix + "embeddings.LayerNorm.weight"], P[prefix + "embeddings.LayerNorm.bias"], 1e-12)
    keep = ids != pad_id                                               # open_clip HFTextEncoder attn_mask
    mask_add = torch.zeros(B, 1, 1, L, dtype=x.dtype).masked_fill(~keep[:, None, None, :], float("-inf"))
    lp = prefix + "encoder.layer."
    depth = 1 + max(int(k[len(lp):].split(".")[0]) for k in P if k.startswith(lp))
    for i in range(depth):
        x = bert_layer(x, _sub(P, f"{lp}{i}."), heads, mask_add, lora=lora)
    return x


def bert_text_forward(ids, P, heads=12, prefix="text.", lora=None):
    x = bert_hidden(ids, P, heads, prefix + "transformer.", lora=lora)
    pooled = x[:, 0]                                                   # cls_last_hidden_state_pooler
    h = F.gelu(F.linear(pooled, P[prefix + "proj.0.weight"]))
    return F.linear(h, P[prefix + "proj.2.weight"])


def openai_text_forward(ids, P, heads):
    """CLIP.encode_text (model.py:361-374); P uses the CLIP root key names."""
    B, L = ids.shape
    x = P["token_embedding.weight"][ids] + P["positional_embedding"][:L]
    mask = torch.full((L, L), float("-inf")).triu_(1)                  # :346-352
    bp = "transformer.resblocks."
    depth = 1 + max(int(k[len(bp):].split(".")[0]) for k in P if k.startswith(bp))
    for i in range(depth):
        x = openai_block(x, _sub(P, f"{bp}{i}."), heads, mask)
    D = x.shape[-1]
    x = F.layer_norm(x, (D,), P["ln_final.weight"], P["ln_final.bias"], 1e-5)
    eot = ids.argmax(dim=-1)                                           # :372
    return x[torch.arange(B), eot] @ P["text_projection"]
