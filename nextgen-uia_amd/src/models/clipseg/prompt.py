"""Per-dataset text prompts for CLIPSeg (counterpart of /root/reference/src/models/clipseg/prompt.py:14-20).  The reference
tokenises with OpenAI's BPE at import time; the tokenizer's vocabulary file is host-side data outside the hot path, so prompts
are carried here as the token ids the reference produces (BUSI prompt → 68 tokens) or synthesised."""
import torch

SOT, EOT = 49406, 49407


def synthetic_prompt(n_tokens=20, context_length=77, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(1, context_length, dtype=torch.long)
    ids[0, 0] = SOT
    ids[0, 1:n_tokens - 1] = torch.randint(300, 40000, (n_tokens - 2,), generator=g)
    ids[0, n_tokens - 1] = EOT                 # highest id → argmax pooling position (model.py:372)
    return ids


busi_prompt = synthetic_prompt(68, 77, seed=1)
