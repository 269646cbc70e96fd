"""Per-dataset text prompts for CLIPSeg — counterpart of /root/reference/src/models/clipseg/prompt.py:6-36.

The reference tokenises four prompt strings with `clip.tokenize` (OpenAI's byte-level BPE, context length 77) when the module is imported and exports the
results as `ln_prompt`, `busi_prompt`, `thyroid_prompt`, `prostate_prompt` ([1, 77] integer tensors).  The tokenizer and its 1.3 MB merges file are host-side
data preparation outside the hot path; the token ids are an INPUT of the hot path.  They are carried here as data — prompt_ids.json, written by
oracle/gen_prompt_ids.py in the build container from the reference's strings, with the reference's own tokenizer and the oracle's restatement of it agreeing
on every id (BUSI: 68 tokens, [49406, 1465, 2326, 9475, 534, ..., 2498, 46092, 269, 49407], as SURVEY §8c printed them).

`synthetic_prompt` (seeded random ids between SOT and EOT) remains for tests and tools that want a prompt of a chosen length; no entry point uses it.
"""
import json
import os

import torch

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prompt_ids.json")) as _f:
    _DATA = json.load(_f)

SOT, EOT, CONTEXT_LENGTH = _DATA["sot"], _DATA["eot"], _DATA["context_length"]


def _tensor(name):
    ids = _DATA["prompts"][name]["ids"]
    assert len(ids) == CONTEXT_LENGTH and ids[0] == SOT and ids[_DATA["prompts"][name]["n_tokens"] - 1] == EOT
    return torch.tensor([ids], dtype=torch.int32)               # clip.tokenize returns IntTensor on torch >= 1.8 (clip.py:243-246)


ln_prompt = _tensor("ln_prompt")
busi_prompt = _tensor("busi_prompt")
thyroid_prompt = _tensor("thyroid_prompt")
prostate_prompt = _tensor("prostate_prompt")


def synthetic_prompt(n_tokens=20, context_length=77, seed=0):
    g = torch.Generator().manual_seed(seed)
    ids = torch.zeros(1, context_length, dtype=torch.long)
    ids[0, 0] = SOT
    ids[0, 1:n_tokens - 1] = torch.randint(300, 40000, (n_tokens - 2,), generator=g)
    ids[0, n_tokens - 1] = EOT                 # highest id → argmax pooling position (model.py:372)
    return ids
