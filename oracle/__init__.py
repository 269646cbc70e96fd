"""oracle/ — CPU restatement of the reference's hot-path arithmetic.  TEST INFRASTRUCTURE ONLY.

Nothing under ``nextgen-uia_amd/`` may import this package: the product path is the HIP library
(``libuia_hip.so``) and fails loudly without it.  The only importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` — as the checker,
never as the thing measured or shipped.

Every function restates, in plain fp32 PyTorch ops on the CPU, the arithmetic of one piece of
jinggqu/NextGen-UIA's CLIP-adapter fine-tune path and cites the reference file:line it follows.
The restatement is *functional* (tensors + a flat ``{state_dict_key: tensor}`` mapping in, tensors
out); it shares no code with the reference's nn.Module classes.

Pinning (how we know the oracle is right) — see ``oracle/gen_golden.py`` and ``tests/golden/``:
  * in-tree reference code (src/adapters/{mona,lora}.py, src/losses/losses.py,
    src/third_party/openai_clip/{model,clipseg_adapter}.py) is importable in the build container;
    golden (inputs, params, outputs, grads) vectors were generated from it and are committed.
    PINNED.
  * third-party arithmetic the reference pulls from open_clip 3.2.0 / timm 1.0.20 (BiomedCLIP
    towers) is absent from /root/reference and from the container.  The timm ViT block and the
    HF BERT encoder are restated from their published definitions and cross-checked against the
    installed ``transformers`` ViTModel / BertModel / CLIPSegDecoder (golden vectors committed).
    The reference itself has no tests or golden vectors for them: PARITY UNPINNED BY THE
    REFERENCE for those rows (pinned only by our captured fixtures).
  * MONAI DiceCE / Dice: restated from the published formulae, hand-computed cases only:
    PARITY UNPINNED.
"""

from . import mona_ref, lora_ref, losses_ref, vit_ref, text_ref, train_ref, clipseg_ref  # noqa: F401
