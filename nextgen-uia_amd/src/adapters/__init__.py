"""Adapters for parameter-efficient fine-tuning — the working subset of the reference's exports.

The reference's package __init__ (/root/reference/src/adapters/__init__.py:21-39) also names
``FractionalMona`` and ``prompt_tuning.SimplePromptTuner / create_simple_prompt_tuner``, none of which
exist in the reference tree (its own import fails).  They are not invented here: asking for them
raises an explicit error.
"""
from .mona import (BaselineMona, NoiseAwareMona, FreqEnhancedMona, FreqEnhancedMonaOp, HybridNoiseFreqMona,  # noqa: F401
                   BatchFirstMonaWrapper, inject_mona_variant_to_clip, inject_mona_variant_to_open_clip)
from .lora import LinearLoRA, PlainMultiheadAttentionLoRA, inject_lora_to_clip, inject_lora_to_biomedclip  # noqa: F401

__all__ = ["FreqEnhancedMona", "FreqEnhancedMonaOp", "BaselineMona", "NoiseAwareMona", "HybridNoiseFreqMona",
           "inject_mona_variant_to_clip", "inject_lora_to_clip", "inject_mona_variant_to_open_clip", "inject_lora_to_biomedclip"]

_ABSENT = {"FractionalMona": "src/adapters/mona.py defines no FractionalMona (reference __init__.py:23 imports a missing name)",
           "SimplePromptTuner": "src/adapters/prompt_tuning.py does not exist in the reference (__init__.py:36-39)",
           "create_simple_prompt_tuner": "src/adapters/prompt_tuning.py does not exist in the reference (__init__.py:36-39)"}


def __getattr__(name):
    if name in _ABSENT:
        raise NotImplementedError(f"{name}: {_ABSENT[name]}; no semantics to reproduce")
    raise AttributeError(name)
