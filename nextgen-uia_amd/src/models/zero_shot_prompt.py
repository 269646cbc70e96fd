"""The zero-shot prompt ensembles of the reference (src/models/zero_shot_prompt.py:1-54: ten prompts per class for the lymph-node sets, ten per class for the
breast sets) as DATA: the strings live in zero_shot_prompts.json beside this file (written by oracle/gen_golden_r05.py from the imported reference module) and
are exposed under the reference's names, so that `from src.models.zero_shot_prompt import LN_PROMPTS_ENSEMBLE, BREAST_PROMPTS_ENSEMBLE` (reference
biomedclip/zero_shot.py:22) works unchanged.  BASELINE configs[0] (zero-shot on BUSI) is defined on these prompts."""
import json
import os

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "zero_shot_prompts.json")) as _f:
    _DATA = json.load(_f)

LN_PROMPTS_ENSEMBLE = _DATA["LN_PROMPTS_ENSEMBLE"]
BREAST_PROMPTS_ENSEMBLE = _DATA["BREAST_PROMPTS_ENSEMBLE"]


def ensemble_for(dataset):
    """reference biomedclip/zero_shot.py:168-173: the dataset name picks the ensemble; anything else is an error."""
    name = dataset.lower()
    if "ln" in name:
        return LN_PROMPTS_ENSEMBLE
    if "busi" in name:
        return BREAST_PROMPTS_ENSEMBLE
    raise ValueError(f"Dataset {dataset} not supported")
