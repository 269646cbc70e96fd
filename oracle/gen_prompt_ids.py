"""Writes nextgen-uia_amd/src/models/clipseg/prompt_ids.json: the token ids of the reference's four CLIPSeg dataset prompts, as DATA.

    python oracle/gen_prompt_ids.py            (build container only: reads /root/reference; the ids travel, nothing else does)

The reference tokenises the four prompt strings of src/models/clipseg/prompt.py:6-36 with `clip.tokenize` when the module is imported.  That module cannot be
imported here (clip.py needs torchvision), so the strings are taken from its syntax tree (the literal arguments of the four `clip.tokenize(...)` calls, by the
name each result is bound to) and tokenised twice:
  1. by oracle/bpe_ref.py, the restatement, with the reference's merges file read where it lies;
  2. by the reference's own SimpleTokenizer (src/third_party/openai_clip/simple_tokenizer.py, loaded by file path).  Its `import ftfy` is satisfied by an
     identity `fix_text` — valid because the prompts are asserted to be plain printable ASCII without '&' (SURVEY §8c recorded the same way of importing it).
Both must agree on every id; the BUSI vector must be the 68 tokens SURVEY §8c printed from the reference ([49406, 1465, 2326, 9475, 534, ..., 2498, 46092, 269, 49407]).
"""
import ast
import importlib.util
import json
import os
import sys
import types

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
REF = os.environ.get("UIA_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "nextgen-uia_amd", "src", "models", "clipseg", "prompt_ids.json")


def prompt_strings():
    tree = ast.parse(open(os.path.join(REF, "src/models/clipseg/prompt.py")).read())
    found = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id.endswith("_prompt"):
            lits = [n.value for n in ast.walk(node.value) if isinstance(n, ast.Constant) and isinstance(n.value, str) and len(n.value) > 20]
            assert len(lits) == 1, node.targets[0].id
            found[node.targets[0].id] = lits[0]
    return found


def reference_tokenizer():
    shim = types.ModuleType("ftfy")
    shim.fix_text = lambda s: s
    sys.modules.setdefault("ftfy", shim)
    spec = importlib.util.spec_from_file_location("ref_simple_tokenizer", os.path.join(REF, "src/third_party/openai_clip/simple_tokenizer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.SimpleTokenizer()


def main():
    from oracle.bpe_ref import BPE, EOT, SOT
    strings = prompt_strings()
    assert sorted(strings) == ["busi_prompt", "ln_prompt", "prostate_prompt", "thyroid_prompt"], sorted(strings)
    mine = BPE(os.path.join(REF, "src/third_party/openai_clip/bpe_simple_vocab_16e6.txt.gz"))
    ref = reference_tokenizer()
    out = {"context_length": 77, "sot": SOT, "eot": EOT, "source": "src/models/clipseg/prompt.py:6-36 through clip.tokenize (clip.py:215-257)", "prompts": {}}
    for name, text in sorted(strings.items()):
        assert all(32 <= ord(c) < 127 for c in text) and "&" not in text, name
        ids = mine.tokenize(text)
        r = [SOT] + ref.encode(text) + [EOT]
        assert ids[:len(r)] == r and not any(ids[len(r):]), f"{name}: restatement and reference tokenizer disagree"
        out["prompts"][name] = {"n_tokens": len(r), "ids": ids}
    b = out["prompts"]["busi_prompt"]
    assert b["n_tokens"] == 68 and b["ids"][:5] == [49406, 1465, 2326, 9475, 534] and b["ids"][64:68] == [2498, 46092, 269, 49407], b["ids"][:68]
    with open(OUT, "w") as f:
        prompts = out.pop("prompts")
        head = json.dumps(out)[:-1]
        rows = ",\n".join(f'  "{k}": {json.dumps(v)}' for k, v in prompts.items())
        f.write(head + ', "prompts": {\n' + rows + "\n}}\n")
        out["prompts"] = prompts
    print({k: v["n_tokens"] for k, v in out["prompts"].items()}, "->", os.path.relpath(OUT, ROOT))


if __name__ == "__main__":
    main()
