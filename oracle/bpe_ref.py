"""TEST / FIXTURE INFRASTRUCTURE (never imported by the product): a CPU restatement of OpenAI CLIP's byte-level BPE tokenizer, the arithmetic behind
`clip.tokenize` that /root/reference/src/models/clipseg/prompt.py:6-36 applies to its four dataset prompts at import time.

Follows /root/reference/src/third_party/openai_clip/simple_tokenizer.py:
  :16-35   the byte -> printable-unicode table (printable Latin-1 bytes map to themselves, the other 68 to U+0100...)
  :62-75   vocabulary order: 256 byte symbols, the same 256 with the end-of-word mark, one entry per merge rule (48 894 of them: lines 1..48894 of the
           merges file), then <|startoftext|> = 49406 and <|endoftext|> = 49407
  :77-80   the splitting pattern (special tokens | English contractions | letter runs | single digits | runs of anything else that is not space)
  :82-122  greedy merging: repeatedly fuse EVERY occurrence of the adjacent pair with the lowest merge rank until no ranked pair is left
  :124-130 encode = clean, lower-case, split, bytes -> symbols, merge, look up
and clip.py:215-257 (`tokenize`: [SOT] + ids + [EOT], zero-padded to the context length, an over-long text raises unless truncate).

Cleaning (:50-59) is `ftfy.fix_text` + two rounds of `html.unescape` + whitespace folding.  ftfy is not installed in the build container; for printable-ASCII text
without '&' or '\\' sequences fix_text is the identity (it repairs mojibake, curly quotes, ligatures, control characters and HTML entities — none can occur), so
this restatement REFUSES anything else rather than guess.  The merges file is the reference's own data file and is read where it lies (it never travels: only
the token ids produced from it do — oracle/gen_prompt_ids.py)."""
import gzip
import html

import regex

SOT, EOT = 49406, 49407
N_MERGES = 49152 - 256 - 2
PATTERN = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)
WORD_END = "</w>"


def byte_symbols():
    """simple_tokenizer.py:16-35: 188 printable bytes keep their code point; the rest are numbered from 256 in byte order."""
    keep = set(range(0x21, 0x7F)) | set(range(0xA1, 0xAD)) | set(range(0xAE, 0x100))
    order = sorted(keep, key=lambda b: (0 if b < 0x7F else 1 if b < 0xAD else 2, b))
    table, nxt = {b: chr(b) for b in order}, 256
    for b in range(256):
        if b not in keep:
            table[b] = chr(nxt)
            order.append(b)
            nxt += 1
    return table, order


class BPE:
    def __init__(self, merges_path):
        lines = gzip.open(merges_path).read().decode("utf-8").split("\n")
        rules = [tuple(l.split()) for l in lines[1:N_MERGES + 1]]
        table, order = byte_symbols()
        self.table = table
        symbols = [table[b] for b in order]
        vocab = symbols + [s + WORD_END for s in symbols] + ["".join(r) for r in rules] + ["<|startoftext|>", "<|endoftext|>"]
        self.ids = {s: i for i, s in enumerate(vocab)}
        self.rank = {r: i for i, r in enumerate(rules)}
        assert self.ids["<|startoftext|>"] == SOT and self.ids["<|endoftext|>"] == EOT

    def merge(self, token):
        parts = list(token[:-1]) + [token[-1] + WORD_END]
        while len(parts) > 1:
            best = min(((self.rank.get((a, b), None), k) for k, (a, b) in enumerate(zip(parts, parts[1:])) if (a, b) in self.rank), default=None)
            if best is None:
                break
            a, b = parts[best[1]], parts[best[1] + 1]
            out, k = [], 0
            while k < len(parts):                       # every occurrence of the chosen pair, left to right, non-overlapping (:98-113)
                if k + 1 < len(parts) and parts[k] == a and parts[k + 1] == b:
                    out.append(a + b)
                    k += 2
                else:
                    out.append(parts[k])
                    k += 1
            parts = out
        return parts

    def encode(self, text):
        if any(not (32 <= ord(c) < 127) and c not in "\n\t" for c in text) or "&" in text or "\\" in text:
            raise ValueError("oracle/bpe_ref.py restates the tokenizer for plain printable-ASCII text only (ftfy.fix_text is the identity there)")
        text = regex.sub(r"\s+", " ", html.unescape(html.unescape(text)).strip()).strip().lower()
        out = []
        for tok in PATTERN.findall(text):
            sym = "".join(self.table[b] for b in tok.encode("utf-8"))
            out.extend(self.ids[p] for p in self.merge(sym))
        return out

    def tokenize(self, text, context_length=77):
        ids = [SOT] + self.encode(text) + [EOT]
        if len(ids) > context_length:
            raise RuntimeError(f"Input {text} is too long for context length {context_length}")            # clip.py:249-254, truncate=False
        return ids + [0] * (context_length - len(ids))
