"""Instruction mix of every backward-branch loop of one kernel in a hipcc -S listing: python tools/asm_loops.py file.s <kernel-name-substring> [min_mfma]."""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 4
start = next(i for i, l in enumerate(s) if l.startswith('_Z') and key in l and l.rstrip().split(':')[0].endswith(l.split(':')[0]))
end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
lines = [l for l in s[start:end] if l.strip() and not l.strip().startswith((';', '.s', '.p', '.a', '.t', '.g')) or l.strip().startswith('.LBB')]
labels = {l.strip().split(':')[0]: i for i, l in enumerate(lines) if re.match(r'\s*\.LBB\S+:', l)}
tot = Counter(x.strip().split()[0] for x in lines if not re.match(r'\s*\.LBB\S+:', x) and not x.startswith('_Z'))
print("whole kernel:", len(lines), "instructions;", tot.most_common(12))
for i, l in enumerate(lines):
    m = re.match(r'\s*s_c?branch\w*\s+(\.LBB\S+)', l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        j = labels[m.group(1)]
        c = Counter(x.strip().split()[0] for x in lines[j:i + 1] if not re.match(r'\s*\.LBB\S+:', x))
        nm = sum(v for k, v in c.items() if 'mfma' in k)
        if nm >= min_mfma:
            print(f"loop {m.group(1)}: {i - j} instr, mfma {nm}, valu {sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)}, salu {sum(v for k, v in c.items() if k.startswith('s_') and 'waitcnt' not in k)}, ds {sum(v for k, v in c.items() if k.startswith('ds_'))}, waitcnt {c.get('s_waitcnt', 0)}")
            print("   ", [(k, v) for k, v in c.most_common(40) if 'mfma' not in k])
