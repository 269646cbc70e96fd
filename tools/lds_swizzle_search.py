"""Brute-force search of an XOR swizzle for a [16 rows][128 B] bf16 LDS tile that is bank-conflict-free for both read kinds of the attention\nbackward: ds_read_b128 row fragments and ds_read_b64_tr_b16 transposed fragments (bank rules: MI355X_MICROARCH.md, LDS table)."""
import itertools
# LDS tile: 16 rows x 128 B; 16-B chunk c of row r stored at chunk c ^ s(r). Find s minimizing conflicts for
# (a) ds_read_b128 row reads (lane l: row l&15, chunk (l>>4) [+4]) and (b) ds_read_b64_tr_b16 (lane: row 4g+qq, chunk 2dt+(pp>>1), +8*(pp&1))
B128_GROUPS = [list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32))]
B128_GROUPS += [[l+32 for l in g] for g in B128_GROUPS]
def conflicts_b128(s, kk):
    worst = 0
    for grp in B128_GROUPS:
        banks = {}
        for l in grp:
            li, g = l & 15, l >> 4
            addr = li*128 + (((g + 4*kk) ^ s(li)) << 4)
            for b in range(4):
                bank = ((addr >> 2) + b) % 64
                banks.setdefault(bank, set()).add(addr)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def conflicts_tr(s, dt):
    worst = 0
    for half in range(2):
        banks = {}
        for l in range(32*half, 32*half+32):
            li, g = l & 15, l >> 4
            qq, pp = li >> 2, li & 3
            r = 4*g + qq
            addr = r*128 + (((2*dt + (pp >> 1)) ^ s(r)) << 4) + 8*(pp & 1)
            for b in range(2):
                bank = ((addr >> 2) + b) % 64
                banks.setdefault(bank, set()).add(addr)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def score(s):
    return max(conflicts_b128(s, kk) for kk in range(2)), max(conflicts_tr(s, dt) for dt in range(4))
cur = lambda r: (r >> 1) & 7
print("current (r>>1)&7:", score(cur))
best = []
# linear maps: s bit i = parity(r & m_i)
for m in itertools.product(range(16), repeat=3):
    def s(r, m=m):
        v = 0
        for i in range(3):
            v |= (bin(r & m[i]).count('1') & 1) << i
        return v
    sc = score(s)
    if sc == (1, 1):
        best.append(m)
print(len(best), best[:20])
