"""Deterministic synthetic inputs for the benchmark configurations of
BASELINE.json.  Counter-mode splitmix64 so that the streams
are reproducible anywhere with plain numpy:

    r(seed, i) = splitmix64_mix(seed * 0x9E3779B97F4A7C15 + i)

C1: 8 x 1 kb, seed 1001; C2: 64 x 5 kb, seed 2001 -- a base sequence of iid
uniform ACGT and n derived sequences, each position substituted with
probability `sub` by a uniformly chosen different base (stream seed+1+i).
"""
import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_G = np.uint64(0x9E3779B97F4A7C15)
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
# the reference's complement (src/bidirected_graph.rs:73-85): ACGT in either case -> upper-case complement, N/n -> N, other bytes unchanged
_COMP = {**{b: b for b in range(256)}, 65: 84, 97: 84, 84: 65, 116: 65, 67: 71, 99: 71, 71: 67, 103: 67, 78: 78, 110: 78}


def _r(seed: int, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = np.uint64(seed) * _G + idx.astype(np.uint64) + _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def _unit(x: np.ndarray) -> np.ndarray:
    return (x >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def base_sequence(L: int, seed: int) -> np.ndarray:
    """codes 0..3"""
    return (_r(seed, np.arange(L)) & np.uint64(3)).astype(np.uint8)


def substitute(base: np.ndarray, sub: float, seed: int) -> np.ndarray:
    L = len(base)
    u = _unit(_r(seed, 2 * np.arange(L)))
    d = (_r(seed, 2 * np.arange(L) + 1) % np.uint64(3)).astype(np.uint8) + 1
    return np.where(u < sub, (base + d) & 3, base).astype(np.uint8)


def to_bytes(codes: np.ndarray) -> bytes:
    return _ACGT[codes].tobytes()


def reverse_complement(s: bytes) -> bytes:
    return bytes(_COMP[c] for c in reversed(s))


def snp_family(n: int, L: int, sub: float, seed: int, rc_every: int = 0):
    """n sequences derived from one base by independent substitutions.
    rc_every > 0: every rc_every-th sequence (1-based) is reverse-complemented."""
    base = base_sequence(L, seed)
    out = []
    for i in range(n):
        s = to_bytes(substitute(base, sub, seed + 1 + i))
        if rc_every and (i + 1) % rc_every == 0:
            s = reverse_complement(s)
        out.append((f"seq{i}", s))
    return out


def indel_family(n: int, L: int, sub: float, indel: float, seed: int, max_indel: int = 8):
    """substitutions plus short indels (python loop; keep n*L small)"""
    base = base_sequence(L, seed)
    out = []
    for i in range(n):
        sd = seed + 1 + i
        u = _unit(_r(sd, 4 * np.arange(L)))
        d = (_r(sd, 4 * np.arange(L) + 1) % np.uint64(3)).astype(np.uint8) + 1
        ln = (_r(sd, 4 * np.arange(L) + 2) % np.uint64(max_indel)).astype(np.int64) + 1
        ins = _r(sd, 4 * np.arange(L) + 3)
        s = bytearray()
        j = 0
        while j < L:
            x = u[j]
            if x < sub:
                s.append(int(_ACGT[(base[j] + d[j]) & 3])); j += 1
            elif x < sub + indel / 2:
                j += int(ln[j])
            elif x < sub + indel:
                v = int(ins[j])
                for q in range(int(ln[j])):
                    s.append(int(_ACGT[(v >> (2 * q)) & 3]))
                s.append(int(_ACGT[base[j]])); j += 1
            else:
                s.append(int(_ACGT[base[j]])); j += 1
        if not s:
            s = bytearray(b"A")
        out.append((f"seq{i}", bytes(s)))
    return out


def config_c1():
    """BASELINE.json configs[0]: 8 synthetic 1 kb sequences, 5% SNP, seed 1001"""
    return snp_family(8, 1000, 0.05, 1001)


def config_c2(n: int = 64):
    """BASELINE.json configs[1]: 64 synthetic 5 kb sequences (5% SNP), seed 2001"""
    return snp_family(n, 5000, 0.05, 2001)


def invert_segment(s: bytes, start: int, length: int) -> bytes:
    """reverse-complement s[start:start+length] in place (config C5's inversions)"""
    return s[:start] + reverse_complement(s[start:start + length]) + s[start + length:]


def insert_segment(s: bytes, pos: int, length: int, seed: int) -> bytes:
    return s[:pos] + to_bytes(base_sequence(length, seed)) + s[pos:]


def config_c3_like(n: int = 4, L: int = 3000, seed: int = 3001):
    """scaled surrogate of BASELINE.json configs[2] (HLA-zoo DRB1 is not in the container): 3 %
    substitutions, 0.3 % short indels and two larger insertions per sequence"""
    fam = indel_family(n, L, 0.03, 0.003, seed, max_indel=6)
    out = []
    for i, (name, s) in enumerate(fam):
        s = insert_segment(s, (i * 397) % max(1, len(s) - 1), 200 + 50 * i, seed + 100 + i)
        s = insert_segment(s, (i * 911 + 1500) % max(1, len(s) - 1), 300 + 40 * i, seed + 200 + i)
        out.append((name, s))
    return out


def config_c5_like(n: int = 4, L: int = 6000, seed: int = 5001):
    """scaled version of BASELINE.json configs[4]: 2 % substitutions, 0.1 % indels, inverted segments
    in some sequences and one sequence entirely reverse-complemented"""
    fam = indel_family(n, L, 0.02, 0.001, seed, max_indel=4)
    out = []
    for i, (name, s) in enumerate(fam):
        if i % 2 == 1:
            s = invert_segment(s, len(s) // 3, 400 + 100 * i)
        if i == n - 1:
            s = reverse_complement(s)
        out.append((name, s))
    return out


# --------------------------------------------------------------------------- full-size configurations
def indel_family_fast(n: int, L: int, sub: float, indel: float, seed: int, max_indel: int = 8, base=None):
    """vectorised relative of indel_family for large inputs: per base position one event drawn from the counter
    stream -- substitution (prob. sub), deletion of 1..max_indel bases starting there (indel/2), insertion of
    1..max_indel random bases in front of it (indel/2); events that fall inside a deleted stretch are dropped"""
    if base is None:
        base = base_sequence(L, seed)
    L = len(base)
    out = []
    idx = np.arange(L)
    for i in range(n):
        sd = seed + 1 + i
        u = _unit(_r(sd, 4 * idx))
        d = (_r(sd, 4 * idx + 1) % np.uint64(3)).astype(np.uint8) + 1
        ln = (_r(sd, 4 * idx + 2) % np.uint64(max_indel)).astype(np.int64) + 1
        is_sub = u < sub
        is_del = (u >= sub) & (u < sub + indel / 2)
        is_ins = (u >= sub + indel / 2) & (u < sub + indel)
        diff = np.zeros(L + max_indel + 1, dtype=np.int64)
        ds = np.nonzero(is_del)[0]
        np.add.at(diff, ds, 1)
        np.add.at(diff, ds + ln[ds], -1)
        deleted = np.cumsum(diff)[:L] > 0
        codes = np.where(is_sub, (base + d) & 3, base).astype(np.uint8)
        keep = ~deleted
        ins_pos = np.nonzero(is_ins & keep)[0]
        if len(ins_pos):
            reps = ln[ins_pos]
            tot = int(reps.sum())
            ins_codes = (_r(sd + 7919, np.arange(tot)) & np.uint64(3)).astype(np.uint8)
            # position of every kept base in the output = its rank among kept bases + inserted bases before it
            cum_ins = np.zeros(L, dtype=np.int64)
            cum_ins[ins_pos] = reps
            cum_ins = np.cumsum(cum_ins)                      # inclusive: insertion sits in front of its base
            kept_idx = np.nonzero(keep)[0]
            out_len = len(kept_idx) + tot
            res = np.empty(out_len, dtype=np.uint8)
            dst = np.arange(len(kept_idx)) + cum_ins[kept_idx]
            res[dst] = codes[kept_idx]
            mask = np.ones(out_len, dtype=bool)
            mask[dst] = False
            res[mask] = ins_codes
        else:
            res = codes[keep]
        if len(res) == 0:
            res = np.zeros(1, dtype=np.uint8)
        out.append((f"seq{i}", to_bytes(res)))
    return out


def config_c3_surrogate(n: int = 12, L: int = 13500, seed: int = 3001):
    """BASELINE.json configs[2] surrogate at the size SURVEY 8(d) states (HLA-zoo DRB1 itself is not in the
    container): 12 sequences of ~13.5 kb, 3 % substitutions, 0.3 % short indels, two 200-800 bp insertions each"""
    return config_c3_like(n, L, seed)


def config_c4(n: int = 1024, L: int = 2000, clades: int = 16, seed: int = 4001):
    """BASELINE.json configs[3]: 1024 x 2 kb, 16 clades x 64, 8 % between-clade and 1 % within-clade substitutions"""
    root = base_sequence(L, seed)
    per = max(1, n // clades)
    out = []
    for i in range(n):
        c = min(i // per, clades - 1)
        cb = substitute(root, 0.08, seed + 100 + c)
        out.append((f"seq{i}", to_bytes(substitute(cb, 0.01, seed + 1000 + i))))
    return out


def config_c5(n: int = 256, L: int = 50000, seed: int = 5001):
    """BASELINE.json configs[4]: 256 x 50 kb, 2 % substitutions, 0.1 % indels; 25 % of the sequences carry 1-3
    segments of U[1000, 5000] bp reverse-complemented in place, 10 % are reverse-complemented entirely"""
    fam = indel_family_fast(n, L, 0.02, 0.001, seed, max_indel=4)
    out = []
    for i, (name, s) in enumerate(fam):
        r = _r(seed + 50000 + i, np.arange(16))
        if int(r[0] % np.uint64(4)) == 0:                       # 25 %: in-place inversions
            k = 1 + int(r[1] % np.uint64(3))
            for j in range(k):
                ln = 1000 + int(r[2 + 2 * j] % np.uint64(4001))
                st = int(r[3 + 2 * j] % np.uint64(max(1, len(s) - ln)))
                s = invert_segment(s, st, ln)
        if int(r[10] % np.uint64(10)) == 0:                     # 10 %: whole sequence reverse-complemented
            s = reverse_complement(s)
        out.append((name, s))
    return out
