"""ORACLE (second model) — TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.

A second, independently written model of the GeneFuseRust ``Indexer`` hot path,
implemented from the prose specification in SURVEY.md Appendix A (not by
transliterating oracle/indexer_oracle.cc): dict/Counter based, k-mers always
computed from scratch, votes ranked with a sort, mask built as a windowed max.
Its only purpose is to cross-check the C++ restatement: two independent
restatements agreeing on the fixtures is the strongest pin available while the
Rust reference cannot be run (no rustc/cargo in this pipeline).

Pure-Python loops: small cases only.  Only tests/ may import this module.

Reference lines restated (relative to /root/reference):
  src/core/indexer.rs:122-250 (make_index, index_contig, fill_bloom_filter)
  src/core/indexer.rs:252-538 (map_read), :616-679 (segment_mask)
  src/core/indexer.rs:690-732, :789-913 (helpers), src/core/sequence.rs:22-60
"""
from __future__ import annotations

from collections import Counter, defaultdict
from typing import Dict, List, Optional, Sequence, Tuple

K = 16
CODE = {"A": 0, "T": 1, "C": 2, "G": 3}  # indexer.rs:825-841
COMP = {"A": "T", "a": "T", "T": "A", "t": "A", "C": "G", "c": "G", "G": "C", "g": "C"}

HIGH = "HIGH"

Site = Tuple[int, int]  # (contig, position)
Match = Tuple[int, int, int, int]  # (seq_start, seq_end, contig, position)


def kmer_at(s: str, i: int) -> Optional[int]:
    """Appendix A: 2-bit pack of s[i:i+16], first base most significant; None if invalid."""
    v = 0
    for ch in s[i:i + K]:
        c = CODE.get(ch)
        if c is None:
            return None
        v = (v << 2) | c
    return v


def revcomp(s: str) -> str:
    """sequence.rs:22-60 — anything outside ACGTacgt becomes N; output upper case."""
    return "".join(COMP.get(ch, "N") for ch in reversed(s))


def key64(contig: int, diag: int) -> int:
    """indexer.rs:698-706: contig in the high word, position's 32 bit pattern in the low."""
    return (contig << 32) | (diag & 0xFFFFFFFF)


def unkey64(v: int) -> Site:
    """indexer.rs:709-714."""
    c = (v >> 32) & 0xFFFF
    if c >= 0x8000:
        c -= 0x10000
    p = v & 0xFFFFFFFF
    if p >= 0x80000000:
        p -= 1 << 32
    return c, p


class IndexModel:
    """Appendix A.1.  ``genes[c]`` is the raw gene slice or None (chromosome missing)."""

    def __init__(self, genes: Sequence[Optional[str]]):
        occ: Dict[int, List[Site]] = defaultdict(list)
        self.fusion_seq: List[str] = []
        for c, raw in enumerate(genes):
            if raw is None:
                self.fusion_seq.append("")
                continue
            s = raw.upper()
            n = len(s)
            self.fusion_seq.append(s)
            # forward strand: windows 0 .. n-17 (the last window n-16 is not indexed)
            for i in range(0, n - K):
                k = kmer_at(s, i)
                if k is not None:
                    occ[k].append((c, i))
            rc = revcomp(s)
            for i in range(0, n - K):
                k = kmer_at(rc, i)
                if k is not None:
                    occ[k].append((c, i + 1 - n))
        # count 1 -> unique, 2..5 -> all kept, >= 6 -> HIGH (no sites)
        self.table: Dict[int, object] = {}
        for k, sites in occ.items():
            self.table[k] = HIGH if len(sites) >= 6 else list(sites)

    def sites(self, k: Optional[int]) -> List[Site]:
        if k is None:
            return []
        v = self.table.get(k)
        if v is None or v is HIGH:
            return []
        return v  # type: ignore[return-value]

    # Appendix A.2
    def map_read(self, read: str) -> List[Match]:
        L = len(read)
        votes: Counter = Counter()
        for i in range(0, L - K + 1, 2):
            for (c, p) in self.sites(kmer_at(read, i)):
                votes[key64(c, p - i)] += 1
        votes.pop(0, None)  # key 0 is invisible
        ranked = sorted(votes.items(), key=lambda kv: (-kv[1], kv[0]))
        gp1, count1 = ranked[0] if len(ranked) > 0 else (0, 0)
        gp2, count2 = ranked[1] if len(ranked) > 1 else (0, 0)
        if 2 * count1 < 40 or 2 * count2 < 20:
            return []

        wclass = [0] * max(0, L - K + 1)
        for i in range(0, L - K + 1):
            best = 0
            for (c, p) in self.sites(kmer_at(read, i)):
                d = key64(c, p - i)
                if abs(d - gp1) <= 1:
                    f = 3
                elif abs(d - gp2) <= 1:
                    f = 2
                elif d == 0:
                    f = 1
                else:
                    f = 0
                best = max(best, f)
            wclass[i] = best
        mask = [0] * L
        for j in range(L):
            lo, hi = max(0, j - K + 1), min(j, L - K)
            if lo <= hi:
                mask[j] = max(wclass[lo:hi + 1])
        if sum(1 for m in mask if m <= 1) > 10:
            return []
        return segment_mask(mask, unkey64(gp1), unkey64(gp2))


def run_end(mask: Sequence[int], s: int, target: int) -> int:
    """Appendix A.3 inner scan: inclusive end of the run that starts at s."""
    L = len(mask)
    end, g = s + 1, 0
    while g < 10 and end + g < L:
        m = mask[end + g]
        if m > target:
            break
        if m == target:
            end += g + 1
            g = 0
        else:
            g += 1
    return end - 1


def segment_mask(mask: Sequence[int], gp1: Site, gp2: Site) -> List[Match]:
    """Appendix A.3: every start s <= L-2 with mask[s]==target is tried; the first
    longest run wins; it is emitted when end-start > 20."""
    L = len(mask)
    out: List[Match] = []
    for target, gp in ((3, gp1), (2, gp2)):
        best_s, best_e = -1, -1
        for s in range(0, L - 1):
            if mask[s] != target:
                continue
            e = run_end(mask, s, target)
            if e - s > best_e - best_s:
                best_s, best_e = s, e
        if best_e - best_s > 20:
            out.append((best_s, best_e, gp[0], gp[1]))
    return out


def in_required_direction(m: Sequence[Match], reversed_flags: Sequence[bool]) -> bool:
    """indexer.rs:541-608 (with the left-vs-left comparison that is always false)."""
    if len(m) < 2:
        return False
    left, right = (m[0], m[1]) if m[0][0] <= m[1][0] else (m[1], m[0])
    if left[3] > 0 and right[3] > 0:
        return True
    if left[3] < 0 and right[3] < 0:
        return False
    lrev, rrev = bool(reversed_flags[left[2]]), bool(reversed_flags[right[2]])
    if lrev and not rrev:
        return False
    if not lrev and rrev:
        return True
    return left[2] < right[2]


# ---- FusionMapper::map_read tail, written from the prose of Appendix A.4
# and fusion_mapper.rs:196-251; the edit distance is the textbook Levenshtein DP on purpose
# (the reference's bit-parallel routine must equal it).

def levenshtein(a: str, b: str) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def calc_ed(fusion_seq: Sequence[str], seq: str, contig: int, start: int, end: int) -> int:
    if (start >= 0 and end <= 0) or (start <= 0 and end >= 0):
        return -1
    fs = fusion_seq[contig]
    if abs(start) >= len(fs) or abs(end) >= len(fs):
        return -2
    if start < 0:
        seq = revcomp(seq)
        start, end = -end, -start
    return levenshtein(seq, fs[start:end + 1])


def fusion_map_read(fusion_seq: Sequence[str], reversed_flags: Sequence[bool], read: str, mapping: Sequence[Match]):
    """(status, fields): 0 = None/unmapable, 1 = None/mapable, 2 = match."""
    if len(mapping) < 2:
        return 0, None
    if not in_required_direction(mapping, reversed_flags):
        return 1, None
    left, right = sorted(mapping[:2], key=lambda m: m[0]) if mapping[0][0] != mapping[1][0] else (mapping[0], mapping[1])
    brk = (left[1] + right[0]) // 2 if (left[1] + right[0]) >= 0 else -((-(left[1] + right[0])) // 2)
    lpos, rpos = left[3] + brk, right[3] + brk + 1
    left_len, right_len = brk + 1, len(read) - (brk + 1)
    return 2, {
        "read_break": brk, "gap": right[0] - left[1] - 1,
        "left_contig": left[2], "left_position": lpos, "right_contig": right[2], "right_position": rpos,
        "left_distance": calc_ed(fusion_seq, read[:left_len], left[2], lpos - left_len + 1, lpos),
        "right_distance": calc_ed(fusion_seq, read[brk + 1:], right[2], rpos, rpos + right_len - 1),
    }


# ---- SequenceReadPair::fast_merge (read.rs:313-440), from its description:
# smallest overlap >= 30 whose mismatches are all "one side >= Q30, the other <= Q15" and fewer
# than three; merged = left prefix + reverse-complemented right, overlap corrected.

def fast_merge(l_seq: str, l_qual: str, r_seq: str, r_qual: str):
    s2, q2 = revcomp(r_seq), r_qual[::-1]
    n1, n2 = len(l_seq), len(s2)

    def lowq(a: str, b: str) -> bool:
        return (a >= "?" and b <= "0") or (a <= "0" and b >= "?")

    for olen in range(30, min(n1, n2) + 1):
        off = n1 - olen
        mism = [i for i in range(olen) if l_seq[off + i] != s2[i]]
        if all(lowq(l_qual[off + i], q2[i]) for i in mism) and len(mism) < 3:
            seq = list(l_seq[:off] + s2)
            qual = list(l_qual[:off] + q2)
            for i in range(olen):
                a, b = l_qual[off + i], q2[i]
                if l_seq[off + i] != s2[i]:
                    if a >= "?" and b <= "0":
                        seq[off + i], qual[off + i] = l_seq[off + i], a
                    else:
                        seq[off + i], qual[off + i] = s2[i], b
                else:
                    qual[off + i] = chr(min(ord(a) + ord(b) - 33, ord("Z")))
            return "".join(seq), "".join(qual), len(mism)
    return None


# ---- FastqReader::read (fastq_reader.rs:75-147), from its description: four
# lines per record, one trailing newline stripped from each, a last line without newline
# counts, the first incomplete record ends the file.

def fastq_records(text: bytes):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # nothing after the final newline: not a line
    return [tuple(lines[4 * i:4 * i + 4]) for i in range(len(lines) // 4)]


# ---- FusionMapper::filter_matches minus remove_alignables, and the order of sort_matches
# (fusion_mapper.rs:276-384, read_match.rs:203-228), from their description.

def low_complexity(s: str) -> bool:
    return len(s) < 20 or sum(1 for a, b in zip(s, s[1:]) if a != b) < 7


def match_filter(seq: str, read_break: int, left, right, left_distance: int, right_distance: int,
                 deletion_threshold: int = 50) -> int:
    """left/right = (contig, position).  0 kept, 1 complexity, 2 distance, 3 indel."""
    if low_complexity(seq[:read_break + 1]) or low_complexity(seq[read_break + 1:]):
        return 1
    if left_distance + right_distance >= 5:
        return 2
    if left[0] == right[0] and abs(left[1] - right[1]) < deletion_threshold:
        return 3
    return 0


def match_sort(matches):
    """matches: (read_break, read_len, name bytes, payload).  Descending by the ReadMatch
    order = read_break descending, read length ascending, name descending."""
    out = sorted(matches, key=lambda m: m[2], reverse=True)          # name descending (stable)
    out = sorted(out, key=lambda m: m[1])                             # length ascending
    return sorted(out, key=lambda m: m[0], reverse=True)              # read_break descending


# ---- the inputs of make_index: the fusion CSV (fusion.rs:22-86, gene.rs:43-105, :131-212) and the
# FASTA reference (fasta_reader.rs:117-200), written from their description.  Pinned by the
# reference's own tests: fusion.rs:112-150 (four pos2str strings on testdata/fusions.csv) and
# fasta_reader.rs:233-258 (the two contigs of testdata/tinyref.fa.gz).

import re as _re

_GENE_LINE = _re.compile(r"^>(?P<name>[^,]*),(?P<chr>[^:,]*):(?P<a>[^-,:]*)-(?P<b>[^-,:]*)")


def csv_genes(text: str):
    """[(name, chr, start, end, [(id, start, end)...], reversed)]"""
    genes, cur = [], None

    def flush():
        if cur is not None and cur[0] != "invalid" and cur[2] != 0 and cur[3] != 0:
            ex = cur[4]
            genes.append((cur[0], cur[1], cur[2], cur[3], list(ex), len(ex) > 1 and ex[0][1] > ex[1][1]))

    for raw in text.splitlines():
        line = raw.strip()
        f = line.split(",")
        if len(f) < 2 or f[0].startswith("#"):
            continue
        if f[0].startswith(">"):
            flush()
            m = _GENE_LINE.match(line)
            cur = [m.group("name").strip(), m.group("chr").strip(), int(m.group("a")), int(m.group("b")), []] if m \
                else ["invalid", "invalid", 0, 0, []]
        elif len(f) >= 3 and cur is not None:
            cur[4].append((int(f[0]), int(f[1]), int(f[2])))
    flush()
    return genes


def gene_pos2str(gene, pos: int) -> str:
    name, chr_, start, _end, exons, rev = gene
    pp = abs(pos) + start
    label = ""
    for i, (eid, es, ee) in enumerate(exons):
        if es <= pp <= ee:
            label = "exon:%d|" % eid
            break
        if i > 0:
            ps, pe = exons[i - 1][1], exons[i - 1][2]
            if (rev and ee < pp < ps) or (not rev and pe < pp < es):
                label = "intron:%d|" % (eid - 1)
                break
    return "%s:%s%s%s:%d" % (name, label, "+" if pos >= 0 else "-", chr_, pp)


def fasta_contigs(data: bytes, upper: bool = True):
    out = {}
    i = data.find(b">")
    if i < 0:
        return out
    pos = i + 1
    n = len(data)
    while pos < n:
        j = data.find(b">", pos)
        rec = data[pos:] if j < 0 else data[pos:j]
        k = 0
        while k < len(rec) and rec[k] not in (10, 32):
            k += 1
        seq = bytes(b for b in rec[k + 1:] if 65 <= b <= 90 or 97 <= b <= 122 or b in (45, 42))
        out[rec[:k].decode("latin-1")] = seq.upper() if upper else seq
        if j < 0:
            break
        pos = j + 1
    return out


# ---- clustering of the sorted matches into fusion candidates, written from
# the description of fusion_mapper.rs:399-486 / fusion_result.rs:60-510 (first fit within 3 bp,
# fusion point = first gap-free match else the truncated mean, reference windows either side of
# the point, break shifted by -3..3 to the smallest 20+20-base edit distance, qualification).
# Textbook Levenshtein and plain slicing on purpose.  A match is a dict with keys
# seq, brk, left=(contig,pos), right=(contig,pos), gap, ld, rd.

def _window(gene_seq: str, a: int, b: int) -> str:
    """Bases a..b (inclusive, gene coordinates; negative = the reverse strand) or "" when the
    window touches 0, changes sign or leaves the gene."""
    if a == 0 or b == 0 or (a > 0) != (b > 0) or abs(a) >= len(gene_seq) or abs(b) >= len(gene_seq):
        return ""
    n = abs(b - a) + 1
    return gene_seq[a:a + n] if a > 0 else revcomp(gene_seq[-b:-b + n])


def _continues(s1: str, s2: str) -> bool:
    """Do s1 and s2 read the same within a slide of 6 and a tenth of mismatches?"""
    for off in range(-6, 7):
        a0, b0, n = max(off, 0), max(-off, 0), len(s1) - abs(off)
        if a0 >= len(s1) or b0 >= len(s2):
            return True
        if b0 + n > len(s2) or n < 0:
            raise IndexError("window beyond the sequence")
        if levenshtein(s1[a0:a0 + n], s2[b0:b0 + n]) <= int(n / 10):
            return True
    return False


def _runs(s: str) -> int:
    return sum(1 for a, b in zip(s, s[1:]) if a != b)


def cluster_model(groups, genes, fusion_seq, unique_requirement: int = 2, output_deletions: bool = False,
                  output_untranslated: bool = False):
    """groups: lists of matches (each list sorted already); genes: csv_genes() tuples.
    Returns the qualified candidates as dicts, most unique reads first."""
    found = []
    for group in groups:
        clusters = []
        for m in group:
            for c in clusters:
                if any(m["left"][0] == x["left"][0] and m["right"][0] == x["right"][0] and
                       abs(m["left"][1] - x["left"][1]) <= 3 and abs(m["right"][1] - x["right"][1]) <= 3 for x in c):
                    c.append(m)
                    break
            else:
                clusters.append([m])
        for c in clusters:
            exact = [m for m in c if m["gap"] == 0]
            if exact:
                left, right = exact[0]["left"], exact[0]["right"]
            else:
                left = (c[0]["left"][0], int(sum(m["left"][1] for m in c) / len(c)))
                right = (c[0]["right"][0], int(sum(m["right"][1] for m in c) / len(c)))
            ll = max(m["brk"] + 1 for m in c)
            lr = max(len(m["seq"]) - m["brk"] - 1 for m in c)
            gl, gr = fusion_seq[left[0]], fusion_seq[right[0]]
            lref, rref = _window(gl, left[1] - ll + 1, left[1]), _window(gr, right[1], right[1] + lr - 1)
            lext, rext = _window(gl, left[1], left[1] + lr - 1), _window(gr, right[1] - ll + 1, right[1])
            reads = []
            for m in c:
                best = None
                for s in range(-3, 4):
                    k = m["brk"] + s + 1
                    a, b = m["seq"][:k], m["seq"][k:]
                    na, nb = min(len(a), len(lref), 20), min(len(b), len(rref), 20)
                    near = levenshtein(a[len(a) - na:], lref[len(lref) - na:]) + levenshtein(b[:nb], rref[:nb])
                    if best is None or near < best[0]:
                        na, nb = min(len(a), len(lref)), min(len(b), len(rref))
                        best = (near, s, levenshtein(a[len(a) - na:], lref[len(lref) - na:]),
                                levenshtein(b[:nb], rref[:nb]))
                _, s, ld, rd = best
                reads.append(dict(m, brk=m["brk"] + s, left=(m["left"][0], m["left"][1] + s),
                                  right=(m["right"][0], m["right"][1] + s), ld=ld, rd=rd))
            unique = 1 + sum(1 for p, q in zip(reads, reads[1:])
                             if p["brk"] != q["brk"] or len(p["seq"]) != len(q["seq"]))
            deletion = left[0] == right[0] and left[1] != 0 and right[1] != 0 and (left[1] > 0) == (right[1] > 0)
            if unique < unique_requirement:
                continue
            if _continues(lext, rref) or _continues(lref, rext):
                continue
            if len(lref) <= 30 or len(rref) <= 30 or _runs(lref[-10:]) <= 2 or _runs(rref[:10]) <= 2:
                continue
            if deletion and not output_deletions:
                continue
            lfwd = (left[1] < 0) if genes[left[0]][5] else (left[1] > 0)
            rfwd = (right[1] < 0) if genes[right[0]][5] else (right[1] > 0)
            if lfwd != rfwd and not output_untranslated:
                continue
            title = "%s%s___%s  (total: %d, unique:%d)" % (
                "Deletion: " if deletion else "Fusion: ", gene_pos2str(genes[left[0]], left[1]),
                gene_pos2str(genes[right[0]], right[1]), len(reads), unique)
            found.append(dict(title=title, left=left, right=right, unique=unique, reads=reads, left_ref=lref,
                              right_ref=rref, left_ref_ext=lext, right_ref_ext=rext))
    return sorted(found, key=lambda f: (-f["unique"], -len(f["reads"])))


# ---- the reference's Matcher as it is (matcher.rs), loop by loop — the
# `break` that leaves make_kmer at the first base, the roll with the base at the window start,
# the votes shifted by the site's list index, the inverted contains_key — for the vectorised
# restatement in genefuserust_amd/matcher.py to be compared with.

class ModelPanic(Exception):
    pass


_M_CODE = {"A": 0, "T": 1, "C": 2, "G": 3}


def _m_make_kmer(seq: str, pos: int):
    kmer = 0
    for i, base in enumerate(seq[pos:pos + 16]):
        if base not in _M_CODE:
            return 0, False
        kmer += _M_CODE[base]
        break  # (the reference's `break` leaves the for loop, not a switch)
    return kmer, True


def matcher_model_build(contigs, read_seqs):
    """-> (bloom bytes as a dict byte -> bits, index: key -> [(ctg, pos)])"""
    bloom = {}
    for s in read_seqs:
        for t in (s, revcomp(s)):
            if len(t) - 16 + 1 < 0:
                raise ModelPanic("range")
            for i in range(len(t) - 16 + 1):
                kmer, valid = _m_make_kmer(t, i)
                if valid:
                    bloom[kmer >> 3] = bloom.get(kmer >> 3, 0) | (1 << (kmer & 7))
    index = {}
    for ctg, (name, seq) in enumerate(sorted(contigs.items())):
        seq = seq.upper()
        if len(seq) - 16 < 0:
            raise ModelPanic("slice")
        kmer, valid = 0, False
        for i in range(len(seq) - 16):
            base = seq[i]
            if valid:
                if base not in _M_CODE:
                    valid = False
                    continue
                kmer = ((kmer << 2) | _M_CODE[base]) & 0xFFFFFFFF
            else:
                kmer, valid = _m_make_kmer(seq, i)
                if not valid:
                    continue
            if not (bloom.get(kmer >> 3, 0) >> (kmer & 7)) & 1:
                continue
            index.setdefault(kmer, []).append((ctg, i))
    return bloom, index


def matcher_model_map(index, seq: str):
    """None, or raises ModelPanic where the reference panics."""
    n = len(seq)
    if n - 16 + 1 < 0:
        raise ModelPanic("range")
    stat = {0: 0}
    all_kmer, kmer_valid, skipped = [0] * n, [False] * n, [False] * n
    for i in range(n - 16 + 1):
        kmer, valid = _m_make_kmer(seq, i)
        kmer_valid[i] = valid
        if not valid:
            continue
        all_kmer[i] = kmer
        if kmer not in index:
            stat[0] += 1
            continue
        if len(index[kmer]) > 50:
            skipped[i] = True
            continue
        for k, (ctg, pos) in enumerate(index[kmer]):
            g = (ctg << 32) + (pos - k)
            stat[g] = stat.get(g, 0) + 1
    topgp, topcount = [0] * 5, [0] * 5
    for g, cnt in stat.items():
        if g == 0 or cnt <= topcount[4]:
            continue
        topgp[4], topcount[4] = g, cnt
        for t in range(3, -1, -1):
            if cnt > topcount[t]:
                topcount[t + 1], topgp[t + 1] = topcount[t], topgp[t]
                topcount[t], topgp[t] = cnt, g
    for t in range(5):
        if topcount[t] == 0:
            break
        mask = [0] * n
        for i in range(n - 16 + 1):
            if not kmer_valid[i] or all_kmer[i] in index:
                continue
            if not skipped[i]:
                raise ModelPanic("unwrap on None")  # self.m_kmer_positions.get(&kmer).unwrap()
            raise ModelPanic("unwrap on None")       # is_consistent's get(..).unwrap()
        if sum(1 for m in mask if m == 0) < 10:
            return ("match", topgp[t])
    return None


def matcher_model_do_match(index, seq: str):
    a = matcher_model_map(index, seq)
    b = matcher_model_map(index, revcomp(seq))
    return a if a is not None else b
