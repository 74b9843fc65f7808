"""SURVEY.md §8(f)-4 — what the reference does with the sorted match list: first-fit clustering
into fusion candidates, the break-point refinement, qualification and the text / JSON result
(``FusionMapper::cluster_matches``, src/core/fusion_mapper.rs:399-486 and :544-556;
``FusionResult``, src/core/fusion_result.rs:24-511 and :761-798; ``ReadMatch::print``,
src/core/read_match.rs:153-186; ``JsonReporter::run``, src/core/json_reporter.rs:34-123).

Host logic on a handful of reads per run (the reference keeps it on the CPU as well); the only
compute is the edit distance, which is ``gf_edit_distance`` of libgfmatch.so.  Where the
reference would panic (``subchars`` beyond a string's end) this raises ``IndexError``.
The HTML report (protein diagrams) is not here.
"""
from __future__ import annotations

from dataclasses import dataclass, field, replace
from typing import List, Optional, Sequence

from .fusion_mapper import ReadMatch, edit_distance, reverse_complement
from .indexer import Fusion, Gene, GenePos


@dataclass
class Settings:
    """src/aux/global_settings.rs:17-24 (defaults)."""
    unique_requirement: int = 2
    deletion_threshold: int = 50
    output_deletions: bool = False
    output_untranslated: bool = False


def _take(s: bytes, skip: int, n: int) -> bytes:
    """``s.chars().skip(skip as usize).take(n as usize)``: a negative i32 cast to usize is huge."""
    if skip < 0:
        return b""
    return s[skip:] if n < 0 else s[skip:skip + n]


def _subchars(s: bytes, pos: int, n: int) -> bytes:
    """utils/mod.rs:36-45: ``self.get(pos..pos+n).unwrap()`` — out of range panics."""
    if pos < 0 or n < 0 or pos + n > len(s):
        raise IndexError("subchars(%d, %d) of a string of %d" % (pos, n, len(s)))
    return s[pos:pos + n]


def dis_connected_count(s: bytes) -> int:
    """utils/mod.rs:48-56."""
    if len(s) == 0:
        raise IndexError("dis_connected_count of an empty string")
    return sum(1 for i in range(len(s) - 1) if s[i] != s[i + 1])


def get_ref_seq(ref: bytes, start: int, end: int) -> bytes:
    """fusion_result.rs:770-798: both ends on one strand and inside the gene, else ""."""
    if (start >= 0 and end <= 0) or (start <= 0 and end >= 0):
        return b""
    if abs(start) >= len(ref) or abs(end) >= len(ref):
        return b""
    n = abs(end - start) + 1
    if start < 0:
        return reverse_complement(ref[-end:-end + n])
    return ref[start:start + n]


def _trunc_div(a: int, b: int) -> int:
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


@dataclass
class FusionResult:
    """fusion_result.rs:24-58."""
    m_left_gp: GenePos = GenePos(0, 0)
    m_right_gp: GenePos = GenePos(0, 0)
    m_matches: List[ReadMatch] = field(default_factory=list)
    m_unique: int = 0
    m_title: str = ""
    m_left_ref: bytes = b""
    m_right_ref: bytes = b""
    m_left_ref_ext: bytes = b""
    m_right_ref_ext: bytes = b""
    m_left_pos: str = ""
    m_right_pos: str = ""
    m_left_gene: Gene = field(default_factory=Gene)
    m_right_gene: Gene = field(default_factory=Gene)
    m_left_is_exon: bool = False
    m_right_is_exon: bool = False
    m_left_exon_or_intron_id: int = -1
    m_right_exon_or_intron_id: int = -1

    # -- clustering ------------------------------------------------------------------------
    @staticmethod
    def support_same(m1: ReadMatch, m2: ReadMatch) -> bool:  # :426-445
        return (abs(m1.m_left_gp.position - m2.m_left_gp.position) <= 3 and
                abs(m1.m_right_gp.position - m2.m_right_gp.position) <= 3 and
                m1.m_left_gp.contig == m2.m_left_gp.contig and m1.m_right_gp.contig == m2.m_right_gp.contig)

    def support(self, m: ReadMatch) -> bool:  # :416-424
        return any(self.support_same(m, x) for x in self.m_matches)

    def add_match(self, m: ReadMatch) -> None:  # :412-414
        self.m_matches.append(m)

    # -- the fusion point -------------------------------------------------------------------
    def calc_fusion_point(self) -> None:  # :60-86
        if not self.m_matches:
            return
        lt = rt = 0
        for m in self.m_matches:
            if m.m_gap == 0:  # an exact junction wins
                self.m_left_gp, self.m_right_gp = m.m_left_gp, m.m_right_gp
                return
            lt += m.m_left_gp.position
            rt += m.m_right_gp.position
        n = len(self.m_matches)
        self.m_left_gp = GenePos(self.m_matches[0].m_left_gp.contig, _trunc_div(lt, n))
        self.m_right_gp = GenePos(self.m_matches[0].m_right_gp.contig, _trunc_div(rt, n))

    def make_reference(self, ref_l: bytes, ref_r: bytes) -> None:  # :242-297
        longest_left = longest_right = 0
        for m in self.m_matches:
            longest_left = max(longest_left, m.m_read_break + 1)
            longest_right = max(longest_right, len(m.m_read) - (m.m_read_break + 1))
        lp, rp = self.m_left_gp.position, self.m_right_gp.position
        self.m_left_ref = get_ref_seq(ref_l, lp - longest_left + 1, lp)
        self.m_right_ref = get_ref_seq(ref_r, rp, rp + longest_right - 1)
        self.m_left_ref_ext = get_ref_seq(ref_l, lp, lp + longest_right - 1)
        self.m_right_ref_ext = get_ref_seq(ref_r, rp - longest_left + 1, rp)

    def calc_ed(self, m: ReadMatch, shift: int):  # :326-410 -> (total, left_ed, right_ed)
        seq = m.m_read
        left_len = m.m_read_break + shift + 1
        right_len = len(seq) - left_len
        left_seq = _take(seq, 0, left_len)
        right_seq = _take(seq, left_len, right_len)
        # the 20 bases either side of the break decide the shift ...
        lc = min(len(left_seq), len(self.m_left_ref), 20)
        rc = min(len(right_seq), len(self.m_right_ref), 20)
        total = (edit_distance(left_seq[len(left_seq) - lc:], self.m_left_ref[len(self.m_left_ref) - lc:]) +
                 edit_distance(right_seq[:rc], self.m_right_ref[:rc]))
        # ... the whole overlap gives the distances that are reported
        lc = min(left_len, len(self.m_left_ref))
        rc = min(right_len, len(self.m_right_ref))
        left_ed = edit_distance(_take(left_seq, len(left_seq) - lc, lc),
                                _take(self.m_left_ref, len(self.m_left_ref) - lc, lc))
        right_ed = edit_distance(_take(right_seq, 0, rc), _take(self.m_right_ref, 0, rc))
        return total, left_ed, right_ed

    def adjust_fusion_break(self) -> None:  # :299-324
        out = []
        for m in self.m_matches:
            smallest, shift = 0xFFFF, 0
            ld, rd = m.m_left_distance, m.m_right_distance
            for s in range(-3, 4):
                ed, l_ed, r_ed = self.calc_ed(m, s)
                if ed < smallest:
                    smallest, shift, ld, rd = ed, s, l_ed, r_ed
            out.append(replace(m, m_read_break=m.m_read_break + shift,
                               m_left_gp=GenePos(m.m_left_gp.contig, m.m_left_gp.position + shift),
                               m_right_gp=GenePos(m.m_right_gp.contig, m.m_right_gp.position + shift),
                               m_left_distance=ld, m_right_distance=rd))
        self.m_matches = out

    def calc_unique(self) -> None:  # :88-105 (the list is sorted: compare neighbours)
        self.m_unique = 1
        for prev, m in zip(self.m_matches, self.m_matches[1:]):
            if m.m_read_break != prev.m_read_break or len(m.m_read) != len(prev.m_read):
                self.m_unique += 1

    # -- what it is ---------------------------------------------------------------------------
    def is_deletion(self) -> bool:  # :107-118
        l, r = self.m_left_gp, self.m_right_gp
        return l.contig == r.contig and ((l.position > 0 and r.position > 0) or (l.position < 0 and r.position < 0))

    def is_left_protein_forward(self) -> bool:  # :446-452
        p = self.m_left_gp.position
        return p < 0 if self.m_left_gene.is_reversed() else p > 0

    def is_right_protein_forward(self) -> bool:  # :454-460
        p = self.m_right_gp.position
        return p < 0 if self.m_right_gene.is_reversed() else p > 0

    def update_info(self, fusions: Sequence[Fusion]) -> None:  # :196-240
        self.m_left_gene = fusions[self.m_left_gp.contig].m_gene
        self.m_right_gene = fusions[self.m_right_gp.contig].m_gene
        self.m_left_pos = self.m_left_gene.pos2str(self.m_left_gp.position)
        self.m_right_pos = self.m_right_gene.pos2str(self.m_right_gp.position)
        self.m_title = "%s%s___%s  (total: %d, unique:%d)" % (
            "Deletion: " if self.is_deletion() else "Fusion: ", self.m_left_pos, self.m_right_pos,
            len(self.m_matches), self.m_unique)
        e, k = self.m_left_gene.get_exon_intron(self.m_left_gp.position)
        if e is not None:  # (the reference's out-parameters stay as they were otherwise)
            self.m_left_is_exon, self.m_left_exon_or_intron_id = e, k
        e, k = self.m_right_gene.get_exon_intron(self.m_right_gp.position)
        if e is not None:
            self.m_right_is_exon, self.m_right_exon_or_intron_id = e, k

    @staticmethod
    def can_be_matched(s1: bytes, s2: bytes) -> bool:  # :131-161
        n = len(s1)
        for offset in range(-6, 7):
            start1, start2 = max(offset, 0), max(-offset, 0)
            cmplen = n - abs(offset)
            if start1 >= len(s1) or start2 >= len(s2):
                return True
            ed = edit_distance(_subchars(s1, start1, cmplen), _subchars(s2, start2, cmplen))
            if ed <= _trunc_div(cmplen, 10):
                return True
        return False

    def can_be_mapped(self) -> bool:  # :120-129: the two sides continue each other's gene
        return (self.can_be_matched(self.m_left_ref_ext, self.m_right_ref) or
                self.can_be_matched(self.m_left_ref, self.m_right_ref_ext))

    def is_qualified(self, settings: Settings) -> bool:  # :163-194
        if self.m_unique < settings.unique_requirement:
            return False
        if self.can_be_mapped():
            return False
        if len(self.m_left_ref) <= 30 or len(self.m_right_ref) <= 30:
            return False
        if dis_connected_count(_subchars(self.m_left_ref, len(self.m_left_ref) - 10, 10)) <= 2:
            return False
        if dis_connected_count(_subchars(self.m_right_ref, 0, 10)) <= 2:
            return False
        return True

    # -- output -------------------------------------------------------------------------------
    def text(self) -> str:
        """``FusionResult::print`` (:761-767) over ``ReadMatch::print`` (read_match.rs:153-186)."""
        out = ["\n#%s\n" % self.m_title]
        for i, m in enumerate(self.m_matches):
            name = _subchars(m.m_name, 1, len(m.m_name) - 1).decode("latin-1")
            b = m.m_read_break + 1
            out.append(">%d, break:%d, diff:(%d %d), read direction: %s, name: %s\n%s %s\n" % (
                i + 1, b, m.m_left_distance, m.m_right_distance,
                "reversed complement" if m.m_reversed else "original direction", name,
                _subchars(m.m_read, 0, b).decode("latin-1"),
                _subchars(m.m_read, b, len(m.m_read) - b).decode("latin-1")))
        return "".join(out)


def match_group(m: ReadMatch, n_fusions: int) -> int:
    """FusionMapper::add_match (fusion_mapper.rs:253-275): the list a match is kept in."""
    return n_fusions * m.m_right_gp.contig + m.m_left_gp.contig


def cluster_matches(groups: Sequence[Sequence[ReadMatch]], fusions: Sequence[Fusion], fusion_seq: Sequence[str],
                    settings: Optional[Settings] = None) -> List[FusionResult]:
    """fusion_mapper.rs:399-486 + sort_fusion_results (:544-556).  ``groups`` = the reference's
    ``fusion_matches``: one sorted list per (right, left) gene pair, in ``match_group`` order.
    ``fusion_seq`` = ``Indexer.m_fusion_seq``."""
    settings = settings or Settings()
    results: List[FusionResult] = []
    for fm in groups:
        frs: List[FusionResult] = []
        for rm in fm:  # first fit
            for fr in frs:
                if fr.support(rm):
                    fr.add_match(rm)
                    break
            else:
                fr = FusionResult()
                fr.add_match(rm)
                frs.append(fr)
        for fr in frs:
            fr.calc_fusion_point()
            fr.make_reference(fusion_seq[fr.m_left_gp.contig].encode("latin-1"),
                              fusion_seq[fr.m_right_gp.contig].encode("latin-1"))
            fr.adjust_fusion_break()
            fr.calc_unique()
            fr.update_info(fusions)
            if not fr.is_qualified(settings):
                continue
            if not settings.output_deletions and fr.is_deletion():
                continue
            if fr.is_left_protein_forward() != fr.is_right_protein_forward() and not settings.output_untranslated:
                continue
            results.append(fr)
    results.sort(key=lambda fr: (-fr.m_unique, -len(fr.m_matches)))  # stable, like sort_by
    return results


def group_and_sort(matches: Sequence[ReadMatch], n_fusions: int) -> List[List[ReadMatch]]:
    """The reference's ``fusion_matches`` after ``sort_matches`` (fusion_mapper.rs:379-384): the
    non-empty lists only, in index order (empty ones produce nothing in ``cluster_matches``)."""
    from .fusion_mapper import FusionMapper
    by: dict = {}
    for m in matches:
        by.setdefault(match_group(m, n_fusions), []).append(m)
    return [FusionMapper.sort_matches(by[k]) for k in sorted(by)]


def report_text(results: Sequence[FusionResult]) -> str:
    return "".join(fr.text() for fr in results)


def report_json(results: Sequence[FusionResult], command: str, version: str, time: str,
                settings: Optional[Settings] = None) -> str:
    """The bytes ``JsonReporter::run`` writes (json_reporter.rs:34-123), tabs and trailing blanks
    included; ``time`` is whatever the caller wants on the reference's ``Local::now()`` line."""
    settings = settings or Settings()
    f: List[str] = ["{\n", "\t\"command\":\"%s\",\n" % command, "\t\"version\":\"%s\",\n" % version,
                    "\t\"time\":\"%s\",\n" % time, "\t\"fusions\":{"]
    first = True
    for fr in results:
        if not settings.output_deletions and fr.is_deletion():
            continue
        if fr.is_left_protein_forward() != fr.is_right_protein_forward() and not settings.output_untranslated:
            continue
        f.append("\n" if first else ",\n")
        first = False
        f.append("\t\t\"%s\":{\n" % fr.m_title)
        for side, gene, gp, ref, ext, pos, is_exon, eid, fwd in (
                ("left", fr.m_left_gene, fr.m_left_gp, fr.m_left_ref, fr.m_left_ref_ext, fr.m_left_pos,
                 fr.m_left_is_exon, fr.m_left_exon_or_intron_id, fr.is_left_protein_forward()),
                ("right", fr.m_right_gene, fr.m_right_gp, fr.m_right_ref, fr.m_right_ref_ext, fr.m_right_pos,
                 fr.m_right_is_exon, fr.m_right_exon_or_intron_id, fr.is_right_protein_forward())):
            f.append("\t\t\t\"%s\":{\n" % side)
            f.append("\t\t\t\t\"gene_name\":\"%s\",\n" % gene.m_name)
            f.append("\t\t\t\t\"gene_chr\":\"%s\",\n" % gene.m_chr)
            f.append("\t\t\t\t\"position\":%d,\n" % gene.gene_pos_2_chr_pos(gp.position))
            f.append("\t\t\t\t\"reference\":\"%s\",\n" % ref.decode("latin-1"))
            f.append("\t\t\t\t\"ref_ext\":\"%s\",\n" % ext.decode("latin-1"))
            f.append("\t\t\t\t\"pos_str\":\"%s\",\n" % pos)
            f.append("\t\t\t\t\"exon_or_intron\":\"%s\",\n" % ("exon" if is_exon else "intron"))
            f.append("\t\t\t\t\"exon_or_intron_id\":%d,\n" % eid)
            f.append("\t\t\t\t\"strand\":\"%s\"\n" % ("forward" if fwd else "reversed"))
            f.append("\t\t\t}, \n")
        f.append("\t\t\t\"unique\":%d,\n" % fr.m_unique)
        f.append("\t\t\t\"reads\":[\n")
        for k, m in enumerate(fr.m_matches):
            f.append("\t\t\t\t{\n")
            f.append("\t\t\t\t\t\"break\":%d,\n" % m.m_read_break)
            f.append("\t\t\t\t\t\"strand\":\"%s\",\n" % ("reversed" if m.m_reversed else "forward"))
            f.append("\t\t\t\t\t\"seq\":\"%s\",\n" % m.m_read.decode("latin-1"))
            f.append("\t\t\t\t\t\"qual\":\"%s\"\n" % m.m_quality.decode("latin-1"))
            f.append("\t\t\t\t}" + ("," if k != len(fr.m_matches) - 1 else "") + "\n")
        f.append("\t\t\t]\n")
        f.append("\t\t}")
    f.append("\n\t}\n}\n\n")
    return "".join(f)
