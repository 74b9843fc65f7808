"""The front half of ``PairEndScanner::scan`` (src/core/pescanner.rs:78-518) from the reference's
own file formats: FASTA + fusion CSV -> index; R1/R2 FASTQ -> records -> per-pair policy
(merge, map, reverse-complement retries) -> ``ReadMatch`` lists, filtered and sorted the way
``FusionMapper::filter_matches`` (without ``remove_alignables``) and ``sort_matches`` do.

Glue only: every step is one of the mirrors in this package, and all compute goes through
libgfmatch.so.  ``scan_pair_end_report`` adds the back half (clustering, qualification, text and
JSON results: SURVEY.md §8(f)-4, ``fusion_result.py``).  ``remove_alignables`` (§8(f)-3, ``matcher.py``:
the reference's ``Matcher`` as it is — it removes nothing on a genome, and panics on small
references) is applied only on request.
"""
from __future__ import annotations

from typing import List, Tuple

from .fastq import FastqReader, FastqReaderPair, record_lines
from .fusion_mapper import FusionMapper, ReadMatch
from .fusion_result import FusionResult, Settings, cluster_matches, group_and_sort, report_json, report_text
from .indexer import FastaReader, Fusion, Indexer
from .read_pair import finish_pair_hits, scan_pairs_device


def scan_pair_end_files(ref_file: str, fusion_csv: str, read1_file: str, read2_file: str, device: int = -1,
                        deletion_threshold: int = 50, _keep: dict = None,
                        remove_alignables: bool = False) -> Tuple[List[ReadMatch], dict]:
    """Returns (matches kept, in ``sort_matches`` order; counters).  Each match carries the name
    of the read it was found on (for a merged read the R1 name with the " merged_diff_N" suffix of
    read.rs:372)."""
    ref = FastaReader(ref_file, True)
    ref.read_all()
    fusions = Fusion.parse_csv(fusion_csv)
    ix = Indexer(ref.m_all_contigs, fusions, device)
    ix.make_index()
    try:
        (l, ltext), (r, rtext) = FastqReaderPair.from_paths(read1_file, read2_file).read_all_device(ix)
        mapper = FusionMapper(ix)
        # the records never leave HBM between the FASTQ cut and the hit list: one device call for the pack
        n = l.n_records
        max_len = max(l.max_read_len(), r.max_read_len(), 1)
        caps = dict(hits_cap=max(1024, n // 8), bytes_cap=max(1024, n // 8) * 2 * max_len)
        rec, hb, hq, tot = scan_pairs_device(ix, l.bases, l.quals, l.offsets, r.bases, r.quals, r.offsets, max_len,
                                             **caps).download()
        if tot["overflow"]:   # unusually many matches or retries: once more with room for everything
            caps = dict(hits_cap=3 * n, bytes_cap=2 * int(l.bases.numel() + r.bases.numel()) + 64, retry_cap=3 * n)
            rec, hb, hq, tot = scan_pairs_device(ix, l.bases, l.quals, l.offsets, r.bases, r.quals, r.offsets, max_len,
                                                 **caps).download()
        found: List[ReadMatch] = []
        for i, m in finish_pair_hits(mapper, rec, hb, hq):
            # a match on R2 (or its reverse complement) carries R2's name; anything else R1's
            m.m_name = record_lines(r, rtext, i)[0] if m.m_source == "r2" else record_lines(l, ltext, i)[0]
            if m.m_source == "merged":
                m.m_name += b" merged_diff_%d" % m.m_merge_diff
            found.append(m)
        kept, removed = mapper.filter_matches(found, deletion_threshold)
        if remove_alignables:  # (the reference always does: a whole-genome scan that removes nothing)
            kept, removed["alignables"] = mapper.remove_alignables(kept)
        counters = {"pairs": l.n_records, "matches_before_filtering": len(found), "merged_pairs": tot["merged_pairs"],
                    "retried_reads": tot["retried_reads"], **removed}
        if _keep is not None:
            _keep.update(fusions=fusions, fusion_seq=list(ix.m_fusion_seq))
        return FusionMapper.sort_matches(kept), counters
    finally:
        ix.close()


def scan_pair_end_report(ref_file: str, fusion_csv: str, read1_file: str, read2_file: str, device: int = -1,
                         settings: Settings = None) -> Tuple[List[FusionResult], dict]:
    """The whole of ``PairEndScanner::scan`` up to the reporters (pescanner.rs:78-176, :335-337):
    files -> matches -> filter -> per-gene-pair sort -> cluster -> qualified fusions, most
    supported first.  ``report_text`` / ``report_json`` of fusion_result.py turn the list into
    the reference's stdout block and JSON file."""
    settings = settings or Settings()
    keep: dict = {}
    kept, counters = scan_pair_end_files(ref_file, fusion_csv, read1_file, read2_file, device,
                                         settings.deletion_threshold, keep)
    groups = group_and_sort(kept, len(keep["fusions"]))
    results = cluster_matches(groups, keep["fusions"], keep["fusion_seq"], settings)
    counters["fusions"] = len(results)
    return results, counters


def scan_single_end_files(ref_file: str, fusion_csv: str, read1_file: str, device: int = -1,
                          deletion_threshold: int = 50, _keep: dict = None) -> Tuple[List[ReadMatch], dict]:
    """``SingleEndScanner`` (src/core/sescanner.rs:62-195) up to the sorted, filtered match list:
    every read is mapped, then its reverse complement when it was mapable without a match."""
    ref = FastaReader(ref_file, True)
    ref.read_all()
    fusions = Fusion.parse_csv(fusion_csv)
    ix = Indexer(ref.m_all_contigs, fusions, device)
    ix.make_index()
    try:
        b, text = FastqReader(read1_file).read_all_device(ix)
        off = b.offsets.cpu().numpy()
        bases, quals = b.bases.cpu().numpy().tobytes(), b.quals.cpu().numpy().tobytes()
        reads = [bases[off[i]:off[i + 1]] for i in range(b.n_records)]
        mapper = FusionMapper(ix)
        found: List[ReadMatch] = []
        for i, m in enumerate(mapper.scan_single_end(reads)):
            if m is None:
                continue
            q = quals[off[i]:off[i + 1]]
            m.m_quality = q[::-1] if m.m_reversed else q
            m.m_name = record_lines(b, text, i)[0]
            m.m_source = "r1"
            found.append(m)
        kept, removed = mapper.filter_matches(found, deletion_threshold)
        counters = {"reads": b.n_records, "matches_before_filtering": len(found), **removed}
        if _keep is not None:
            _keep.update(fusions=fusions, fusion_seq=list(ix.m_fusion_seq))
        return FusionMapper.sort_matches(kept), counters
    finally:
        ix.close()


def scan_single_end_report(ref_file: str, fusion_csv: str, read1_file: str, device: int = -1,
                           settings: Settings = None) -> Tuple[List[FusionResult], dict]:
    """``SingleEndScanner::scan`` up to the reporters: files -> qualified fusions."""
    settings = settings or Settings()
    keep: dict = {}
    kept, counters = scan_single_end_files(ref_file, fusion_csv, read1_file, device, settings.deletion_threshold, keep)
    results = cluster_matches(group_and_sort(kept, len(keep["fusions"])), keep["fusions"], keep["fusion_seq"], settings)
    counters["fusions"] = len(results)
    return results, counters
