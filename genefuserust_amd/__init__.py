"""genefuserust_amd — MI355X-native k-mer seed matcher behind the GeneFuseRust
``Indexer`` interface (src/core/indexer.rs of the reference).

Only the hot path lives here: index build + per-read seed mapping as HIP
kernels for gfx950 (csrc/), the C ABI (include/gfmatch.h) and this host-side
mirror of the reference interface.  See DESIGN.md.
"""
from .indexer import (Exon, FastaReader, Fusion, Gene, GenePos, Indexer, SeqMatch, resolve_gene_slice,  # noqa: F401
                      unpack_matches)
from .fusion_mapper import FusionMapper, ReadMatch, edit_distance, reverse_complement  # noqa: F401
from .read_pair import MergedRead, SequenceReadPair, fast_merge_batch, fast_merge_device, scan_pair_end  # noqa: F401
from .fusion_result import (FusionResult, Settings, cluster_matches, group_and_sort, report_json,  # noqa: F401
                            report_text)
from .matcher import Matcher, MatcherPanic, remove_alignables  # noqa: F401
from .fastq import FastqBatch, FastqReader, FastqReaderPair, fastq_cut_device, record_lines  # noqa: F401

__version__ = "0.1.0"
