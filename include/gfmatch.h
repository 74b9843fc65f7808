/* gfmatch — MI355X-native k-mer seed matcher: C ABI.
 *
 * Drop-in boundary for the one hot path of Crispy13/GeneFuseRust: the `Indexer`
 * of src/core/indexer.rs (index build + per-read seed mapping).  The reference
 * has no FFI of its own; the seam is the Rust method set that `FusionMapper`
 * uses.  Each entry point below names the reference interface it
 * replaces (paths relative to the reference root).  INTEGRATION.md shows the
 * Rust `extern "C"` block and the `Indexer` wrapper a maintainer would add.
 *
 * Conventions: plain pointers and sizes, caller owns every buffer, no panics —
 * functions return GF_OK (0) or a negative GF_ERR_* code and gf_last_error()
 * holds a message for the calling thread.  There is no CPU fallback: every
 * compute entry point fails with GF_ERR_NO_DEVICE when no HIP device is usable.
 * Concurrent gf_map_* / gf_scan_* / gf_stream_* calls on one index are allowed (the index is
 * read-only after gf_index_build, like `&self` in Indexer::map_read, which the reference calls
 * from t-1 consumer threads, pescanner.rs:296-311): host-buffer calls run on up to 8 internal
 * lanes (stream + arena each) so that callers overlap; device-buffer calls on different streams
 * are independent, on one stream they run in call order.  Profiling (gf_set_profiling) and
 * gf_set_map_variant are single-threaded switches for experiments.
 *
 * Linking: libgfmatch.so needs libamdhip64 and librccl (the one exchange between GPUs, gf_comm_* below, calls RCCL
 * itself); a process that has loaded another copy of either under the same SONAME (PyTorch-ROCm bundles both) shares it.
 *
 * Device buffers: the kernels load whole aligned 16-byte chunks around a batch's span of bases —
 * up to 15 bytes before d_bases + offsets[0] and up to 15 (FASTQ/merge: 63) past the last base are
 * READ (never written, never interpreted).  Inside any whole allocation that is always
 * addressable; a sub-range of a larger buffer is fine too; only a span ending exactly at the end
 * of a mapping that is not 16-byte padded would not be.
 */
#ifndef GFMATCH_H
#define GFMATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GF_OK 0
#define GF_ERR_ARG (-1)           /* null pointer, negative size, bad offsets */
#define GF_ERR_HIP (-2)           /* a HIP runtime call failed */
#define GF_ERR_NO_DEVICE (-3)     /* no usable HIP device */
#define GF_ERR_CAPACITY (-4)      /* gene set too large for the 29-bit site space */
#define GF_ERR_READ_TOO_LONG (-5) /* a read exceeds GF_MAX_READ_LEN */

/* Longest read accepted.  The reference cannot see reads above 1000 bases
 * (src/core/fastq_reader.rs:27 caps a FASTQ line at 1000 bytes and panics
 * beyond), merged pairs stay below 2000. */
#define GF_MAX_READ_LEN 4096

/* Marker written to out_counts for a read longer than the limit given to
 * gf_map_reads_device (the host entry points return GF_ERR_READ_TOO_LONG). */
#define GF_COUNT_TOO_LONG 255

typedef struct gf_index gf_index; /* opaque; replaces `struct Indexer` (indexer.rs:67-78) */

/* SeqMatch + GenePos flattened (indexer.rs:41-45, common.rs:4-7).
 * [seq_start, seq_end] is the inclusive read interval; (contig, position) is
 * start_gp: contig = index into the gene list, negative position = reverse strand. */
typedef struct gf_seqmatch {
  int32_t seq_start;
  int32_t seq_end;
  int32_t position;
  int16_t contig;
  int16_t pad;
} gf_seqmatch;

/* One read's non-empty result, as exchanged between ranks and handed to the
 * host-side scoring (`FusionMapper::map_read`, fusion_mapper.rs:93-132). */
typedef struct gf_hit {
  int64_t read_id; /* read_id_base + index of the read in the batch */
  int32_t n;       /* 1 or 2 valid entries in m[] (TOP first, then SECOND) */
  int32_t pad;
  gf_seqmatch m[2];
} gf_hit;

typedef struct gf_options {
  int32_t device;        /* HIP device ordinal; -1 = the calling thread's current device */
  int32_t reserved[7];   /* must be zero */
} gf_options;

typedef struct gf_index_info {
  int64_t n_genes;
  int64_t total_bp;       /* sum of gene lengths */
  int64_t n_sites;        /* valid (key, site) pairs enumerated over both strands */
  int64_t n_keys;         /* distinct k-mers = m_kmer_pos.len() */
  int64_t n_unique;       /* keys with exactly one site */
  int64_t n_dupe_keys;    /* keys with 2..5 sites (DUPE_NORMAL_LEVEL) */
  int64_t n_high_keys;    /* keys with >= 6 sites (DUPE_HIGH_LEVEL) */
  int64_t n_dupe_sites;   /* sites stored in duplicate lists */
  int64_t n_buckets;      /* 64-byte buckets in the device table */
  int64_t table_bytes;    /* device bytes: buckets + duplicate lists */
  int32_t device;
  int32_t pad;
} gf_index_info;

/* --- index lifetime --------------------------------------------------------
 * gf_index_build replaces Indexer::new / Indexer::with_loaded_ref followed by
 * Indexer::make_index (indexer.rs:81, :100, :122-177) from the point where each
 * gene's slice has been cut out of its chromosome (indexer.rs:154-158):
 * gene_seqs[c] points at gene_lens[c] raw bytes (any case; upper-cased inside,
 * like :159).  gene_lens[c] < 0 marks a gene whose chromosome was not found
 * (:149-150): nothing is indexed and its fusion sequence is "".
 * Builds forward and reverse-complement sites, classifies keys into unique /
 * 2..5 / >=6 (:179-241) and leaves the table resident in HBM. */
int gf_index_build(const char* const* gene_seqs, const int64_t* gene_lens, int32_t n_genes,
                   const gf_options* opts, gf_index** out_index);
void gf_index_free(gf_index* idx);
int gf_index_info_get(const gf_index* idx, gf_index_info* out);

/* Long-lived hosts: the library keeps grow-only device memory between calls — a workspace per
 * (device, stream) used by gf_map_reads_device / gf_scan_pairs_device (64-112 bytes per read of the
 * largest span mapped on that stream, at most 2^30 reads per span; owned by the process so that it
 * survives the index rebuilds of multi-CSV mode) and, per index, the arenas of the host-buffer entry
 * points.  gf_index_trim waits for the work queued on those streams and frees them all for the
 * index's device (they grow again on demand).  gf_index_free waits for the device and hands the index's
 * device blocks to a cache that the next gf_index_build on that device draws from (multi-CSV mode rebuilds the
 * index per CSV); at most GF_INDEX_CACHE_MIB MiB (environment, default 2048, 0 = none) stay cached, and
 * gf_index_trim frees those too. */
int gf_index_trim(gf_index* idx);

/* Test hook: one of the index's derived device arrays, copied to the host as it is.  Returns its size in
 * bytes (or a negative error) and copies min(cap, size) bytes to out when out != NULL.
 *   GF_EXPORT_GDU      uint32 pairs, one per 16 site codes: (both strands of the genes as 2-bit codes A0 C1 T2 G3,
 *                      the sites' flags: even bit = "only site of its key", odd bit = "its key has six sites or more"
 *                      — the keys that never vote, indexer.rs:202-239)
 *   GF_EXPORT_FILTER   the presence filter over canonical 14-mers, uint32 words
 *   GF_EXPORT_LIN_BASE uint32 per gene: the site code of its forward base 0 (forward base f = code + f,
 *                      reverse-strand base f = code - f)
 * tests/test_index_arrays.py rebuilds all three from the gene slices and compares every word. */
enum { GF_EXPORT_GDU = 0, GF_EXPORT_FILTER = 1, GF_EXPORT_LIN_BASE = 2 };
int64_t gf_index_export(const gf_index* idx, int32_t what, void* out, int64_t cap);

/* Indexer.m_fusion_seq[c] (indexer.rs:77, :170; read by fusion_mapper.rs:230,
 * :439-452): the upper-cased gene slice, kept on the host.  Returns its length
 * (or a negative error) and copies min(cap, length) bytes to out when out != NULL. */
int64_t gf_index_fusion_seq(const gf_index* idx, int32_t contig, char* out, int64_t cap);

/* Test/diagnostic query of the device table: for each reference-coded k-mer
 * (2 bits per base, first base most significant, A=0 T=1 C=2 G=3,
 * indexer.rs:789-913) report what m_kmer_pos / m_dupe_list hold
 * (indexer.rs:300-320): out_count[i] = 0 absent, -2 HIGH, else 1..5 sites
 * written to out_contig[5*i..], out_position[5*i..] in ascending (contig,
 * position) order.  Host buffers. */
int gf_index_lookup(const gf_index* idx, const uint32_t* kmers, int64_t n, int32_t* out_count,
                    int16_t* out_contig, int32_t* out_position);

/* Test/diagnostic: segment_mask (indexer.rs:616-679) as the device computes it, on caller-given
 * class masks (values 0..3, mask r = masks[offsets[r] .. offsets[r+1]), 1..GF_MAX_READ_LEN long,
 * offsets[0] = 0) with gp1 / gp2 as the reference's i64 keys (indexer.rs:698-706).  out_counts[r] =
 * segments returned (0..2), out_matches[2r + k] = them.  Host buffers. */
int gf_segment_mask_test(const gf_index* idx, const uint8_t* masks, const int64_t* offsets, int64_t n,
                         const int64_t* gp1, const int64_t* gp2, int32_t* out_counts, gf_seqmatch* out_matches);

/* --- mapping ---------------------------------------------------------------
 * gf_map_reads replaces the per-pack loop over Indexer::map_read
 * (indexer.rs:252-538 called from fusion_mapper.rs:100 inside
 * pescanner.rs:430-515): read r is bases[offsets[r] .. offsets[r+1]) (ASCII, not
 * case-folded, like the reference).  out_counts[r] = len of the returned
 * Vec<SeqMatch> (0..2); out_matches[2*r + k] its elements in order (TOP, SECOND).
 * Entries beyond out_counts[r] are left untouched.  Host buffers; copies in,
 * launches, copies out, synchronises. */
int gf_map_reads(const gf_index* idx, const char* bases, const int64_t* offsets, int64_t n,
                 int32_t* out_counts, gf_seqmatch* out_matches);

/* Indexer::map_read for one read (indexer.rs:252): returns the number of
 * SeqMatch written to out (0..2) or a negative error.  Same kernel, n = 1. */
int gf_map_read(const gf_index* idx, const char* seq, int64_t len, gf_seqmatch out[2]);

/* Same, host buffers in, only the non-empty results out, in read order:
 * out_hits[0..*out_n) with read_id = read_id_base + r.  If more than cap reads
 * hit, *out_n holds the total and only cap records are written. */
int gf_map_reads_hits(const gf_index* idx, const char* bases, const int64_t* offsets, int64_t n,
                      int64_t read_id_base, gf_hit* out_hits, int64_t cap, int64_t* out_n);

/* Device-resident form (the measured path): every pointer is device memory on
 * the index's device, `stream` is a hipStream_t (NULL = default stream), nothing
 * is synchronised.  d_counts is uint8[n] (0..2, or GF_COUNT_TOO_LONG);
 * d_matches is gf_seqmatch[2*n] (entries beyond the count untouched).
 * max_read_len = upper bound of the read lengths in the batch (<= GF_MAX_READ_LEN);
 * it selects the LDS footprint of the kernel; longer reads get GF_COUNT_TOO_LONG. */
int gf_map_reads_device(const gf_index* idx, const void* d_bases, const void* d_offsets, int64_t n,
                        int32_t max_read_len, void* d_counts, void* d_matches, void* stream);

/* The same for a batch of reads of ONE length laid back to back (read r = bases r * read_len .. (r+1) * read_len - 1;
 * read_len <= 320): no offsets array — the kernels compute what they would load (8 bytes per read less to move: 2.3 %
 * of the mapping time at 150 bases).  Results identical to gf_map_reads_device with offsets[r] = r * read_len. */
int gf_map_reads_fixed_device(const gf_index* idx, const void* d_bases, int64_t n, int32_t read_len, void* d_counts,
                              void* d_matches, void* stream);

/* --- packed hand-over ---------------------------------------------------------------------------
 * The mapping kernels work on 2 bits per base + 1 "not A/C/G/T" bit; with ASCII input they convert
 * while they stage.  A host that maps the same reads more than once (multi-CSV mode: one index per
 * CSV over a resident read set, fusion_scan.rs:62-188) or produces the reads on the device can hand
 * over that form instead: gf_pack_bases_device converts a whole `bases` buffer (chunk c of d_pk
 * (uint32) / d_iv (uint16) = bases 16c .. 16c+15 of the buffer; gf_packed_chunks(n_bases) elements
 * each), and gf_map_reads_packed_device maps reads given by the SAME offsets (they count bases) —
 * results identical to gf_map_reads_device on the ASCII buffer; 6 bytes fetched per 16 bases
 * instead of 16. */
int64_t gf_packed_chunks(int64_t n_bases);
int gf_pack_bases_device(const gf_index* idx, const void* d_bases, int64_t n_bases, void* d_pk, void* d_iv, void* stream);
int gf_map_reads_packed_device(const gf_index* idx, const void* d_pk, const void* d_iv, const void* d_offsets, int64_t n,
                               int32_t max_read_len, void* d_counts, void* d_matches, void* stream);

/* The same conversion on the HOST (AVX2 where the CPU has it), for hosts that ship their reads over the link: the
 * packed form is 6 bytes per 16 bases where ASCII is 16, and the link — not the kernels — bounds a host-fed GPU.
 * pk / iv: gf_packed_chunks(n_bases) elements each, the words gf_pack_bases_device would write; n_threads host
 * threads (<= 0: one).  Needs no device. */
int gf_pack_bases_host(const char* bases, int64_t n_bases, uint32_t* pk, uint16_t* iv, int32_t n_threads);

/* Ordered compaction of the dense result (device): writes gf_hit records for the
 * reads with count 1..2, ascending read index, to d_hits (capacity hits_cap
 * records) and the total to *d_n_hits (int64 on device).  d_workspace must hold
 * gf_compact_workspace_bytes(n) bytes. */
int64_t gf_compact_workspace_bytes(int64_t n);
int gf_compact_hits_device(const gf_index* idx, const void* d_counts, const void* d_matches,
                           int64_t n, int64_t read_id_base, void* d_hits, int64_t hits_cap,
                           void* d_n_hits, void* d_workspace, void* stream);

/* Indexer::in_required_direction (indexer.rs:541-608), host logic.
 * gene_reversed[c] = Fusion::is_reversed() of gene c (gene.rs:98-107).
 * Returns 1/0, or a negative error. */
int gf_in_required_direction(const gf_seqmatch* matches, int32_t n, const uint8_t* gene_reversed,
                             int32_t n_genes);

/* --- the immediate caller, host logic ---------------------
 * gf_fusion_map_read is FusionMapper::map_read after its call of Indexer::map_read
 * (fusion_mapper.rs:100-131): `mapping` is the Vec<SeqMatch> of the read.  Returns
 *   GF_RM_NONE          None, *mapable = false   (fewer than 2 segments, :107-115)
 *   GF_RM_NONE_MAPABLE  None, *mapable = true    (wrong direction, :118-123: the caller
 *                                                 retries the reverse complement,
 *                                                 pescanner.rs:458-468, sescanner.rs:188-195)
 *   GF_RM_MATCH         Some(ReadMatch) written to *out: make_match (:154-194) and
 *                       calc_distance / calc_ed (:196-251) with edit_distance
 *                       (edit_distance.rs:12-197; -1 = ends on different strands, -2 = off the gene)
 * or a negative error.  fusion_seqs/fusion_lens = Indexer.m_fusion_seq (upper-cased gene
 * slices), gene_reversed[c] = Fusion::is_reversed().  Runs on the ~0.1 % of reads that
 * return two segments; stays on the host like the reference's. */
#define GF_RM_NONE 0
#define GF_RM_NONE_MAPABLE 1
#define GF_RM_MATCH 2

typedef struct gf_readmatch {
  int32_t read_break;     /* ReadMatch.m_read_break */
  int32_t gap;            /* m_gap */
  int32_t left_distance;  /* m_left_distance */
  int32_t right_distance; /* m_right_distance */
  int32_t left_position;  /* m_left_gp.position */
  int32_t right_position; /* m_right_gp.position */
  int16_t left_contig;    /* m_left_gp.contig */
  int16_t right_contig;   /* m_right_gp.contig */
} gf_readmatch;

int gf_fusion_map_read(const char* const* fusion_seqs, const int64_t* fusion_lens, int32_t n_genes,
                       const uint8_t* gene_reversed, const char* seq, int64_t len,
                       const gf_seqmatch* mapping, int32_t n_mapping, gf_readmatch* out);
/* FusionMapper::filter_matches without remove_alignables (fusion_mapper.rs:276-376), for one
 * match: 0 = kept, 1 = removed by remove_by_complexity (either side of the break shorter than
 * 20 bases or with fewer than 7 changes of base, :298-320 and :559-569, utils/mod.rs:48-56),
 * 2 = by remove_by_distance (left + right edit distance >= 5, :322-348), 3 = by
 * remove_indels (same gene and |left - right position| < deletion_threshold, the reference's
 * default is 50, :350-376) — the first of the three that applies, in the reference's order. */
int gf_readmatch_filter(const gf_readmatch* rm, const char* seq, int64_t len, int32_t deletion_threshold);
/* The order sort_matches gives (fusion_mapper.rs:378-384: sort_by(|a,b| b.partial_cmp(a)) over
 * read_match.rs:203-228): read_break descending, then shorter read first, then name
 * descending (bytes).  Returns < 0 when a comes before b, 0 when they tie, > 0 otherwise. */
int gf_readmatch_order(int32_t a_break, int64_t a_len, const char* a_name, int64_t a_name_len, int32_t b_break,
                       int64_t b_len, const char* b_name, int64_t b_name_len);
/* the same with the fusion sequences of an index */
int gf_index_fusion_map_read(const gf_index* idx, const uint8_t* gene_reversed, const char* seq, int64_t len,
                             const gf_seqmatch* mapping, int32_t n_mapping, gf_readmatch* out);
/* edit_distance (edit_distance.rs:159-193): Levenshtein distance, bit-parallel up to
 * 640 symbols of the longer string, dynamic programming beyond */
int64_t gf_edit_distance(const char* a, int64_t alen, const char* b, int64_t blen);

/* --- the step before the path ----------------------------
 * SequenceReadPair::fast_merge (read.rs:313-440) for a batch of pairs, device buffers on the
 * index's device (the index names the device and owns the per-stream workspace; the merge
 * does not look at the table).  R1 = l_*, R2 = r_* as read from the FASTQ files (R2 not
 * reverse-complemented), ASCII bases and Phred+33 qualities, offsets int64[n+1] (pair p =
 * bytes offsets[p] .. offsets[p+1] of its buffers, the same offsets for bases and qualities).
 *
 * gf_fast_merge_find_device: d_out_len[p] (int32) = length of the merged read, 0 when the
 *   pair does not merge; d_out_diff[p] (int32) = the N of the reference's "merged_diff_N"
 *   name suffix (read.rs:372).  max_read_len sizes the workspace (packed copies of both
 *   buffers); pairs with a longer read are still merged exactly, by a slower byte loop.
 * gf_fast_merge_write_device: writes merged read p (d_len[p] > 0, as produced by _find) at
 *   d_out_pos[p] (int64[n], chosen by the caller — typically the prefix sum of d_len, which
 *   packs the merged reads back to back in the layout gf_map_reads_device takes) of
 *   d_out_bases / d_out_quals.  d_out_quals may be NULL: the bases alone (qualities decide a base only
 *   where the reads disagree inside the overlap and are fetched there; half the traffic).
 * gf_fast_merge_device: both steps, for callers whose d_out_pos does not depend on the
 *   lengths (e.g. one slot of len1+len2 bytes per pair). */
int gf_fast_merge_find_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                              const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                              const void* d_r_offsets, int64_t n, int32_t max_read_len, void* d_out_len,
                              void* d_out_diff, void* stream);
int gf_fast_merge_write_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals,
                               const void* d_l_offsets, const void* d_r_bases, const void* d_r_quals,
                               const void* d_r_offsets, int64_t n, const void* d_len, const void* d_out_pos,
                               void* d_out_bases, void* d_out_quals, void* stream);
int gf_fast_merge_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals, const void* d_l_offsets,
                         const void* d_r_bases, const void* d_r_quals, const void* d_r_offsets, int64_t n,
                         int32_t max_read_len, const void* d_out_pos, void* d_out_bases, void* d_out_quals,
                         void* d_out_len, void* d_out_diff, void* stream);
/* One pair, host buffers (out_seq/out_qual: capacity len1+len2).  Returns 1 merged, 0 not
 * merged, or a negative error.  Needs a HIP device like every compute entry point. */
int gf_fast_merge(const gf_index* idx, const char* l_seq, const char* l_qual, int32_t len1, const char* r_seq,
                  const char* r_qual, int32_t len2, char* out_seq, char* out_qual, int32_t* out_len,
                  int32_t* out_diff);

/* FastqReader::read (fastq_reader.rs:75-147) for a whole FASTQ text resident in HBM (plain text:
 * gzip stays on the host).  A line is the bytes up to, not including, a '\n' — a '\r' stays
 * in the line, as in the reference; the last line may lack its '\n'; record i owns lines
 * 4i (name), 4i+1 (sequence), 4i+2 (strand), 4i+3 (quality); a trailing group of fewer than
 * four lines is no record.
 *
 * gf_fastq_index_device: d_nl_pos[k] (int64, k < cap_lines) = byte offset of the k-th '\n';
 *   d_n_lines (int64[2]) = {lines in the text, newlines in the text}.  Records = lines / 4.
 * gf_fastq_gather_device: copies the sequence and quality lines of records 0..n_records-1
 *   back to back into d_bases / d_quals (capacity cap_bytes each; n_bytes always suffices)
 *   with d_offsets (int64[n_records+1]) — the layout gf_map_reads_device and
 *   gf_fast_merge_*_device take.  A quality line is cut or padded with '!' to its sequence's
 *   length; *d_n_bad (uint64) counts the records where that happened.  Name and strand lines
 *   are located by d_nl_pos for whoever needs them.
 * d_workspace: gf_fastq_workspace_bytes(n_bytes) bytes, the same buffer for both calls. */
int64_t gf_fastq_workspace_bytes(int64_t n_bytes);
int gf_fastq_index_device(const gf_index* idx, const void* d_text, int64_t n_bytes, void* d_nl_pos, int64_t cap_lines,
                          void* d_n_lines, void* d_workspace, void* stream);
int gf_fastq_gather_device(const gf_index* idx, const void* d_text, int64_t n_bytes, const void* d_nl_pos,
                           int64_t n_newlines, int64_t n_records, void* d_offsets, void* d_bases, void* d_quals,
                           int64_t cap_bytes, void* d_n_bad, void* d_workspace, void* stream);
/* gf_fastq_gather_lean_device: the sequence lines only; d_qual_off[r] (int64[n_records]) = byte offset of record r's
 * quality line in the text, for gf_scan_pairs_text_device.  *d_n_bad still counts the records whose quality line
 * has another length than the sequence: with any of those, gather the full way (the in-place form cannot pad). */
int gf_fastq_gather_lean_device(const gf_index* idx, const void* d_text, int64_t n_bytes, const void* d_nl_pos,
                                int64_t n_newlines, int64_t n_records, void* d_offsets, void* d_bases, int64_t cap_bytes,
                                void* d_qual_off, void* d_n_bad, void* d_workspace, void* stream);

/* --- the pair policy, device resident --------------------
 * PairEndScanner::scan_pair_end (pescanner.rs:427-518) for a pack of n pairs whose records are in
 * HBM (the layout gf_fastq_gather_device writes and gf_fast_merge_*_device take; l_bytes / r_bytes =
 * bytes in the R1 / R2 base buffers = offsets[n]):
 *   merged = pair.fast_merge() (:438); a pair that merged is searched as its merged read only
 *   (:446-471), the others as R1 and then R2 (:473-515); a read that mapped to two places in the
 *   wrong direction (`mapable` without a match, fusion_mapper.rs:107-123) is searched again as its
 *   reverse complement (SequenceRead::reverse_complement, read.rs:243-261: qualities reversed).
 * Output: one gf_pair_hit per read on which FusionMapper::map_read would call make_match — two
 * segments in the required direction — in the order the reference pushes them (pair, then merged |
 * R1, R2), with that read's bases and qualities (the reverse complement's when the match is on it)
 * copied to d_hit_bases / d_hit_quals at seq_offset: what gf_index_fusion_map_read needs to finish
 * the ReadMatch on the host.  Nothing is synchronised and nothing goes through the host between the
 * steps (merge-find, slots, merge-write, three mapping passes, classification, reverse
 * complements, a fourth mapping pass over those, ordered compaction).
 * flags bit 0: the match is on the reverse complement; bit 1: ReadMatch.m_reversed as the reference
 * sets it — for R1 / R2 (:489,:511) but NOT for a merged read (:465-468).
 * d_totals (int64[8], device): [0] hits, [1] bytes of their reads, [2] pairs that merged,
 * [3] reads searched again reversed, [4] overflow bits — 1: more retries than retry_cap reads (the
 * retry pass was emptied: run the pack again with retry_cap = 3 n), 2: more hits / bytes than the
 * output capacities (totals [0], [1] say how many).  retry_cap <= 0: gf_scan_pairs_retry_capacity(n).
 * gene_reversed flags for the direction rule come from gf_index_set_gene_reversed (all false until set). */
typedef struct gf_pair_hit {
  int64_t pair_id;     /* pair_id_base + index of the pair in the pack */
  int32_t source;      /* 0 = merged read, 1 = R1, 2 = R2 */
  int32_t flags;
  int32_t read_len;    /* length of the matched read (bases at seq_offset) */
  int32_t merge_diff;  /* source 0: the N of the " merged_diff_N" name suffix (read.rs:372) */
  int64_t seq_offset;  /* into d_hit_bases / d_hit_quals */
  gf_seqmatch m[2];    /* Indexer::map_read of that read: TOP, SECOND */
} gf_pair_hit;

int gf_index_set_gene_reversed(gf_index* idx, const uint8_t* gene_reversed, int32_t n_genes);
/* The host-side tail for the records of gf_scan_pairs_device (host copies of the records and of
 * d_hit_bases): make_match + calc_distance (fusion_mapper.rs:154-251) per record on n_threads host
 * threads — out[k] the ReadMatch fields, out_status[k] = GF_RM_MATCH.  (The direction gate of
 * :118-123 has passed on the device.) */
int gf_pair_hits_finish(const gf_index* idx, const gf_pair_hit* hits, int64_t n, const char* hit_bases,
                        int64_t hit_bytes, gf_readmatch* out, int32_t* out_status, int32_t n_threads);
/* The same tail on the DEVICE, for the records while they are still in HBM (d_hits / d_hit_bases / d_totals as
 * gf_scan_pairs_device wrote them: d_totals[0] = records, at most hits_cap of which exist): a wavefront per record and
 * side of the break, Levenshtein distance by the block-based bit-vector recurrence of edit_distance.rs:12-92 with the
 * match masks taken by wave ballots.  d_out: gf_readmatch[hits_cap], d_status: int32[hits_cap] (GF_RM_MATCH, or
 * GF_ERR_ARG for a record whose contigs or segments are out of range).  Queued on `stream`, nothing synchronised. */
int gf_pair_hits_finish_device(const gf_index* idx, const void* d_hits, const void* d_totals, int64_t hits_cap,
                               const void* d_hit_bases, void* d_out, void* d_status, void* stream);
int64_t gf_scan_pairs_retry_capacity(int64_t n);
int gf_scan_pairs_device(const gf_index* idx, const void* d_l_bases, const void* d_l_quals, const void* d_l_offsets,
                         int64_t l_bytes, const void* d_r_bases, const void* d_r_quals, const void* d_r_offsets,
                         int64_t r_bytes, int64_t n, int32_t max_read_len, int64_t pair_id_base, int64_t retry_cap,
                         void* d_hits, int64_t hits_cap, void* d_hit_bases, void* d_hit_quals, int64_t hit_bytes_cap,
                         void* d_totals, void* stream);
/* The same with the qualities left in the FASTQ texts (gf_fastq_gather_lean_device): d_l_text / d_r_text are the
 * texts and d_l_qual_off / d_r_qual_off (int64[n]) where each record's quality line starts.  The pipeline reads
 * qualities for the mismatching columns of an overlap (read.rs:380-428), for the reads it searches again reversed
 * and for the hit records — a few bytes per thousand pairs — so copying 1.5 GB of them per 10 M records first
 * (half of what gf_fastq_gather_device writes) buys nothing.  Same results, record for record. */
int gf_scan_pairs_text_device(const gf_index* idx, const void* d_l_bases, const void* d_l_text, const void* d_l_qual_off,
                              const void* d_l_offsets, int64_t l_bytes, const void* d_r_bases, const void* d_r_text,
                              const void* d_r_qual_off, const void* d_r_offsets, int64_t r_bytes, int64_t n,
                              int32_t max_read_len, int64_t pair_id_base, int64_t retry_cap, void* d_hits, int64_t hits_cap,
                              void* d_hit_bases, void* d_hit_quals, int64_t hit_bytes_cap, void* d_totals, void* stream);

/* --- streaming host entry ----------------------------------------------------------------------
 * For a host that produces packs of reads while earlier packs are being mapped (the reference's
 * producer / consumer loop, pescanner.rs:255-311, with the queue on the device side of the link):
 * gf_stream_submit queues copy-in, mapping, ordered compaction and the copy of the hit records back
 * on one of `depth` slots (each with its own HIP stream, device arena and pinned result block) and
 * returns; gf_stream_collect waits for the OLDEST submitted pack and hands out its hits (ascending
 * read id = read_id_base + index in the pack; *out_n = their number, of which min(cap, *out_n) are
 * written).  With depth >= 2 the copy of pack k+1 overlaps the kernels of pack k.
 * `bases` / `offsets` as for gf_map_reads.  When they live in pinned memory (gf_host_alloc) the copy
 * is asynchronous and the buffers must stay untouched until that pack is collected; from pageable
 * memory the runtime stages the copy before gf_stream_submit returns.
 * A gf_stream belongs to one thread at a time; several streams may share an index.
 * LIFETIME: a gf_stream keeps a pointer to its index — close every stream before gf_index_free.  The library counts
 * the open streams of an index: gf_index_free on an index that still has one prints a message and does NOT free it
 * (a leak, where the alternative is a use-after-free in gf_stream_collect / gf_stream_close).
 * Errors: GF_ERR_CAPACITY when a pack exceeds max_reads / max_bytes or every slot is in flight. */
typedef struct gf_stream gf_stream;
int gf_stream_open(const gf_index* idx, int64_t max_reads, int64_t max_bytes, int32_t depth, gf_stream** out);
int gf_stream_submit(gf_stream* s, const char* bases, const int64_t* offsets, int64_t n, int64_t read_id_base);
/* The pack in packed form (gf_pack_bases_host): pk / iv are the packed stream of a base buffer, offsets (int64[n+1])
 * count bases in that buffer as for gf_stream_submit; the chunks covering offsets[0] .. offsets[n] cross the link
 * (0.375 bytes per base).  max_bytes of gf_stream_open bounds the pack's bases as before. */
int gf_stream_submit_packed(gf_stream* s, const uint32_t* pk, const uint16_t* iv, const int64_t* offsets, int64_t n,
                            int64_t read_id_base);
int gf_stream_collect(gf_stream* s, gf_hit* out_hits, int64_t cap, int64_t* out_n);
void gf_stream_close(gf_stream* s);
/* pinned host memory for the buffers handed to gf_stream_submit / gf_map_reads* (hipHostMalloc) */
void* gf_host_alloc(int64_t bytes);
void gf_host_free(void* p);
/* hipMemcpyAsync host -> device on `stream`, for hosts that hold no HIP binding of their own (a
 * streamed FASTQ: genefuserust_amd/scan_stream.py).  Asynchronous when h_src is pinned. */
int gf_copy_from_host_device(const gf_index* idx, const void* h_src, void* d_dst, int64_t nbytes, void* stream);

/* --- multi-GPU: the one exchange of the path ----------------------------------------
 * One process per GPU; reads shard into contiguous ranges (rank r maps reads [r n / R, (r+1) n / R) with
 * read_id_base = its first read), every rank builds the same index, nothing is exchanged while mapping.
 * The per-rank gf_hit lists (gf_compact_hits_device: ascending read id) are merged by ONE all-gather over
 * RCCL (xGMI inside a node); concatenated in rank order they are the list one GPU would produce, on
 * every rank.  The reference has no counterpart (one process: consumer threads push under a mutex,
 * fusion_mapper.rs:253-275); its split of the work over workers is fusion_scan.rs:103-116,143-181.
 *
 * gf_comm_unique_id: rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by
 *   whatever the host has (a file, a socket, MPI, torch.distributed); gf_comm_init is collective over the
 *   `world` ranks that were given that id (ncclCommInitRank on `device`).  Groups are just communicators
 *   made from another id (multi-CSV mode: the ranks that share a CSV).
 * gf_allgather_hits_device: every rank passes its list (d_hits, *d_n_hits records, device memory, 16-byte
 *   aligned) and the same cap.  Queued on `stream`: stage -> ncclAllGather of (cap + 1) records per rank
 *   -> pack.  d_merged (world * cap records) receives the merged list; d_totals (int64[2 + world]):
 *   [0] records in it, [1] 1 when a rank had more than cap records (that rank's list is cut at cap: run
 *   the batch again with a larger cap), [2 + r] rank r's own count.  d_workspace:
 *   gf_allgather_workspace_bytes(world, cap) bytes.  Nothing is synchronised: the exchange of one batch
 *   can run on a side stream while the next batch is being mapped.
 * gf_pack_gathered_hits_device: the pack step alone, on a receive buffer of `world` blocks of (cap + 1)
 *   records whose record 0 holds the block's count in read_id (what the all-gather delivers) — for hosts
 *   that bring their own transport, and for the tests. */
#define GF_COMM_ID_BYTES 128
#define GF_ERR_COMM (-6)          /* an RCCL call failed */
typedef struct gf_comm gf_comm;
int gf_comm_unique_id(void* out_id);
int gf_comm_init(const void* id, int32_t rank, int32_t world, int32_t device, gf_comm** out);
int gf_comm_rank(const gf_comm* comm, int32_t* rank, int32_t* world);
void gf_comm_free(gf_comm* comm);
int64_t gf_allgather_workspace_bytes(int32_t world, int64_t cap);
int gf_allgather_hits_device(gf_comm* comm, const void* d_hits, const void* d_n_hits, int64_t cap, void* d_merged,
                             void* d_totals, void* d_workspace, void* stream);
int gf_pack_gathered_hits_device(const void* d_recv, int32_t world, int64_t cap, void* d_merged, void* d_totals,
                                 void* stream);

/* --- instrumentation -------------------------------------------------------
 * With profiling on, gf_map_reads_device brackets its mapping kernel with HIP
 * events on the launch stream; gf_last_map_kernel_ms synchronises on them and
 * returns the duration of the most recent launch (negative if none). */
int gf_set_profiling(gf_index* idx, int32_t on);

/* First pass for short reads (up to 320 bases for 0, 256 for the others): 0 (default) = flat
 * pipeline (thread per read: pack + seed + verify in one kernel, filter and bucket passes over
 * the undecided, exact wave-per-read kernel on the survivors), 1 = wave-per-read kernel probing
 * every window, 2 = wave-per-read kernel with seed + verify.  All are exact and return identical
 * results; the switch exists for A/B timing and for the tests that check exactly that. */
int gf_set_map_variant(gf_index* idx, int32_t variant);
/* Host-buffer calls (gf_map_reads, gf_map_reads_hits, gf_stream_submit) of up to `reads` reads take the zero-copy
 * route: the pack is staged in pinned memory (or read where it is, when it lies in a gf_host_alloc block), ONE launch
 * of the exact wave-per-read kernels fetches it over the link and writes the results back, the host waits on a word
 * the kernel stores — a pack of the reference's size (PACK_SIZE = 1000 pairs, common.rs:23) costs one launch instead
 * of two copies, seven launches and two copies back.  Larger calls take the batch route (copies + the flat pipeline).
 * reads < 0 restores the default (8192, or GF_PACK_CALL_READS from the environment); 0 = batch route for every call
 * beyond 64 reads.  Same results either way (tests/test_boundary_gpu.py). */
int gf_set_pack_call_reads(gf_index* idx, int64_t reads);
float gf_last_map_kernel_ms(gf_index* idx);

const char* gf_last_error(void);
const char* gf_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GFMATCH_H */
