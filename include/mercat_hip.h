/* mercat_hip.h -- C ABI of libmercat_hip.so: the MI355X (gfx950) k-mer counting engine that
 * stands in for MerCat2's counting path.
 *
 * Drop-in boundary (reference file:line, relative to the reference checkout):
 *   find_kmers(file, kmer, min_count) -> {kmer: count}      lib/mercat2_kmers.py:32-78
 *   run_mercat2(basename, files, out_file, kmer, min_count) bin/mercat2.py:115-137
 *   chunk_files / Chunker.stream_delim                      bin/mercat2.py:86-106, lib/mercat2_Chunker.py:39-59
 * The Python host layer (mercat2_amd/) mirrors those three callables on top of this ABI with
 * ctypes; INTEGRATION.md shows the stub a MerCat2 maintainer would add.
 *
 * Conventions: every function returns an int status (MK_OK == 0, negative = error); the text
 * of the last error of a context is mk_last_error(ctx).  The caller owns every buffer it
 * passes in or receives into; a context owns its device memory and its HIP stream.  A context
 * is not thread-safe; different contexts may be used from different host threads.  There is
 * no CPU fallback anywhere behind this ABI: without a usable HIP device mk_create fails.
 */
#ifndef MERCAT_HIP_H
#define MERCAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MK_OK 0
#define MK_ERR_ARG (-1)       /* bad argument (k < 1, unknown alphabet, null pointer, ...) */
#define MK_ERR_HIP (-2)       /* a HIP runtime call failed (message has the call and hipError) */
#define MK_ERR_NOMEM (-3)     /* device or host allocation failed */
#define MK_ERR_STATE (-4)     /* call sequence error (feed outside begin/end, ...) */
#define MK_ERR_NON_ASCII (-5) /* a SEQUENCE line holds bytes >= 0x80: the reference would decode them as
                                 multi-byte characters; refused rather than counted wrongly.  Header lines
                                 may hold any bytes: they never enter a k-mer */
#define MK_ERR_IO (-6)        /* file could not be written */
#define MK_ERR_RANGE (-7)     /* caller buffer too small / value out of range */
#define MK_ERR_UNSUPPORTED (-8) /* clean mode (mk_set_clean): the chunk holds text whose rewrite by removeN the GPU does not
                                 reproduce (a blank inside a sequence line, a '>' that does not start a header line, a
                                 0x7F byte); nothing of the chunk was counted -- count the text mk_remove_n produces instead */

/* Alphabets select the packed fast path; they never restrict the input.  Windows holding a
 * symbol outside the alphabet are still counted, exactly, by the by-reference kernel
 * (the reference counts "any character": lib/mercat2_kmers.py:56-60). */
#define MK_ALPHABET_NT2 0 /* A C G T -> 2-bit codes 0..3 (ASCII order)            */
#define MK_ALPHABET_AA5 1 /* 'A'..'Z' -> 5-bit codes 0..25 (ASCII order)          */
#define MK_ALPHABET_RAW 2 /* no packing: every window by reference (any k, any text) */

typedef struct mk_ctx mk_ctx;

/* Counters of one context, cumulative since mk_create / mk_reset_stats. */
typedef struct mk_stats_t {
  uint64_t raw_bytes;      /* FASTA bytes fed                                               */
  uint64_t symbols;        /* sequence characters kept by the parser (the "bases")          */
  uint64_t windows;        /* length-k windows counted (all paths)                          */
  uint64_t exotic_windows; /* of those, windows counted by the by-reference path            */
  uint64_t chunks;         /* chunks ended                                                  */
  uint64_t survivors;      /* chunk-table entries that passed their chunk's min_count       */
  uint64_t rows;           /* distinct k-mers now in the running (merged) table             */
  uint64_t table_slots;    /* slots (or dense bins) of the chunk table of the last chunk    */
  int32_t mode;            /* 0 dense-LDS, 1 hash64 (one-word packed keys), 2 hash128 (nucleotide 33..64-mers:
                              two-word packed keys), 3 by-reference on bytes only                 */
  int32_t profiled;        /* 1 when per-kernel HIP-event timing is on (mk_set_profiling)   */
  /* HIP-event time per kernel family, milliseconds, and launches (only when profiled) */
  double ms_parse, ms_pack, ms_count, ms_exotic, ms_filter, ms_export;
  uint64_t n_parse, n_pack, n_count, n_exotic, n_filter, n_export;
  /* partition stage (bucket histogram + scan + scatter) that feeds the LDS count kernel;
   * ms_count / n_count are the count kernel alone */
  double ms_part;
  uint64_t n_part;
  uint64_t records;  /* super-k-mer records written (0 on other paths) */
  uint64_t distinct; /* distinct packed keys seen per chunk, summed over chunks */
  uint64_t part_retries; /* chunks partitioned twice: the sampled bucket sizes were too small somewhere */
  uint64_t part_reused;  /* chunks that inherited the bucket regions of the chunk before them (no histogram, no scan) */
  uint64_t fused_chunks; /* chunks whose count kernel put the survivors into the running table itself (ABI 4)        */
  uint64_t fuse_spilled; /* ... survivors of those it set aside instead (table filling up), imported afterwards        */
  uint64_t parse_retries; /* chunks parsed a second time by the general parser (a blank inside a sequence line)           */
} mk_stats_t;

/* ---- lifetime ------------------------------------------------------------------------- */
/* One context per GPU and per (alphabet, k).  Replaces the per-task state of
 * countKmers/find_kmers (bin/mercat2.py:112-114). */
int mk_create(int device, int alphabet, int k, mk_ctx** out);
/* HIP devices this process sees (0 when there is none or the runtime cannot start): the GPUs a host may spread a
 * sample's chunks over, or give one small sample each (bin/mercat2.py:217 sizes its Ray pool by -n the same way). */
int mk_device_count(void);
void mk_destroy(mk_ctx* ctx);
const char* mk_last_error(const mk_ctx* ctx);
/* Forget the running (merged) table: start the next sample (run_mercat2's `kmers = dict()`,
 * bin/mercat2.py:117). */
int mk_reset(mk_ctx* ctx);
/* mk_reset that also sizes the (emptied) packed table for about expect_rows distinct keys when that is less than its
 * present size: what an owner does between handing its rows out and taking in the rows of its own key range
 * (1/N of the table: a table that fits the caches takes the imports several times faster). */
int mk_reset_for(mk_ctx* ctx, uint64_t expect_rows);
/* Opt-in extension, NOT reference behaviour (the reference counts forward-strand substrings,
 * lib/mercat2_kmers.py:56-60): with on != 0 every ACGT-only window is counted under
 * min(kmer, reverse-complement(kmer)).  Windows holding other characters keep their own text.
 * Nucleotide alphabet only; call before the first chunk of a sample (or right after mk_reset).
 * Nucleotide k <= 64.  One-word keys (k <= 32) file a window under min(11-mer, its reverse complement) of its
 * minimizer; two-word keys (33 <= k <= 64) under the smallest canonical 11-mer among the candidates of the window's
 * first and last 32 bases -- the set the two strands share.  k > 64 and the raw alphabet count text, which has no
 * complement: MK_ERR_ARG. */
int mk_set_canonical(mk_ctx* ctx, int on);

/* removeN's effect on the count, on the GPU (lib/mercat2_fasta.py:53-119; MerCat2 runs it on every nucleotide FASTA
 * before counting, bin/mercat2.py:239-244): with on != 0 the chunks fed from now on are RAW FASTA and are counted as
 * if removeN had rewritten them first -- every run of upper-case 'N' cuts its record (no window spans or holds it),
 * text in front of the first header line is not counted, and with toupper != 0 lower-case letters count as their
 * upper-case forms (a lower-case 'n' then is an 'N' that IS counted, as in the reference, which splits before it
 * upper-cases).  The N scan shares the parser's and packer's pass over the text; the table does not wait for the
 * rewritten file.  Nucleotide alphabet, one chunk per file (a sample the reference would chunk is cut on the CLEANED
 * text: count mk_remove_n's output for those).  A chunk the mode cannot reproduce is refused with MK_ERR_UNSUPPORTED
 * and nothing of it is counted. */
int mk_set_clean(mk_ctx* ctx, int on, int toupper);
typedef struct mk_clean_gpu_t {
  uint64_t raw_bytes;    /* bytes of the chunks counted in clean mode since mk_reset                              */
  uint64_t symbols;      /* sequence characters kept (N runs excluded): the total_length removeN divides by,
                            without the header lines of split records                                            */
  uint64_t gc_count;     /* 'G' + 'C' among them AFTER -toupper if it is on, header lines of split records not
                            included: NOT the reference's "GC Content" figure, which counts before upper-casing and
                            includes those header lines (lib/mercat2_fasta.py:92-113; mk_remove_n reports that one) */
  uint64_t n_bytes;      /* upper-case 'N' bytes removed                                                         */
  uint64_t n_runs;       /* runs of N = cuts = pieces added                                                      */
  uint64_t header_lines; /* records                                                                             */
  uint64_t last_runs;    /* runs of the last chunk (mk_clean_runs lists them)                                    */
} mk_clean_gpu_t;
int mk_clean_stats(mk_ctx* ctx, mk_clean_gpu_t* out);
/* The N runs of the last chunk as [starts[i], ends[i]) in the parsed stream (kept characters in order, one separator
 * per header line: the stream the k-mer windows slide over), ascending; *n = how many (cap values fit).  A chunk with
 * more than 65536 runs is not listed: MK_ERR_RANGE, *n = 0 (mk_clean_stats still counts them).  What split_sequenceN
 * cuts at (lib/mercat2_fasta.py:35-38). */
int mk_clean_runs(mk_ctx* ctx, uint64_t* starts, uint64_t* ends, size_t cap, size_t* n);

/* ---- one chunk = one find_kmers call (lib/mercat2_kmers.py:32-78) ------------------------ */
int mk_chunk_begin(mk_ctx* ctx);
/* Append raw FASTA bytes (host memory) of the current chunk; may be called repeatedly, the
 * pieces are concatenated.  The buffer may be reused on return. */
int mk_chunk_feed(mk_ctx* ctx, const uint8_t* text, size_t n);
/* Same, from device memory of this context's GPU (used when the text is already in HBM). */
int mk_chunk_feed_device(mk_ctx* ctx, const uint8_t* d_text, size_t n);
/* Parse + pack + count the chunk, keep entries with count >= min_count
 * (lib/mercat2_kmers.py:73-76) and add them into the running table
 * (the dict sum of bin/mercat2.py:121-127). */
int mk_chunk_end(mk_ctx* ctx, uint64_t min_count);
/* Count a chunk in place from device memory without the staging copy (begin+feed+end).
 * d_text must stay valid until the call returns. */
int mk_count_device(mk_ctx* ctx, const uint8_t* d_text, size_t n, uint64_t min_count);

/* ---- one file of a sample, from disk: chunk_files + a countKmers task per chunk + the dict sum
 *      (bin/mercat2.py:86-106, 112-127), with the file reading of find_kmers
 *      (lib/mercat2_kmers.py:47-50) and no chunk files ------------------------------------ */
typedef struct mk_file_stats_t {
  uint64_t disk_bytes; /* size of the file on disk                                           */
  uint64_t text_bytes; /* (inflated) bytes read and fed                                      */
  uint64_t chunks;     /* chunks counted (1 when the file was not chunked)                   */
  int32_t gz;          /* 1: the file was inflated (last suffix ".gz")                        */
  int32_t chunked;     /* 1: disk_bytes >= chunk_bytes, the Chunker rule was applied         */
  int32_t members;     /* gzip members seen                                                  */
  int32_t threads;     /* reader threads used                                                */
  int32_t contexts;    /* contexts that counted chunks                                       */
  int32_t devices;     /* GPUs those contexts are on                                         */
  int32_t split_pieces;/* > 0: the file was one filter unit counted in that many pieces on several GPUs, filtered after the sum */
  int32_t pad_;
  double s_wait_io;    /* seconds the dispatching thread waited for file blocks              */
  double s_wait_gpu;   /* seconds it waited for a context to finish its previous chunk       */
  double s_total;      /* wall seconds of the call                                           */
  double s_merge;      /* of those, the sum of the contexts' tables at the end                */
  /* (ABI 4) where the dispatching thread's time went; s_setup + s_wait_io + s_wait_gpu + s_scan + s_feed + s_retire +
   * s_drain + s_merge ~= s_total */
  double s_setup;      /* open, ring (pinned memory), worker and reader threads started       */
  double s_scan;       /* the Chunker rule applied to the blocks                              */
  double s_feed;       /* inside the host-to-device copy calls                                */
  double s_retire;     /* waiting for copies out of ring blocks the readers want back         */
  double s_drain;      /* after the last block: copies, readers and the last chunks' counting */
} mk_file_stats_t;
/* Reads `path` (gzip iff its name ends in ".gz", as the reference decides), splits it as
 * Chunker(path, dest, chunk_bytes, '>') would iff its on-disk size is >= chunk_bytes > 0
 * (chunk_bytes == 0: never), counts every chunk with its own min_count filter and adds the
 * survivors to the running table of ctxs[0].  With nctx > 1 (same alphabet, k, canonical mode; on one
 * GPU or on several: mk_plan_contexts gives the order) chunk i goes to ctxs[i mod nctx] and is counted
 * there, filtered on its own, concurrently with the reading; at the end the contexts of one GPU are summed
 * on that GPU, the GPUs' tables are summed into ctxs[0] (mk_merge_devices, MK_MERGE_GATHER) and the other
 * contexts are reset.  A file that is NOT chunked (one filter unit, lib/mercat2_kmers.py:73-76) but large
 * (>= 64 MiB of text) is, with contexts on several GPUs, cut into one piece per context at record starts
 * (mk_record_cuts), the pieces are counted unfiltered, summed, and min_count is applied to the sum
 * (single-chunk sample).  st->split_pieces says so.  threads = reader
 * threads (<= 0: pick): plain files are read, BGZF blocks and -- from 16 MiB on -- ordinary gzip
 * streams are decoded by that many threads; 1 decodes a gzip stream front to back.  st may be NULL. */
int mk_count_file(mk_ctx* const* ctxs, int nctx, const char* path, uint64_t chunk_bytes, uint64_t min_count,
                  int threads, mk_file_stats_t* st);

/* ---- result of the sample: sorted(kmers.items()) (bin/mercat2.py:130-133) --------------- */
int mk_export_size(mk_ctx* ctx, size_t* rows);
/* kmers: rows*k ASCII bytes (no terminators), counts: rows values; both caller-allocated.
 * Rows are in Python sorted(str) order == byte-wise order. */
int mk_export(mk_ctx* ctx, uint8_t* kmers, uint64_t* counts, size_t rows_cap);
/* Writes "k-mer\t{basename}_Count\n" + rows; writes NO file and sets *rows = 0 when the
 * table is empty (bin/mercat2.py:128-137). */
int mk_write_tsv(mk_ctx* ctx, const char* path, const char* basename, size_t* rows);

/* Where the last mk_export / mk_write_tsv of this context spent its time (the sorted() + print loop of
 * bin/mercat2.py:130-133 is as expensive as the counting for the reference: 17 s + 13 s at 7.7 M rows). */
typedef struct mk_export_stats_t {
  uint64_t rows;     /* rows exported (packed + kept as text)                                         */
  uint64_t bytes;    /* bytes of TSV text written (0 for mk_export)                                   */
  double s_sort;     /* device: compaction of the table + radix sort of the packed keys, waited for   */
  double s_d2h;      /* sorted rows to the host (+ the rows kept as text, sorted on the device)         */
  double s_format;   /* host: keys decoded to text, counts to decimal, merged with the text rows       */
  double s_write;    /* host: inside write(2) (the file is written while it is formatted)               */
  double s_total;
} mk_export_stats_t;
int mk_export_stats(mk_ctx* ctx, mk_export_stats_t* out);

/* ---- several samples side by side: merge_tsv (lib/mercat2_report.py:98-156) from the tables --- */
/* The combined table of n samples (contexts with the same k; each on its own GPU or all on one):
 * every k-mer present in any of them, in sorted(str) order, with its count in each sample (0 where
 * absent).  kmers: rows*k bytes, matrix: rows*n counts, row-major, both caller-allocated; call with
 * kmers = matrix = NULL to get *rows. */
int mk_merged_export(mk_ctx* const* ctxs, int n, uint8_t* kmers, uint64_t* matrix, size_t rows_cap, size_t* rows);
/* The same as the file merge_tsv writes: "<first_column>\t<names[0]>\t...\n" then one line per k-mer.
 * names are the column titles, in the order of ctxs (the reference sorts the sample names). */
int mk_write_merged_tsv(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column,
                        const char* path, size_t* rows);
/* The same file with the rows exactly as merge_tsv's streaming loop produces them: that loop looks for the next k-mer
 * only among the samples that advanced in the current step and writes a sample's pending count under the k-mer at
 * hand whenever its own key is not greater (lib/mercat2_report.py:131-150), so a key held only by samples that did
 * not advance gets no row of its own and its count lands in a later row.  Tables that share (nearly) all their keys
 * -- k = 5 over genomes, the reference's committed runs -- come out the same either way. */
int mk_write_merged_tsv_as_reference(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column,
                                     const char* path, size_t* rows);
/* merge_tsv_T (lib/mercat2_report.py:160-194), the transposed table beta diversity reads (bin/mercat2.py:354-355):
 * "sample\t<k-mer>\t...\n", then "<names[s]>\t<count>...\n" per sample.  The reference orders the k-mer columns
 * as a Python set iterates (not reproducible); here they are in sorted(str) order.  *rows = k-mer columns. */
int mk_write_merged_tsv_t(mk_ctx* const* ctxs, int n, const char* const* names, const char* path, size_t* rows);
/* ---- alpha diversity of a sample: the moments of its count column (lib/mercat2_diversity.py:13-53
 *      computes nine scikit-bio metrics from exactly these), reduced on the GPU ------------------ */
typedef struct mk_alpha_t {
  uint64_t observed; /* rows (distinct k-mers)                                   */
  uint64_t total;    /* sum of the counts                                        */
  uint64_t freq[11]; /* freq[i], i = 1..10: rows whose count is i (freq[0] unused) */
  double sum_sq;     /* sum of count^2                                           */
  double sum_clnc;   /* sum of count * ln(count)                                 */
} mk_alpha_t;
int mk_alpha_stats(mk_ctx* ctx, mk_alpha_t* out);

/* Free the per-chunk working memory of a context and keep its running table (for samples that wait
 * for mk_merged_export while others are being counted). */
int mk_trim(mk_ctx* ctx);

/* ---- multi-GPU merge plumbing (replaces ray.get + dict sum across workers) -------------- */
/* Packed-key view of the running table for exchange over RCCL: *rows entries sorted by key
 * into caller-provided DEVICE buffers.  d_keys holds mk_words_per_key(ctx) 64-bit words per row,
 * row after row (1 for dense/hash64; 2 for hash128: {hi, lo}, hi = bases 0..31, lo = the rest,
 * both left-aligned, so (hi, lo) order is the byte order of the k-mer text); cap counts rows.
 * By-reference (exotic) rows are exchanged through mk_export_exotic / mk_import_exotic. */
int mk_export_pairs_device(mk_ctx* ctx, uint64_t* d_keys, uint64_t* d_counts, size_t cap, size_t* rows);
/* insert-add (key,count) pairs from DEVICE buffers into the running table (same layout; the same key
 * may come more than once, e.g. rows received from several ranks). */
int mk_import_pairs_device(mk_ctx* ctx, const uint64_t* d_keys, const uint64_t* d_counts, size_t rows);
int mk_export_exotic(mk_ctx* ctx, uint8_t* kmers, uint64_t* counts, size_t cap, size_t* rows);
int mk_import_exotic(mk_ctx* ctx, const uint8_t* kmers, const uint64_t* counts, size_t rows);
int mk_words_per_key(const mk_ctx* ctx);
/* Add every row of src's running table into dst's (both on the same GPU, same alphabet, k and
 * canonical mode); src is left unchanged.  Lets a host deal the chunks of one sample to several
 * contexts (= HIP streams) that count concurrently and sum them at the end, on the device. */
int mk_merge_from(mk_ctx* dst, mk_ctx* src);

/* Several contexts of ONE GPU, one running table (round 4): from now on the count kernels of ctx put the survivors of its
 * chunks straight into owner's table (fused launches, one-word keys; whatever ctx still merges on its own -- a new
 * context's first chunk, rows kept as text -- stays in ctx's table).  Sum as before at the end: mk_merge_from(owner, ctx)
 * now finds little to add.  owner == NULL: ctx goes back to its own table.  Same GPU, alphabet, k, canonical mode; one
 * level deep.  The caller orders mk_reset(owner) against the sharers' chunks (reset the owner first). */
int mk_share_table(mk_ctx* ctx, mk_ctx* owner);

/* ---- one process, several GPUs: the Ray fan-out of a sample's chunks over workers and the dict sum of their
 *      results (bin/mercat2.py:119-127, 336-339) with the workers being the GPUs of one node ------------- */
/* Equal key ranges: bounds[i-1] = first key (first 64-bit word of the packed key, key_bits wide: bits*k for
 * one-word keys, 64 for two-word keys) owned by owner i, i = 1..n-1; owner 0 starts at 0.  Host helper. */
int mk_owner_bounds(int key_bits, int n, uint64_t* bounds);
/* The order in which to create contexts for ndev devices with `streams` contexts each, so that mk_count_file's
 * rule "chunk i goes to ctxs[i mod nctx]" sends chunk i to device devices[i mod ndev] and successive
 * chunks of one device to its different streams: ctx_device[j] = devices[j mod ndev], j < ndev*streams.  Host helper. */
int mk_plan_contexts(const int* devices, int ndev, int streams, int* ctx_device);
/* The rows of the running table grouped by owner (owner of a row = number of bounds <= the first word of its
 * key; n owners, n-1 ascending bounds), written to the DEVICE buffer d_rows as interleaved rows of
 * mk_words_per_key()+1 words {key word(s), count}, owner after owner; counts[j] (HOST, n values) = rows of owner j.
 * One pass for the histogram, one for the rows: no sort.  d_rows == NULL: only the counts.  The table is unchanged.
 * Dense bins travel as {bin, count}; rows kept as text are not included (mk_export_exotic). */
int mk_bucket_rows_device(mk_ctx* ctx, const uint64_t* bounds, int n, uint64_t* d_rows, size_t cap_rows, uint64_t* counts);
/* insert-add interleaved rows (that layout) from a DEVICE buffer of this context's GPU into the running table. */
int mk_import_rows_device(mk_ctx* ctx, const uint64_t* d_rows, size_t rows);

/* About one in `stride` rows of the running table -- the first word of their keys, in no order -- into the HOST
 * buffer out (cap values; *n = how many): the sample owner bounds with about equal rows per owner are made of
 * (SURVEY 8e "optionally sampled splitters").  The table is hashed, so every stride-th slot is a uniform sample. */
int mk_sample_keys(mk_ctx* ctx, size_t stride, uint64_t* out, size_t cap, size_t* n);
/* Dense mode (k * bits <= 15): the bins as one array of nbins = 4^k / 32^k counts, copied out to (store = 0) or in
 * from (store != 0) a DEVICE buffer of this context's GPU -- several GPUs sum dense tables with one reduce of that
 * array instead of exchanging rows. */
int mk_dense_bins_device(mk_ctx* ctx, uint64_t* d_bins, size_t nbins, int store);

#define MK_MERGE_RANGES 0   /* afterwards ctxs[i] holds exactly the rows of key range i: the concatenation of the
                               contexts' sorted exports, in order, is the sorted table (mk_*_multi below)        */
#define MK_MERGE_GATHER 1   /* afterwards ctxs[0] holds every row and the others are empty                        */
#define MK_MERGE_BALANCED 2 /* (with RANGES) owner bounds from a sample of the keys (about equal rows per owner)
                               instead of equal key ranges                                                        */
#define MK_MERGE_RCCL 4     /* the segments travel by grouped ncclSend / ncclRecv (RCCL over xGMI; one communicator per
                               GPU, made on first use) instead of peer copies; MK_ERR_UNSUPPORTED when librccl.so cannot
                               be loaded                                                                          */
typedef struct mk_merge_stats_t {
  uint64_t rows_in;      /* rows of all contexts before the merge (the same key counted once per context) */
  uint64_t rows_out;     /* rows of all contexts after it (distinct keys)                                  */
  uint64_t rows_moved;   /* rows copied between different contexts                                          */
  uint64_t bytes_moved;  /* ... in bytes                                                                    */
  uint64_t max_owned;    /* rows of the fullest owner afterwards                                            */
  int32_t contexts, devices;
  int32_t peer_direct;   /* device pairs with direct peer access (xGMI) among the pairs that exchanged rows */
  int32_t rccl;          /* 1: the rows travelled over RCCL (MK_MERGE_RCCL)                                  */
  double s_bucket, s_copy, s_import, s_total; /* wall seconds of the phases */
} mk_merge_stats_t;
/* Sum the running tables of n contexts (same alphabet, k, canonical mode; on n different GPUs, or several on one)
 * in this one process: every context groups its rows by owner (mk_bucket_rows_device), the segments go straight
 * to their owners' GPUs (peer copies: each pair of GPUs has its own xGMI link, all pairs at once), every owner
 * insert-adds what it received.  Rows kept as text end up in ctxs[0].  st may be NULL. */
int mk_merge_devices(mk_ctx* const* ctxs, int n, int flags, mk_merge_stats_t* st);
/* sorted(kmers.items()) of a table spread over n contexts by key range (after MK_MERGE_RANGES): every context
 * sorts its own range on its own GPU, at once; the host concatenates in order.  Same outputs as mk_export_size /
 * mk_export / mk_write_tsv.  MK_ERR_STATE if the contexts' ranges are not ascending and disjoint. */
int mk_export_size_multi(mk_ctx* const* ctxs, int n, size_t* rows);
int mk_export_multi(mk_ctx* const* ctxs, int n, uint8_t* kmers, uint64_t* counts, size_t rows_cap);
int mk_write_tsv_multi(mk_ctx* const* ctxs, int n, const char* path, const char* basename, size_t* rows);

/* Drop the rows of the running table whose count is below min_count.  For a sample that is ONE chunk
 * (one filter unit, lib/mercat2_kmers.py:73-76) but was counted in pieces without a filter -- its records
 * split over several GPUs -- and merged: the filter comes after the merge. */
int mk_filter_min(mk_ctx* ctx, uint64_t min_count);

/* ---- statistics / profiling ----------------------------------------------------------- */
int mk_set_profiling(mk_ctx* ctx, int on);
int mk_get_stats(mk_ctx* ctx, mk_stats_t* out);
int mk_reset_stats(mk_ctx* ctx);

/* ---- host helpers (no GPU needed) ------------------------------------------------------- */
/* Virtual Chunker (lib/mercat2_Chunker.py:39-59): byte offsets into `text` (decompressed file
 * bytes) at which the reference would start chunk 1, 2, ...: a line that contains '>' starts
 * a new chunk once the newline-normalised bytes written to the current chunk are >=
 * chunksize.  cuts[0..*ncuts) ascending; chunk i is [cuts[i-1], cuts[i]) with cuts[-1] = 0
 * and cuts[ncuts] = n.  Returns MK_ERR_RANGE (and the needed size in *ncuts) if cap is short. */
int mk_chunk_cuts(const uint8_t* text, size_t n, uint64_t chunksize, uint64_t* cuts, size_t cap, size_t* ncuts);
/* The same cuts computed by the streaming scanner mk_count_file uses, with the text handed over
 * `block` bytes at a time (a self-check for tests: MK_ERR_STATE if the scanner lost, repeated or
 * misplaced a byte). */
int mk_stream_cuts(const uint8_t* text, size_t n, uint64_t chunksize, size_t block, uint64_t* cuts, size_t cap,
                   size_t* ncuts);
/* Where mk_count_file cuts ONE filter unit (a file below the chunk size) that it spreads over several GPUs: pieces
 * of at least `piece` bytes, each ending where a record starts -- a line whose first non-blank byte is '>', what
 * find_kmers takes for a header (lib/mercat2_kmers.py:51-52); a line that merely contains '>' is not one.
 * Through the streaming scanner, `block` bytes at a time, self-checked like mk_stream_cuts. */
int mk_record_cuts(const uint8_t* text, size_t n, uint64_t piece, size_t block, uint64_t* cuts, size_t cap, size_t* ncuts);
/* The file reader's own gzip decoder (csrc/mk_inflate.h) over a whole '.gz' file held in memory,
 * producing `block` bytes per step as the reader does, every member's CRC-32 and length checked.
 * A self-check for tests (against zlib): out must hold the whole text (cap bytes).
 * MK_ERR_IO: corrupt or not gzip, MK_ERR_RANGE: truncated, MK_ERR_NOMEM: cap too small. */
int mk_gunzip(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, size_t block, size_t* written, int* members);
/* The same through the parallel decoder (csrc/mk_pgunzip.h): `threads` pieces of `piece_bytes` compressed
 * bytes are decoded at once, each from a block start found by search, and stitched together. */
int mk_gunzip_parallel(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, int threads, size_t piece_bytes,
                       size_t* written, int* members);
/* zlib's crc32(seed, p, n) as the gzip reader computes it (carry-less multiplies; csrc/mk_crc32.h). */
uint32_t mk_crc32_of(const uint8_t* p, size_t n, uint32_t seed);
/* removeN (lib/mercat2_fasta.py:53-119 with split_sequenceN :21-49), the rewrite MerCat2 applies to every
 * nucleotide FASTA before counting (bin/mercat2.py:239-244): records holding 'N' are cut at every run of N
 * into ">{name}_{i} {info}" pieces re-wrapped at 80 columns, the others are written back line by line;
 * toupper != 0 upper-cases sequence lines on output.  text = the (decompressed) file; *out receives a
 * malloc'ed buffer with the cleaned text (release it with mk_free), st the figures GC content is made of.
 * MK_ERR_RANGE: a record to be split has an empty header (the reference raises IndexError).  A sequence to be
 * split that holds blanks, tabs or hyphens is wrapped by the standard library's word rules, as textwrap.wrap does it
 * for the reference (restated in csrc/mk_host.cpp, checked against textwrap itself: mk_textwrap).  Header lines may
 * hold any bytes; a byte >= 0x80 in a SEQUENCE line is MK_ERR_NON_ASCII (st->unsupported_record names the record),
 * as in the counting calls. */
typedef struct mk_clean_stats_t {
  uint64_t gc_count;      /* 'G' + 'C' as the reference counts them (header lines of split records included) */
  uint64_t total_length;  /* the length it divides by: GC content = 100 * gc_count / total_length            */
  uint64_t records, split_records, pieces, n_runs;
  int64_t unsupported_record; /* -1, or the index of the first record this function does not rewrite */
} mk_clean_stats_t;
int mk_remove_n(const uint8_t* text, size_t n, int toupper, uint8_t** out, size_t* out_len, mk_clean_stats_t* st);
void mk_free(void* p);
/* textwrap.wrap(text, width) of CPython 3.10 for ASCII text, as mk_remove_n applies it to the pieces of a split
 * sequence: the lines, each followed by '\n', in a malloc'ed buffer (mk_free).  A self-check for tests. */
int mk_textwrap(const uint8_t* text, size_t n, size_t width, uint8_t** out, size_t* out_len);
/* Deterministic synthetic reads: genome of `genome_len` iid ACGT from
 * splitmix64(genome_seed); `reads` reads of `read_len` from uniform starts, reverse-complemented
 * on a coin flip, per-base substitution with probability sub_ppm/1e6, all from
 * splitmix64(read_seed); records ">r{i}\n{seq}\n" with i starting at first_index.
 * Writes at most cap bytes to out (host) and the size to *written (call with out = NULL to size). */
int mk_synth_reads(uint64_t genome_len, uint64_t genome_seed, uint64_t reads, uint32_t read_len,
                   uint64_t read_seed, uint32_t sub_ppm, uint64_t first_index, uint8_t* out, size_t cap,
                   size_t* written);
const char* mk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MERCAT_HIP_H */
