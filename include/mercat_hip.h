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
} mk_stats_t;

/* ---- lifetime ------------------------------------------------------------------------- */
/* One context per GPU and per (alphabet, k).  Replaces the per-task state of
 * countKmers/find_kmers (bin/mercat2.py:112-114). */
int mk_create(int device, int alphabet, int k, mk_ctx** out);
void mk_destroy(mk_ctx* ctx);
const char* mk_last_error(const mk_ctx* ctx);
/* Forget the running (merged) table: start the next sample (run_mercat2's `kmers = dict()`,
 * bin/mercat2.py:117). */
int mk_reset(mk_ctx* ctx);
/* Opt-in extension, NOT reference behaviour (the reference counts forward-strand substrings,
 * lib/mercat2_kmers.py:56-60): with on != 0 every ACGT-only window is counted under
 * min(kmer, reverse-complement(kmer)).  Windows holding other characters keep their own text.
 * Nucleotide alphabet only; call before the first chunk of a sample (or right after mk_reset).
 * Nucleotide k <= 64.  One-word keys (k <= 32) file a window under min(11-mer, its reverse complement) of its
 * minimizer; two-word keys (33 <= k <= 64) under the smallest canonical 11-mer among the candidates of the window's
 * first and last 32 bases -- the set the two strands share.  k > 64 and the raw alphabet count text, which has no
 * complement: MK_ERR_ARG. */
int mk_set_canonical(mk_ctx* ctx, int on);

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
  int32_t pad_;
  double s_wait_io;    /* seconds the dispatching thread waited for file blocks              */
  double s_wait_gpu;   /* seconds it waited for a context to finish its previous chunk       */
  double s_total;      /* wall seconds of the call                                           */
} mk_file_stats_t;
/* Reads `path` (gzip iff its name ends in ".gz", as the reference decides), splits it as
 * Chunker(path, dest, chunk_bytes, '>') would iff its on-disk size is >= chunk_bytes > 0
 * (chunk_bytes == 0: never), counts every chunk with its own min_count filter and adds the
 * survivors to the running table of ctxs[0].  With nctx > 1 (same device, alphabet, k) the chunks
 * are dealt to the contexts in turn and counted concurrently with the reading; the other
 * contexts' tables are added into ctxs[0] and reset before the call returns.  threads = reader
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
 * MK_ERR_RANGE: a record to be split has an empty header (the reference raises IndexError).  When a
 * sequence to be split holds a blank, tab or hyphen (textwrap would treat it as a word break) nothing is
 * produced and st->unsupported_record names the record: the Python host layer then rewrites that file. */
typedef struct mk_clean_stats_t {
  uint64_t gc_count;      /* 'G' + 'C' as the reference counts them (header lines of split records included) */
  uint64_t total_length;  /* the length it divides by: GC content = 100 * gc_count / total_length            */
  uint64_t records, split_records, pieces, n_runs;
  int64_t unsupported_record; /* -1, or the index of the first record this function does not rewrite */
} mk_clean_stats_t;
int mk_remove_n(const uint8_t* text, size_t n, int toupper, uint8_t** out, size_t* out_len, mk_clean_stats_t* st);
void mk_free(void* p);
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
