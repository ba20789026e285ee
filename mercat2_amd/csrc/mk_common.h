// mk_common.h -- shared declarations of the MI355X k-mer engine (internal, not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <mutex>
#include <shared_mutex>
#include <vector>
#include "../../include/mercat_hip.h"

// ------------------------------------------------------------------------------------------
// Device data layout (all in HBM, owned by the context)
//
//   raw      u8[n]            the chunk's FASTA bytes as fed
//   seq      u8[<=n]          parsed stream: kept sequence characters, records separated by
//                             one SEP byte ('\n' can never be a kept character)
//   codes    u64[...]         packed symbols, MSB first: 32 x 2-bit (nt) or 12 x 5-bit (aa,
//                             bits 63..4) per word
//   bad      u64[...]         1 bit per symbol (LSB first): symbol outside the alphabet, a
//                             separator, or beyond the end of seq
//   table    Slot[2^m]        open-addressed, linear probing, 16-byte slots {key, count}
//   table128 Slot128[2^m]     32-byte slots {hi, lo, count, pad} for 33..64-mers
//   bins     u64[4^k | 32^k]  dense histogram (small k)
// ------------------------------------------------------------------------------------------

// Smallest nucleotide k that takes the super-k-mer partition (mk_skmer.hip): k - 10 minimizer candidates per window.
// Round 3: 12 (measured on an S2 chunk, Gbases/s against the 8-byte-key partition: k = 12 50 / 42, 14 74 / 47, 16 90 / 46,
// 17 96 / 21); the environment variable MK_SK_MIN_K (tests / A-B runs) may raise it back to 18, the round-2 threshold.
#ifndef MK_SK_MIN_K
#define MK_SK_MIN_K 12
#endif
#define MK_SEP 0x0Au
#define MK_EMPTY 0xFFFFFFFFFFFFFFFFull

struct __attribute__((aligned(16))) MkSlot {
  unsigned long long key;
  unsigned long long cnt;
};

// Running table of two-word keys (nucleotide 33..64-mers: hi = bases 0..31, lo = the rest, left-aligned).
// There is no 128-bit compare-and-swap, so the COUNT word is the slot's state: 0 = free,
// MK_LOCK128 = claimed, key words being written, anything else = the count (key words final).
#define MK_LOCK128 0xFFFFFFFFFFFFFFFFull
struct __attribute__((aligned(32))) MkSlot128 {
  unsigned long long hi;
  unsigned long long lo;
  unsigned long long cnt;
  unsigned long long pad;
};

// Device-side scalars of one chunk (one struct in HBM, read back once per chunk).
struct MkChunkInfo {
  unsigned long long seq_len;       // bytes in seq (symbols + separators)
  unsigned long long symbols;       // kept sequence characters
  unsigned long long non_ascii;     // kept (sequence) bytes >= 0x80; header lines may hold any bytes
  unsigned long long bad_symbols;   // kept characters outside the alphabet
  unsigned long long windows;       // windows counted by the packed/dense path
  unsigned long long exotic;        // windows counted by the by-reference path
  unsigned long long survivors;     // packed entries passing min_count (this chunk)
  unsigned long long survivors_ref; // by-reference entries passing min_count
  unsigned long long side;          // count of the one key that equals MK_EMPTY (all-T 32-mer)
  unsigned long long new_rows;      // rows added to the running table by this chunk
  unsigned long long new_rows_ref;
  unsigned long long distinct;      // distinct packed keys seen in this chunk (partitioned path)
  unsigned long long errors;        // non-zero: a kernel hit a condition it cannot handle
  unsigned long long parse_fallback;  // fast parser saw a blank in a sequence line: re-parse generally
  unsigned long long records;       // super-k-mer records written (partitioned nt path)
  unsigned long long part_overflow; // a bucket region sized from a sampled histogram was too small: partition again, exactly
  unsigned long long spilled;       // fused upsert (mk_skcount.hip): survivors written to the spill list instead of the running table
};

enum MkMode { MK_MODE_DENSE = 0, MK_MODE_HASH64 = 1, MK_MODE_HASH128 = 2, MK_MODE_BYREF = 3 };
enum MkKernelId { MK_K_PARSE = 0, MK_K_PACK, MK_K_COUNT, MK_K_EXOTIC, MK_K_FILTER, MK_K_EXPORT, MK_K_PART, MK_K_NUM };

// 64-bit finaliser (splitmix64 / murmur3 style): bijective, mixes every input bit into every
// output bit -- used to pick the home slot of a packed key.
__host__ __device__ static inline unsigned long long mk_mix64(unsigned long long x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

#define MK_POLY_B 0x9E3779B97F4A7C15ull  // odd multiplier of the rolling polynomial hash

struct MkDevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

// Simple growable device buffer helpers (implemented in mk_api.cpp)
struct mk_ctx;
int mk_buf_reserve(mk_ctx* c, MkDevBuf& b, size_t bytes, bool keep = false);
// mk_api.hip, for mk_ingest.hip: append host bytes to the open chunk without waiting for the copy
// (the source must stay untouched until the context's stream has passed it) unless wait is set
int mk_feed_host_async(mk_ctx* c, const uint8_t* p, size_t n, bool wait);
int mk_reserve_raw(mk_ctx* c, size_t bytes);

struct MkEventPair {
  hipEvent_t a, b;
  int id;
};

struct mk_ctx {
  int device = 0;
  int alphabet = 0;
  int k = 0;
  int bits = 0;           // bits per symbol (2, 5, 0 for raw)
  int syms_per_word = 0;  // 32, 12
  int mode = 0;
  hipStream_t stream = nullptr;
  std::string err;
  bool in_chunk = false;
  bool profile = false;

  // chunk staging
  MkDevBuf raw;
  size_t raw_len = 0;
  MkDevBuf seq, codes, bad;
  MkDevBuf tile_maps;   // parse scratch
  MkDevBuf info;        // MkChunkInfo on device
  MkChunkInfo* h_info = nullptr;  // pinned host copy (two structs: [1] receives the read-back that is not waited for)
  bool pending_rows = false;      // h_info[1] holds (or will hold, once the stream has passed the copy) the row totals
                                  // of the last chunk's merge, not yet added to run_rows / run128_rows / run_ref_rows
  int use_speculation = 1;        // one read-back per chunk on the partitioned paths (mk_api.hip: process_chunk_fast)

  // chunk tables
  MkDevBuf ctab;        // MkSlot[] (hash64) / MkSlot128[] (hash128) / u64 bins (dense)
  size_t ctab_slots = 0;
  MkDevBuf rtab_chunk;  // by-reference chunk table MkSlot[] (key = tag|pos)
  size_t rtab_chunk_slots = 0;

  // running (merged) tables
  MkDevBuf run;         // same layout as ctab (dense: u64 bins)
  size_t run_slots = 0;
  size_t run_rows = 0;
  unsigned long long run_side = 0;  // count of the MK_EMPTY-valued key
  MkDevBuf run_ref;     // by-reference running table MkSlot[] (key = tag|arena index)
  size_t run_ref_slots = 0;
  size_t run_ref_rows = 0;
  MkDevBuf arena;       // k bytes per by-reference row
  size_t arena_rows_cap = 0;
  MkDevBuf run128;      // MkSlot128[]: packed two-word keys (mode MK_MODE_HASH128), u64 counts
  size_t run128_slots = 0;
  size_t run128_rows = 0;

  // partitioned counting (hash64): keys bucketed by hash, counted per bucket in LDS
  MkDevBuf part;        // u64 keys, bucket after bucket
  MkDevBuf part_meta;   // u64 hist[P1] | start[P1+1] | cursor[P1]
  MkDevBuf surv_keys, surv_cnts;  // (key,count) survivors of the chunk
  MkDevBuf surv_keys2;            // second key word of the survivors (33..64-mers)
  int use_superkmer2 = 1;
  int p1_log2 = 10;
  double dup_hint = 1.0;  // windows per distinct key seen in the previous chunk
  bool dup_known = false; // ... of THIS sample (mk_reset forgets it)
  int use_partition = 1;
  int use_fast_parse = 1;
  int canonical = 0;      // opt-in: count min(kmer, revcomp) (nt only)
  int use_superkmer = 1;
  int run_bucket_major = 0;  // one-word running table addressed bucket-major (mk_table.hip: RunAddr; experiment, MK_BUCKET_MAJOR=1)
  int sk_min_k = MK_SK_MIN_K;  // nucleotide k from which the super-k-mer partition is used (below: 8-byte-key partition)
  bool part_sampled = false;  // the last super-k-mer partition sized its buckets from a sample
  // bucket regions of the previous chunk kept for the next one (mk_skmer.hip): same size, same min_count, no overflow
  bool part_reuse_ok = false;
  bool part_dirty = false;    // a partition was launched and its chunk has not been seen to end well (cursors may be anywhere)
  size_t part_prev_len = 0;
  int part_prev_p1 = 0;
  int part_nseg = 1;            // regions per bucket of the last one-word partition (1, or 8: one per XCD)
  unsigned long long part_prev_minc = 0;
  int part_cooldown = 0;      // chunks that size their buckets afresh after a chunk overflowed inherited regions
  int use_reuse = 1;
  int surv_regions = 0;   // survivors of the last chunk are laid out per bucket (kstart/nsurv in part_meta)
  double nk_hint = 8.0;   // windows per super-k-mer record seen in the previous chunk
  double items_hint = 0;  // records per analysis thread (32 positions) seen in the previous chunk; 0 = not known yet
  // fused upsert (mk_skcount.hip): the count kernel puts a chunk's survivors into the running table itself
  int use_fused = 1;                  // MK_NO_FUSE=1 turns it off
  int fuse_cap = 0;                   // list entries per sweep the NEXT count launch may use (0: survivors go to their regions)
  bool fused_last = false;            // the last count launch was a fused one
  unsigned long long surv_hint = 0;   // survivors of the previous chunk of this sample
  bool surv_hint_ok = false;
  // one table for several contexts of a device (mk_share_table): the fused launches of this context upsert into the owner's
  mk_ctx* share_owner = nullptr;
  mk_ctx* fuse_target = nullptr;      // the context whose table the NEXT / last fused launch of this one upserts into (this or share_owner)
  std::vector<mk_ctx*> sharers;       // contexts whose fused launches upsert into THIS context's table
  std::shared_mutex table_mu;         // shared: a launch reads run.p / run_slots; exclusive: the table is replaced (grown) or cleared
  std::mutex rows_mu;                 // run_rows of an owner is added to by its sharers' host threads

  // export scratch
  MkDevBuf ex_keys, ex_cnts, ex_keys2, ex_cnts2, ex_tmp;
  MkDevBuf ex128, ex128_out;  // two-word rows: compacted {hi, lo, count} + sort scratch; sorted rows for the host

  // clean mode (mk_clean.hip): the raw file is counted as removeN would leave it
  int clean_mode = 0, clean_upper = 0;
  MkDevBuf clean_meta, clean_runs;
  unsigned long long* h_clean = nullptr;   // pinned: the 8 meta words of the last chunk
  unsigned long long clean_n_runs = 0, clean_n_bytes = 0, clean_gc = 0, clean_symbols = 0, clean_raw = 0, clean_headers = 0;
  unsigned long long clean_last_runs = 0;  // runs of the last chunk (listed in clean_runs up to its capacity)

  // multi-GPU merge staging (mk_multi.hip): rows grouped by owner going out, rows received from the peers,
  // owner bounds + histogram + cursors
  MkDevBuf xfer_out, xfer_in, xfer_meta;

  // pinned block ring of mk_count_file (mk_ingest.hip), kept between files
  void* ingest_ring = nullptr;
  size_t ingest_ring_bytes = 0;
  // two blocks the TSV text is copied out through (mk_write_tsv): ordinary (cached) memory, registered with the driver --
  // the CPU READS these, and it read hipHostMalloc'ed memory at 3.5 GB/s
  void* tsv_pin = nullptr;
  size_t tsv_pin_bytes = 0;

  // stats
  mk_stats_t st{};
  mk_export_stats_t ex_st{};  // of the last mk_export / mk_write_tsv (mk_export_stats)
  std::vector<MkEventPair> events;
  std::vector<hipEvent_t> event_pool;
};

#define MK_HIP(call)                                                                         \
  do {                                                                                       \
    hipError_t e__ = (call);                                                                 \
    if (e__ != hipSuccess) {                                                                 \
      c->err = std::string(#call) + ": " + hipGetErrorString(e__);                           \
      return MK_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

// ---- kernel launchers (each in its own translation unit) ---------------------------------
// parse: raw[n] -> seq, info (seq_len, symbols, non_ascii)
int mk_launch_parse(mk_ctx* c, const uint8_t* d_raw, size_t n);
// fast parse (mk_fparse.hip): same output; sets info.parse_fallback when its assumption fails
int mk_launch_fparse(mk_ctx* c, const uint8_t* d_raw, size_t begin, size_t len, bool fuse_pack_nt, bool write_seq = true);
// pack: seq -> codes, bad (+ info.bad_symbols)
int mk_launch_pack(mk_ctx* c, size_t seq_cap);
// counting
int mk_launch_count_dense(mk_ctx* c, size_t seq_cap);
int mk_launch_count_hash64(mk_ctx* c, size_t seq_cap);
int mk_launch_count_byref(mk_ctx* c, size_t seq_cap, bool exotic_only);
// nt 33..64-mers without bad symbols: by reference with packed hashing/compare (mode MK_MODE_HASH128)
int mk_launch_count_ref128(mk_ctx* c, size_t seq_len);
// partitioned hash64 path: windows -> hash buckets -> per-bucket LDS tables -> survivors (count >= min_count)
int mk_launch_count_partitioned(mk_ctx* c, size_t seq_len, uint64_t min_count);
// super-k-mer form of the same (nt, 18 <= k <= 32): mk_skmer.hip
int mk_launch_count_superkmer(mk_ctx* c, size_t seq_len, uint64_t min_count, bool exact = false);
// mk_skmer.hip: bucket regions (records and survivors) from an exact or sampled histogram, in one kernel
bool mk_part_inherit(mk_ctx* c, size_t seq_len, int p1_log2, uint64_t min_count, bool sampled, bool exact);  // mk_skmer.hip
void mk_launch_sk_scan(mk_ctx* c, const unsigned long long* hist, const unsigned long long* khist, unsigned long long* start,
                       unsigned* cursor, unsigned long long* kstart, int p1_log2, int sample_log2, int nkmax,
                       unsigned long long surv_div, unsigned long long part_cap, unsigned long long surv_cap, float sigmas,
                       int nseg);  // nseg: regions per bucket (1, or 8: one per XCD, mk_skmer.hip)
// mk_skcount.hip: the count kernel of the one-word super-k-mer path over the bucket regions the scatter filled
int mk_launch_sk_count(mk_ctx* c, const unsigned long long* start, unsigned* cursor, const unsigned long long* kstart,
                       unsigned long long* nsurv, uint64_t min_count, int nkmax, size_t p1, bool exact, int nseg);
// mk_skcount_small.hip: the same with 512-thread workgroups and 4096-slot tables (experiment MK_CORES)
int mk_launch_sk_count_small(mk_ctx* c, const unsigned long long* start, unsigned* cursor, const unsigned long long* kstart,
                             unsigned long long* nsurv, uint64_t min_count, int nkmax, size_t p1, bool exact, int nseg);
// nt 33 <= k <= 64, two-word keys: mk_skmer2.hip; survivors {hi,lo,count} per bucket region
int mk_launch_count_superkmer2(mk_ctx* c, size_t seq_len, uint64_t min_count, bool exact = false);
int mk_launch_import_ref128_regions(mk_ctx* c, const uint64_t* hi, const uint64_t* lo, const uint64_t* cnts,
                                    const uint64_t* kstart, const uint64_t* nsurv, size_t p1);
// tables
int mk_launch_clear_slots(mk_ctx* c, MkSlot* t, size_t slots);
int mk_launch_count_survivors(mk_ctx* c, uint64_t min_count);
// mk_sort.hip: arena rows (k bytes each) in byte order; *d_order = row indices, sorted (lives in c->ex_cnts2)
int mk_sort_rows(mk_ctx* c, const uint8_t* d_arena, size_t rows, int k, uint64_t** d_order);
int mk_launch_rows_by_slot(mk_ctx* c, const uint64_t* slot_keys, const uint64_t* slot_cnts, size_t rows, uint64_t* cnt_by_row, uint64_t* d_bad);
int mk_launch_rows_gather(mk_ctx* c, const uint8_t* arena, const uint64_t* order, const uint64_t* cnt_by_row, size_t rows, int k,
                          uint8_t* out_rows, uint64_t* out_cnts);
int mk_launch_alpha(mk_ctx* c, unsigned long long* d_out);  // 16 words: see mk_alpha_k
int mk_launch_accumulate(mk_ctx* c, uint64_t min_count);
// every row of another one-word table (same device) added into c's running table; *new_rows counts the new keys
int mk_launch_merge_table64(mk_ctx* c, const MkSlot* from, size_t from_slots);
int mk_launch_rehash64(mk_ctx* c, const MkSlot* from, size_t from_slots, MkSlot* to, size_t to_slots);
int mk_launch_rehash_ref(mk_ctx* c, const MkSlot* from, size_t from_slots, MkSlot* to, size_t to_slots);
int mk_launch_import_pairs(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_counts, size_t rows, bool distinct = false);
int mk_launch_import_regions(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* kstart,
                             const uint64_t* nsurv, size_t p1, size_t survivors);
int mk_launch_import_ref(mk_ctx* c, const uint8_t* d_kmers, const uint64_t* d_counts, size_t rows);
// two-word keys (mk_table.hip): survivors {hi, lo, count} per bucket region / rows {hi, lo} interleaved -> run128
int mk_launch_import128_regions(mk_ctx* c, const uint64_t* hi, const uint64_t* lo, const uint64_t* cnts, const uint64_t* kstart,
                                const uint64_t* nsurv, size_t p1);
int mk_launch_import128_pairs(mk_ctx* c, const uint64_t* d_keys2, const uint64_t* d_counts, size_t rows);
int mk_launch_refilter64(mk_ctx* c, const MkSlot* from, MkSlot* to, size_t slots, uint64_t min_count, uint64_t* d_kept);
int mk_launch_refilter128(mk_ctx* c, const MkSlot128* from, MkSlot128* to, size_t slots, uint64_t min_count, uint64_t* d_kept);
int mk_launch_refilter_dense(mk_ctx* c, uint64_t* bins, size_t nbins, uint64_t min_count);
int mk_launch_rehash128(mk_ctx* c, const MkSlot128* from, size_t from_slots, MkSlot128* to, size_t to_slots);
int mk_launch_compact128(mk_ctx* c, const MkSlot128* t, size_t slots, uint64_t* hi, uint64_t* lo, uint64_t* cnts, size_t cap,
                         uint64_t* d_cursor);
// mk_sort.hip: rows {hi[i], lo[i], cnt[i]} -> sorted by (hi, lo): keys2_out = {hi, lo} interleaved, cnts_out; scratch = 4 * n words
int mk_sort_pairs128(mk_ctx* c, const uint64_t* hi, const uint64_t* lo, const uint64_t* cnts, size_t n, int lo_bits,
                     uint64_t* scratch, uint64_t* keys2_out, uint64_t* cnts_out);
// export helpers
int mk_launch_compact(mk_ctx* c, const MkSlot* t, size_t slots, uint64_t* d_keys, uint64_t* d_counts, size_t cap,
                      uint64_t* d_cursor);
int mk_sort_pairs(mk_ctx* c, const uint64_t* keys_in, const uint64_t* vals_in, uint64_t* keys_out, uint64_t* vals_out,
                  size_t n, int key_bits);

// mk_api.hip, for mk_multi.hip
int mk_settle(mk_ctx* c);                      // fold row totals that were read back without waiting
int mk_pull_info(mk_ctx* c);                   // MkChunkInfo -> h_info, stream idle afterwards
int mk_grow_run(mk_ctx* c, size_t more_rows);  // room in the packed running table for more_rows further keys
// mk_table.hip: interleaved rows {key word(s), count} -> running table (dense: {bin, count})
int mk_launch_import_rows(mk_ctx* c, const uint64_t* d_rows, size_t rows);

// mk_clean.hip
int mk_launch_clean_pre(mk_ctx* c, uint8_t* d_raw, size_t n);
int mk_launch_clean_post(mk_ctx* c, size_t seq_cap);

void mk_prof_begin(mk_ctx* c, int id);
void mk_prof_end(mk_ctx* c);
