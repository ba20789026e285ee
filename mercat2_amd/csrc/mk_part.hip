// mk_part.hip -- partitioned k-mer counting: the fast form of the hash64 path.
//
// Same job as the reference's dict update `kmerlist[seq[i:i+k]] += 1` followed by the
// per-file filter `count >= min_count` (lib/mercat2_kmers.py:56-60, 73-76), organised for the
// memory system of MI355X instead of as one random read-modify-write per window:
//
//   1 mk_part_hist     every valid window -> key -> bucket = top bits of mix64(key);
//                      per-workgroup LDS histogram, flushed with contiguous atomics
//   2 mk_part_scan     exclusive scan of the P1 bucket sizes (one workgroup)
//   3 mk_part_scatter  per tile of 8192 windows: rank of every key inside its (tile,bucket)
//                      run by an LDS counter, ONE contiguous block of cursor atomics per tile
//                      to reserve the runs, then the keys are stored run by run (the stores of a
//                      run come from one workgroup back to back and merge in its XCD's L2)
//   4 mk_part_count    one workgroup per bucket: stream the bucket's keys (coalesced 16-byte
//                      loads) into an open-addressed table in LDS (ds_cmpst_b64 claim +
//                      ds_add_u32), emit the entries with count >= min_count, clear, next
//                      sub-range. A bucket holding more distinct keys than the LDS table takes
//                      is split by further hash bits (the bucket is re-read, from L2), and a
//                      sub-range that still overflows is halved again -- any input works.
//   5 (mk_table.hip)   survivors -> running table.
//
// All counting atomics are LDS atomics; HBM sees the packed symbols once or twice, each key
// written once and read once (plus re-reads served by L2), and the survivors.
#include "mk_common.h"
#include "mk_device.h"
#include <cstdlib>

#define PART_THREADS 256
#define PART_MAX_P1 4096
#define CNT_SLOTS 4096                      // LDS table slots per workgroup
#define CNT_LOADCAP (CNT_SLOTS * 3 / 4)     // distinct keys accepted per sub-range
#define CNT_THREADS 256
#define SUB_BITS 24                         // hash bits available for splitting a bucket

static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// --------------------------------------------------------------------------- 1 histogram
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(PART_THREADS) void mk_part_hist_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                               MkChunkInfo* __restrict__ info, u64* __restrict__ hist,
                                                               int p1_log2, int k, size_t nthreads_total) {
  __shared__ unsigned lh[PART_MAX_P1];
  constexpr int R = SPW * WPT;
  const unsigned p1 = 1u << p1_log2;
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const size_t seq_len = info->seq_len;
  const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
  u64 mine = 0, side = 0;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nthreads_total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t p0 = t * R;
    if (p0 >= seq_len) break;
    u64 w[WPT + 1];
#pragma unroll
    for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
    const u64 badw = bad_window(bad, p0);
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        if (((badw >> (i * SPW + s)) & kmask) == 0) {
          const u64 key = window_key<BITS, SPW>(w[i], w[i + 1], s, k);
          ++mine;
          if (key == MK_EMPTY) ++side;
          else atomicAdd(&lh[(unsigned)(mk_mix64(key) >> (64 - p1_log2))], 1u);
        }
      }
    }
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < p1; b += blockDim.x) {
    const unsigned v = lh[b];
    if (v) atomicAdd(&hist[b], (u64)v);
  }
  wave_add(&info->windows, mine);
  wave_add(&info->side, side);
}

// -------------------------------------------------------------------------------- 2 scan
__global__ __launch_bounds__(1024) void mk_part_scan_k(const u64* __restrict__ hist, u64* __restrict__ start,
                                                       u64* __restrict__ cursor, int p1_log2) {
  __shared__ u64 sums[1024];
  const unsigned p1 = 1u << p1_log2;
  const unsigned per = (p1 + 1023) / 1024;
  const unsigned lo = threadIdx.x * per;
  u64 acc = 0;
  for (unsigned i = lo; i < lo + per && i < p1; ++i) acc += hist[i];
  sums[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 run = 0;
    for (int i = 0; i < 1024; ++i) {
      const u64 v = sums[i];
      sums[i] = run;
      run += v;
    }
    start[p1] = run;
  }
  __syncthreads();
  u64 run = sums[threadIdx.x];
  for (unsigned i = lo; i < lo + per && i < p1; ++i) {
    start[i] = run;
    cursor[i] = run;
    run += hist[i];
  }
}

// ----------------------------------------------------------------------------- 3 scatter
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(PART_THREADS) void mk_part_scatter_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                  const MkChunkInfo* __restrict__ info,
                                                                  u64* __restrict__ cursor, u64* __restrict__ part,
                                                                  int p1_log2, int k, size_t ntiles) {
  __shared__ unsigned lh[PART_MAX_P1];
  __shared__ u64 gbase[PART_MAX_P1];
  constexpr int R = SPW * WPT;
  const unsigned p1 = 1u << p1_log2;
  const size_t seq_len = info->seq_len;
  const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const size_t t = tile * PART_THREADS + threadIdx.x;
    const size_t p0 = t * R;
    u64 w[WPT + 1];
    u64 badw = ~0ull;
    unsigned short rank[R];
    if (p0 < seq_len) {
#pragma unroll
      for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
      badw = bad_window(bad, p0);
    } else {
#pragma unroll
      for (int i = 0; i <= WPT; ++i) w[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        rank[i * SPW + s] = 0;
        if (((badw >> (i * SPW + s)) & kmask) == 0) {
          const u64 key = window_key<BITS, SPW>(w[i], w[i + 1], s, k);
          if (key != MK_EMPTY)
            rank[i * SPW + s] = (unsigned short)atomicAdd(&lh[(unsigned)(mk_mix64(key) >> (64 - p1_log2))], 1u);
        }
      }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < p1; b += blockDim.x) {
      const unsigned v = lh[b];
      gbase[b] = v ? atomicAdd(&cursor[b], (u64)v) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
#pragma unroll
      for (int s = 0; s < SPW; ++s) {
        if (((badw >> (i * SPW + s)) & kmask) == 0) {
          const u64 key = window_key<BITS, SPW>(w[i], w[i + 1], s, k);
          if (key != MK_EMPTY) part[gbase[(unsigned)(mk_mix64(key) >> (64 - p1_log2))] + rank[i * SPW + s]] = key;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------- 4 count
// Sub-range (s, i) of a bucket: keys whose SUB_BITS-bit field f = (h >> sub_shift) & mask has
// its top s bits equal to i. Depth-first walk over the binary tree of sub-ranges: a sub-range
// whose distinct keys do not fit the LDS table is replaced by its two halves.
__global__ __launch_bounds__(CNT_THREADS) void mk_part_count_k(const u64* __restrict__ part, const u64* __restrict__ start,
                                                               MkChunkInfo* __restrict__ info, u64 min_count,
                                                               u64* __restrict__ out_keys, u64* __restrict__ out_cnts,
                                                               int p1_log2, double dup_hint) {
  __shared__ u64 tkey[CNT_SLOTS];
  __shared__ unsigned tcnt[CNT_SLOTS];
  __shared__ unsigned s_distinct, s_overflow;
  const u64 lo = start[blockIdx.x], hi = start[blockIdx.x + 1];
  const u64 n = hi - lo;
  if (n == 0) return;
  if (n >> 32) {  // LDS counters are 32-bit
    if (threadIdx.x == 0) atomicAdd(&info->errors, 1ull);
    return;
  }
  const int sub_shift = 64 - p1_log2 - SUB_BITS;  // field sits right below the bucket bits
  for (unsigned i = threadIdx.x; i < CNT_SLOTS; i += blockDim.x) { tkey[i] = MK_EMPTY; tcnt[i] = 0; }
  if (threadIdx.x == 0) { s_distinct = 0; s_overflow = 0; }
  __syncthreads();
  // starting depth from the expected number of distinct keys in this bucket
  int s0 = 0;
  {
    const double expect = (double)n / (dup_hint > 1.0 ? dup_hint : 1.0);
    while (s0 < SUB_BITS && expect / (double)(1u << s0) > 0.8 * CNT_LOADCAP) ++s0;
    if (n <= CNT_LOADCAP) s0 = 0;
  }
  int s = s0;
  unsigned idx = 0;
  u64 distinct_total = 0;
  const u64* __restrict__ src = part + lo;
  for (;;) {
    // ---- fill the table with the keys of sub-range (s, idx)
    for (u64 j = threadIdx.x; j < n; j += CNT_THREADS) {
      const u64 key = src[j];
      const u64 h = mk_mix64(key);
      if (s && (unsigned)(((h >> sub_shift) & ((1u << SUB_BITS) - 1)) >> (SUB_BITS - s)) != idx) continue;
      unsigned slot = (unsigned)h & (CNT_SLOTS - 1);
      bool done = false;
      for (int probe = 0; probe < CNT_SLOTS; ++probe) {
        u64 cur = tkey[slot];
        if (cur == MK_EMPTY) {
          cur = atomicCAS(&tkey[slot], MK_EMPTY, key);
          if (cur == MK_EMPTY) {
            cur = key;
            if (atomicAdd(&s_distinct, 1u) >= CNT_LOADCAP) s_overflow = 1;
          }
        }
        if (cur == key) { atomicAdd(&tcnt[slot], 1u); done = true; break; }
        slot = (slot + 1) & (CNT_SLOTS - 1);
      }
      if (!done) s_overflow = 1;
      if (*(volatile unsigned*)&s_overflow) break;  // somebody saw the table fill up: this attempt is void
    }
    __syncthreads();
    const bool over = s_overflow != 0;
    const unsigned found = s_distinct;
    // ---- emit (when complete) and clear
    for (unsigned base = 0; base < CNT_SLOTS; base += CNT_THREADS) {
      const unsigned i = base + threadIdx.x;
      const u64 key = tkey[i];
      const unsigned c = tcnt[i];
      const bool keep = !over && key != MK_EMPTY && (u64)c >= min_count;
      const u64 m = __ballot(keep);
      if (m) {
        const int lane = threadIdx.x & 63;
        u64 at = 0;
        if (lane == 0) at = atomicAdd(&info->survivors, (u64)__popcll(m));
        at = __shfl(at, 0);
        if (keep) {
          const u64 pos = at + __popcll(m & ((1ull << lane) - 1));
          out_keys[pos] = key;
          out_cnts[pos] = c;
        }
      }
      tkey[i] = MK_EMPTY;
      tcnt[i] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_distinct = 0; s_overflow = 0; }
    __syncthreads();
    // ---- next sub-range (uniform across the workgroup)
    if (over) {
      if (s >= SUB_BITS) {  // cannot split further: report, never give a wrong count silently
        if (threadIdx.x == 0) atomicAdd(&info->errors, 1ull);
        return;
      }
      s += 1;
      idx <<= 1;
    } else {
      distinct_total += found;
      while (s > s0 && (idx & 1u)) { idx >>= 1; --s; }
      if (s == s0) {
        ++idx;
        if (idx >= (1u << s0)) break;
      } else {
        ++idx;
      }
    }
  }
  if (threadIdx.x == 0) atomicAdd(&info->distinct, distinct_total);
}

// ------------------------------------------------------------------------------ launcher
int mk_launch_count_partitioned(mk_ctx* c, size_t seq_len, uint64_t min_count) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  // bucket count: aim at ~16K keys per bucket, between 256 and PART_MAX_P1
  int p1_log2 = 8;
  while (p1_log2 < 12 && (seq_len >> p1_log2) > 16384) ++p1_log2;
  if (const char* e = getenv("MK_P1_LOG2")) { int v = atoi(e); if (v >= 4 && v <= 12) p1_log2 = v; }
  c->p1_log2 = p1_log2;
  const size_t p1 = (size_t)1 << p1_log2;
  int rc;
  if ((rc = mk_buf_reserve(c, c->part_meta, (3 * p1 + 8) * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->part, (seq_len + 64) * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_keys, (seq_len + 64) * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_cnts, (seq_len + 64) * sizeof(u64))) != MK_OK) return rc;
  u64* hist = (u64*)c->part_meta.p;
  u64* start = hist + p1;
  u64* cursor = start + p1 + 1;
  MK_HIP(hipMemsetAsync(hist, 0, p1 * sizeof(u64), c->stream));
  mk_prof_begin(c, MK_K_COUNT);
  if (c->alphabet == MK_ALPHABET_NT2) {
    const size_t threads = div_up(seq_len, 32), tiles = div_up(threads, PART_THREADS);
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL((mk_part_hist_k<2, 32, 1>), dim3(grid), dim3(PART_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, p1_log2, c->k, threads);
    hipLaunchKernelGGL(mk_part_scan_k, dim3(1), dim3(1024), 0, c->stream, (const u64*)hist, start, cursor, p1_log2);
    hipLaunchKernelGGL((mk_part_scatter_k<2, 32, 1>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(PART_THREADS), 0,
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, cursor, (u64*)c->part.p, p1_log2,
                       c->k, tiles);
  } else {
    const size_t threads = div_up(seq_len, 36), tiles = div_up(threads, PART_THREADS);
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL((mk_part_hist_k<5, 12, 3>), dim3(grid), dim3(PART_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, p1_log2, c->k, threads);
    hipLaunchKernelGGL(mk_part_scan_k, dim3(1), dim3(1024), 0, c->stream, (const u64*)hist, start, cursor, p1_log2);
    hipLaunchKernelGGL((mk_part_scatter_k<5, 12, 3>), dim3((unsigned)(tiles < 4096 ? tiles : 4096)), dim3(PART_THREADS), 0,
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, cursor, (u64*)c->part.p, p1_log2,
                       c->k, tiles);
  }
  hipLaunchKernelGGL(mk_part_count_k, dim3((unsigned)p1), dim3(CNT_THREADS), 0, c->stream, (const u64*)c->part.p,
                     (const u64*)start, info, (u64)min_count, (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, p1_log2,
                     c->dup_hint);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
