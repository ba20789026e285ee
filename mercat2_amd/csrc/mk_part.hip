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
#define PART_MAX_P1 8192
#define CNT_SLOTS 8192                      // LDS table slots per workgroup (96 KB: one workgroup per CU)
#define CNT_LOADCAP (CNT_SLOTS / 2)         // distinct keys accepted per sub-range
#define CNT_TARGET (CNT_SLOTS * 3 / 10)     // distinct keys aimed at per sub-range (short probe chains)
#define CNT_THREADS 1024
#define SUB_BITS 24                         // hash bits available for splitting a bucket

static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// --------------------------------------------------------------------------- 1 histogram
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(PART_THREADS) void mk_part_hist_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                               MkChunkInfo* __restrict__ info, u64* __restrict__ hist,
                                                               int p1_log2, int k, size_t nthreads_total, int canon) {
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
          const u64 key = mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
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
  block_add(&info->windows, mine);
  wave_add(&info->side, side);  // (almost always zero: no add)
}

// -------------------------------------------------------------------------------- 2 scan
// `div` > 1 scans ceil(hist/div) instead of hist: a bucket with m k-mers has at most m/c entries with
// count >= c, which is all the room its survivor region needs.
__global__ __launch_bounds__(1024) void mk_part_scan_k(const u64* __restrict__ hist, u64* __restrict__ start,
                                                       u64* __restrict__ cursor, int p1_log2, u64 div) {
  __shared__ u64 sums[1024];
  const unsigned p1 = 1u << p1_log2;
  const unsigned per = (p1 + 1023) / 1024;
  const unsigned lo = threadIdx.x * per;
  u64 acc = 0;
  for (unsigned i = lo; i < lo + per && i < p1; ++i) acc += (hist[i] + div - 1) / div;
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
    run += (hist[i] + div - 1) / div;
  }
}

// ----------------------------------------------------------------------------- 3 scatter
// One tile = SUBT x 256 threads x R windows. Pass 1 counts the tile's keys per bucket in LDS,
// then the workgroup reserves all its runs with one sweep of (lane-contiguous) cursor atomics,
// pass 2 re-derives the keys (the packed words are L1/L2-hot), takes each key's rank from a
// second LDS counter and stores it into its run.
#define SCAT_SUBT 2
#define SCAT_THREADS 1024
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(SCAT_THREADS) void mk_part_scatter_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                  const MkChunkInfo* __restrict__ info,
                                                                  u64* __restrict__ cursor, u64* __restrict__ part,
                                                                  int p1_log2, int k, size_t ntiles, int dbg, int canon) {
  __shared__ unsigned lh[PART_MAX_P1];
  __shared__ u64 gbase[PART_MAX_P1];
  constexpr int R = SPW * WPT;
  constexpr int NB = PART_MAX_P1 / SCAT_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const int hshift = 64 - p1_log2;
  const size_t seq_len = info->seq_len;
  const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- pass 1: bucket sizes of this tile
#pragma unroll 1
    for (int st = 0; st < SCAT_SUBT; ++st) {
      const size_t t = (tile * SCAT_SUBT + st) * SCAT_THREADS + threadIdx.x;
      const size_t p0 = t * R;
      if (p0 >= seq_len) continue;
      u64 w[WPT + 1];
#pragma unroll
      for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
      const u64 badw = bad_window(bad, p0);
#pragma unroll
      for (int i = 0; i < WPT; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
          if (((badw >> (i * SPW + s)) & kmask) == 0) {
            const u64 key = mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
            if (key != MK_EMPTY) atomicAdd(&lh[(unsigned)(mk_mix64(key) >> hshift)], 1u);
          }
        }
      }
    }
    __syncthreads();
    // ---- reserve the runs: all cursor atomics of the tile in flight together
    {
      unsigned v[NB];
      u64 r[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SCAT_THREADS;
        v[i] = b < p1 ? lh[b] : 0u;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SCAT_THREADS;
        r[i] = (v[i] && !(dbg & 2)) ? atomicAdd(&cursor[b], (u64)v[i]) : (u64)b * 16;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SCAT_THREADS;
        if (b < p1) { gbase[b] = r[i]; lh[b] = 0; }
      }
    }
    __syncthreads();
    // ---- pass 2: rank inside the run, store
#pragma unroll 1
    for (int st = 0; st < SCAT_SUBT; ++st) {
      const size_t t = (tile * SCAT_SUBT + st) * SCAT_THREADS + threadIdx.x;
      const size_t p0 = t * R;
      if (p0 >= seq_len) continue;
      u64 w[WPT + 1];
#pragma unroll
      for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
      const u64 badw = bad_window(bad, p0);
#pragma unroll
      for (int i = 0; i < WPT; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
          if (((badw >> (i * SPW + s)) & kmask) == 0) {
            const u64 key = mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
            if (key != MK_EMPTY) {
              const unsigned b = (unsigned)(mk_mix64(key) >> hshift);
              const u64 at = gbase[b] + atomicAdd(&lh[b], 1u);
              if (!(dbg & 1)) part[at] = key;
            }
          }
        }
      }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------- 4 count
// Sub-range (s, i) of a bucket: keys whose SUB_BITS-bit field f = (h >> sub_shift) & mask has
// its top s bits equal to i. Depth-first walk over the binary tree of sub-ranges: a sub-range
// whose distinct keys do not fit the LDS table is replaced by its two halves.
#define CNT_U 4  // keys per thread in flight

#define CNT_MAX_PROBE 48  // longer chains mean the table is too full for this sub-range: split it
__device__ __forceinline__ void lds_insert_slow(u64* tkey, unsigned* tcnt, unsigned* s_overflow, u64 key, unsigned slot,
                                                u64 cur) {
#pragma unroll 1  // (unrolled 48 times and inlined at every call site this loop was most of the kernel's 23 KB of code)
  for (int probe = 0; probe < CNT_MAX_PROBE; ++probe) {
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&tkey[slot], MK_EMPTY, key);
      if (cur == MK_EMPTY) cur = key;
    }
    if (cur == key) { atomicAdd(&tcnt[slot], 1u); return; }
    slot = (slot + 1) & (CNT_SLOTS - 1);
    cur = tkey[slot];
  }
  *(volatile unsigned*)s_overflow = 1;
}

__global__ __launch_bounds__(CNT_THREADS) void mk_part_count_k(const u64* __restrict__ part, const u64* __restrict__ start,
                                                               MkChunkInfo* __restrict__ info, u64 min_count,
                                                               u64* __restrict__ out_keys, u64* __restrict__ out_cnts,
                                                               int p1_log2, double dup_hint, int dbg) {
  __shared__ u64 tkey[CNT_SLOTS];
  __shared__ unsigned tcnt[CNT_SLOTS];
  __shared__ unsigned s_distinct, s_overflow, s_emit;
  __shared__ u64 s_base;
  const u64 lo = start[blockIdx.x], hi = start[blockIdx.x + 1];
  const u64 n = hi - lo;
  if (n == 0) return;
  if (n >> 32) {  // LDS counters are 32-bit
    if (threadIdx.x == 0) atomicAdd(&info->errors, 1ull);
    return;
  }
  const int sub_shift = 64 - p1_log2 - SUB_BITS;  // field sits right below the bucket bits
  for (unsigned i = threadIdx.x; i < CNT_SLOTS; i += blockDim.x) { tkey[i] = MK_EMPTY; tcnt[i] = 0; }
  if (threadIdx.x == 0) { s_distinct = 0; s_overflow = 0; s_emit = 0; }
  __syncthreads();
  // starting depth from the expected number of distinct keys in this bucket
  int s0 = 0;
  {
    const double expect = (double)n / (dup_hint > 1.0 ? dup_hint : 1.0);
    while (s0 < SUB_BITS && expect / (double)(1u << s0) > (double)CNT_TARGET) ++s0;
    if (n <= CNT_LOADCAP) s0 = 0;
    if (s0 > 3) s0 = 3;  // (see mk_skmer.hip: the estimate may be far too high for this bucket; overflows split further)
  }
  int s = s0;
  unsigned idx = 0;
  u64 distinct_total = 0;
  const u64* __restrict__ src = part + lo;
  for (;;) {
    // ---- fill the table with the keys of sub-range (s, idx): CNT_U loads in flight per thread,
    //      first probe of all of them issued together, the rare collisions take the slow path
    const unsigned sel_shift = SUB_BITS - s;
    for (u64 base = 0; base < n; base += (u64)CNT_THREADS * CNT_U) {
      u64 kk[CNT_U];
#pragma unroll
      for (int u = 0; u < CNT_U; ++u) {
        const u64 j = base + (u64)u * CNT_THREADS + threadIdx.x;
        kk[u] = j < n ? src[j] : MK_EMPTY;
      }
      unsigned slot[CNT_U];
      u64 cur[CNT_U];
#pragma unroll
      for (int u = 0; u < CNT_U; ++u) {
        const u64 h = mk_mix64(kk[u]);
        slot[u] = (unsigned)h & (CNT_SLOTS - 1);
        if (s && (unsigned)(((h >> sub_shift) & ((1u << SUB_BITS) - 1)) >> sel_shift) != idx) kk[u] = MK_EMPTY;
      }
#pragma unroll
      for (int u = 0; u < CNT_U; ++u) cur[u] = tkey[slot[u]];
#pragma unroll
      for (int u = 0; u < CNT_U; ++u) {
        if (kk[u] == MK_EMPTY) continue;
        if (cur[u] == kk[u]) atomicAdd(&tcnt[slot[u]], 1u);
        else lds_insert_slow(tkey, tcnt, &s_overflow, kk[u], slot[u], cur[u]);
      }
      if (*(volatile unsigned*)&s_overflow) break;  // the table filled up: this attempt is void
    }
    __syncthreads();
    const bool over = s_overflow != 0;
    // ---- emit (when complete) and clear: count the keepers, reserve their output range with ONE
    //      global atomic per workgroup, then place them by an LDS cursor
    {
      constexpr int PER = CNT_SLOTS / CNT_THREADS;
      u64 ek[PER];
      unsigned ec[PER];
      unsigned mine = 0, occ = 0;
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const unsigned i = q * CNT_THREADS + threadIdx.x;
        ek[q] = tkey[i];
        ec[q] = tcnt[i];
        tkey[i] = MK_EMPTY;
        tcnt[i] = 0;
        occ += ek[q] != MK_EMPTY;
        if (over || ek[q] == MK_EMPTY || (u64)ec[q] < min_count) ek[q] = MK_EMPTY;
        mine += ek[q] != MK_EMPTY;
      }
      for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d);
      if ((threadIdx.x & 63) == 0 && occ && !over) atomicAdd(&s_distinct, occ);
      if (mine) atomicAdd(&s_emit, mine);
      __syncthreads();
      if (threadIdx.x == 0) {
        const unsigned tot = s_emit;
        s_base = (tot && !(dbg & 4)) ? atomicAdd(&info->survivors, (u64)tot) : 0ull;
        s_emit = 0;
      }
      __syncthreads();
      distinct_total += s_distinct;
      if (mine) {
        const u64 at = s_base + atomicAdd(&s_emit, mine);
        unsigned o = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          if (ek[q] != MK_EMPTY) {
            out_keys[at + o] = ek[q];
            out_cnts[at + o] = ec[q];
            ++o;
          }
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { s_distinct = 0; s_overflow = 0; s_emit = 0; }
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

void mk_launch_part_scan(mk_ctx* c, const u64* hist, u64* start, u64* cursor, int p1_log2, u64 div) {
  hipLaunchKernelGGL(mk_part_scan_k, dim3(1), dim3(1024), 0, c->stream, hist, start, cursor, p1_log2, div ? div : 1);
}

// ------------------------------------------------------------------------------ launcher
int mk_launch_count_partitioned(mk_ctx* c, size_t seq_len, uint64_t min_count) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  // bucket count: aim at ~8K keys per bucket, between 256 and PART_MAX_P1
  int p1_log2 = 8;
  while (p1_log2 < 13 && (seq_len >> p1_log2) > 8192) ++p1_log2;
  if (const char* e = getenv("MK_P1_LOG2")) { int v = atoi(e); if (v >= 4 && v <= 13) p1_log2 = v; }
  c->p1_log2 = p1_log2;
  const int dbg = getenv("MK_DBG") ? atoi(getenv("MK_DBG")) : 0;
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
  mk_prof_begin(c, MK_K_PART);
  if (c->alphabet == MK_ALPHABET_NT2) {
    const size_t threads = div_up(seq_len, 32), tiles = div_up(threads, PART_THREADS);
    const size_t stiles = div_up(threads, (size_t)SCAT_THREADS * SCAT_SUBT);
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL((mk_part_hist_k<2, 32, 1>), dim3(grid), dim3(PART_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, p1_log2, c->k, threads, c->canonical);
    hipLaunchKernelGGL(mk_part_scan_k, dim3(1), dim3(1024), 0, c->stream, (const u64*)hist, start, cursor, p1_log2, (u64)1);
    hipLaunchKernelGGL((mk_part_scatter_k<2, 32, 1>), dim3((unsigned)(stiles < 4096 ? stiles : 4096)), dim3(SCAT_THREADS), 0,
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, cursor, (u64*)c->part.p, p1_log2,
                       c->k, stiles, dbg, c->canonical);
  } else {
    const size_t threads = div_up(seq_len, 36), tiles = div_up(threads, PART_THREADS);
    const size_t stiles = div_up(threads, (size_t)SCAT_THREADS * SCAT_SUBT);
    const unsigned grid = (unsigned)(tiles < 2048 ? tiles : 2048);
    hipLaunchKernelGGL((mk_part_hist_k<5, 12, 3>), dim3(grid), dim3(PART_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, p1_log2, c->k, threads, c->canonical);
    hipLaunchKernelGGL(mk_part_scan_k, dim3(1), dim3(1024), 0, c->stream, (const u64*)hist, start, cursor, p1_log2, (u64)1);
    hipLaunchKernelGGL((mk_part_scatter_k<5, 12, 3>), dim3((unsigned)(stiles < 4096 ? stiles : 4096)), dim3(SCAT_THREADS), 0,
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, cursor, (u64*)c->part.p, p1_log2,
                       c->k, stiles, dbg, c->canonical);
  }
  mk_prof_end(c);
  mk_prof_begin(c, MK_K_COUNT);
  hipLaunchKernelGGL(mk_part_count_k, dim3((unsigned)p1), dim3(CNT_THREADS), 0, c->stream, (const u64*)c->part.p,
                     (const u64*)start, info, (u64)min_count, (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, p1_log2,
                     c->dup_hint, dbg);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
