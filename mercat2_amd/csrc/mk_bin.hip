// mk_bin.hip -- counting by DIRECT INDEX for keys of 16..26 bits (nucleotide 8 <= k <= 11 below the super-k-mer path,
// protein k = 4, 5: the reference's own protein runs are k = 5, results/run-tests.sh:14-28).
//
// Same arithmetic as every other path (lib/mercat2_kmers.py:56-60: every window +1; :73-76: keep count >= min_count,
// per chunk) -- but a key this short needs no hash table.  Its top bits name a bucket, its low `low` bits (<= 13) a bin:
//   1 mk_bin_hist     windows per bucket (LDS histogram per workgroup, one global add per bucket and workgroup)
//   2 mk_bin_scan     bucket regions of the item buffer
//   3 mk_bin_scatter  per tile: rank of every window inside its (tile, bucket) run from an LDS counter, one sweep of
//                     packed 32-bit cursor atomics reserves the runs, the LOW BITS of the key -- two bytes -- are stored
//   4 mk_bin_count    persistent workgroups walk the buckets: one LDS add per item into 2^low bins, then the bins with
//                     count >= min_count are written out as (key, count) pairs; no compare-and-swap, no probing, no
//                     sub-range passes, and a bucket of any size or skew counts in one pass (32-bit bins: a chunk holds
//                     fewer than 2^32 windows)
// Up to round 3 these shapes took the 8-byte-key partition (mk_part.hip: two 64-bit mixing hashes per window in the
// scatter, a compare-and-swap table in the count kernel): 28 Gresidues/s at protein k = 5.
#include "mk_common.h"
#include "mk_device.h"
#include <algorithm>
#include <cstdlib>

#define BIN_THREADS 512
#define BINC_THREADS 1024  // the count kernel: one workgroup per CU (128 KB of bins)
#define BIN_MAX_NB 8192   // (histogram / scan kernels: buckets they can hold)
#define BIN_SC_NB 2048    // buckets at most: 1024, or 2048 for keys of 26 bits
#define BIN_MAX_LOW 15    // bins per bucket 2^low <= 32768 (128 KB of 32-bit counters in the count kernel's LDS)

static size_t bin_div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// f(key) for every clean window of the R = SPW * WPT symbols thread t owns
template <int BITS, int SPW, int WPT, class F>
__device__ __forceinline__ void bin_windows(const u64* __restrict__ codes, const u64* __restrict__ bad, size_t t, int k, u64 kmask,
                                            bool canon, F&& f) {
  u64 w[WPT + 1];
#pragma unroll
  for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
  const u64 badw = bad_window(bad, t * (size_t)(SPW * WPT));
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
#pragma unroll
    for (int s = 0; s < SPW; ++s) {
      if (((badw >> (i * SPW + s)) & kmask) == 0)
        f((unsigned)mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon));
    }
  }
}

template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(BIN_THREADS) void mk_bin_hist_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                             const MkChunkInfo* __restrict__ info, unsigned* __restrict__ hist, int low,
                                                             unsigned nb, int k, size_t nthreads_total, int canon) {
  __shared__ unsigned lh[BIN_MAX_NB];
  for (unsigned i = threadIdx.x; i < nb; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  const size_t seq_len = info->seq_len;
  const u64 kmask = (1ull << k) - 1;
  for (size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x; t < nthreads_total; t += (size_t)gridDim.x * blockDim.x) {
    if (t * (size_t)(SPW * WPT) >= seq_len) break;
    bin_windows<BITS, SPW, WPT>(codes, bad, t, k, kmask, canon != 0, [&](unsigned key) { atomicAdd(&lh[key >> low], 1u); });
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < nb; b += blockDim.x) {
    const unsigned v = lh[b];
    if (v) atomicAdd(&hist[b], v);
  }
}

// start[b] (nb + 1 values) and cursor[b] = start[b]; one workgroup
__global__ __launch_bounds__(1024) void mk_bin_scan_k(const unsigned* __restrict__ hist, unsigned* __restrict__ start,
                                                      unsigned* __restrict__ cursor, unsigned nb) {
  constexpr int PER = BIN_MAX_NB / 1024;
  __shared__ unsigned wsum[16];
  const unsigned per = (nb + 1023) / 1024, lo = threadIdx.x * per;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned v[PER], acc = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned i = lo + q;
    v[q] = ((unsigned)q < per && i < nb) ? hist[i] : 0u;
    acc += v[q];
  }
  unsigned inc = acc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned a = __shfl_up(inc, d);
    if (lane >= d) inc += a;
  }
  if (lane == 63) wsum[wv] = inc;
  __syncthreads();
  unsigned run = inc - acc;
  for (int w = 0; w < wv; ++w) run += wsum[w];
  if (threadIdx.x == 1023) start[nb] = run + acc;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned i = lo + q;
    if ((unsigned)q < per && i < nb) {
      start[i] = run;
      cursor[i] = run;
      run += v[q];
    }
  }
}

// A tile's items are first SORTED BY BUCKET IN LDS and then written run by run.  Stored straight from the lanes that
// found them (the first version) every two-byte item was a memory transaction of its own -- 64 lanes, 64 buckets, 64
// lines per store instruction -- and the scatter ran at the L2's transaction rate, 610 us for 60 M protein 5-mers, wherever
// the lines went (tile classes with per-XCD queues changed nothing).  With <= 2048 buckets a tile of 18 K items holds runs
// of 9-18 items per bucket: the copy-out's lanes write neighbouring addresses, a store instruction touches 4-8 lines.
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(BIN_THREADS) void mk_bin_scatter_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                const MkChunkInfo* __restrict__ info, unsigned* __restrict__ cursor,
                                                                unsigned short* __restrict__ items, int low, unsigned nb, int k,
                                                                size_t ntiles, int canon) {
  constexpr unsigned TILE_ITEMS = BIN_THREADS * SPW * WPT;
  __shared__ unsigned cnt[BIN_SC_NB];    // pass 1: the tile's windows per bucket; then the running place inside the stage
  __shared__ unsigned off0[BIN_SC_NB];   // where the bucket's run starts in the stage
  __shared__ unsigned gbase[BIN_SC_NB];  // where it starts in the item buffer (reserved with one add per bucket and tile)
  __shared__ unsigned stage[TILE_ITEMS]; // bucket << 16 | low bits of the key, sorted by bucket
  __shared__ unsigned wsum[BIN_THREADS / 64];
  const size_t seq_len = info->seq_len;
  const u64 kmask = (1ull << k) - 1;
  const unsigned lowmask = (1u << low) - 1;
  const unsigned per = nb / BIN_THREADS;  // buckets per thread in the scan (nb is 1024 or 2048: 2 or 4)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    for (unsigned i = threadIdx.x; i < nb; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const size_t t = tile * BIN_THREADS + threadIdx.x;
    const bool mine = t * (size_t)(SPW * WPT) < seq_len;
    if (mine) bin_windows<BITS, SPW, WPT>(codes, bad, t, k, kmask, canon != 0, [&](unsigned key) { atomicAdd(&cnt[key >> low], 1u); });
    __syncthreads();
    // reservations (regions are exact: the histogram counted the same windows) and the exclusive scan of the counts
    unsigned v[4] = {0, 0, 0, 0}, sum = 0;
#pragma unroll
    for (unsigned q = 0; q < 4; ++q)
      if (q < per) {
        const unsigned b = threadIdx.x * per + q;
        v[q] = cnt[b];
        gbase[b] = v[q] ? atomicAdd(&cursor[b], v[q]) : 0u;
        sum += v[q];
      }
    const unsigned inc = mk_wave_scan_incl(sum);
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    unsigned run = inc - sum;
    for (int w = 0; w < wv; ++w) run += wsum[w];
#pragma unroll
    for (unsigned q = 0; q < 4; ++q)
      if (q < per) {
        const unsigned b = threadIdx.x * per + q;
        off0[b] = run;
        cnt[b] = run;
        run += v[q];
      }
    unsigned total = 0;
    for (int w = 0; w < BIN_THREADS / 64; ++w) total += wsum[w];
    __syncthreads();
    if (mine)
      bin_windows<BITS, SPW, WPT>(codes, bad, t, k, kmask, canon != 0, [&](unsigned key) {
        const unsigned b = key >> low;
        stage[atomicAdd(&cnt[b], 1u)] = (b << 16) | (key & lowmask);
      });
    __syncthreads();
    for (unsigned i = threadIdx.x; i < total; i += BIN_THREADS) {
      const unsigned e = stage[i], b = e >> 16;
      items[(size_t)gbase[b] + (i - off0[b])] = (unsigned short)e;
    }
    __syncthreads();
  }
}

// Persistent: workgroup w takes buckets w, w + gridDim.x, ...  Survivors go to out_keys / out_cnts at positions reserved
// with one add to info->survivors per bucket (the import of mk_table.hip takes them from there).
__global__ __launch_bounds__(BINC_THREADS) void mk_bin_count_k(const unsigned short* __restrict__ items, const unsigned* __restrict__ start,
                                                              MkChunkInfo* __restrict__ info, u64 min_count, u64* __restrict__ out_keys,
                                                              u64* __restrict__ out_cnts, int low, unsigned nb) {
  __shared__ unsigned bins[1 << BIN_MAX_LOW];
  __shared__ unsigned s_keep[2], s_occ[2];
  __shared__ unsigned long long s_base[2];
  const unsigned nbins = 1u << low;
  for (unsigned i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
  if (threadIdx.x < 2) { s_keep[threadIdx.x] = 0; s_occ[threadIdx.x] = 0; }
  __syncthreads();
  u64 windows = 0, distinct = 0;
  unsigned par = 0;
  const int lane = threadIdx.x & 63;
  for (unsigned b = blockIdx.x; b < nb; b += gridDim.x) {
    const unsigned lo = start[b], n = start[b + 1] - lo;
    if (!n) continue;  // (uniform: every thread reads the same bounds)
    const unsigned short* __restrict__ src = items + lo;
    // (the bucket's items, two bytes each: the head up to an 8-byte boundary one by one, then four per load)
    const unsigned head = (unsigned)((8u - ((uintptr_t)src & 7u)) & 7u) / 2u;
    const unsigned h = head < n ? head : n;
    if (threadIdx.x < h) atomicAdd(&bins[src[threadIdx.x]], 1u);
    const unsigned quads = (n - h) / 4;
    const ushort4* __restrict__ q4 = reinterpret_cast<const ushort4*>(src + h);
    unsigned i = threadIdx.x;
    for (; i + 3 * BINC_THREADS < quads; i += 4 * BINC_THREADS) {  // (four loads in flight per lane)
      const ushort4 v0 = q4[i], v1 = q4[i + BINC_THREADS], v2 = q4[i + 2 * BINC_THREADS], v3 = q4[i + 3 * BINC_THREADS];
      atomicAdd(&bins[v0.x], 1u); atomicAdd(&bins[v0.y], 1u); atomicAdd(&bins[v0.z], 1u); atomicAdd(&bins[v0.w], 1u);
      atomicAdd(&bins[v1.x], 1u); atomicAdd(&bins[v1.y], 1u); atomicAdd(&bins[v1.z], 1u); atomicAdd(&bins[v1.w], 1u);
      atomicAdd(&bins[v2.x], 1u); atomicAdd(&bins[v2.y], 1u); atomicAdd(&bins[v2.z], 1u); atomicAdd(&bins[v2.w], 1u);
      atomicAdd(&bins[v3.x], 1u); atomicAdd(&bins[v3.y], 1u); atomicAdd(&bins[v3.z], 1u); atomicAdd(&bins[v3.w], 1u);
    }
    for (; i < quads; i += BINC_THREADS) {
      const ushort4 v = q4[i];
      atomicAdd(&bins[v.x], 1u);
      atomicAdd(&bins[v.y], 1u);
      atomicAdd(&bins[v.z], 1u);
      atomicAdd(&bins[v.w], 1u);
    }
    const unsigned done = h + 4 * quads;
    if (threadIdx.x < n - done) atomicAdd(&bins[src[done + threadIdx.x]], 1u);
    if (threadIdx.x == 0) { windows += n; s_keep[par ^ 1] = 0; s_occ[par ^ 1] = 0; }
    __syncthreads();
    // sweep: count the bins in use and the ones that stay, reserve the survivors' places, write them, clear
    unsigned mine = 0, occ = 0;
    for (unsigned i = threadIdx.x; i < nbins; i += BINC_THREADS) {
      const unsigned v = bins[i];
      occ += v != 0;
      mine += (v && (u64)v >= min_count) ? 1u : 0u;
    }
    for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d);
    if (lane == 0 && occ) atomicAdd(&s_occ[par], occ);
    const unsigned at0 = mine ? atomicAdd(&s_keep[par], mine) : 0u;
    __syncthreads();
    if (threadIdx.x == 0) {
      distinct += s_occ[par];
      if (s_keep[par]) s_base[par] = atomicAdd(&info->survivors, (u64)s_keep[par]);
    }
    __syncthreads();
    if (mine) {
      u64 at = s_base[par] + at0;
      for (unsigned i = threadIdx.x; i < nbins; i += BINC_THREADS) {
        const unsigned v = bins[i];
        if (v && (u64)v >= min_count) {
          out_keys[at] = ((u64)b << low) | (u64)i;
          out_cnts[at] = (u64)v;
          ++at;
        }
      }
    }
    for (unsigned i = threadIdx.x; i < nbins; i += BINC_THREADS) bins[i] = 0;
    __syncthreads();
    par ^= 1;
  }
  if (threadIdx.x == 0) {
    if (windows) atomicAdd(&info->windows, windows);
    if (distinct) atomicAdd(&info->distinct, distinct);
  }
}

// key bits this path takes (the caller checks the alphabet and that no longer path is faster)
bool mk_binned_takes(const mk_ctx* c) {
  static const bool off = getenv("MK_NO_BINNED") != nullptr;
  const int kb = c->bits * c->k;
  return !off && c->mode == MK_MODE_HASH64 && kb >= 16 && kb <= 26;
}

int mk_launch_count_binned(mk_ctx* c, size_t seq_len, uint64_t min_count) {
  if (seq_len == 0) return MK_OK;
  if (seq_len >= 0xFFFFFF00ull) { c->err = "mk_launch_count_binned: chunk of 4 G symbols or more"; return MK_ERR_RANGE; }
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const int kb = c->bits * c->k;
  const int low = std::min(BIN_MAX_LOW, std::max(6, kb - 10));  // 1024 buckets (2048 for 26 bits), <= 32768 bins each
  const unsigned nb = 1u << (kb - low);
  c->p1_log2 = kb - low;
  int rc;
  if ((rc = mk_buf_reserve(c, c->part_meta, (3 * (size_t)nb + 16) * sizeof(unsigned))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->part, (seq_len + 64) * sizeof(unsigned short))) != MK_OK) return rc;
  // (survivors: at most one per window and at most one per possible key)
  const size_t surv_cap = std::min<size_t>(seq_len + 64, ((size_t)1 << kb) + 64);
  if ((rc = mk_buf_reserve(c, c->surv_keys, surv_cap * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_cnts, surv_cap * sizeof(u64))) != MK_OK) return rc;
  unsigned* hist = (unsigned*)c->part_meta.p;
  unsigned* start = hist + nb;
  unsigned* cursor = start + nb + 1;
  MK_HIP(hipMemsetAsync(hist, 0, nb * sizeof(unsigned), c->stream));
  int ncu = 256;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
  mk_prof_begin(c, MK_K_PART);
#define BIN_LAUNCH(BITS, SPW, WPT)                                                                                          \
  do {                                                                                                                      \
    const size_t threads = bin_div_up(seq_len, (size_t)(SPW) * (WPT));                                                      \
    const size_t tiles = bin_div_up(threads, (size_t)BIN_THREADS);                                                          \
    const unsigned hgrid = (unsigned)std::min<size_t>(bin_div_up(threads, BIN_THREADS), (size_t)ncu * 4);                   \
    hipLaunchKernelGGL((mk_bin_hist_k<BITS, SPW, WPT>), dim3(hgrid), dim3(BIN_THREADS), 0, c->stream, (const u64*)c->codes.p, \
                       (const u64*)c->bad.p, info, hist, low, nb, c->k, threads, c->canonical);                             \
    hipLaunchKernelGGL(mk_bin_scan_k, dim3(1), dim3(1024), 0, c->stream, (const unsigned*)hist, start, cursor, nb);         \
    hipLaunchKernelGGL((mk_bin_scatter_k<BITS, SPW, WPT>), dim3((unsigned)std::min<size_t>(tiles, 4096)), dim3(BIN_THREADS), 0, \
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, cursor, (unsigned short*)c->part.p, low, \
                       nb, c->k, tiles, c->canonical);                                                                      \
  } while (0)
  if (c->alphabet == MK_ALPHABET_NT2) BIN_LAUNCH(2, 32, 1);
  else BIN_LAUNCH(5, 12, 3);
#undef BIN_LAUNCH
  mk_prof_end(c);
  mk_prof_begin(c, MK_K_COUNT);
  hipLaunchKernelGGL(mk_bin_count_k, dim3((unsigned)std::min<size_t>(nb, (size_t)ncu)), dim3(BINC_THREADS), 0, c->stream,
                     (const unsigned short*)c->part.p, (const unsigned*)start, info, (u64)min_count, (u64*)c->surv_keys.p,
                     (u64*)c->surv_cnts.p, low, nb);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
