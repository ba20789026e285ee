// mk_skmer.hip -- super-k-mer partitioned counting for nucleotide k-mers, 12 <= k <= 32 (round 2: from 18).
//
// Same arithmetic as mk_part.hip (every window +1, keep count >= min_count;
// lib/mercat2_kmers.py:56-60, 73-76) but the unit that travels through HBM is not the 8-byte
// key of ONE window: it is a 16-byte record holding a run of up to 8 CONSECUTIVE windows that
// share their minimizer (a "super-k-mer": nk + k - 1 bases, 2 bits each, plus nk in 6 bits).
//
//   minimizer of a window = the 11-mer inside it with the smallest hash (leftmost on ties): a
//   function of the window's content only, so equal k-mers always meet in the same bucket;
//   bucket = hash2(minimizer).  Consecutive windows usually keep their minimizer, so a run of
//   ~8 windows costs one 16-byte store instead of eight scattered 8-byte stores, and the
//   partition traffic drops from 8 B to ~2 B per window.
//
//   1 mk_sk_hist     per thread 32 windows: 11-mers, hashes, sliding minimum (doubling), runs;
//                    records and k-mers per bucket in LDS histograms.  Chunks of >= 8 Mbases look at one
//                    thread in eight only (a sample)
//   2 mk_sk_scan     bucket regions of the record buffer and of the survivor buffer in one pass; sampled
//                    counts become capacities with room for the sampling error
//   3 mk_sk_scatter  same walk over tiles of 2 x 1024 threads (the first sub-tile's analysis parked in
//                    LDS); rank of each record inside its (tile,bucket) run from an LDS counter, one sweep
//                    of cursor atomics per tile checked against the region ends, 16-byte record stores
//   4 mk_sk_count    (mk_skcount.hip) persistent, one workgroup per CU walks the buckets: expands the records into k-mers
//                    and counts them in an LDS open-addressing table (claim-or-compare with one
//                    compare-and-swap), emits entries with count >= min_count into the bucket's survivor
//                    region, splits a bucket by further hash bits when its distinct keys do not fit
//   A region that turns out too small (sampled sizes only) is never written past: the chunk is flagged
//   (MkChunkInfo.part_overflow) and partitioned again from the exact histogram by the caller.
#include "mk_skmer_dev.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

#define SK_NKMAX 8              // windows per record (<= 62 - k)
#ifndef SK_HIST_THREADS
#define SK_HIST_THREADS 256
#endif
#ifndef SK_HIST_GRID
#define SK_HIST_GRID 512
#endif
#ifndef SK_SCAT_THREADS
#define SK_SCAT_THREADS 1024
#endif
#ifndef SK_SCAT_GRID
#define SK_SCAT_GRID 4096
#endif
#ifndef SK_SCAT_SUBT
#define SK_SCAT_SUBT 2
#endif
#ifndef SK_MAX_P1_LOG2
#define SK_MAX_P1_LOG2 14       // most buckets a chunk is cut into (2^13 by default: the launcher's choice; 2^14 with MK_CORES)
#endif
#define SK_MAX_P1 (1 << SK_MAX_P1_LOG2)
#define SK_LH_LOG2 13           // bucket counters the queue scatter keeps in LDS at a time: with more buckets than that a tile
#define SK_LH (1 << SK_LH_LOG2) // is worked off in rounds of 2^13 buckets (the LDS stays at 72-80 KB: two workgroups per CU)
#ifndef SK_BUCKET_SYMS
#define SK_BUCKET_SYMS 8192     // symbols of the chunk per bucket the bucket count aims at (~1.2K records, ~10K windows)
#endif
#define SK_NOFIT 0xFF000000u    // lh[] value of a (tile, bucket) run that does not fit its region: nothing is stored
static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

template <int W, class F>
__device__ __forceinline__ void sk_for_each_record(u64 w0, u64 w1, u64 badw, int k, int nkmax, bool canon, F&& emit) {
  const SkRuns r = sk_analyse<W>(w0, w1, sk_valid32(badw, k), canon);
  sk_walk(r, w0, w1, nkmax, canon, emit);
}

__device__ __forceinline__ ulonglong2 sk_make_record(u64 w0, u64 w1, int jstart, int nk, int k) {
  // (w0:w1) << 2 jstart, its top 2 L bits kept, nk in the low bits -- selects, no branches: the walk's lanes diverge enough
  const int s = 2 * jstart;  // 0 .. 62
  u64 hi = (w0 << s) | ((w1 >> 1) >> (63 - s));
  u64 lo = w1 << s;
  const int t = 2 * (nk + k - 1);  // bits of the record's bases, <= 122
  const u64 mh = t >= 64 ? ~0ull : (~0ull << ((64 - t) & 63));
  const u64 ml = t <= 64 ? 0ull : (~0ull << ((128 - t) & 63));
  hi &= mh;
  lo = (lo & ml) | (u64)nk;
  return make_ulonglong2(hi, lo);
}

// ------------------------------------------------------------------------------ 1 hist
// Records and k-mers per bucket.  With sample_log2 = s > 0 only one analysis thread of every 2^s
// (a pseudo-random member of each group, so that no period of the text can hide from the sample)
// is looked at: the scan below turns the sampled counts into capacities with room for the sampling
// error, the scatter checks every reservation against them, and a chunk whose estimate was too
// small anywhere is partitioned again with s = 0 (exact).
template <int W, bool CANON>
__global__ __launch_bounds__(SK_HIST_THREADS) void mk_sk_hist_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                const MkChunkInfo* __restrict__ info, u64* __restrict__ hist,
                                                                u64* __restrict__ khist, int p1_log2, int k, int nkmax,
                                                                size_t nthreads_total, int canon, int sample_log2) {
  extern __shared__ unsigned sk_hist_lds[];  // 2 x p1 words (launcher): records per bucket, then k-mers per bucket
  const unsigned p1 = 1u << p1_log2;
  unsigned* const lh = sk_hist_lds;        // records per bucket
  unsigned* const lk = sk_hist_lds + p1;   // k-mers per bucket (bounds the bucket's survivors)
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) { lh[i] = 0; lk[i] = 0; }
  __syncthreads();
  const size_t seq_len = info->seq_len;
  const size_t ngroups = (nthreads_total + ((size_t)1 << sample_log2) - 1) >> sample_log2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
    size_t t = g;
    if (sample_log2) t = (g << sample_log2) + (((unsigned)g * 0x9E3779B1u >> 7) & ((1u << sample_log2) - 1));
    const size_t p0 = t * SK_R;
    if (t >= nthreads_total || p0 >= seq_len) continue;
    const u64 w0 = codes[t], w1 = codes[t + 1];
    const u64 badw = bad_window(bad, p0);
    (void)canon;
    sk_for_each_record<W>(w0, w1, badw, k, nkmax, CANON, [&](int, int nk, unsigned mm) {
      const unsigned b = sk_bucket(mm, p1_log2);
      atomicAdd(&lh[b], 1u);
      atomicAdd(&lk[b], (unsigned)nk);
    });
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < p1; b += blockDim.x) {
    const unsigned v = lh[b];
    // one global add per bucket: records in the low word, k-mers in the high word (a chunk holds fewer than 2^32
    // symbols, so neither half can carry) -- the per-workgroup flush is most of this kernel's HBM traffic
    if (v) atomicAdd(&hist[b], (u64)v | ((u64)lk[b] << 32));
  }
}

// ------------------------------------------------------------------------------ 2 scan
// Bucket regions of the record buffer (start/cursor) and of the survivor buffer (kstart) in one pass.
// Exact histogram: a bucket gets exactly its records, and room for ceil(kmers / min_count) survivors
// (no more entries than that can reach min_count).  Sampled histogram (S = 2^sample_log2): the
// estimate S*h plus six standard deviations of it plus a floor.  The sampling unit is a thread, which
// can put up to SK_R / SK_NKMAX records (SK_R k-mers) into one bucket, so the deviation is taken as
// sqrt(S * estimate * that weight) -- measured: with weight 1 (as if records were sampled one by one) a
// bucket in ~10^4 overflowed.  If the totals do not fit the buffers the chunk is flagged for the exact pass.
#ifndef SK_TU_CANON  // (mk_skmer_canon.hip compiles this file again for the canonical instances only)
__device__ __forceinline__ u64 sk_cap(u64 h, int sample_log2, u64 weight, float sigmas) {
  if (sample_log2 == 0) return h;
  const u64 est = h << sample_log2;
  // single precision is plenty for a margin (the square root is rounded up by the +1)
  const float dev = sigmas * __builtin_sqrtf((float)((u64)weight << sample_log2) * (float)est);
  return est + (u64)dev + 1 + (sigmas > 0 ? 16 * weight : 0);
}
// Room for one XCD's share of a bucket whose sample held h records (eight of these also make the bucket's shared region,
// which is then never smaller than sk_cap's room for the whole bucket): an eighth of the estimate, plus `sigmas` deviations of
// (a) the share's own spread around that eighth and (b) an eighth of the estimate's error -- variances weight * est / nseg
// and weight * S * est / nseg^2 (S = 2^sample_log2) -- plus a small floor.  Tiles go to the XCDs in turn (workgroup i runs on
// XCD i mod 8), so for shuffled reads a share is as good as a 1-in-8 sample of the bucket; a run that does not fit goes to
// the bucket's shared region, and only one that does not fit there either sends the chunk to the exact partition.
__device__ __forceinline__ u64 sk_seg_cap(u64 h, int sample_log2, u64 weight, float sigmas, int nseg) {
  const float est = (float)(h << sample_log2);
  const float var = (float)weight * est * (1.0f / (float)nseg + (float)(1u << sample_log2) / (float)(nseg * nseg));
  return (u64)(est / (float)nseg) + (u64)(sigmas * __builtin_sqrtf(var)) + 1 + (sigmas > 0 ? 4 * weight : 0);
}
__global__ __launch_bounds__(1024) void mk_sk_scan_k(const u64* __restrict__ hist, const u64* __restrict__ khist,
                                                     u64* __restrict__ start, SkCursor* __restrict__ cursor, u64* __restrict__ kstart,
                                                     MkChunkInfo* __restrict__ info, int p1_log2, int sample_log2, int nkmax,
                                                     u64 div, u64 part_cap, u64 surv_cap, float sigmas, int nseg) {
  // nseg = 9 (sampled sizes only): every bucket gets NINE regions -- start[x * p1 + b], cursor[x * p1 + b], all regions
  // x = 0 first: one per XCD (x = 0..7), each sized for an eighth of the bucket (sk_seg_cap), and a shared one (x = 8)
  // as large as the eight together, for the runs that do not fit their XCD's region (a repeat or a homopolymer puts all of
  // a bucket's records into one tile, so into one XCD's region)
  constexpr int PER = SK_MAX_P1 / 1024;  // buckets per thread (p1 <= SK_MAX_P1)
  __shared__ u64 wsum[16], wksum[16];
  const unsigned p1 = 1u << p1_log2;
  const unsigned per = (p1 + 1023) / 1024;
  const unsigned lo = threadIdx.x * per;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  u64 cap[PER], kcap[PER];
  u64 acc = 0, kacc = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned i = lo + q;
    const bool on = (unsigned)q < per && i < p1;
    const u64 hk = on ? hist[i] : 0ull;  // records | k-mers << 32 (see the histogram kernels' flush)
    cap[q] = !on ? 0 : (nseg > 1 ? sk_seg_cap(hk & 0xFFFFFFFFull, sample_log2, SK_R / SK_NKMAX, sigmas, 8)
                                 : sk_cap(hk & 0xFFFFFFFFull, sample_log2, SK_R / SK_NKMAX, sigmas));
    kcap[q] = on ? (sk_cap(hk >> 32, sample_log2, SK_R, sigmas) + div - 1) / div : 0;
    acc += cap[q];
    kacc += kcap[q];
  }
  // exclusive scan over the 1024 threads: inside each wave by shuffles, then over the 16 wave totals
  u64 inc = acc, kinc = kacc;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 a = __shfl_up(inc, d), b = __shfl_up(kinc, d);
    if (lane >= d) { inc += a; kinc += b; }
  }
  if (lane == 63) { wsum[wv] = inc; wksum[wv] = kinc; }
  __syncthreads();
  u64 run = inc - acc, krun = kinc - kacc;
  for (int w = 0; w < wv; ++w) { run += wsum[w]; krun += wksum[w]; }
  __shared__ u64 s_total;
  if (threadIdx.x == 1023) {
    const u64 total = run + acc, ktotal = krun + kacc;
    s_total = total;
    const u64 all = nseg > 1 ? 16 * total : total;  // (8 regions + a shared one of 8 times the size)
    start[(size_t)nseg * p1] = all;
    kstart[p1] = ktotal;
    if (all > part_cap) atomicOr(&info->part_overflow, 1ull);  // (only a sampled estimate can get here)
    if (ktotal > surv_cap) atomicOr(&info->part_overflow, 2ull);
  }
  // (every bucket's place goes through LDS so that the nseg x p1 starts and cursors are written by all threads with
  // consecutive addresses: written bucket by bucket from the thread that owns it -- 9 x 8 x 2 stores at a 64-byte stride,
  // one workgroup -- the kernel took 79 us instead of 20, for every chunk that does not inherit its regions)
  __shared__ unsigned s_run[SK_MAX_P1];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned i = lo + q;
    if ((unsigned)q < per && i < p1) {
      s_run[i] = (unsigned)run;  // (record indices stay below 2^32: the callers check)
      kstart[i] = krun;
      run += cap[q];
      krun += kcap[q];
    }
  }
  __syncthreads();
  const u64 total = s_total;
  for (int x = 0; x < nseg; ++x)
    for (unsigned i = threadIdx.x; i < p1; i += 1024) {
      const u64 at = x < 8 ? (u64)x * total + s_run[i] : 8 * total + 8 * (u64)s_run[i];
      start[(size_t)x * p1 + i] = at;
      cursor[(size_t)x * p1 + i] = (SkCursor)at;
    }
}

#endif

// --------------------------------------------------------------------------- 3 scatter
template <int W, bool CANON>
__global__ __launch_bounds__(1024) void mk_sk_scatter_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                   MkChunkInfo* __restrict__ info, const u64* __restrict__ start,
                                                                   SkCursor* __restrict__ cursor, ulonglong2* __restrict__ part,
                                                                   int p1_log2, int k, int nkmax, size_t ntiles, int canon) {
  // lh[b]: pass 1 counts the tile's records of bucket b; after the reservation it holds the record index at which
  // the tile's run in that bucket starts (the launcher keeps indices below SK_NOFIT) and pass 2's atomic add hands
  // out base + rank in one step -- one array instead of two, which is what lets a tile park a second analysis
  __shared__ unsigned lh[SK_MAX_P1];
  constexpr int PK = SK_SCAT_SUBT > 1 ? SK_SCAT_SUBT - 1 : 1;  // parked analyses (56 bytes per thread each)
  __shared__ uint2 pk_mask[PK][SK_SCAT_THREADS];
  __shared__ ulonglong2 pk_w[PK][SK_SCAT_THREADS];
  __shared__ ulonglong2 pk_pos[PK][2][SK_SCAT_THREADS];
  __shared__ unsigned s_abort;  // (read once per workgroup: other workgroups of this launch may set the flag meanwhile)
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;  // the regions do not fit the buffers: nothing may be written
  unsigned spilled = 0;
  constexpr int NB = SK_MAX_P1 / SK_SCAT_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const size_t seq_len = info->seq_len;
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // pass 1: analyse every sub-tile once and size the runs. The analysis of the last sub-tile stays
    // in registers for pass 2, that of the others is parked in LDS (a second register copy spills):
    // twice the records per tile halves the cursor atomics per record, which run at the L2's limit.
    SkRuns runs;
    u64 ww0 = 0, ww1 = 0;
#pragma unroll
    for (int st = 0; st < SK_SCAT_SUBT; ++st) {
      const size_t t = (tile * SK_SCAT_SUBT + st) * SK_SCAT_THREADS + threadIdx.x;
      const size_t p0 = t * SK_R;
      runs.valid = 0;
      runs.starts = 0;
      runs.pos[0] = runs.pos[1] = runs.pos[2] = runs.pos[3] = 0;
      ww0 = ww1 = 0;
      if (p0 < seq_len) {
        ww0 = codes[t];
        ww1 = codes[t + 1];
        runs = sk_analyse<W>(ww0, ww1, sk_valid32(bad_window(bad, p0), k), CANON);
        sk_walk(runs, ww0, ww1, nkmax, CANON,
                [&](int, int, unsigned mm) { atomicAdd(&lh[sk_bucket(mm, p1_log2)], 1u); });
      }
      if (st + 1 < SK_SCAT_SUBT) {
        pk_mask[st][threadIdx.x] = make_uint2(runs.valid, runs.starts);
        pk_w[st][threadIdx.x] = make_ulonglong2(ww0, ww1);
        pk_pos[st][0][threadIdx.x] = make_ulonglong2(runs.pos[0], runs.pos[1]);
        pk_pos[st][1][threadIdx.x] = make_ulonglong2(runs.pos[2], runs.pos[3]);
      }
    }
    __syncthreads();
    {
      unsigned v[NB];
      u64 r[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SK_SCAT_THREADS;
        v[i] = b < p1 ? lh[b] : 0u;
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SK_SCAT_THREADS;
#ifdef SK_ABL_NOCURSOR  // (timing ablation only: no reservation, runs land on top of each other)
        r[i] = v[i] ? start[b] : 0ull;
#elif defined(SK_ABL_WGSCOPE)  // (timing ablation only: the add is performed in this XCD's L2 -- not coherent across XCDs)
        r[i] = v[i] ? __hip_atomic_fetch_add(&cursor[b], (u64)v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0ull;
#else
        r[i] = v[i] ? atomicAdd(&cursor[b], (u64)v[i]) : 0ull;
#endif
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const unsigned b = threadIdx.x + i * SK_SCAT_THREADS;
        if (b < p1) {
          // a run that would cross the end of its bucket's region (sampled sizes only) is not written
          const bool fits = v[i] == 0 || r[i] + v[i] <= start[b + 1];
          spilled |= fits ? 0u : 1u;
          lh[b] = fits ? (unsigned)r[i] : SK_NOFIT;
        }
      }
    }
    __syncthreads();
    // pass 2: rank of every record inside its run, store (registers first, then the parked sub-tiles)
#pragma unroll
    for (int st = SK_SCAT_SUBT - 1; st >= 0; --st) {
      if (st + 1 < SK_SCAT_SUBT) {
        const uint2 m = pk_mask[st][threadIdx.x];
        const ulonglong2 w = pk_w[st][threadIdx.x], pa = pk_pos[st][0][threadIdx.x], pb = pk_pos[st][1][threadIdx.x];
        runs.valid = m.x;
        runs.starts = m.y;
        runs.pos[0] = pa.x; runs.pos[1] = pa.y; runs.pos[2] = pb.x; runs.pos[3] = pb.y;
        ww0 = w.x;
        ww1 = w.y;
      }
      const u64 w0 = ww0, w1 = ww1;
      (void)canon;
      sk_walk(runs, w0, w1, nkmax, CANON, [&](int jstart, int nk, unsigned mm) {
        const unsigned b = sk_bucket(mm, p1_log2);
        const unsigned at = atomicAdd(&lh[b], 1u);  // base + rank
        // (the record is built while the LDS answers: pinned here, or the compiler sinks it behind the test of `at`)
        ulonglong2 rec = sk_make_record(w0, w1, jstart, nk, k);
        asm volatile("" : "+v"(rec.x), "+v"(rec.y));
#ifdef SK_ABL_NOSTORE   // (timing ablation only: the record is built and dropped)
        if (at == 0xFFFFFFFFu) part[(size_t)at] = rec;
#else
        if (at < SK_NOFIT) part[(size_t)at] = rec;
#endif
      });
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
  if (spilled) atomicOr(&info->part_overflow, 4ull);
}

// The same scatter with the walks FLATTENED.  In the kernel above every lane walks its own runs, and a loop over
// runs lasts as long as the lane with the most of them: 4.9 runs per thread on average, about 10 for the slowest of
// 64 lanes, so the two walks (two thirds of the kernel's instructions) run half empty.  Here a lane only LISTS its
// runs -- one 32-bit item {lane, first window, windows, minimizer position} per record, written into the wave's
// queue in LDS at a position from a wave prefix sum -- and the wave then works the queue off 64 items at a time with
// every lane busy: pass 1 finds each item's bucket (the words of the lane that listed it come from LDS), counts it
// and writes the bucket into the item; pass 2 needs nothing but the item and those words.  Nothing of the analysis
// lives across the reservation.  A wave whose runs do not fit its queue (8 or 5.9 items per lane, the shape chosen so that the
// mean is well below) walks that sub-tile the old way and analyses it again in pass 2 -- rare, content-dependent, exact.
#ifdef MK_STAMP
static __device__ u64 skq_dbg[1024 * 8];  // per workgroup: time of wave 0 in each phase of mk_sk_scatterq_k
#endif
// Two shapes of a tile (template parameters): 2 sub-tiles with 512-item queues (8 items per lane), and 3 sub-tiles with
// 376-item queues for chunks whose lanes list fewer than ~5.2 records on average (a quarter fewer (tile, bucket)
// reservations: S2 at k = 31, 4.9 per lane: 10.46-10.79 -> 10.32-10.44 ms per step); either keeps the workgroup's LDS
// under 80 KB.  The launcher picks by what the chunk before listed (mk_ctx::items_hint).
#define SKQ_CAP 512      // (the larger of the two: MK_SKQ_CAP is clamped to the shape's own)
#if defined(SK_ABL_COARSE) && !defined(SK_PLAIN_CURSORS)
#error "SK_ABL_COARSE needs -DSK_PLAIN_CURSORS (the ablation's region ends differ from the ones sk_reserve8k checks)"
#endif
#ifndef SKQ_THREADS
#define SKQ_THREADS 512  // two workgroups per CU (72 KB of LDS each): one's reservations and record stores -- memory-side --
#endif                   // run under the other's analysis; 1024: one workgroup per CU, a quarter fewer reservations
#define SKQ_WAVES (SKQ_THREADS / 64)
#define SKQ_WALKED 0xFFFFFFFFu
template <int W, bool CANON, int SKQ_SUBT, int SKQ_QCAP>
__global__ __launch_bounds__(SKQ_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void mk_sk_scatterq_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                         MkChunkInfo* __restrict__ info, const u64* __restrict__ start,
                                                         SkCursor* __restrict__ cursor, ulonglong2* __restrict__ part,
                                                         int p1_log2, int k, int nkmax, size_t ntiles, unsigned qcap, int nseg) {
  // nseg = 9: a bucket has one region per XCD (and a shared one, below) and this workgroup fills the regions of the XCD
  // it runs on: the sectors of a line then reach the memory side from one L2 instead of eight (scatter -10..-17 %; the bytes
  // written stay what they were, 32 per 16-byte record: a partly written sector leaves the L2 long before its neighbour
  // record arrives -- DESIGN.md 4.2, tools/xcd_scatter_probe.hip).
  // The XCD is read from the hardware register, not derived from blockIdx: which region a run lands in is then right
  // whatever the dispatcher does (the cursors are ordinary device-wide atomics either way); only the regions' sizes
  // assume that the tiles go round the XCDs evenly.
  SkCursor* const shared_cursor = cursor + ((size_t)8 << p1_log2);  // (nseg > 1: the buckets' shared regions, see mk_sk_scan_k)
  const u64* const shared_start = start + ((size_t)8 << p1_log2);
  if (nseg > 1) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const size_t seg = xcc & 7u;
    cursor += seg << p1_log2;
    start += seg << p1_log2;
  }
  __shared__ unsigned lh[SK_LH];  // as above: counts, then base + rank -- of the 2^13 buckets of the current round
  // every thread's first word; its second is the next lane's first, and a wave keeps the second word of its last lane
  // itself (pass 1 runs between wave barriers only: a wave must not read what another wave writes)
  __shared__ u64 pk_x[SKQ_SUBT][SKQ_WAVES][65];
  __shared__ unsigned queue[SKQ_SUBT][SKQ_WAVES][SKQ_QCAP];        // items: lane | j << 6 | nk << 11 | (position, then bucket) << 16
  __shared__ unsigned s_abort;
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;  // the regions do not fit the buffers: nothing may be written
  unsigned spilled = 0;
  constexpr int NB = SK_LH / SKQ_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const unsigned lhn = p1 < (unsigned)SK_LH ? p1 : (unsigned)SK_LH;  // counters in use
  const unsigned nround = p1 / lhn;                                  // 1, or 2 with 2^14 buckets
  const size_t seq_len = info->seq_len;
  const int lane = threadIdx.x & 63;
  // (the wave's number, computed again at every use from an empty asm: kept in a register -- with the queue and word-row
  // addresses the compiler derives from it for every sub-tile -- it was spilled to scratch under the kernel's 128 registers
  // and loaded back once per sub-tile)
#define wv sk_wave_id()
  auto sk_wave_id = [&]() { int w = (int)threadIdx.x; asm volatile("" : "+v"(w)); return w >> 6; };
  for (unsigned i = threadIdx.x; i < lhn; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  u64 tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tF = 0, t0 = 0, ntile = 0;
  (void)tA; (void)tB; (void)tC; (void)tD; (void)tE; (void)tF; (void)t0; (void)ntile;
  STAMP(t0);
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    ntile += 1;
    unsigned qn[SKQ_SUBT];  // items queued per sub-tile (wave-uniform), or SKQ_WALKED
#pragma unroll
    for (int st = 0; st < SKQ_SUBT; ++st) {
      const size_t t = (tile * SKQ_SUBT + st) * SKQ_THREADS + threadIdx.x;
      const size_t p0 = t * SK_R;
      SkRuns runs;
      runs.valid = 0;
      runs.starts = 0;
      runs.pos[0] = runs.pos[1] = runs.pos[2] = runs.pos[3] = 0;
      u64 ww0 = 0, ww1 = 0;
      if (p0 < seq_len) {
        ww0 = codes[t];
        ww1 = codes[t + 1];
        runs = sk_analyse<W>(ww0, ww1, sk_valid32(bad_window(bad, p0), k), CANON);
      } else if (p0 < seq_len + SK_R) {
        ww0 = codes[t];  // (the thread before this one is the last with windows: this is its second word)
      }
      pk_x[st][wv][lane] = ww0;
      if (lane == 63) pk_x[st][wv][64] = ww1;
      const unsigned s2 = sk_cut_starts(runs.starts, runs.valid, nkmax);
      const unsigned cnt = __popc(s2);
      const unsigned inc = mk_wave_scan_incl(cnt);  // inclusive scan over the wave
      const unsigned total = mk_wave_last(inc);
      unsigned* const myq = queue[st][wv];
      if (total <= qcap) {  // (qcap <= SKQ_QCAP; tests lower it to walk some or all waves)
        unsigned todo = s2, at = inc - cnt;
        while (todo) {
          const int j = __ffs(todo) - 1;
          todo &= todo - 1;
          const unsigned stop = (s2 | ~runs.valid) & ~((2u << j) - 1);
          const int nk = (stop ? (__ffs(stop) - 1) : SK_R) - j;
          const u64 pw = j < 10 ? runs.pos[0] : (j < 20 ? runs.pos[1] : (j < 30 ? runs.pos[2] : runs.pos[3]));
          const unsigned best = (unsigned)(pw >> (6 * (j % 10))) & 63u;
          myq[at++] = (unsigned)lane | ((unsigned)j << 6) | ((unsigned)nk << 11) | (best << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the wave's own LDS writes, read by other lanes below)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (unsigned base = 0; base < total; base += 64) {
          const unsigned i = base + lane;
          if (i < total) {
            const unsigned it = myq[i];
            const u64* const wp = &pk_x[st][wv][it & 63u];
            const unsigned mm = sk_canon_mmer(sk_mmer(wp[0], wp[1], (int)(it >> 16)), CANON);
#ifdef SK_ABL_COARSE  // (timing ablation only: the first level of a two-level partition -- p1 / 64 coarse buckets)
            const unsigned b = sk_bucket(mm, p1_log2) & ~63u;
#else
            const unsigned b = sk_bucket(mm, p1_log2);
#endif
            if ((b >> SK_LH_LOG2) == 0) atomicAdd(&lh[b & (SK_LH - 1)], 1u);  // (round 0's buckets: counted as they are found)
            myq[i] = (it & 0xFFFFu) | (b << 16);
          }
        }
        qn[st] = total;
      } else {
        if (p0 < seq_len)
          sk_walk(runs, ww0, ww1, nkmax, CANON, [&](int, int, unsigned mm) {
            const unsigned b = sk_bucket(mm, p1_log2);
            if ((b >> SK_LH_LOG2) == 0) atomicAdd(&lh[b & (SK_LH - 1)], 1u);
          });
        qn[st] = SKQ_WALKED;
      }
    }
    // A tile's records are placed in rounds of 2^13 buckets (one round unless the chunk is cut into 2^14): count the
    // round's records per bucket (round 0: done above), reserve their runs, store them, clear the counters.
#pragma unroll 1
    for (unsigned round = 0; round < nround; ++round) {
    SkCursor* const rcursor = cursor + (size_t)round * SK_LH;
    const u64* const rstart = start + (size_t)round * SK_LH;
    if (round) {
#pragma unroll
      for (int st = 0; st < SKQ_SUBT; ++st) {
        if (qn[st] != SKQ_WALKED) {
          const unsigned total = qn[st];
          const unsigned* const myq = queue[st][wv];
          for (unsigned base = 0; base < total; base += 64) {
            const unsigned i = base + lane;
            if (i < total) {
              const unsigned b = myq[i] >> 16;
              if ((b >> SK_LH_LOG2) == round) atomicAdd(&lh[b & (SK_LH - 1)], 1u);
            }
          }
        } else {
          const size_t t = (tile * SKQ_SUBT + st) * SKQ_THREADS + threadIdx.x;
          const size_t p0 = t * SK_R;
          if (p0 < seq_len) {
            const ulonglong2 w = make_ulonglong2(pk_x[st][wv][lane], pk_x[st][wv][lane + 1]);
            const SkRuns runs = sk_analyse<W>(w.x, w.y, sk_valid32(bad_window(bad, p0), k), CANON);
            sk_walk(runs, w.x, w.y, nkmax, CANON, [&](int, int, unsigned mm) {
              const unsigned b = sk_bucket(mm, p1_log2);
              if ((b >> SK_LH_LOG2) == round) atomicAdd(&lh[b & (SK_LH - 1)], 1u);
            });
          }
        }
      }
    }
    STAMP_ADD(tA, t0);
    __syncthreads();
    STAMP_ADD(tB, t0);
#ifndef SK_PLAIN_CURSORS
    if (lhn == SK_LH) {  // (8192 buckets, 8 per thread: all reservations in flight together, mk_skmer_dev.h)
      static_assert(NB % 8 == 0, "sk_reserve8");
#pragma unroll
      for (int h = 0; h < NB; h += 8) {  // (512 threads: buckets 0..4095, then 4096..8191)
        unsigned v[8], at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lh[threadIdx.x + (h + i) * SKQ_THREADS];
        unsigned nofit = sk_reserve8<SKQ_THREADS>(v, rcursor + h * SKQ_THREADS, rstart + h * SKQ_THREADS, SK_NOFIT, at);
        if (nseg > 1 && __any(nofit)) {  // runs that do not fit the XCD's region: once more, in the bucket's shared region
          // The cursor of the XCD's region has moved past the region's end by now and the count kernel reads a region up
          // to its end then: the records between the last run that did fit and the end are nobody's (every later run
          // starts past the end) -- the one run that found them free fills them with empty records.
          unsigned v2[8], at2[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            v2[i] = at[i] >= SK_NOFIT ? v[i] : 0u;
            const unsigned pad = at[i] >= SK_NOFIT ? at[i] & 0xFFFFFFu : 0u;
            if (pad) {
              const u64 end = rstart[threadIdx.x + (h + i) * SKQ_THREADS + 1];
              for (unsigned q = 0; q < pad; ++q) part[end - pad + q] = make_ulonglong2(0, 0);
            }
          }
          nofit = sk_reserve8<SKQ_THREADS>(v2, shared_cursor + (size_t)round * SK_LH + h * SKQ_THREADS,
                                           shared_start + (size_t)round * SK_LH + h * SKQ_THREADS, SK_NOFIT, at2);
#pragma unroll
          for (int i = 0; i < 8; ++i) at[i] = v2[i] ? at2[i] : at[i];
        }
        spilled |= nofit;
#pragma unroll
        for (int i = 0; i < 8; ++i) lh[threadIdx.x + (h + i) * SKQ_THREADS] = at[i];
      }
    } else
#endif
    {  // (fewer buckets than the most -- small chunks: a plain loop, one bucket at a time)
      for (unsigned b = threadIdx.x; b < lhn; b += SKQ_THREADS) {
        const unsigned v = lh[b];
        u64 r = v ? (u64)atomicAdd(&rcursor[b], v) : 0ull;
#ifdef SK_ABL_COARSE
        const bool fits = v == 0 || r + v <= rstart[b + 64 < p1 ? b + 64 : p1];
#else
        bool fits = v == 0 || r + v <= rstart[b + 1];
        if (!fits && nseg > 1) {  // (the bucket's shared region; what was free at the end of the XCD's: empty records, as above)
          for (u64 q = r; q < rstart[b + 1]; ++q) part[q] = make_ulonglong2(0, 0);
          r = (u64)atomicAdd(&shared_cursor[b], v);
          fits = r + v <= shared_start[b + 1];
        }
#endif
        spilled |= fits ? 0u : 1u;
        lh[b] = fits ? (unsigned)r : SK_NOFIT;
      }
    }
    STAMP_ADD(tC, t0);
    __syncthreads();
    STAMP_ADD(tD, t0);
#pragma unroll
    for (int st = 0; st < SKQ_SUBT; ++st) {
      if (qn[st] != SKQ_WALKED) {
        const unsigned total = qn[st];
        const unsigned* const myq = queue[st][wv];
        for (unsigned base = 0; base < total; base += 64) {
          const unsigned i = base + lane;
          if (i < total && ((myq[i] >> 16) >> SK_LH_LOG2) == round) {
            const unsigned it = myq[i];
            const unsigned at = atomicAdd(&lh[(it >> 16) & (SK_LH - 1)], 1u);  // base + rank
            const u64* const wp = &pk_x[st][wv][it & 63u];
            ulonglong2 rec = sk_make_record(wp[0], wp[1], (int)((it >> 6) & 31u), (int)((it >> 11) & 31u), k);
            asm volatile("" : "+v"(rec.x), "+v"(rec.y));  // (built while the LDS answers)
#ifdef SK_NT_STORE  // (timing experiment: streaming stores -- no L2 allocation for the record lines)
            typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
            if (at < SK_NOFIT) __builtin_nontemporal_store(u64x2_t{rec.x, rec.y}, (u64x2_t*)&part[(size_t)at]);
#else
            if (at < SK_NOFIT) part[(size_t)at] = rec;
#endif
          }
        }
      } else {
        const size_t t = (tile * SKQ_SUBT + st) * SKQ_THREADS + threadIdx.x;
        const size_t p0 = t * SK_R;
        if (p0 < seq_len) {
          const ulonglong2 w = make_ulonglong2(pk_x[st][wv][lane], pk_x[st][wv][lane + 1]);
          const SkRuns runs = sk_analyse<W>(w.x, w.y, sk_valid32(bad_window(bad, p0), k), CANON);
          sk_walk(runs, w.x, w.y, nkmax, CANON, [&](int jstart, int nk, unsigned mm) {
            const unsigned b = sk_bucket(mm, p1_log2);
            if ((b >> SK_LH_LOG2) != round) return;
            const unsigned at = atomicAdd(&lh[b & (SK_LH - 1)], 1u);
            if (at < SK_NOFIT) part[(size_t)at] = sk_make_record(w.x, w.y, jstart, nk, k);
          });
        }
      }
    }
    STAMP_ADD(tE, t0);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < lhn; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    STAMP_ADD(tF, t0);
    }  // rounds of 2^13 buckets
  }
#ifdef MK_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 1024) { u64* d = skq_dbg + (size_t)blockIdx.x * 8; d[0] = tA; d[1] = tB; d[2] = tC; d[3] = tD; d[4] = tE; d[5] = tF; d[6] = ntile; }
#endif
  if (spilled) atomicOr(&info->part_overflow, 4ull);
}
#undef wv

#ifdef SK_EXP_SORT
// EXPERIMENT (timing only, not in the product build): the records of every bucket sorted by their number of windows,
// longest first (classes = 8: every length its own class; classes = 2: five windows or more first) -- what a scatter
// that files records by length class would hand the count kernel.  One workgroup per bucket, counting sort through LDS.
__global__ __launch_bounds__(256) void mk_sk_expsort_k(ulonglong2* __restrict__ part, const u64* __restrict__ start,
                                                       const SkCursor* __restrict__ cursor, unsigned p1, int classes) {
  __shared__ ulonglong2 buf[6144];
  __shared__ unsigned cnt[64], base[64];
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 lo = start[b], n = cursor[b] - start[b];
    if (n == 0 || n > 6144) continue;
    if (threadIdx.x < 64) cnt[threadIdx.x] = 0;
    __syncthreads();
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) {
      const ulonglong2 r = part[lo + i];
      buf[i] = r;
      const int nk = (int)(r.y & 63);
      const int cls = classes == 2 ? (nk >= 5 ? 0 : 1) : (32 - nk);
      atomicAdd(&cnt[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) { unsigned a = 0; for (int q = 0; q < 64; ++q) { base[q] = a; a += cnt[q]; } }
    __syncthreads();
    for (u64 i = threadIdx.x; i < n; i += blockDim.x) {
      const ulonglong2 r = buf[i];
      const int nk = (int)(r.y & 63);
      const int cls = classes == 2 ? (nk >= 5 ? 0 : 1) : (32 - nk);
      part[lo + atomicAdd(&base[cls], 1u)] = r;
    }
    __syncthreads();
  }
}
#endif

// ------------------------------------------------------------------------------ launcher

#ifndef SK_TU_CANON
void mk_launch_sk_scan(mk_ctx* c, const u64* hist, const u64* khist, u64* start, SkCursor* cursor, u64* kstart, int p1_log2,
                       int sample_log2, int nkmax, u64 surv_div, u64 part_cap, u64 surv_cap, float sigmas, int nseg) {
  hipLaunchKernelGGL(mk_sk_scan_k, dim3(1), dim3(1024), 0, c->stream, hist, khist, start, cursor, kstart,
                     (MkChunkInfo*)c->info.p, p1_log2, sample_log2, nkmax, surv_div, part_cap, surv_cap, sigmas, nseg);
}

#endif

template <int W, bool CANON>
static void launch_wc(mk_ctx* c, size_t seq_len, int p1_log2, int nkmax, int sample_log2, u64 surv_div, u64 part_cap, u64 surv_cap,
                     u64* hist, u64* start, SkCursor* cursor, u64* khist, u64* kstart, bool reuse, int nseg) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  float sigmas = 6.0f;  // MK_SAMPLE_SIGMAS=0 makes the sampled sizes too small on purpose (tests of the exact second pass)
  if (const char* e = getenv("MK_SAMPLE_SIGMAS")) sigmas = (float)atof(e);
  const size_t threads = div_up(seq_len, SK_R);
  const size_t tiles = div_up(div_up(threads, (size_t)1 << sample_log2), SK_HIST_THREADS);
  const size_t stiles = div_up(threads, (size_t)SK_SCAT_THREADS * SK_SCAT_SUBT);
  // few, long-lived workgroups: each one flushes 2 x p1 global atomics at its end (fewer still for a sample)
  const size_t hist_grid = sample_log2 ? SK_HIST_GRID / 2 : SK_HIST_GRID;
  if (!reuse) {  // (reuse: the regions of the previous chunk stand as they are, cursors back at their starts)
    hipLaunchKernelGGL((mk_sk_hist_k<W, CANON>), dim3((unsigned)(tiles < hist_grid ? tiles : hist_grid)), dim3(SK_HIST_THREADS),
                       2 * sizeof(unsigned) << p1_log2, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, hist, khist, p1_log2, c->k, nkmax, threads, c->canonical,
                       sample_log2);
    mk_launch_sk_scan(c, (const u64*)hist, (const u64*)khist, start, cursor, kstart, p1_log2, sample_log2, nkmax, surv_div, part_cap,
                      surv_cap, sigmas, nseg);
  }
  static const bool walked = getenv("MK_SCATTER_WALK") != nullptr;  // (the per-lane walks of the first version, for A/B runs)
  // three sub-tiles when the lanes of the chunk before listed few enough records for the shorter queues (mean + 3 sigma
  // of a wave's total under 376: sigma ~ 2 per lane); MK_SKQ_SUBT=2|3 forces a shape
  const int force_subt = getenv("MK_SKQ_SUBT") ? atoi(getenv("MK_SKQ_SUBT")) : 0;
  const bool three = force_subt == 3 || (force_subt != 2 && c->items_hint > 0 && c->items_hint * 64.0 + 48.0 < 376.0);
  unsigned qcap = three ? 376u : 512u;
  if (const char* e = getenv("MK_SKQ_CAP")) { const int v = atoi(e); if (v >= 0 && (unsigned)v < qcap) qcap = (unsigned)v; }
  if (walked)
    hipLaunchKernelGGL((mk_sk_scatter_k<W, CANON>), dim3((unsigned)(stiles < SK_SCAT_GRID ? stiles : SK_SCAT_GRID)), dim3(SK_SCAT_THREADS), 0,
                       c->stream, (const u64*)c->codes.p, (const u64*)c->bad.p, info, (const u64*)start, cursor,
                       (ulonglong2*)c->part.p, p1_log2, c->k, nkmax, stiles, c->canonical);
  else {
    const size_t qtiles = div_up(threads, (size_t)SKQ_THREADS * (three ? 3 : 2));
    const dim3 qgrid((unsigned)(qtiles < SK_SCAT_GRID ? qtiles : SK_SCAT_GRID));
    if (three)
      hipLaunchKernelGGL((mk_sk_scatterq_k<W, CANON, 3, 376>), qgrid, dim3(SKQ_THREADS), 0, c->stream, (const u64*)c->codes.p,
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (ulonglong2*)c->part.p, p1_log2, c->k, nkmax, qtiles, qcap, nseg);
    else
      hipLaunchKernelGGL((mk_sk_scatterq_k<W, CANON, 2, 512>), qgrid, dim3(SKQ_THREADS), 0, c->stream, (const u64*)c->codes.p,
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (ulonglong2*)c->part.p, p1_log2, c->k, nkmax, qtiles, qcap, nseg);
  }
#ifdef MK_STAMP
  if (!walked) {
    (void)hipStreamSynchronize(c->stream);
    const size_t qtiles = div_up(threads, (size_t)SKQ_THREADS * (three ? 3 : 2));
    const unsigned g = (unsigned)(qtiles < SK_SCAT_GRID ? qtiles : SK_SCAT_GRID);
    std::vector<u64> h(8 * 1024);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(skq_dbg), h.size() * 8);
    double a[7] = {0, 0, 0, 0, 0, 0, 0};
    for (unsigned w = 0; w < g && w < 1024; ++w) for (int q = 0; q < 7; ++q) a[q] += (double)h[w * 8 + q] / g;
    fprintf(stderr, "[stamp scatterq] per WG (wave 0): analyse+list+pass1=%.0f barrier1=%.0f cursors=%.0f barrier2=%.0f pass2=%.0f barrier3+clear=%.0f tiles=%.2f grid=%u\n",
            a[0], a[1], a[2], a[3], a[4], a[5], a[6], g);
  }
#endif
}


// histogram + scan + scatter for W = k - 10 minimizer candidates per window; false: W out of range
template <bool CANON>
static bool sk_partition(int W, mk_ctx* c, size_t seq_len, int p1_log2, int nkmax, int sample_log2, u64 surv_div, u64 part_cap, u64 surv_cap,
                         u64* hist, u64* start, SkCursor* cursor, u64* khist, u64* kstart, bool reuse, int nseg) {
  switch (W) {
#define SK_CASE(W_)                                                                                                      \
  case W_: launch_wc<W_, CANON>(c, seq_len, p1_log2, nkmax, sample_log2, surv_div, part_cap, surv_cap, hist, start, cursor, khist, kstart, reuse, nseg); return true;
    SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7)
    SK_CASE(8) SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
    SK_CASE(17) SK_CASE(18) SK_CASE(19) SK_CASE(20) SK_CASE(21) SK_CASE(22)
#undef SK_CASE
    default: return false;
  }
}

#ifdef SK_TU_CANON
bool mk_sk_partition_canon(int W, mk_ctx* c, size_t seq_len, int p1_log2, int nkmax, int sample_log2, u64 surv_div, u64 part_cap,
                           u64 surv_cap, u64* hist, u64* start, SkCursor* cursor, u64* khist, u64* kstart, bool reuse, int nseg) {
  return sk_partition<true>(W, c, seq_len, p1_log2, nkmax, sample_log2, surv_div, part_cap, surv_cap, hist, start, cursor, khist, kstart, reuse, nseg);
}
#else
bool mk_sk_partition_canon(int W, mk_ctx* c, size_t seq_len, int p1_log2, int nkmax, int sample_log2, u64 surv_div, u64 part_cap,
                           u64 surv_cap, u64* hist, u64* start, SkCursor* cursor, u64* khist, u64* kstart, bool reuse, int nseg);

// Chunks of one sample are equally long (the Chunker cuts at the first record past the size) and drawn from the
// same text: a chunk whose predecessor sized its buckets from the sampled histogram -- the estimate plus six
// standard deviations plus a floor, about twice the mean -- and did not overflow them INHERITS those regions
// (the count kernel puts every cursor back to its region's start): no histogram, no scan.  A chunk that overflows
// inherited regions is partitioned again exactly, like one that overflows sampled ones, and the next few chunks
// size their buckets afresh.  Called once per partition launch (both key widths); true = inherit.
bool mk_part_inherit(mk_ctx* c, size_t seq_len, int p1_log2, uint64_t min_count, bool sampled, bool exact) {
  if (c->part_cooldown > 0 && !exact) c->part_cooldown -= 1;
  const bool reuse = c->use_reuse && !exact && sampled && c->part_reuse_ok && !c->part_dirty && c->part_cooldown == 0 &&
                     p1_log2 == c->part_prev_p1 && (unsigned long long)min_count == c->part_prev_minc &&
                     seq_len <= c->part_prev_len + c->part_prev_len / 100 && seq_len >= c->part_prev_len - c->part_prev_len / 50;
  if (reuse) c->st.part_reused += 1;
  c->part_dirty = true;  // (until the caller has read the chunk's flags back: process_chunk, process_chunk_fast)
  if (exact) { c->part_reuse_ok = false; c->part_cooldown = 4; }
  else if (sampled && !reuse) {
    c->part_reuse_ok = true;
    c->part_prev_len = seq_len;
    c->part_prev_p1 = p1_log2;
    c->part_prev_minc = (unsigned long long)min_count;
  } else if (!reuse) c->part_reuse_ok = false;
  return reuse;
}

int mk_launch_count_superkmer(mk_ctx* c, size_t seq_len, uint64_t min_count, bool exact) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const int k = c->k;
  // ~1.2K records (~10K windows) per bucket, between 256 and SK_MAX_P1 buckets
  // MK_CORES=1 (experiment, round 4): half-sized buckets -- up to 2^14 of them -- counted by 512-thread workgroups with
  // 4096-slot tables (mk_skcount_small.hip: 68-76 KB of LDS, so that two of them, or one and a scatter workgroup of the
  // other context, share a CU)
  static const bool cores = getenv("MK_CORES") != nullptr;
  const int max_log2 = cores ? SK_MAX_P1_LOG2 : SK_LH_LOG2;
  const size_t bucket_syms = cores ? SK_BUCKET_SYMS / 2 : SK_BUCKET_SYMS;
  int p1_log2 = 8;
  while (p1_log2 < max_log2 && (seq_len >> p1_log2) > bucket_syms) ++p1_log2;
  if (const char* e = getenv("MK_P1_LOG2")) { int v = atoi(e); if (v >= 4 && v <= SK_MAX_P1_LOG2) p1_log2 = v; }
  c->p1_log2 = p1_log2;
  const size_t p1 = (size_t)1 << p1_log2;
  // Runs are cut into records of at most SK_NKMAX windows: the count kernel expands one record per
  // thread, SKC_B k-mers per round, so short uniform records keep its lanes busy (measured: 31 ->
  // 8 trades 1.5x more records for 2.7x fewer expansion rounds).
  int nkmax = 62 - k;
  if (nkmax > SK_NKMAX) nkmax = SK_NKMAX;
  if (const char* e = getenv("MK_NKMAX")) { int v = atoi(e); if (v >= 1 && v <= 31 && v <= 62 - k) nkmax = v; }
  // Bucket sizes from a 1-in-2^s sample of the analysis threads (big chunks only: the exact histogram
  // costs as much as the scatter's own analysis). MK_SAMPLE_LOG2=0 turns it off, MK_SAMPLE_MIN moves
  // the size threshold (tests).
  int sample_log2 = 0;
  {
    int want = 3;
    size_t min_len = (size_t)8 << 20;
    if (const char* e = getenv("MK_SAMPLE_LOG2")) { int v = atoi(e); if (v >= 0 && v <= 6) want = v; }
    if (const char* e = getenv("MK_SAMPLE_MIN")) min_len = (size_t)atoll(e);
    if (!exact && seq_len >= min_len) sample_log2 = want;
  }
  c->part_sampled = sample_log2 != 0;
  // regions per bucket: one per XCD when the sizes come from a sample (the exact partition keeps one: a bucket's exact
  // size says nothing exact about its eighths); MK_XSEG=0 keeps one everywhere (A/B runs, tests)
  const bool xseg_on = !(getenv("MK_XSEG") && atoi(getenv("MK_XSEG")) == 0);
  static const bool walked_env = getenv("MK_SCATTER_WALK") != nullptr;
  static const bool force_pre = getenv("MK_FORCE_PREFILTER") != nullptr;  // (that count kernel reads one region per bucket)
  // (chunks of up to 1 GiB: the nine regions take room for two records per window, 32 bytes per byte of text -- a sample
  // counted unchunked, -s 0, keeps the single region and its 16)
  const int nseg = (xseg_on && sample_log2 != 0 && !walked_env && !force_pre && p1_log2 <= SK_LH_LOG2 &&
                    seq_len <= ((size_t)1 << 30)) ? 9 : 1;
  if (nseg != c->part_nseg) c->part_reuse_ok = false;  // (regions of the other layout cannot be inherited)
  c->part_nseg = nseg;
  const bool reuse = mk_part_inherit(c, seq_len, p1_log2, min_count, sample_log2 != 0, exact);
  int rc;
  // hist p1 | start p1 + 1 | cursor (p1 words) | khist p1 | kstart p1 + 1 | (p1) | nsurv p1 | start of the 9 p1 regions + 1 | their cursors
  if ((rc = mk_buf_reserve(c, c->part_meta, (7 * p1 + 16 + 14 * p1 + 8) * sizeof(u64))) != MK_OK) return rc;
  // worst case one record per window (nine regions per bucket: the shared ones alone have that much, the eight others as
  // much again -- untouched memory for the most part)
  const size_t part_cap = nseg > 1 ? 2 * seq_len + 64 : seq_len + 64;
  if ((rc = mk_buf_reserve(c, c->part, part_cap * sizeof(ulonglong2))) != MK_OK) return rc;
  // a bucket with m k-mers has at most ceil(m / min_count) survivors: that bounds its region
  // (regions sized from a sample carry its error margin: half as much room again plus the per-bucket floor)
  const u64 surv_div = min_count > 1 ? (u64)min_count : 1;
  size_t surv_cap = seq_len / surv_div + p1 + 64;
  if (sample_log2) {
    // sum over the buckets of (estimate + 6 sigma + floor) <= 1.25 L + 6 sqrt(p1 * S * SK_R * 1.25 L) + floor * p1
    // (Cauchy-Schwarz on the sum of square roots; L = seq_len bounds the k-mers)
    const double L = 1.25 * (double)seq_len, S = (double)(1u << sample_log2), w = (double)SK_R;
    surv_cap = (size_t)((L + 6.0 * sqrt((double)p1 * S * w * L) + 16.0 * w * (double)p1) / (double)surv_div) + 2 * p1 + 64;
  }
  if ((rc = mk_buf_reserve(c, c->surv_keys, surv_cap * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_cnts, surv_cap * sizeof(u64))) != MK_OK) return rc;
  u64* hist = (u64*)c->part_meta.p;
  u64* start = hist + p1;
  SkCursor* cursor = (SkCursor*)(start + p1 + 1);  // (packed 32-bit, in the space of p1 64-bit words)
  u64* khist = start + p1 + 1 + p1;
  u64* kstart = khist + p1;
  u64* nsurv = kstart + p1 + 1 + p1;  // (the p1 words in between: a cursor array the 8-byte-key path uses)
  if (nseg > 1) {
    start = hist + 7 * p1 + 16;
    cursor = (SkCursor*)(start + 9 * p1 + 8);
  }
  if (!reuse) MK_HIP(hipMemsetAsync(hist, 0, (7 * p1 + 8) * sizeof(u64), c->stream));
  mk_prof_begin(c, MK_K_PART);
  // (the canonical instances live in a translation unit of their own, mk_skmer_canon.hip: half the compile time each)
  if (c->canonical) {
    if (!mk_sk_partition_canon(k - SK_M + 1, c, seq_len, p1_log2, nkmax, sample_log2, surv_div, (u64)part_cap, (u64)surv_cap, hist, start, cursor, khist, kstart, reuse, nseg)) {
      c->err = "mk_launch_count_superkmer: k out of range";
      return MK_ERR_ARG;
    }
  } else if (!sk_partition<false>(k - SK_M + 1, c, seq_len, p1_log2, nkmax, sample_log2, surv_div, (u64)part_cap, (u64)surv_cap, hist, start, cursor, khist, kstart, reuse, nseg)) {
    c->err = "mk_launch_count_superkmer: k out of range";
    return MK_ERR_ARG;
  }
  mk_prof_end(c);
#ifdef SK_EXP_SORT
  if (const char* e = getenv("MK_EXP_SORT"))
    hipLaunchKernelGGL(mk_sk_expsort_k, dim3(2048), dim3(256), 0, c->stream, (ulonglong2*)c->part.p, (const u64*)start, (const SkCursor*)cursor,
                       (unsigned)p1, atoi(e));
#endif
  {
    const int rc_count = cores ? mk_launch_sk_count_small(c, (const u64*)start, cursor, (const u64*)kstart, nsurv, min_count, nkmax, p1, exact, nseg)
                               : mk_launch_sk_count(c, (const u64*)start, cursor, (const u64*)kstart, nsurv, min_count, nkmax, p1, exact, nseg);
    if (rc_count) return rc_count;
  }
  MK_HIP(hipGetLastError());
  c->surv_regions = 1;
  return MK_OK;
}
#endif  // !SK_TU_CANON
