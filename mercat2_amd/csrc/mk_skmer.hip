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
//   4 mk_sk_count    persistent, one workgroup per CU walks the buckets: expands the records into k-mers
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
#define SK_MAX_P1_LOG2 13
#endif
#define SK_MAX_P1 (1 << SK_MAX_P1_LOG2)
#ifndef SK_BUCKET_SYMS
#define SK_BUCKET_SYMS 8192     // symbols of the chunk per bucket the bucket count aims at (~1.2K records, ~10K windows)
#endif
#define SK_NOFIT 0xFF000000u    // lh[] value of a (tile, bucket) run that does not fit its region: nothing is stored
#ifndef SKC_SLOTS
#define SKC_SLOTS 8192          // LDS table slots of one workgroup (12 bytes each)
#endif
#ifndef SKC_THREADS
#define SKC_THREADS 1024
#endif
#ifndef SKC_WGS
#define SKC_WGS 1               // workgroups per CU the grid is sized for
#endif
#ifndef SKC_TARGET_PCT
#define SKC_TARGET_PCT 40
#endif
#ifndef SKC_LB
#define SKC_LB SKC_THREADS      // launch bound the register budget is derived from
#endif
#ifndef SKC_CAS_FIRST
#define SKC_CAS_FIRST 1  // claim-or-compare with ONE compare-and-swap per key instead of read + conditional swap: the insert is bound by LDS instruction issue, not by active lanes (count kernel -5 %)
#endif
#ifndef SKC_PRE
#define SKC_PRE (2048 / SKC_THREADS)   // record batches (one record per thread each) per load round
#endif
#define SKC_LOADCAP (SKC_SLOTS / 2)
#define SKC_TARGET (SKC_SLOTS * SKC_TARGET_PCT / 100)
#define SKC_SUB_BITS 16
#define SKC_S0_MAX 3               // deepest sub-range split a bucket STARTS with (it splits on as tables overflow)

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
  __shared__ unsigned lh[SK_MAX_P1];  // records per bucket
  __shared__ unsigned lk[SK_MAX_P1];  // k-mers per bucket (bounds the bucket's survivors)
  const unsigned p1 = 1u << p1_log2;
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
__device__ __forceinline__ u64 sk_cap(u64 h, int sample_log2, u64 weight, float sigmas) {
  if (sample_log2 == 0) return h;
  const u64 est = h << sample_log2;
  // single precision is plenty for a margin (the square root is rounded up by the +1)
  const float dev = sigmas * __builtin_sqrtf((float)((u64)weight << sample_log2) * (float)est);
  return est + (u64)dev + 1 + (sigmas > 0 ? 16 * weight : 0);
}
__global__ __launch_bounds__(1024) void mk_sk_scan_k(const u64* __restrict__ hist, const u64* __restrict__ khist,
                                                     u64* __restrict__ start, SkCursor* __restrict__ cursor, u64* __restrict__ kstart,
                                                     MkChunkInfo* __restrict__ info, int p1_log2, int sample_log2, int nkmax,
                                                     u64 div, u64 part_cap, u64 surv_cap, float sigmas) {
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
    cap[q] = on ? sk_cap(hk & 0xFFFFFFFFull, sample_log2, SK_R / SK_NKMAX, sigmas) : 0;
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
  if (threadIdx.x == 1023) {
    const u64 total = run + acc, ktotal = krun + kacc;
    start[p1] = total;
    kstart[p1] = ktotal;
    if (total > part_cap) atomicOr(&info->part_overflow, 1ull);  // (only a sampled estimate can get here)
    if (ktotal > surv_cap) atomicOr(&info->part_overflow, 2ull);
  }
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const unsigned i = lo + q;
    if ((unsigned)q < per && i < p1) {
      start[i] = run;
      cursor[i] = run;
      kstart[i] = krun;
      run += cap[q];
      krun += kcap[q];
    }
  }
}

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
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); __builtin_amdgcn_sched_barrier(0); var = t__; }
#define STAMP_ADD(acc, t0) { unsigned long long t1__; STAMP(t1__); acc += t1__ - t0; t0 = t1__; }
#else
#define STAMP(var)
#define STAMP_ADD(acc, t0)
#endif
#ifdef MK_STAMP
__device__ u64 skq_dbg[1024 * 8];  // per workgroup: time of wave 0 in each phase of mk_sk_scatterq_k
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
                                                         int p1_log2, int k, int nkmax, size_t ntiles, unsigned qcap) {
  __shared__ unsigned lh[SK_MAX_P1];  // as above: counts, then base + rank
  // every thread's first word; its second is the next lane's first, and a wave keeps the second word of its last lane
  // itself (pass 1 runs between wave barriers only: a wave must not read what another wave writes)
  __shared__ u64 pk_x[SKQ_SUBT][SKQ_WAVES][65];
  __shared__ unsigned queue[SKQ_SUBT][SKQ_WAVES][SKQ_QCAP];        // items: lane | j << 6 | nk << 11 | (position, then bucket) << 16
  __shared__ unsigned s_abort;
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;  // the regions do not fit the buffers: nothing may be written
  unsigned spilled = 0;
  constexpr int NB = SK_MAX_P1 / SKQ_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const size_t seq_len = info->seq_len;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
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
      unsigned inc = cnt;  // inclusive scan over the wave
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
      }
      const unsigned total = __shfl(inc, 63);
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
            atomicAdd(&lh[b], 1u);
            myq[i] = (it & 0xFFFFu) | (b << 16);
          }
        }
        qn[st] = total;
      } else {
        if (p0 < seq_len)
          sk_walk(runs, ww0, ww1, nkmax, CANON, [&](int, int, unsigned mm) { atomicAdd(&lh[sk_bucket(mm, p1_log2)], 1u); });
        qn[st] = SKQ_WALKED;
      }
    }
    STAMP_ADD(tA, t0);
    __syncthreads();
    STAMP_ADD(tB, t0);
#ifndef SK_PLAIN_CURSORS
    if (p1 == SK_MAX_P1) {  // (8192 buckets, 8 per thread: all reservations in flight together, mk_skmer_dev.h)
      static_assert(NB % 8 == 0, "sk_reserve8");
#pragma unroll
      for (int h = 0; h < NB; h += 8) {  // (512 threads: buckets 0..4095, then 4096..8191)
        unsigned v[8], at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lh[threadIdx.x + (h + i) * SKQ_THREADS];
        spilled |= sk_reserve8<SKQ_THREADS>(v, cursor + h * SKQ_THREADS, start + h * SKQ_THREADS, SK_NOFIT, at);
#pragma unroll
        for (int i = 0; i < 8; ++i) lh[threadIdx.x + (h + i) * SKQ_THREADS] = at[i];
      }
    } else
#endif
    {  // (fewer buckets than the most -- small chunks: a plain loop, one bucket at a time)
      for (unsigned b = threadIdx.x; b < p1; b += SKQ_THREADS) {
        const unsigned v = lh[b];
        const u64 r = v ? (u64)atomicAdd(&cursor[b], v) : 0ull;
#ifdef SK_ABL_COARSE
        const bool fits = v == 0 || r + v <= start[b + 64 < p1 ? b + 64 : p1];
#else
        const bool fits = v == 0 || r + v <= start[b + 1];
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
          if (i < total) {
            const unsigned it = myq[i];
            const unsigned at = atomicAdd(&lh[it >> 16], 1u);  // base + rank
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
            const unsigned at = atomicAdd(&lh[sk_bucket(mm, p1_log2)], 1u);
            if (at < SK_NOFIT) part[(size_t)at] = sk_make_record(w.x, w.y, jstart, nk, k);
          });
        }
      }
    }
    STAMP_ADD(tE, t0);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    STAMP_ADD(tF, t0);
  }
#ifdef MK_STAMP
  if (threadIdx.x == 0 && blockIdx.x < 1024) { u64* d = skq_dbg + (size_t)blockIdx.x * 8; d[0] = tA; d[1] = tB; d[2] = tC; d[3] = tD; d[4] = tE; d[5] = tF; d[6] = ntile; }
#endif
  if (spilled) atomicOr(&info->part_overflow, 4ull);
}

// ------------------------------------------------------------------------------ 4 count
// What the insert costs (measured, tools/lds_probe.hip and the ISA of the round-1 kernel): the kernel is
// bound by VALU issue, not by the LDS.  One compare-and-swap plus one add per key take ~30 clocks of the
// CU's LDS per 64 keys; the round-1 kernel spent ~150.  Where it went: (a) every key that did not find its
// home slot took a serial probe loop, inlined and unrolled 16 x 48 times (110 KB of code, 1072 spilled
// SGPRs), and nearly every wave has a few such keys in every slot of its batch, so the whole wave walked
// eight probe loops with a handful of active lanes; (b) three quarter-rate 32-bit multiplies per key.
// Now: (a) a key whose home slot holds another key is DEFERRED: it goes onto a per-wave stack in LDS (slot
// positions from ballots, no atomic) and the wave probes 64 deferred keys at a time, every lane busy;
// (b) the slot hash is three full-rate 24-bit multiplies.

// 32-bit hash of a packed key for the LDS table: bits 31..19 pick the slot, bits 15..0 the sub-range.
// Three 24-bit multiplies (v_mul_u32_u24 issues at full rate; a 32-bit multiply at a quarter) over the three
// 24-bit pieces of the key; as even as a random function on the keys of a bucket (windows of the same loci,
// shifted by one base: tools/hash_quality.py).
__device__ __forceinline__ unsigned skc_hash(u64 key) {
  const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
  const unsigned mid = __funnelshift_r(lo, hi, 24);  // bits 24..55 (the multiply takes its low 24)
  return __umul24(lo, 0x9E3779u) ^ __umul24(mid, 0x85EBCBu) ^ __umul24(hi >> 16, 0xC2B2AFu);
}

#define SKC_MAX_PROBE 48  // longer chains mean the table is too full for this sub-range: split it
// home slot of a hash and the slot d steps further (any table size; a power of two costs a shift and a mask)
__device__ __forceinline__ unsigned skc_home(unsigned h) {
  if constexpr ((SKC_SLOTS & (SKC_SLOTS - 1)) == 0) return h >> (32 - __builtin_ctz(SKC_SLOTS));
  else return (unsigned)(((u64)h * SKC_SLOTS) >> 32);
}
__device__ __forceinline__ unsigned skc_step(unsigned slot, unsigned d) {
  slot += d;
  if constexpr ((SKC_SLOTS & (SKC_SLOTS - 1)) == 0) return slot & (SKC_SLOTS - 1);
  else return slot >= SKC_SLOTS ? slot - SKC_SLOTS : slot;
}

// Which record of a load round a thread takes: batch h, record h * SKC_THREADS + ..; odd batches hand the 64-record
// groups to the waves in reverse order, so that when a bucket's records come sorted by length (longest first) every
// wave gets a long group and a short one
#ifdef SKC_DYN
#define SKC_JMAP(h) ((u64)(h) * SKC_THREADS + (((h) & 1) ? (unsigned)(SKC_THREADS - 64 - (threadIdx.x & ~63u)) + (threadIdx.x & 63u) : threadIdx.x))
#else
#define SKC_JMAP(h) ((u64)(h) * SKC_THREADS + threadIdx.x)
#endif
// A bucket's records are read once, front to back: loaded past the L2's replacement order (SKC_NT_LOAD), so that the
// 243 MB a chunk's count kernel streams do not push the open lines of the OTHER context's scatter out of the L2s.
__device__ __forceinline__ ulonglong2 skc_ldrec(const ulonglong2* __restrict__ p) {
#ifdef SKC_NT_LOAD
  typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
  const u64x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u64x2_t*>(p));
  return make_ulonglong2(v.x, v.y);
#else
  return *p;
#endif
}
#define SKC_B 8          // k-mers of a record expanded and probed together
#define SKC_WAVES (SKC_THREADS / 64)
#ifndef SKC_PUSH
#define SKC_PUSH 4       // slots whose deferred keys are pushed before the stack is looked at again
#endif
#define SKC_QCAP (64 + 64 * SKC_PUSH)  // deferred keys a wave can hold: < 64 left over + SKC_PUSH slots x 64 lanes

__device__ __forceinline__ unsigned skc_lane_rank(u64 mask) {  // set bits of mask below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// The top n (<= 64) deferred keys of this wave's stack: linear probing from the slot after the home slot
// (the home slot is known to hold another key), one key per lane.
__device__ __forceinline__ void skc_drain(u64* tkey, unsigned* tcnt, const u64* q, unsigned& qcount, unsigned n,
                                              unsigned* s_overflow) {
  const unsigned lane = threadIdx.x & 63;
  qcount -= n;
  if (lane < n) {
    const u64 key = q[qcount + lane];
    unsigned slot = skc_step(skc_home(skc_hash(key)), 1);
    bool placed = false;
#pragma unroll 1
    for (int probe = 0; probe < SKC_MAX_PROBE; ++probe) {
      u64 cur = tkey[slot];
      if (cur == MK_EMPTY) {
        cur = atomicCAS(&tkey[slot], MK_EMPTY, key);
        if (cur == MK_EMPTY) cur = key;
      }
      if (cur == key) {
        atomicAdd(&tcnt[slot], 1u);
        placed = true;
        break;
      }
      slot = skc_step(slot, 1);
    }
    if (!placed) atomicOr(s_overflow, 1u);  // (a plain volatile LDS store here trips a gfx950 backend assertion in ROCm 7.2)
  }
}


// Persistent: gridDim.x workgroups (one per CU) walk the buckets b = blockIdx.x, +gridDim.x, ...
// The next bucket's bounds and its first two record batches are loaded while the current
// bucket is being emitted, so no global-memory latency sits on the critical path.
// K32: k == 32, the only k whose keys can equal the free-slot mark (32 x 'T'): that key is counted aside.
template <bool CANON, bool K32>
__global__ __launch_bounds__(SKC_LB) void mk_sk_count_k(const ulonglong2* __restrict__ part, const u64* __restrict__ start,
                                                             SkCursor* __restrict__ cursor,
                                                             const u64* __restrict__ kstart, u64* __restrict__ nsurv,
                                                             MkChunkInfo* __restrict__ info, u64 min_count,
                                                             u64* __restrict__ out_keys, u64* __restrict__ out_cnts,
                                                             int k, unsigned p1, double dup_hint, double nk_hint, u64* __restrict__ dbg,
                                                             int dflags) {
  __shared__ __attribute__((aligned(16))) u64 tkey[SKC_SLOTS];
  __shared__ __attribute__((aligned(16))) unsigned tcnt[SKC_SLOTS];
  __shared__ __attribute__((aligned(16))) u64 wq[SKC_WAVES][SKC_QCAP];  // deferred keys, one stack per wave
  // per-pass flags, double-buffered by pass parity so that resetting them needs no extra barrier
  __shared__ unsigned s_distinct[2], s_overflow[2], s_emit[2];
  __shared__ unsigned long long s_windows;
  __shared__ unsigned s_abort;  // (read once per workgroup: other workgroups of this launch may set the flag meanwhile)
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;  // the scatter did not fit its (sampled) regions: the chunk is partitioned again
  for (unsigned i = threadIdx.x; i < SKC_SLOTS; i += blockDim.x) { tkey[i] = MK_EMPTY; tcnt[i] = 0; }
  if (threadIdx.x < 2) { s_distinct[threadIdx.x] = 0; s_overflow[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; }
  if (threadIdx.x == 0) s_windows = 0;
  __syncthreads();
  unsigned par = 0;
  const int kshift = 64 - 2 * k;
  const int lane = threadIdx.x & 63;
  u64* const myq = wq[threadIdx.x >> 6];
  u64 distinct_total = 0, side = 0, survivors_total = 0, nerr = 0;
  u64 windows = 0, records_total = 0;  // what the chunk held (every record is expanded at least once)
  u64 tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tF = 0, t0 = 0, npass = 0;
  STAMP(t0);

  // prefetched state of the bucket about to be processed
  unsigned bn = blockIdx.x;
  u64 lo_n = 0, hi_n = 0, ks_n = 0, ke_n = 0;  // records [lo_n, hi_n), survivor region [ks_n, ke_n)
  ulonglong2 pre[SKC_PRE];
#pragma unroll
  for (int h = 0; h < SKC_PRE; ++h) pre[h] = make_ulonglong2(0, 0);
  if (bn < p1) {
    lo_n = start[bn];
    hi_n = cursor[bn];
    ks_n = kstart[bn];
    ke_n = kstart[bn + 1];
#pragma unroll
    for (int h = 0; h < SKC_PRE; ++h) {
      const u64 j = SKC_JMAP(h);
      if (j < hi_n - lo_n) pre[h] = skc_ldrec(part + lo_n + j);
    }
  }
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 lo = lo_n, n = hi_n - lo_n;  // records of this bucket
    u64* __restrict__ my_keys = out_keys + ks_n;
    u64* __restrict__ my_cnts = out_cnts + ks_n;
    const u64 region = ke_n - ks_n;
    ulonglong2 first[SKC_PRE];
#pragma unroll
    for (int h = 0; h < SKC_PRE; ++h) first[h] = pre[h];
    // bounds of the next bucket: in flight while this one is counted
    bn = b + gridDim.x;
    if (bn < p1) {
      lo_n = start[bn];
      hi_n = cursor[bn];
      ks_n = kstart[bn];
      ke_n = kstart[bn + 1];
    }
    unsigned emitted = 0;
    if (n >> 27) {  // 2^27 records x 31 k-mers would overflow the 32-bit LDS counters
      ++nerr;
    } else if (n) {
      int s0 = 0;
      {
        const double expect = (double)n * nk_hint / (dup_hint > 1.0 ? dup_hint : 1.0);
        while (s0 < SKC_SUB_BITS && expect / (double)(1u << s0) > (double)SKC_TARGET) ++s0;
        if ((double)n * 31.0 <= (double)SKC_LOADCAP) s0 = 0;
        // The estimate knows nothing about THIS bucket: a homopolymer puts millions of windows of one k-mer
        // here, and 2^s0 passes over them took 30 s for 2 Mbases of poly-A.  Start no deeper than 8 sub-ranges;
        // a sub-range that overflows is split further anyway, and its pass stops at the first overflow.
        if (s0 > SKC_S0_MAX) s0 = SKC_S0_MAX;
        if (dflags & 16) s0 = 0;  // (timing experiments only)
      }
      int s = s0;
      unsigned idx = 0;
      records_total += n;
      u64 side_pass = 0, win_pass = 0;
      bool side_done = false;
      bool first_pass = true;
      const ulonglong2* __restrict__ src = part + lo;
      for (;;) {
        const unsigned sel_shift = SKC_SUB_BITS - s;
        side_pass = 0;  // the all-ones key (32 x 'T') is counted aside, once per bucket
        win_pass = 0;
        bool over = false;
        unsigned* const ovf = &s_overflow[par];
        unsigned qcount = 0;  // this wave's deferred keys (wave-uniform)
        for (u64 rb2 = 0; rb2 < n && !over; rb2 += SKC_PRE * SKC_THREADS) {
          ulonglong2 recs2[SKC_PRE];
          if (first_pass && rb2 == 0) {
#pragma unroll
            for (int h = 0; h < SKC_PRE; ++h) recs2[h] = first[h];
          } else {
#pragma unroll
            for (int h = 0; h < SKC_PRE; ++h) {
              const u64 j = rb2 + SKC_JMAP(h);
              recs2[h] = j < n ? skc_ldrec(src + j) : make_ulonglong2(0, 0);
            }
          }
          STAMP_ADD(tF, t0);
#pragma unroll
          for (int h = 0; h < SKC_PRE; ++h) {
            // ---- one record per thread, expanded 8 k-mers at a time: the 8 compare-and-swaps on the home
            //      slots are issued together, then the adds of the keys that found (or claimed) their slot;
            //      the others are deferred
            const ulonglong2 rec = recs2[h];
            const int nk = (int)(rec.y & 63);
            win_pass += side_done ? 0 : (u64)nk;
            u64 x = rec.x, y = rec.y;
            // (the whole wave walks the loop together -- lanes without a record or with a short one just have
            // no live slots -- because the deferred-key stack below is the wave's: every lane takes part)
            // One round = up to NB consecutive k-mers of every lane's record.  NB is a compile-time constant of the
            // body; with SKC_DYN the wave picks the body that fits its LONGEST record (4, 6 or 8 slots: scalar
            // branch, no per-slot tests), which pays when the records a wave holds are about equally long.
            auto round = [&](auto nb_tag, int base) {
              constexpr int NB = decltype(nb_tag)::value;
              u64 kk[NB], cur[NB];
              unsigned hh[NB];
              unsigned live = 0;  // bit u: slot u holds a key of this pass
              // every key's compare-and-swap is issued as soon as its slot is known, so that the hashing of the
              // later keys runs while the earlier ones are on their way through the LDS
              // canonical keys: the reverse complement ROLLS with the window -- the base that enters the key on the
              // right enters its reverse complement, complemented, on the left -- one full reversal per round
              u64 rcv = CANON ? mk_revcomp2(x >> kshift, k) : 0ull;
#pragma unroll
              for (int u = 0; u < NB; ++u) {
                const u64 fw = x >> kshift;
                kk[u] = (CANON && rcv < fw) ? rcv : fw;
                x = (x << 2) | (y >> 62);
                y <<= 2;
                if (CANON) rcv = (rcv >> 2) | ((((x >> kshift) & 3ull) ^ 3ull) << (2 * k - 2));
                hh[u] = skc_hash(kk[u]);
                bool on = base + u < nk;
                if (K32 && on && kk[u] == MK_EMPTY) {
                  side_pass += side_done ? 0 : 1;
                  on = false;
                }
                if (s && ((hh[u] & ((1u << SKC_SUB_BITS) - 1)) >> sel_shift) != idx) on = false;
                live |= on ? (1u << u) : 0u;
                if (on) cur[u] = atomicCAS(&tkey[skc_home(hh[u])], MK_EMPTY, kk[u]);
#ifdef SKC_SCHED_FENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
              }
              unsigned fail = 0;
#pragma unroll
              for (int u = 0; u < NB; ++u)
                if ((live >> u) & 1u) {
                  if (cur[u] == MK_EMPTY || cur[u] == kk[u]) atomicAdd(&tcnt[skc_home(hh[u])], 1u);
                  else fail |= 1u << u;
                }
              // deferred keys -> the wave's stack (positions from ballots: no atomic), four slots at a time
              // so that the stack never holds more than SKC_QCAP; full groups of 64 are probed right away
#pragma unroll
              for (int half = 0; half < NB; half += SKC_PUSH) {
#pragma unroll
                for (int u = half; u < half + SKC_PUSH && u < NB; ++u) {
                  const bool f = (fail >> u) & 1u;
                  const u64 m = __ballot(f);
                  if (m) {
                    if (f) myq[qcount + skc_lane_rank(m)] = kk[u];
                    qcount += (unsigned)__popcll(m);
                  }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                while (qcount >= 64) skc_drain(tkey, tcnt, myq, qcount, 64u, ovf);
              }
            };
#ifdef SKC_DYN
            int wmax = nk;  // the wave's longest record (wave-uniform)
            for (int d = 32; d > 0; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d));
            wmax = __builtin_amdgcn_readfirstlane(wmax);
            for (int base = 0; base < wmax;) {
              const int left = wmax - base;
              if (left <= 4) { round(std::integral_constant<int, 4>{}, base); base += 4; }
              else if (left <= 6) { round(std::integral_constant<int, 6>{}, base); base += 6; }
              else { round(std::integral_constant<int, SKC_B>{}, base); base += SKC_B; }
            }
#else
            for (int base = 0; __any(base < nk); base += SKC_B) round(std::integral_constant<int, SKC_B>{}, base);
#endif
          }
          STAMP_ADD(tC, t0);
          if (*(volatile unsigned*)ovf) over = true;  // hint only; decided after the barrier below
        }
        STAMP_ADD(tA, t0);
        if (qcount) skc_drain(tkey, tcnt, myq, qcount, qcount, ovf);  // (< 64 left)
        first_pass = false;
        STAMP_ADD(tB, t0);
        __syncthreads();  // A: every insert of the pass is in the table
        // (every wave has long read this bucket's bounds: put its cursor back to the region's start, so that the next
        // chunk can inherit the regions without a histogram and a scan -- see the launcher)
        if (threadIdx.x == 0) cursor[b] = lo;
        STAMP_ADD(tF, t0);
        ++npass;
        over = s_overflow[par] != 0;
        if (threadIdx.x == 0) { s_distinct[par ^ 1] = 0; s_overflow[par ^ 1] = 0; s_emit[par ^ 1] = 0; }  // next pass's set
        // will this be the bucket's last pass? then start loading the next bucket's records now
        bool last = false;
        if (!over) {
          int s2 = s;
          unsigned i2 = idx;
          while (s2 > s0 && (i2 & 1u)) { i2 >>= 1; --s2; }
          last = (s2 == s0) && (i2 + 1 >= (1u << s0));
        }
        if (last && bn < p1) {
#pragma unroll
          for (int h = 0; h < SKC_PRE; ++h) {
            const u64 j = SKC_JMAP(h);
            pre[h] = (j < hi_n - lo_n) ? skc_ldrec(part + lo_n + j) : make_ulonglong2(0, 0);
          }
        }
        // ---- emit (when complete) into the bucket's own region, and clear.  The sweep reads the COUNTS only
        //      (two neighbouring slots per access) and the key of a slot only when its count reaches min_count
        //      -- with -c 10 that is one slot in a hundred; a slot is occupied iff its count is not zero
        {
          constexpr int PER = SKC_SLOTS / SKC_THREADS;
          static_assert(PER % 2 == 0, "the sweep takes slot pairs");
          unsigned ec[PER];
          unsigned keep = 0;  // bit q: slot q of this thread survives
          unsigned occ = 0;
#pragma unroll
          for (int q = 0; q < PER; q += 2) {
            const unsigned i = (q * SKC_THREADS + 2 * threadIdx.x);
            const uint2 cp = *reinterpret_cast<const uint2*>(&tcnt[i]);
            ec[q] = cp.x;
            ec[q + 1] = cp.y;
            occ += (cp.x != 0) + (cp.y != 0);
            keep |= (!over && cp.x && (u64)cp.x >= min_count) ? (1u << q) : 0u;
            keep |= (!over && cp.y && (u64)cp.y >= min_count) ? (2u << q) : 0u;
          }
          for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d);
          if (lane == 0 && occ && !over) atomicAdd(&s_distinct[par], occ);
          if (keep) {
            unsigned mine = (unsigned)__popc(keep);
            const unsigned at = emitted + atomicAdd(&s_emit[par], mine);  // LDS cursor inside the region
            unsigned o = 0;
            if ((u64)at + mine > region) {  // only a region sized from a sampled histogram can be too small
              atomicOr(&info->part_overflow, 8ull);
              mine = 0;
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              if (mine && ((keep >> q) & 1u)) {
                my_keys[at + o] = tkey[(q & ~1) * SKC_THREADS + 2 * threadIdx.x + (q & 1)];
                my_cnts[at + o] = ec[q];
                ++o;
              }
            }
          }
#pragma unroll
          for (int q = 0; q < PER; q += 2) {
            const unsigned i = (q * SKC_THREADS + 2 * threadIdx.x);
            *reinterpret_cast<ulonglong2*>(&tkey[i]) = make_ulonglong2(MK_EMPTY, MK_EMPTY);
            *reinterpret_cast<uint2*>(&tcnt[i]) = make_uint2(0u, 0u);
          }
        }
        __syncthreads();  // B: table is clear, counters of this pass are final
        emitted += s_emit[par];
        distinct_total += s_distinct[par];
        par ^= 1;
        STAMP_ADD(tD, t0);
        if (over) {
          if (s >= SKC_SUB_BITS) { ++nerr; break; }
          s += 1;
          idx <<= 1;
        } else {
          side += side_pass;
          windows += win_pass;
          side_done = true;
          while (s > s0 && (idx & 1u)) { idx >>= 1; --s; }
          if (s == s0) {
            ++idx;
            if (idx >= (1u << s0)) break;
          } else {
            ++idx;
          }
        }
      }
    }
    if (n == 0 || (n >> 27)) {
      // nothing was prefetched for the next bucket by a "last pass": do it here
      if (bn < p1) {
#pragma unroll
        for (int h = 0; h < SKC_PRE; ++h) {
          const u64 j = SKC_JMAP(h);
          pre[h] = (j < hi_n - lo_n) ? skc_ldrec(part + lo_n + j) : make_ulonglong2(0, 0);
        }
      }
    }
    if (threadIdx.x == 0) nsurv[b] = emitted;
    survivors_total += emitted;
    STAMP_ADD(tE, t0);
  }
  {  // one global add per workgroup (adds to one address are serialised by the L2: ~4 ns each)
    for (int d = 32; d > 0; d >>= 1) windows += __shfl_down(windows, d);
    if (lane == 0 && windows) atomicAdd(&s_windows, (unsigned long long)windows);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (s_windows) atomicAdd(&info->windows, (u64)s_windows);
    if (records_total) atomicAdd(&info->records, records_total);
    if (distinct_total) atomicAdd(&info->distinct, distinct_total);
    if (survivors_total) atomicAdd(&info->survivors, survivors_total);
    if (nerr) atomicAdd(&info->errors, nerr);
#ifdef MK_STAMP
    if (dbg) { u64* d = dbg + (size_t)blockIdx.x * 8; d[0] = tA; d[1] = tB; d[2] = tC; d[3] = tD; d[4] = tE; d[5] = tF; d[6] = npass; }
#endif
  }
  if (K32) wave_add(&info->side, side);
}

// ------------------------------------------------------------------- count with a counting pre-filter
// With -c well above the mean count of a key (S2: 75 M windows over 19.6 M distinct keys per chunk, -c 10) nearly every
// insert of the kernel above -- compare-and-swap of the 64-bit key, add, deferred-key stack -- feeds a slot that the
// emit sweep throws away.  Here every bucket is walked twice (as in mk_skmer2.hip, where the case is made at length):
//   P  every key adds 1 to one of 16 384 32-bit counters in LDS (a count-min row): never below the count of a key that
//      maps to it;
//   Q  keys whose counter reached min_count (every key that can survive, plus the few that share a counter) are
//      inserted into a small exact table, every occurrence of them; the rest costs one LDS read.
// `distinct` = counters in use (a lower bound).
// MEASURED (round 3, S2 chunk, k = 31, -c 10): 437 us against the exact kernel's 300 -- for one-word keys the tuned
// single pass (compare-and-swap as soon as a slot is known, deferred-key stacks, the next bucket's records prefetched)
// beats two plain passes; it is the two-word kernel, with its lock / write / publish protocol and its sub-range passes,
// that the pre-filter more than halves (mk_skmer2.hip).  So this kernel is NOT the default: MK_FORCE_PREFILTER=1 selects
// it (tests keep it exact: tests/test_gpu_parity.py::test_counting_prefilter_kernels_are_exact).
#define SKP_CNT 16384
#define SKP_SLOTS 2048
#define SKP_MAX_PROBE 64
#define SKP_QCAP 128  // candidates a wave can hold: < 64 left over + one slot x 64 lanes

__device__ __forceinline__ void skp_insert(u64* tkey, unsigned* tcnt, unsigned* ovf, u64 key, unsigned h) {
  unsigned slot = h >> (32 - 11);  // SKP_SLOTS = 2^11
  bool done = false;
#pragma unroll 1
  for (int probe = 0; probe < SKP_MAX_PROBE && !done; ++probe) {
    u64 cur = tkey[slot];
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&tkey[slot], MK_EMPTY, key);
      if (cur == MK_EMPTY) cur = key;
    }
    if (cur == key) {
      atomicAdd(&tcnt[slot], 1u);
      done = true;
    } else {
      slot = (slot + 1) & (SKP_SLOTS - 1);
    }
  }
  if (!done) atomicOr(ovf, 1u);
}

template <bool CANON, bool K32>
__global__ __launch_bounds__(SKC_THREADS) void mk_sk_countp_k(const ulonglong2* __restrict__ part, const u64* __restrict__ start,
                                                              SkCursor* __restrict__ cursor, const u64* __restrict__ kstart,
                                                              u64* __restrict__ nsurv, MkChunkInfo* __restrict__ info, u64 min_count,
                                                              u64* __restrict__ out_keys, u64* __restrict__ out_cnts, int k, unsigned p1) {
  __shared__ unsigned cnt32[SKP_CNT];
  __shared__ __attribute__((aligned(16))) u64 tkey[SKP_SLOTS];
  __shared__ unsigned tcnt[SKP_SLOTS];
  __shared__ __attribute__((aligned(16))) u64 cq[SKC_WAVES][SKP_QCAP];  // candidates, one stack per wave (positions from ballots)
  __shared__ unsigned s_distinct[2], s_overflow[2], s_emit[2];
  __shared__ unsigned long long s_windows;
  __shared__ unsigned s_abort;
  if (threadIdx.x == 0) { s_abort = info->part_overflow != 0; s_windows = 0; }
  __syncthreads();
  if (s_abort) return;
  for (unsigned i = threadIdx.x; i < SKP_CNT; i += blockDim.x) cnt32[i] = 0;
  for (unsigned i = threadIdx.x; i < SKP_SLOTS; i += blockDim.x) { tkey[i] = MK_EMPTY; tcnt[i] = 0; }
  if (threadIdx.x < 2) { s_distinct[threadIdx.x] = 0; s_overflow[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; }
  __syncthreads();
  unsigned par = 0;
  const int kshift = 64 - 2 * k;
  const int lane = threadIdx.x & 63;
  const unsigned need = min_count > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)min_count;
  u64 distinct_total = 0, side = 0, survivors_total = 0, nerr = 0, windows = 0, records_total = 0;
  // as in mk_sk_count_k: the next bucket's bounds and its first records are loaded while this one is swept, and a
  // bucket's first SKC_PRE x 1024 records (nearly always all of them) stay in registers from P to Q
  unsigned bn = blockIdx.x;
  u64 lo_n = 0, hi_n = 0, ks_n = 0, ke_n = 0;
  ulonglong2 pre[SKC_PRE];
#pragma unroll
  for (int h = 0; h < SKC_PRE; ++h) pre[h] = make_ulonglong2(0, 0);
  if (bn < p1) {
    lo_n = start[bn];
    hi_n = cursor[bn];
    ks_n = kstart[bn];
    ke_n = kstart[bn + 1];
#pragma unroll
    for (int h = 0; h < SKC_PRE; ++h) {
      const u64 j = (u64)h * SKC_THREADS + threadIdx.x;
      if (j < hi_n - lo_n) pre[h] = part[lo_n + j];
    }
  }
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 lo = lo_n, n = hi_n - lo_n;
    u64* __restrict__ my_keys = out_keys + ks_n;
    u64* __restrict__ my_cnts = out_cnts + ks_n;
    const u64 region = ke_n - ks_n;
    ulonglong2 first[SKC_PRE];
#pragma unroll
    for (int h = 0; h < SKC_PRE; ++h) first[h] = pre[h];
    bn = b + gridDim.x;
    if (bn < p1) {
      lo_n = start[bn];
      hi_n = cursor[bn];
      ks_n = kstart[bn];
      ke_n = kstart[bn + 1];
    }
    unsigned emitted = 0;
    bool counted = false;
    bool fetched = false;  // the next bucket's records are in pre[]
    records_total += n;
    if (n >> 27) {  // 2^27 records x 31 k-mers would overflow the 32-bit LDS counters
      ++nerr;
    } else if (n) {
      int s = 0;
      unsigned idx = 0;
      const ulonglong2* __restrict__ src = part + lo;
      for (;;) {
        const unsigned sel_shift = 32 - s;
        unsigned* const ovf = &s_overflow[par];
        u64 win_pass = 0, side_pass = 0;
        // ---- P: eight fire-and-forget LDS adds per record, nothing waits for an answer.  The hashes (and which slots
        //      are live) of the records that stay in registers are kept for Q: that pass then costs a read and a compare
        //      per key, and the key itself is rebuilt only for the rare candidate
        unsigned hs[SKC_PRE][SKC_B], lives[SKC_PRE];
        for (u64 rb2 = 0; rb2 < n; rb2 += SKC_PRE * SKC_THREADS) {
#pragma unroll
          for (int h = 0; h < SKC_PRE; ++h) {
            ulonglong2 rec;
            if (rb2 == 0) rec = first[h];
            else {
              const u64 j = rb2 + (u64)h * SKC_THREADS + threadIdx.x;
              rec = j < n ? src[j] : make_ulonglong2(0, 0);
            }
            const int nk = (int)(rec.y & 63);
            win_pass += counted ? 0 : (u64)nk;
            u64 x = rec.x, y = rec.y;
            u64 rcv = CANON ? mk_revcomp2(x >> kshift, k) : 0ull;
            unsigned live = 0;
#pragma unroll
            for (int u = 0; u < SKC_B; ++u) {
              const u64 fw = x >> kshift;
              const u64 key = (CANON && rcv < fw) ? rcv : fw;
              x = (x << 2) | (y >> 62);
              y <<= 2;
              if (CANON) rcv = (rcv >> 2) | ((((x >> kshift) & 3ull) ^ 3ull) << (2 * k - 2));
              const unsigned hv = skc_hash(key);
              bool on = u < nk && (!s || (hv >> sel_shift) == idx);
              if (K32 && on && key == MK_EMPTY) {
                side_pass += counted ? 0 : 1;
                on = false;
              }
              if (on) atomicAdd(&cnt32[hv & (SKP_CNT - 1)], 1u);
              live |= on ? (1u << u) : 0u;
              if (rb2 == 0) hs[h][u] = hv;
            }
            if (rb2 == 0) lives[h] = live;
          }
        }
        __syncthreads();
        // ---- Q: eight independent LDS reads per record; the rare candidate goes onto the wave's stack, and the wave
        //      inserts 64 of them at a time with every lane busy (one by one in the lane that found them, the inserts'
        //      LDS round trips ran one after the other: that alone made the kernel slower than the exact one)
        u64* const myq = cq[threadIdx.x >> 6];
        unsigned qcount = 0;
        for (u64 rb2 = 0; rb2 < n; rb2 += SKC_PRE * SKC_THREADS) {
#pragma unroll
          for (int h = 0; h < SKC_PRE; ++h) {
            ulonglong2 rec;
            if (rb2 == 0) rec = first[h];
            else {
              const u64 j = rb2 + (u64)h * SKC_THREADS + threadIdx.x;
              rec = j < n ? src[j] : make_ulonglong2(0, 0);
            }
            unsigned hh[SKC_B], cv[SKC_B];
            unsigned live = 0;
            if (rb2 == 0) {
              live = lives[h];
#pragma unroll
              for (int u = 0; u < SKC_B; ++u) hh[u] = hs[h][u];
            } else {
              const int nk = (int)(rec.y & 63);
              u64 x = rec.x, y = rec.y;
              u64 rcv = CANON ? mk_revcomp2(x >> kshift, k) : 0ull;
#pragma unroll
              for (int u = 0; u < SKC_B; ++u) {
                const u64 fw = x >> kshift;
                const u64 key = (CANON && rcv < fw) ? rcv : fw;
                x = (x << 2) | (y >> 62);
                y <<= 2;
                if (CANON) rcv = (rcv >> 2) | ((((x >> kshift) & 3ull) ^ 3ull) << (2 * k - 2));
                hh[u] = skc_hash(key);
                bool on = u < nk && (!s || (hh[u] >> sel_shift) == idx);
                if (K32 && key == MK_EMPTY) on = false;
                live |= on ? (1u << u) : 0u;
              }
            }
#pragma unroll
            for (int u = 0; u < SKC_B; ++u) cv[u] = ((live >> u) & 1u) ? cnt32[hh[u] & (SKP_CNT - 1)] : 0u;
            unsigned cand = 0;
#pragma unroll
            for (int u = 0; u < SKC_B; ++u) cand |= (cv[u] >= need && cv[u]) ? (1u << u) : 0u;
            if (__any(cand != 0)) {
#pragma unroll
              for (int u = 0; u < SKC_B; ++u) {
                const bool f = (cand >> u) & 1u;
                const u64 m = __ballot(f);
                if (m) {
                  if (f) {  // window u of the record: its 2k bits start 2u bits into (x : y)
                    const u64 sx = u ? ((rec.x << (2 * u)) | (rec.y >> (64 - 2 * u))) : rec.x;
                    myq[qcount + skc_lane_rank(m)] = mk_canon2(sx >> kshift, k, CANON);
                  }
                  qcount += (unsigned)__popcll(m);
                  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                  if (qcount >= 64) {
                    qcount -= 64;
                    const u64 key = myq[qcount + lane];
                    skp_insert(tkey, tcnt, ovf, key, skc_hash(key) * 0x9E3779B1u);
                  }
                }
              }
            }
          }
        }
        if (qcount) {  // (< 64 left)
          if ((unsigned)lane < qcount) {
            const u64 key = myq[lane];
            skp_insert(tkey, tcnt, ovf, key, skc_hash(key) * 0x9E3779B1u);
          }
          qcount = 0;
        }
        __syncthreads();  // A
        if (threadIdx.x == 0) cursor[b] = lo;
        const bool over = s_overflow[par] != 0;
        if (threadIdx.x == 0) { s_distinct[par ^ 1] = 0; s_overflow[par ^ 1] = 0; s_emit[par ^ 1] = 0; }
        // the bucket's last pass? then the next bucket's records start loading now
        if (!over && !fetched) {
          int s2 = s;
          unsigned i2 = idx;
          while (s2 > 0 && (i2 & 1u)) { i2 >>= 1; --s2; }
          if (s2 == 0) {
            fetched = true;
            if (bn < p1) {
#pragma unroll
              for (int h = 0; h < SKC_PRE; ++h) {
                const u64 j = (u64)h * SKC_THREADS + threadIdx.x;
                pre[h] = (j < hi_n - lo_n) ? part[lo_n + j] : make_ulonglong2(0, 0);
              }
            }
          }
        }
        {
          unsigned occ = 0;
#pragma unroll
          for (int q = 0; q < SKP_CNT / SKC_THREADS; q += 4) {
            const unsigned i = (q * SKC_THREADS + 4 * threadIdx.x);
            const uint4 c4 = *reinterpret_cast<const uint4*>(&cnt32[i]);
            occ += (c4.x != 0) + (c4.y != 0) + (c4.z != 0) + (c4.w != 0);
            *reinterpret_cast<uint4*>(&cnt32[i]) = make_uint4(0u, 0u, 0u, 0u);
          }
          for (int d = 32; d > 0; d >>= 1) occ += __shfl_down(occ, d);
          if (lane == 0 && occ && !over) atomicAdd(&s_distinct[par], occ);
          constexpr int PER = SKP_SLOTS / SKC_THREADS;
          unsigned ec[PER];
          unsigned mine = 0;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const unsigned i = q * SKC_THREADS + threadIdx.x;
            ec[q] = tcnt[i];
            if (over || (u64)ec[q] < min_count) ec[q] = 0;
            mine += ec[q] != 0;
          }
          if (mine) {
            const unsigned at = emitted + atomicAdd(&s_emit[par], mine);
            unsigned o = 0;
            if ((u64)at + mine > region) {
              atomicOr(&info->part_overflow, 8ull);
              mine = 0;
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              if (mine && ec[q]) {
                my_keys[at + o] = tkey[q * SKC_THREADS + threadIdx.x];
                my_cnts[at + o] = ec[q];
                ++o;
              }
            }
          }
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const unsigned i = q * SKC_THREADS + threadIdx.x;
            tkey[i] = MK_EMPTY;
            tcnt[i] = 0;
          }
        }
        __syncthreads();  // B
        emitted += s_emit[par];
        distinct_total += s_distinct[par];
        par ^= 1;
        if (over) {
          if (s >= 16) { ++nerr; break; }
          s += 1;
          idx <<= 1;
        } else {
          windows += win_pass;
          side += side_pass;
          counted = true;
          while (s > 0 && (idx & 1u)) { idx >>= 1; --s; }
          if (s == 0) break;
          ++idx;
        }
      }
    }
    if (!fetched && bn < p1) {  // (an empty or refused bucket: nothing was prefetched by a last pass)
#pragma unroll
      for (int h = 0; h < SKC_PRE; ++h) {
        const u64 j = (u64)h * SKC_THREADS + threadIdx.x;
        pre[h] = (j < hi_n - lo_n) ? part[lo_n + j] : make_ulonglong2(0, 0);
      }
    }
    if (threadIdx.x == 0) nsurv[b] = emitted;
    survivors_total += emitted;
  }
  {
    for (int d = 32; d > 0; d >>= 1) windows += __shfl_down(windows, d);
    if (lane == 0 && windows) atomicAdd(&s_windows, (unsigned long long)windows);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (s_windows) atomicAdd(&info->windows, (u64)s_windows);
    if (records_total) atomicAdd(&info->records, records_total);
    if (distinct_total) atomicAdd(&info->distinct, distinct_total);
    if (survivors_total) atomicAdd(&info->survivors, survivors_total);
    if (nerr) atomicAdd(&info->errors, nerr);
  }
  if (K32) wave_add(&info->side, side);
}

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

void mk_launch_sk_scan(mk_ctx* c, const u64* hist, const u64* khist, u64* start, SkCursor* cursor, u64* kstart, int p1_log2,
                       int sample_log2, int nkmax, u64 surv_div, u64 part_cap, u64 surv_cap, float sigmas) {
  hipLaunchKernelGGL(mk_sk_scan_k, dim3(1), dim3(1024), 0, c->stream, hist, khist, start, cursor, kstart,
                     (MkChunkInfo*)c->info.p, p1_log2, sample_log2, nkmax, surv_div, part_cap, surv_cap, sigmas);
}

template <int W, bool CANON>
static void launch_wc(mk_ctx* c, size_t seq_len, int p1_log2, int nkmax, int sample_log2, u64 surv_div, u64 part_cap, u64 surv_cap,
                     u64* hist, u64* start, SkCursor* cursor, u64* khist, u64* kstart, bool reuse) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  float sigmas = 6.0f;  // MK_SAMPLE_SIGMAS=0 makes the sampled sizes too small on purpose (tests of the exact second pass)
  if (const char* e = getenv("MK_SAMPLE_SIGMAS")) sigmas = (float)atof(e);
  const size_t threads = div_up(seq_len, SK_R);
  const size_t tiles = div_up(div_up(threads, (size_t)1 << sample_log2), SK_HIST_THREADS);
  const size_t stiles = div_up(threads, (size_t)SK_SCAT_THREADS * SK_SCAT_SUBT);
  // few, long-lived workgroups: each one flushes 2 x p1 global atomics at its end (fewer still for a sample)
  const size_t hist_grid = sample_log2 ? SK_HIST_GRID / 2 : SK_HIST_GRID;
  if (!reuse) {  // (reuse: the regions of the previous chunk stand as they are, cursors back at their starts)
    hipLaunchKernelGGL((mk_sk_hist_k<W, CANON>), dim3((unsigned)(tiles < hist_grid ? tiles : hist_grid)), dim3(SK_HIST_THREADS), 0, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, hist, khist, p1_log2, c->k, nkmax, threads, c->canonical,
                       sample_log2);
    hipLaunchKernelGGL(mk_sk_scan_k, dim3(1), dim3(1024), 0, c->stream, (const u64*)hist, (const u64*)khist, start, cursor, kstart,
                       info, p1_log2, sample_log2, nkmax, surv_div, part_cap, surv_cap, sigmas);
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
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (ulonglong2*)c->part.p, p1_log2, c->k, nkmax, qtiles, qcap);
    else
      hipLaunchKernelGGL((mk_sk_scatterq_k<W, CANON, 2, 512>), qgrid, dim3(SKQ_THREADS), 0, c->stream, (const u64*)c->codes.p,
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (ulonglong2*)c->part.p, p1_log2, c->k, nkmax, qtiles, qcap);
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

#ifdef MK_STAMP
u64* mk_dbg_ptr = nullptr;
#endif

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
  int p1_log2 = 8;
  while (p1_log2 < SK_MAX_P1_LOG2 && (seq_len >> p1_log2) > SK_BUCKET_SYMS) ++p1_log2;
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
  const bool reuse = mk_part_inherit(c, seq_len, p1_log2, min_count, sample_log2 != 0, exact);
  int rc;
  if ((rc = mk_buf_reserve(c, c->part_meta, (7 * p1 + 16) * sizeof(u64))) != MK_OK) return rc;
  // worst case one record per window
  const size_t part_cap = seq_len + 64;
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
  if (!reuse) MK_HIP(hipMemsetAsync(hist, 0, (7 * p1 + 8) * sizeof(u64), c->stream));
  mk_prof_begin(c, MK_K_PART);
  switch (k - SK_M + 1) {
#define SK_CASE(W)                                                                                                      \
  case W:                                                                                                               \
    if (c->canonical) launch_wc<W, true>(c, seq_len, p1_log2, nkmax, sample_log2, surv_div, (u64)part_cap, (u64)surv_cap, hist, start, cursor, khist, kstart, reuse); \
    else launch_wc<W, false>(c, seq_len, p1_log2, nkmax, sample_log2, surv_div, (u64)part_cap, (u64)surv_cap, hist, start, cursor, khist, kstart, reuse); \
    break;
    SK_CASE(2) SK_CASE(3) SK_CASE(4) SK_CASE(5) SK_CASE(6) SK_CASE(7)
    SK_CASE(8) SK_CASE(9) SK_CASE(10) SK_CASE(11) SK_CASE(12) SK_CASE(13) SK_CASE(14) SK_CASE(15) SK_CASE(16)
    SK_CASE(17) SK_CASE(18) SK_CASE(19) SK_CASE(20) SK_CASE(21) SK_CASE(22)
#undef SK_CASE
    default:
      c->err = "mk_launch_count_superkmer: k out of range";
      return MK_ERR_ARG;
  }
  mk_prof_end(c);
#ifdef SK_EXP_SORT
  if (const char* e = getenv("MK_EXP_SORT"))
    hipLaunchKernelGGL(mk_sk_expsort_k, dim3(2048), dim3(256), 0, c->stream, (ulonglong2*)c->part.p, (const u64*)start, (const SkCursor*)cursor,
                       (unsigned)p1, atoi(e));
#endif
  mk_prof_begin(c, MK_K_COUNT);
  {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    const unsigned grid = (unsigned)((size_t)ncu * SKC_WGS < p1 ? (size_t)ncu * SKC_WGS : p1);
    u64* dbgbuf = nullptr;
#ifdef MK_STAMP
    if (!mk_dbg_ptr) (void)hipMalloc((void**)&mk_dbg_ptr, 8 * 8 * 1024);
    dbgbuf = mk_dbg_ptr;
#endif
    const int dflags = getenv("MK_DBG") ? atoi(getenv("MK_DBG")) : 0;
    static const bool no_pre = getenv("MK_NO_PREFILTER") != nullptr;
    static const bool force_pre = getenv("MK_FORCE_PREFILTER") != nullptr;
    const bool pre = !no_pre && !exact && min_count >= 2 && nkmax <= SKC_B && force_pre;  // (opt-in only: see mk_sk_countp_k)
#define SKP_LAUNCH(CANON, K32)                                                                                          \
  hipLaunchKernelGGL((mk_sk_countp_k<CANON, K32>), dim3(grid), dim3(SKC_THREADS), 0, c->stream, (const ulonglong2*)c->part.p, \
                     (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count,                        \
                     (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, k, (unsigned)p1)
    if (pre) {
      if (c->canonical) { if (k == 32) SKP_LAUNCH(true, true); else SKP_LAUNCH(true, false); }
      else { if (k == 32) SKP_LAUNCH(false, true); else SKP_LAUNCH(false, false); }
    } else {
#define SKC_LAUNCH(CANON, K32)                                                                                          \
  hipLaunchKernelGGL((mk_sk_count_k<CANON, K32>), dim3(grid), dim3(SKC_THREADS), 0, c->stream, (const ulonglong2*)c->part.p, \
                     (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count,                        \
                     (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, k, (unsigned)p1, c->dup_hint, c->nk_hint, dbgbuf, dflags)
    if (c->canonical) { if (k == 32) SKC_LAUNCH(true, true); else SKC_LAUNCH(true, false); }
    else { if (k == 32) SKC_LAUNCH(false, true); else SKC_LAUNCH(false, false); }
#undef SKC_LAUNCH
    }
#undef SKP_LAUNCH
  }
  mk_prof_end(c);
#ifdef MK_STAMP
  {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    std::vector<u64> h(8 * ncu);
    (void)hipStreamSynchronize(c->stream);
    u64* d = nullptr;
    {
      static u64* s_dbg2 = nullptr; (void)s_dbg2;
    }
    d = mk_dbg_ptr;
    if (d) { (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
      double a[7] = {0,0,0,0,0,0,0}; for (int w = 0; w < ncu; ++w) for (int q = 0; q < 7; ++q) a[q] += (double)h[w * 8 + q] / ncu;
      fprintf(stderr, "[stamp] per-WG cycles: loop_exit=%.0f last_drain=%.0f insert=%.0f emit=%.0f bucket_tail=%.0f loads+barrierA=%.0f passes=%.1f\n", a[0], a[1], a[2], a[3], a[4], a[5], a[6]); }
  }
#endif
  MK_HIP(hipGetLastError());
  c->surv_regions = 1;
  return MK_OK;
}
