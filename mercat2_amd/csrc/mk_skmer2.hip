// mk_skmer2.hip -- super-k-mer partitioned counting for nucleotide 33 <= k <= 64 (two-word keys).
//
// Same scheme as mk_skmer.hip (which see) with these differences:
//   * a window's bucket comes from the minimizer of its FIRST 32 bases (22 candidate 11-mers):
//     still a function of the window's content only, and it lets the analysis reuse the k = 32
//     instantiation; runs are cut into records of <= 8 windows;
//   * a record is 32 bytes: up to 8 + 63 = 71 bases (142 bits) in three words + the run length;
//   * the LDS table holds 128-bit keys {hi, lo} + a 32-bit count. There is no 128-bit
//     compare-and-swap, so the COUNT word is the slot's state: 0 = free, LOCK = being written,
//     otherwise the count. A lane claims a free slot by CAS(count, 0 -> LOCK), writes hi/lo and
//     publishes with one atomic add (LOCK -> 1); lanes that meet LOCK look again on their next
//     loop iteration (the owner finishes within the iteration in which it won, so lanes of the
//     same wave cannot deadlock);
//   * survivors leave the kernel as {hi, lo, count} in per-bucket regions and are added to the
//     by-reference running table (text keys, mk_table.hip) by mk_import_ref128_regions_k -- the
//     running table, the export and the multi-GPU merge are the ones the other large-k paths use.
// Reference semantics: lib/mercat2_kmers.py:56-60 (every window +1), :73-76 (count >= min_count
// per chunk).
#include "mk_skmer_dev.h"
#include <cstdlib>

#define SK2_R SK_R
#define SK2_HIST_THREADS 1024   // (two 64 KB LDS histograms per workgroup: one workgroup per CU)
#define SK2_SCAT_THREADS 1024
#ifndef SK2_MAX_P1_LOG2
#define SK2_MAX_P1_LOG2 13        // 8192 buckets (14 measured: the count kernel saves 1-2 ms per S2 step, the scatter
#endif                           // loses 2: twice the cursor atomics and half the records per run)
#define SK2_MAX_P1 (1 << SK2_MAX_P1_LOG2)
#define SK2_NKMAX 8
#ifndef SK2C_SLOTS
#define SK2C_SLOTS 6144   // 20 bytes each: 120 KB of LDS, + 32 KB of deferred-key stacks (4096: 25 % slower at k = 63, twice the sub-range passes)
#endif
#define SK2C_WAVES (SK2C_THREADS / 64)
#define SK2C_QCAP 128     // deferred keys a wave can hold: < 64 left over + one slot x 64 lanes pushed at once
#define SK2C_THREADS 1024
#ifndef SK2C_TARGET
#define SK2C_TARGET (SK2C_SLOTS * 6 / 10)
#endif
#ifndef SK2C_LOADCAP
#define SK2C_LOADCAP (SK2C_SLOTS * 3 / 4)
#endif
#define SK2C_SUB_BITS 16
#define SK2C_MAX_PROBE 64
#define SK2C_LOCK 0x80000000u

static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

// The minimizer of a window is taken over the 22 candidate 11-mers of its FIRST 32 bases: the analysis is the
// k = 32 instantiation of mk_skmer_dev.h's.
typedef SkRuns Sk2Runs;
__device__ __forceinline__ Sk2Runs sk2_analyse(u64 w0, u64 w1, unsigned valid) { return sk_analyse<22>(w0, w1, valid, false); }
template <class F>
__device__ __forceinline__ void sk2_walk(const Sk2Runs& r, u64 w0, u64 w1, F&& emit) { sk_walk(r, w0, w1, SK2_NKMAX, false, emit); }
__device__ __forceinline__ unsigned sk2_bucket(unsigned mm, int p1_log2) { return sk_bucket(mm, p1_log2); }

// ---- canonical mode (opt-in extension, mk_set_canonical): a window and its reverse complement must be filed in the
// same bucket.  The minimizer of the first 32 bases is not shared by the two strands; the smallest canonical 11-mer
// among the candidates of the FIRST 32 and of the LAST 32 bases is: position p of a window is position k - 11 - p of
// its reverse complement, so the two candidate sets {0..21} and {k-32..k-11} swap, and a canonical 11-mer is its own
// mirror image.  A window is filed under that VALUE (its order hash, a bijection of the 11-mer): runs are windows with
// equal values.
template <class F>
__device__ __forceinline__ void sk2c_region(u64 a0, u64 a1, F&& put) {
  constexpr int W = 22, NQ = SK_R + W - 1, P = 16;
  unsigned ord[NQ];
  unsigned mm = sk_mmer(a0, a1, 0);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q) {
      const int pos = q + SK_M - 1;
      const unsigned base = (unsigned)((pos < 32 ? (a0 >> (62 - 2 * pos)) : (a1 >> (62 - 2 * (pos - 32)))) & 3u);
      mm = ((mm << 2) | base) & SK_MASK;
    }
    ord[q] = sk_order_hash(sk_canon_mmer(mm, true));
  }
#pragma unroll
  for (int step = 1; step < P; step <<= 1) {
#pragma unroll
    for (int q = 0; q + step < NQ; ++q) ord[q] = min(ord[q], ord[q + step]);
  }
#pragma unroll
  for (int j = 0; j < SK_R; ++j) put(j, min(ord[j], ord[j + W - P]));
}

struct Sk2CRuns {
  unsigned valid, starts;
  unsigned hv[SK_R];  // filing value of every window
};

__device__ __forceinline__ Sk2CRuns sk2c_analyse(u64 w0, u64 w1, u64 w2, unsigned valid, int k) {
  Sk2CRuns r;
  r.valid = valid;
  sk2c_region(w0, w1, [&](int j, unsigned v) { r.hv[j] = v; });
  const int d = k - 32;  // 1 .. 32: where the last 32 bases of window 0 start
  const u64 v0 = d == 32 ? w1 : ((w0 << (2 * d)) | (w1 >> (64 - 2 * d)));
  const u64 v1 = d == 32 ? w2 : ((w1 << (2 * d)) | (w2 >> (64 - 2 * d)));
  sk2c_region(v0, v1, [&](int j, unsigned v) { r.hv[j] = min(r.hv[j], v); });
  unsigned starts = valid & 1u;
#pragma unroll
  for (int j = 1; j < SK_R; ++j) {
    const bool ok = (valid >> j) & 1u, prev = (valid >> (j - 1)) & 1u;
    starts |= (ok && (!prev || r.hv[j] != r.hv[j - 1])) ? (1u << j) : 0u;
  }
  r.starts = starts;
  return r;
}

template <class F>
__device__ __forceinline__ void sk2c_walk(const Sk2CRuns& r, F&& emit) {
#pragma unroll
  for (int j = 0; j < SK_R; ++j) {  // (static indices into hv: it lives in registers)
    if (!((r.starts >> j) & 1u)) continue;
    const unsigned stop = (r.starts | ~r.valid) & ~((2u << j) - 1);
    int nk = (stop ? (__ffs(stop) - 1) : SK_R) - j;
    int at = j;
    while (nk > 0) {
      const int take = nk < SK2_NKMAX ? nk : SK2_NKMAX;
      emit(at, take, r.hv[j]);
      at += take;
      nk -= take;
    }
  }
}

// min(key, reverse complement) of a two-word key: k bases left-aligned in {hi, lo}.
__device__ __forceinline__ u64 sk2_revpairs(u64 x) {  // the 32 two-bit groups of x in reverse order
  const u64 y = __brevll(x);
  return ((y & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((y & 0x5555555555555555ull) << 1);
}
__device__ __forceinline__ void sk2_canon128(u64& hi, u64& lo, int k) {
  // complement, reverse all 64 groups (the 64 - k padding groups, now 'T', come first), shift the k bases back up
  const u64 a = sk2_revpairs(~lo), b = sk2_revpairs(~hi);
  const int s = 128 - 2 * k;  // 0 .. 62
  const u64 nh = s ? ((a << s) | (b >> (64 - s))) : a;
  const u64 nl = b << s;
  if (nh < hi || (nh == hi && nl < lo)) { hi = nh; lo = nl; }
}

// Windows j = 0..31 whose k (<= 64) bases are clean; p0 is a multiple of 32.
__device__ __forceinline__ unsigned sk2_valid32(const u64* __restrict__ bad, size_t p0, int k) {
  const size_t bi = p0 >> 6;
  u64 b_lo, b_hi;
  if (p0 & 63) {
    const u64 x0 = bad[bi], x1 = bad[bi + 1], x2 = bad[bi + 2];
    b_lo = (x0 >> 32) | (x1 << 32);
    b_hi = (x1 >> 32) | (x2 << 32);
  } else {
    b_lo = bad[bi];
    b_hi = bad[bi + 1];
  }
  if ((b_lo | (b_hi & 0xFFFFFFFFull)) == 0) return ~0u;  // bits 0..95 are all the 32 windows can touch
  const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
  unsigned v = 0;
#pragma unroll
  for (int j = 0; j < SK2_R; ++j) {
    const u64 win = (b_lo >> j) | (j ? (b_hi << (64 - j)) : 0ull);  // bad bits j .. j+63
    v |= ((win & kmask) == 0) ? (1u << j) : 0u;
  }
  return v;
}

struct __attribute__((aligned(32))) Sk2Rec {
  u64 r0, r1, r2, nk;
};

__device__ __forceinline__ Sk2Rec sk2_make_record(u64 w0, u64 w1, u64 w2, u64 w3, int jstart, int nk, int k) {
  const int s = 2 * jstart;
  Sk2Rec r;
  r.r0 = s ? ((w0 << s) | (w1 >> (64 - s))) : w0;
  r.r1 = s ? ((w1 << s) | (w2 >> (64 - s))) : w1;
  r.r2 = s ? ((w2 << s) | (w3 >> (64 - s))) : w2;
  const int bits = 2 * (nk + k - 1);  // 66 .. 142
  if (bits <= 128) { r.r2 = 0; if (bits < 128) r.r1 &= ~0ull << (128 - bits); }
  else r.r2 &= ~0ull << (192 - bits);
  r.nk = (u64)nk;
  return r;
}

// ------------------------------------------------------------------------------ hist / scatter
// (sample_log2 > 0: one pseudo-randomly chosen analysis thread of every 2^sample_log2, as in mk_skmer.hip)
template <bool CANON>
__global__ __launch_bounds__(SK2_HIST_THREADS) void mk_sk2_hist_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                  const MkChunkInfo* __restrict__ info, u64* __restrict__ hist,
                                                                  u64* __restrict__ khist, int p1_log2, int k,
                                                                  size_t nthreads_total, int sample_log2) {
  __shared__ unsigned lh[SK2_MAX_P1];
  __shared__ unsigned lk[SK2_MAX_P1];
  const unsigned p1 = 1u << p1_log2;
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) { lh[i] = 0; lk[i] = 0; }
  __syncthreads();
  const size_t seq_len = info->seq_len;
  const size_t ngroups = (nthreads_total + ((size_t)1 << sample_log2) - 1) >> sample_log2;
  for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * blockDim.x) {
    size_t t = g;
    if (sample_log2) t = (g << sample_log2) + (((unsigned)g * 0x9E3779B1u >> 7) & ((1u << sample_log2) - 1));
    const size_t p0 = t * SK2_R;
    if (t >= nthreads_total || p0 >= seq_len) continue;
    const u64 w0 = codes[t], w1 = codes[t + 1];
    auto tally = [&](int, int nk, unsigned mm) {
      const unsigned b = sk2_bucket(mm, p1_log2);
      atomicAdd(&lh[b], 1u);
      atomicAdd(&lk[b], (unsigned)nk);
    };
    if constexpr (CANON) {
      const Sk2CRuns r = sk2c_analyse(w0, w1, codes[t + 2], sk2_valid32(bad, p0, k), k);
      sk2c_walk(r, tally);
    } else {
      const Sk2Runs r = sk2_analyse(w0, w1, sk2_valid32(bad, p0, k));
      sk2_walk(r, w0, w1, tally);
    }
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < p1; b += blockDim.x) {
    const unsigned v = lh[b];
    // one global add per bucket: records in the low word, k-mers in the high word (a chunk holds fewer than 2^32
    // symbols, so neither half can carry) -- the per-workgroup flush is most of this kernel's HBM traffic
    if (v) atomicAdd(&hist[b], (u64)v | ((u64)lk[b] << 32));
  }
}

template <bool CANON>
__global__ __launch_bounds__(SK2_SCAT_THREADS) void mk_sk2_scatter_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                     MkChunkInfo* __restrict__ info, const u64* __restrict__ start,
                                                                     SkCursor* __restrict__ cursor, Sk2Rec* __restrict__ part,
                                                                     int p1_log2, int k, size_t ntiles) {
  __shared__ unsigned lh[SK2_MAX_P1];
  __shared__ unsigned gbase[SK2_MAX_P1];  // (record indices stay below 2^32: the caller checks the chunk size)
  __shared__ unsigned s_abort;  // (read once per workgroup: other workgroups of this launch may set the flag meanwhile)
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;  // the regions do not fit the buffers: nothing may be written
  unsigned spilled = 0;
  constexpr int NB = SK2_MAX_P1 / SK2_SCAT_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const size_t seq_len = info->seq_len;
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const size_t t = tile * SK2_SCAT_THREADS + threadIdx.x;
    const size_t p0 = t * SK2_R;
    Sk2Runs runs;
    Sk2CRuns cruns;  // (canonical mode)
    runs.valid = 0;
    runs.starts = 0;
    cruns.valid = 0;
    cruns.starts = 0;
    u64 w0 = 0, w1 = 0;
    if (p0 < seq_len) {
      w0 = codes[t];
      w1 = codes[t + 1];
      auto tally = [&](int, int, unsigned mm) { atomicAdd(&lh[sk2_bucket(mm, p1_log2)], 1u); };
      if constexpr (CANON) {
        cruns = sk2c_analyse(w0, w1, codes[t + 2], sk2_valid32(bad, p0, k), k);
        sk2c_walk(cruns, tally);
      } else {
        runs = sk2_analyse(w0, w1, sk2_valid32(bad, p0, k));
        sk2_walk(runs, w0, w1, tally);
      }
    }
    __syncthreads();
#ifndef SK_PLAIN_CURSORS
    if (p1 == SK2_MAX_P1) {  // (8192 buckets, 8 per thread: all reservations in flight together, mk_skmer_dev.h)
      static_assert(NB == 8 && SK2_SCAT_THREADS == 1024, "sk_reserve8");
      unsigned v[NB], at[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) v[i] = lh[threadIdx.x + i * SK2_SCAT_THREADS];
      spilled |= sk_reserve8<SK2_SCAT_THREADS>(v, cursor, start, ~0u, at);
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        gbase[threadIdx.x + i * SK2_SCAT_THREADS] = at[i];
        lh[threadIdx.x + i * SK2_SCAT_THREADS] = 0;
      }
    } else
#endif
    {  // (small chunks, fewer buckets: one at a time)
      for (unsigned b = threadIdx.x; b < p1; b += SK2_SCAT_THREADS) {
        const unsigned v = lh[b];
        const u64 r = v ? (u64)atomicAdd(&cursor[b], v) : 0ull;
        // a run that would cross the end of its bucket's region (sampled sizes only) is not written
        const bool fits = v == 0 || r + v <= start[b + 1];
        spilled |= fits ? 0u : 1u;
        gbase[b] = fits ? (unsigned)r : ~0u;
        lh[b] = 0;
      }
    }
    __syncthreads();
    if (CANON ? cruns.starts : runs.starts) {
      const u64 w2 = codes[t + 2], w3 = codes[t + 3];
      auto store = [&](int jstart, int nk, unsigned mm) {
        const unsigned b = sk2_bucket(mm, p1_log2);
        const unsigned base = gbase[b];
        const unsigned rank = atomicAdd(&lh[b], 1u);
        if (base != ~0u) part[(size_t)base + rank] = sk2_make_record(w0, w1, w2, w3, jstart, nk, k);
      };
      if constexpr (CANON) sk2c_walk(cruns, store);
      else sk2_walk(runs, w0, w1, store);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
  if (spilled) atomicOr(&info->part_overflow, 4ull);
}

// The same scatter with two sub-tiles per tile (half the cursor atomics per record) and the walks FLATTENED through an
// LDS queue, as mk_sk_scatterq_k does for one-word keys (mk_skmer.hip, where the case is made): a lane only LISTS its
// runs -- one 32-bit item {lane, first window, windows, minimizer position, later the bucket} per record -- and the wave
// works its queue off 64 items at a time with every lane busy.  A record needs FOUR packed words here; thread t holds
// words t and t + 1 (its 32 bases and the 32 after them), so its words 2 and 3 are the SECOND words of its two right
// neighbours: every sub-tile parks its threads' word pairs in LDS plus the pairs of the two threads to its right.  Forward-strand keys only: the canonical analysis files a window under a VALUE, not
// under a position in its first 32 bases, and keeps the kernel above.
#define SK2Q_CAP 512  // (the larger of the two tile shapes: 2 sub-tiles x 512 items, 3 x 376, as in mk_skmer.hip)
#ifndef SK2Q_THREADS
#define SK2Q_THREADS 512  // two workgroups per CU, as mk_sk_scatterq_k (72 KB of LDS each)
#endif
#define SK2Q_WAVES (SK2Q_THREADS / 64)
#define SK2Q_WALKED 0xFFFFFFFFu
#define SK2_NOFIT 0xFF000000u
template <int SK2Q_SUBT, int SK2Q_QCAP>
__global__ __launch_bounds__(SK2Q_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void mk_sk2_scatterq_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                                      MkChunkInfo* __restrict__ info, const u64* __restrict__ start,
                                                                      SkCursor* __restrict__ cursor, Sk2Rec* __restrict__ part,
                                                                      int p1_log2, int k, size_t ntiles, unsigned qcap) {
  __shared__ unsigned lh[SK2_MAX_P1];  // counts, then base + rank (record indices stay below SK2_NOFIT: the launcher checks)
  // every thread's first word, wave by wave, and the three words after the wave's last lane: a thread's words 1..3 are
  // the first words of the three threads to its right (pass 1 runs between wave barriers only: a wave reads its own row)
  __shared__ u64 pk_x[SK2Q_SUBT][SK2Q_WAVES][67];
  __shared__ unsigned queue[SK2Q_SUBT][SK2Q_WAVES][SK2Q_QCAP];  // items: lane | j << 6 | nk << 11 | (position, then bucket) << 16
  __shared__ unsigned s_abort;
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0;
  __syncthreads();
  if (s_abort) return;
  unsigned spilled = 0;
  constexpr int NB = SK2_MAX_P1 / SK2Q_THREADS;
  const unsigned p1 = 1u << p1_log2;
  const size_t seq_len = info->seq_len;
  const int lane = threadIdx.x & 63;
#define wv sk2_wave_id()  /* (computed again at every use, as in mk_sk_scatterq_k: no spill of the addresses derived from it) */
  auto sk2_wave_id = [&]() { int w = (int)threadIdx.x; asm volatile("" : "+v"(w)); return w >> 6; };
  for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
  __syncthreads();
  for (size_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    unsigned qn[SK2Q_SUBT];
#pragma unroll
    for (int st = 0; st < SK2Q_SUBT; ++st) {
      const size_t t = (tile * SK2Q_SUBT + st) * SK2Q_THREADS + threadIdx.x;
      const size_t p0 = t * SK2_R;
      Sk2Runs runs;
      runs.valid = 0;
      runs.starts = 0;
      runs.pos[0] = runs.pos[1] = runs.pos[2] = runs.pos[3] = 0;
      u64 w0 = 0, w1 = 0;
      if (p0 < seq_len) {
        w0 = codes[t];
        w1 = codes[t + 1];
        runs = sk2_analyse(w0, w1, sk2_valid32(bad, p0, k));
      }
      pk_x[st][wv][lane] = w0;
      if (lane == 63) {
        pk_x[st][wv][64] = w1;
        pk_x[st][wv][65] = p0 < seq_len ? codes[t + 2] : 0ull;
        pk_x[st][wv][66] = p0 < seq_len ? codes[t + 3] : 0ull;
      }
      const unsigned s2 = sk_cut_starts(runs.starts, runs.valid, SK2_NKMAX);
      const unsigned cnt = __popc(s2);
      const unsigned inc = mk_wave_scan_incl(cnt);  // inclusive scan over the wave (DPP: mk_device.h)
      const unsigned total = mk_wave_last(inc);
      unsigned* const myq = queue[st][wv];
      if (total <= qcap) {
        unsigned todo = s2, at = inc - cnt;
        while (todo) {
          const int j = __ffs(todo) - 1;
          todo &= todo - 1;
          const unsigned stop = (s2 | ~runs.valid) & ~((2u << j) - 1);
          const int nk = (stop ? (__ffs(stop) - 1) : SK2_R) - j;
          const u64 pw = j < 10 ? runs.pos[0] : (j < 20 ? runs.pos[1] : (j < 30 ? runs.pos[2] : runs.pos[3]));
          const unsigned best = (unsigned)(pw >> (6 * (j % 10))) & 63u;
          myq[at++] = (unsigned)lane | ((unsigned)j << 6) | ((unsigned)nk << 11) | (best << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (unsigned base = 0; base < total; base += 64) {
          const unsigned i = base + lane;
          if (i < total) {
            const unsigned it = myq[i];
            const u64* const wp = &pk_x[st][wv][it & 63u];
            const unsigned b = sk2_bucket(sk_mmer(wp[0], wp[1], (int)(it >> 16)), p1_log2);
            atomicAdd(&lh[b], 1u);
            myq[i] = (it & 0xFFFFu) | (b << 16);
          }
        }
        qn[st] = total;
      } else {
        if (p0 < seq_len) sk2_walk(runs, w0, w1, [&](int, int, unsigned mm) { atomicAdd(&lh[sk2_bucket(mm, p1_log2)], 1u); });
        qn[st] = SK2Q_WALKED;
      }
    }
    __syncthreads();
#ifndef SK_PLAIN_CURSORS
    if (p1 == SK2_MAX_P1) {
      static_assert(NB % 8 == 0, "sk_reserve8");
#pragma unroll
      for (int h = 0; h < NB; h += 8) {
        unsigned v[8], at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lh[threadIdx.x + (h + i) * SK2Q_THREADS];
        spilled |= sk_reserve8<SK2Q_THREADS>(v, cursor + h * SK2Q_THREADS, start + h * SK2Q_THREADS, SK2_NOFIT, at);
#pragma unroll
        for (int i = 0; i < 8; ++i) lh[threadIdx.x + (h + i) * SK2Q_THREADS] = at[i];
      }
    } else
#endif
    {  // (small chunks, fewer buckets: one at a time)
      for (unsigned b = threadIdx.x; b < p1; b += SK2Q_THREADS) {
        const unsigned v = lh[b];
        const u64 r = v ? (u64)atomicAdd(&cursor[b], v) : 0ull;
        const bool fits = v == 0 || r + v <= start[b + 1];
        spilled |= fits ? 0u : 1u;
        lh[b] = fits ? (unsigned)r : SK2_NOFIT;
      }
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < SK2Q_SUBT; ++st) {
      if (qn[st] != SK2Q_WALKED) {
        const unsigned total = qn[st];
        const unsigned* const myq = queue[st][wv];
        for (unsigned base = 0; base < total; base += 64) {
          const unsigned i = base + lane;
          if (i < total) {
            const unsigned it = myq[i];
            const unsigned at = atomicAdd(&lh[it >> 16], 1u);  // base + rank
            const u64* const wp = &pk_x[st][wv][it & 63u];
            const Sk2Rec rec = sk2_make_record(wp[0], wp[1], wp[2], wp[3], (int)((it >> 6) & 31u), (int)((it >> 11) & 31u), k);
            if (at < SK2_NOFIT) part[(size_t)at] = rec;
          }
        }
      } else {
        const size_t t = (tile * SK2Q_SUBT + st) * SK2Q_THREADS + threadIdx.x;
        const size_t p0 = t * SK2_R;
        if (p0 < seq_len) {
          const u64* const wp = &pk_x[st][wv][lane];
          const ulonglong2 wa = make_ulonglong2(wp[0], wp[1]);
          const u64 w2 = wp[2], w3 = wp[3];
          const Sk2Runs runs = sk2_analyse(wa.x, wa.y, sk2_valid32(bad, p0, k));
          sk2_walk(runs, wa.x, wa.y, [&](int jstart, int nk, unsigned mm) {
            const unsigned at = atomicAdd(&lh[sk2_bucket(mm, p1_log2)], 1u);
            if (at < SK2_NOFIT) part[(size_t)at] = sk2_make_record(wa.x, wa.y, w2, w3, jstart, nk, k);
          });
        }
      }
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < p1; i += blockDim.x) lh[i] = 0;
    __syncthreads();
  }
  if (spilled) atomicOr(&info->part_overflow, 4ull);
}
#undef wv

// ------------------------------------------------------------------------------------- count
// Slot hash of a two-word key: six full-rate 24-bit multiplies over its 24-bit pieces (a 32-bit multiply issues at a
// quarter of the rate; see skc_hash in mk_skmer.hip).  Bits 31.. pick the slot, bits 15..0 the sub-range.
__device__ __forceinline__ unsigned sk2c_hash(u64 hi, u64 lo) {
  const unsigned h0 = (unsigned)hi, h1 = (unsigned)(hi >> 32), l0 = (unsigned)lo, l1 = (unsigned)(lo >> 32);
  unsigned h = __umul24(h0, 0x9E3779u) ^ __umul24(__funnelshift_r(h0, h1, 24), 0x85EBCBu) ^ __umul24(h1 >> 16, 0xC2B2AFu);
  h ^= __umul24(l0, 0x27D4EBu) ^ __umul24(__funnelshift_r(l0, l1, 24), 0x165667u) ^ __umul24(l1 >> 16, 0x2C1B3Du);
  return h ^ (h >> 15);
}

// Home slot of a hash (any table size).
__device__ __forceinline__ unsigned sk2c_home(unsigned h) { return (unsigned)(((u64)h * SK2C_SLOTS) >> 32); }

__device__ __forceinline__ unsigned sk2c_lane_rank(u64 mask) {  // set bits of mask below this lane
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Insert one 128-bit key, probing from its home slot (see the protocol in the header).  The lane that wins a slot
// writes the key and publishes inside the loop iteration in which it won.  tkey holds {hi, lo} side by side: one
// 16-byte LDS access per key.
__device__ __forceinline__ void sk2c_insert(ulonglong2* tkey, unsigned* tcnt, unsigned* ovf, u64 hi, u64 lo, unsigned h) {
  unsigned slot = sk2c_home(h);
  bool done = false;
#pragma unroll 1
  for (int probe = 0; probe < SK2C_MAX_PROBE && !done;) {
    unsigned c = __hip_atomic_load(&tcnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (c == 0) {
      c = atomicCAS(&tcnt[slot], 0u, SK2C_LOCK);
      if (c == 0) {  // ours: write the key, then publish with count 1
        tkey[slot] = make_ulonglong2(hi, lo);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicAdd(&tcnt[slot], 1u - SK2C_LOCK);
        done = true;
      }
    }
    if (!done && !(c & SK2C_LOCK)) {  // (a slot that is being written is looked at again)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const ulonglong2 o = tkey[slot];
      if (o.x == hi && o.y == lo) {
        atomicAdd(&tcnt[slot], 1u);
        done = true;
      } else {
        slot = slot + 1 == SK2C_SLOTS ? 0u : slot + 1;
        ++probe;
      }
    }
  }
  if (!done) atomicOr(ovf, 1u);
}

// The top n (<= 64) deferred keys of this wave's stack, one per lane, through the general insert.
__device__ __forceinline__ void sk2c_drain(ulonglong2* tkey, unsigned* tcnt, const ulonglong2* q, unsigned& qcount, unsigned n,
                                           unsigned* ovf) {
  const unsigned lane = threadIdx.x & 63;
  qcount -= n;
  if (lane < n) {
    const ulonglong2 key = q[qcount + lane];
    sk2c_insert(tkey, tcnt, ovf, key.x, key.y, sk2c_hash(key.x, key.y));
  }
}

template <bool CANON>
__global__ __launch_bounds__(SK2C_THREADS) void mk_sk2_count_k(const Sk2Rec* __restrict__ part, const u64* __restrict__ start,
                                                               SkCursor* __restrict__ cursor,
                                                               const u64* __restrict__ kstart, u64* __restrict__ nsurv,
                                                               MkChunkInfo* __restrict__ info, u64 min_count,
                                                               u64* __restrict__ out_hi, u64* __restrict__ out_lo,
                                                               u64* __restrict__ out_cnt, int k, unsigned p1,
                                                               double dup_hint, double nk_hint) {
  __shared__ __attribute__((aligned(16))) ulonglong2 tkey[SK2C_SLOTS];  // {hi, lo}
  __shared__ unsigned tcnt[SK2C_SLOTS];                                   // 0 free, LOCK being written, else the count
  __shared__ __attribute__((aligned(16))) ulonglong2 wq[SK2C_WAVES][SK2C_QCAP];  // deferred keys, one stack per wave
  __shared__ unsigned s_distinct[2], s_overflow[2], s_emit[2];
  __shared__ unsigned long long s_windows;
  __shared__ unsigned s_abort;  // (read once per workgroup: other workgroups of this launch may set the flag meanwhile)
  if (threadIdx.x == 0) { s_abort = info->part_overflow != 0; s_windows = 0; }
  __syncthreads();
  if (s_abort) return;  // the scatter did not fit its (sampled) regions: the chunk is partitioned again
  for (unsigned i = threadIdx.x; i < SK2C_SLOTS; i += blockDim.x) tcnt[i] = 0;
  if (threadIdx.x < 2) { s_distinct[threadIdx.x] = 0; s_overflow[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; }
  __syncthreads();
  unsigned par = 0;
  const int lane = threadIdx.x & 63;
  const u64 lomask = (k >= 64) ? ~0ull : (~0ull << (128 - 2 * k));
  u64 distinct_total = 0, survivors_total = 0, nerr = 0, windows = 0, records_total = 0;
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 lo_r = start[b], n = cursor[b] - lo_r;  // (the scatter's cursor ends where the bucket's records end)
    u64* __restrict__ my_hi = out_hi + kstart[b];
    u64* __restrict__ my_lo = out_lo + kstart[b];
    u64* __restrict__ my_cnt = out_cnt + kstart[b];
    const u64 region = kstart[b + 1] - kstart[b];
    unsigned emitted = 0;
    bool counted = false;  // the bucket's windows have been added to the chunk's total
    records_total += n;
    if (n >> 27) {
      ++nerr;
    } else if (n) {
      int s0 = 0;
      {
        const double expect = (double)n * nk_hint / (dup_hint > 1.0 ? dup_hint : 1.0);
        while (s0 < SK2C_SUB_BITS && expect / (double)(1u << s0) > (double)SK2C_TARGET) ++s0;
        if ((double)n * SK2_NKMAX <= (double)SK2C_LOADCAP) s0 = 0;
        if (s0 > 3) s0 = 3;  // (see mk_skmer.hip: the estimate may be far too high for this bucket; overflows split further)
      }
      int s = s0;
      unsigned idx = 0;
      const Sk2Rec* __restrict__ src = part + lo_r;
      for (;;) {
        const unsigned sel_shift = SK2C_SUB_BITS - s;
        unsigned* const ovf = &s_overflow[par];
        u64 win_pass = 0;
        ulonglong2* const myq = wq[threadIdx.x >> 6];
        unsigned qcount = 0;  // this wave's deferred keys (wave-uniform)
        // (the whole wave walks the record loop together: lanes past the end hold an empty record)
        for (u64 jb = 0; jb < n; jb += SK2C_THREADS) {
          const u64 j = jb + threadIdx.x;
          Sk2Rec rec;
          rec.r0 = rec.r1 = rec.r2 = rec.nk = 0;
          if (j < n) rec = src[j];
          const int nk = (int)rec.nk;  // <= SK2_NKMAX
          win_pass += counted ? 0 : (u64)nk;
          u64 khi[SK2_NKMAX], klo[SK2_NKMAX];
          unsigned hh[SK2_NKMAX], st[SK2_NKMAX];
          unsigned alive = 0;
          {
            u64 x0 = rec.r0, x1 = rec.r1, x2 = rec.r2;
#pragma unroll
            for (int u = 0; u < SK2_NKMAX; ++u) {
              khi[u] = x0;
              klo[u] = x1 & lomask;
              if constexpr (CANON) sk2_canon128(khi[u], klo[u], k);
              x0 = (x0 << 2) | (x1 >> 62);
              x1 = (x1 << 2) | (x2 >> 62);
              x2 <<= 2;
              hh[u] = sk2c_hash(khi[u], klo[u]);
              const bool mine = u < nk && (!s || ((hh[u] & ((1u << SK2C_SUB_BITS) - 1)) >> sel_shift) == idx);
              alive |= mine ? (1u << u) : 0u;
            }
          }
          // batched first probe, compare-and-swap first: a free home slot (most keys of a read set at 33 <= k <= 64 are
          // new) is claimed by the very first LDS operation, the key goes in with one 16-byte write and the count is
          // published; a slot that holds a count has its key read (one 16-byte read) and compared.  Whatever does not
          // settle at its home slot (another key there, or a slot mid-write) is DEFERRED onto the wave's stack and
          // probed 64 keys at a time, every lane busy.
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u)
            st[u] = ((alive >> u) & 1u) ? atomicCAS(&tcnt[sk2c_home(hh[u])], 0u, SK2C_LOCK) : 0u;
          unsigned defer = 0;
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            if (((alive >> u) & 1u) && st[u] == 0) {
              const unsigned slot = sk2c_home(hh[u]);
              tkey[slot] = make_ulonglong2(khi[u], klo[u]);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
              atomicAdd(&tcnt[slot], 1u - SK2C_LOCK);
              alive &= ~(1u << u);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          ulonglong2 ok[SK2_NKMAX];
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            const bool look = ((alive >> u) & 1u) && !(st[u] & SK2C_LOCK);
            ok[u] = look ? tkey[sk2c_home(hh[u])] : make_ulonglong2(0ull, 0ull);
          }
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            if (!((alive >> u) & 1u)) continue;
            if (!(st[u] & SK2C_LOCK) && ok[u].x == khi[u] && ok[u].y == klo[u]) atomicAdd(&tcnt[sk2c_home(hh[u])], 1u);
            else defer |= 1u << u;
          }
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            const bool f = (defer >> u) & 1u;
            const u64 m = __ballot(f);
            if (m) {
              if (f) myq[qcount + sk2c_lane_rank(m)] = make_ulonglong2(khi[u], klo[u]);
              qcount += (unsigned)__popcll(m);
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
              if (qcount >= 64) sk2c_drain(tkey, tcnt, myq, qcount, 64u, ovf);
            }
          }
          if (__hip_atomic_load(ovf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        }
        if (qcount) sk2c_drain(tkey, tcnt, myq, qcount, qcount, ovf);  // (< 64 left)
        __syncthreads();  // A
        if (threadIdx.x == 0) cursor[b] = lo_r;  // back to the region's start: the next chunk may inherit the regions (mk_skmer.hip)
        const bool over = s_overflow[par] != 0;
        if (threadIdx.x == 0) { s_distinct[par ^ 1] = 0; s_overflow[par ^ 1] = 0; s_emit[par ^ 1] = 0; }
        {
          // the sweep reads the counts only; a slot's key only when its count reaches min_count
          constexpr int PER = SK2C_SLOTS / SK2C_THREADS;
          unsigned ec[PER];
          unsigned mine = 0, occ = 0;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const unsigned i = q * SK2C_THREADS + threadIdx.x;
            ec[q] = tcnt[i];
            tcnt[i] = 0;
            occ += ec[q] != 0;
            if (over || (u64)ec[q] < min_count) ec[q] = 0;
            mine += ec[q] != 0;
          }
          occ = mk_wave_sum(occ);
          if (lane == 0 && occ && !over) atomicAdd(&s_distinct[par], occ);
          if (mine) {
            const unsigned at = emitted + atomicAdd(&s_emit[par], mine);
            unsigned o = 0;
            if ((u64)at + mine > region) {  // only a region sized from a sampled histogram can be too small
              atomicOr(&info->part_overflow, 8ull);
              mine = 0;
            }
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              if (mine && ec[q]) {
                const ulonglong2 key = tkey[q * SK2C_THREADS + threadIdx.x];
                my_hi[at + o] = key.x;
                my_lo[at + o] = key.y;
                my_cnt[at + o] = ec[q];
                ++o;
              }
            }
          }
        }
        __syncthreads();  // B
        emitted += s_emit[par];
        distinct_total += s_distinct[par];
        par ^= 1;
        if (over) {
          if (s >= SK2C_SUB_BITS) { ++nerr; break; }
          s += 1;
          idx <<= 1;
        } else {
          windows += win_pass;  // (a pass that ran to its end has seen every record once)
          counted = true;
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
    if (threadIdx.x == 0) nsurv[b] = emitted;
    survivors_total += emitted;
  }
  {  // one global add per workgroup
    for (int d = 32; d > 0; d >>= 1) windows += __shfl_down(windows, d);
    if (lane == 0 && windows) atomicAdd(&s_windows, (unsigned long long)windows);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    // windows / survivors of THIS kernel go to the fields the one-word path uses; the caller folds them into
    // the by-reference totals once the chunk is known to be complete (they may have to be discarded)
    if (s_windows) atomicAdd(&info->windows, (u64)s_windows);
    if (records_total) atomicAdd(&info->records, records_total);
    if (distinct_total) atomicAdd(&info->distinct, distinct_total);
    if (survivors_total) atomicAdd(&info->survivors, survivors_total);
    if (nerr) atomicAdd(&info->errors, nerr);
  }
}

// ------------------------------------------------------------------- count with a counting pre-filter
// When min_count is well above the mean count of a key -- BASELINE config 5: S3 at -c 10, a chunk covers its genome
// 1.9 times, 43 M of its 57 M windows are keys seen once -- nearly every insert of the kernel above is wasted: the key
// goes into the table (lock, 16-byte write, publish), forces a second sub-range pass because 5 250 distinct keys do not
// fit 6 144 slots at 60 %, and is thrown away by the emit sweep.  Here every (sub-range) pass runs TWICE over the
// bucket's records:
//   P  every key adds 1 to ONE of 16 384 32-bit counters in LDS (index = bits of a hash of the key): a count-min row.
//      A counter is never below the count of any key that maps to it.
//   Q  the keys whose counter reached min_count -- candidates: all keys that can survive, plus a few that share a
//      counter with others -- are inserted into a small exact table (2 048 slots), every occurrence of them, so the
//      counts the emit sweep sees are exact; the rest is dropped after one LDS read.
// No key is written, nothing is locked, and the bucket needs no sub-range split (the counters have no capacity to
// overflow; the candidates' table overflows only if thousands of keys reach min_count, and then splits as above).
// `distinct` is the number of counters in use (keys that share a counter count once: a lower bound).
#define SK2P_CNT 16384
#define SK2P_SLOTS 2048
#define SK2P_MAX_PROBE 64

// 32-bit hash for the counters and the sub-range: the 128 key bits folded to 64, then three 24-bit multiplies
__device__ __forceinline__ unsigned sk2p_hash(u64 hi, u64 lo) {
  const u64 f = hi ^ ((lo >> 23) | (lo << 41));  // (no 64-bit multiply: a quarter-rate v_mul_lo_u32 each)
  const unsigned a = (unsigned)f, b = (unsigned)(f >> 32);
  unsigned h = __umul24(a, 0x9E3779u) ^ __umul24(__funnelshift_r(a, b, 24), 0x85EBCBu) ^ __umul24(b >> 16, 0xC2B2AFu);
  return h ^ (h >> 15);
}

__device__ __forceinline__ void sk2p_insert(ulonglong2* tkey, unsigned* tcnt, unsigned* ovf, u64 hi, u64 lo, unsigned h) {
  unsigned slot = (unsigned)(((u64)h * SK2P_SLOTS) >> 32);
  bool done = false;
#pragma unroll 1
  for (int probe = 0; probe < SK2P_MAX_PROBE && !done;) {
    unsigned c = __hip_atomic_load(&tcnt[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (c == 0) {
      c = atomicCAS(&tcnt[slot], 0u, SK2C_LOCK);
      if (c == 0) {
        tkey[slot] = make_ulonglong2(hi, lo);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        atomicAdd(&tcnt[slot], 1u - SK2C_LOCK);
        done = true;
      }
    }
    if (!done && !(c & SK2C_LOCK)) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const ulonglong2 o = tkey[slot];
      if (o.x == hi && o.y == lo) {
        atomicAdd(&tcnt[slot], 1u);
        done = true;
      } else {
        slot = slot + 1 == SK2P_SLOTS ? 0u : slot + 1;
        ++probe;
      }
    }
  }
  if (!done) atomicOr(ovf, 1u);
}

template <bool CANON>
__global__ __launch_bounds__(SK2C_THREADS) void mk_sk2_countp_k(const Sk2Rec* __restrict__ part, const u64* __restrict__ start,
                                                                SkCursor* __restrict__ cursor,
                                                                const u64* __restrict__ kstart, u64* __restrict__ nsurv,
                                                                MkChunkInfo* __restrict__ info, u64 min_count,
                                                                u64* __restrict__ out_hi, u64* __restrict__ out_lo,
                                                                u64* __restrict__ out_cnt, int k, unsigned p1) {
  __shared__ unsigned cnt32[SK2P_CNT];
  __shared__ __attribute__((aligned(16))) ulonglong2 tkey[SK2P_SLOTS];
  __shared__ unsigned tcnt[SK2P_SLOTS];
  __shared__ __attribute__((aligned(16))) ulonglong2 cq[SK2C_WAVES][SK2C_QCAP];  // candidates, one stack per wave
  __shared__ unsigned s_distinct[2], s_overflow[2], s_emit[2];
  __shared__ unsigned long long s_windows;
  __shared__ unsigned s_abort;
  if (threadIdx.x == 0) { s_abort = info->part_overflow != 0; s_windows = 0; }
  __syncthreads();
  if (s_abort) return;
  for (unsigned i = threadIdx.x; i < SK2P_CNT; i += blockDim.x) cnt32[i] = 0;
  for (unsigned i = threadIdx.x; i < SK2P_SLOTS; i += blockDim.x) tcnt[i] = 0;
  if (threadIdx.x < 2) { s_distinct[threadIdx.x] = 0; s_overflow[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; }
  __syncthreads();
  unsigned par = 0;
  const int lane = threadIdx.x & 63;
  const u64 lomask = (k >= 64) ? ~0ull : (~0ull << (128 - 2 * k));
  const unsigned need = min_count > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)min_count;
  u64 distinct_total = 0, survivors_total = 0, nerr = 0, windows = 0, records_total = 0;
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 lo_r = start[b], n = cursor[b] - lo_r;
    u64* __restrict__ my_hi = out_hi + kstart[b];
    u64* __restrict__ my_lo = out_lo + kstart[b];
    u64* __restrict__ my_cnt = out_cnt + kstart[b];
    const u64 region = kstart[b + 1] - kstart[b];
    unsigned emitted = 0;
    bool counted = false;
    records_total += n;
    if (n >> 27) {
      ++nerr;
    } else if (n) {
      int s = 0;
      unsigned idx = 0;
      const Sk2Rec* __restrict__ src = part + lo_r;
      for (;;) {
        const unsigned sel_shift = 32 - s;  // the sub-range is picked by the TOP bits of the hash, the counter by the low ones
        unsigned* const ovf = &s_overflow[par];
        u64 win_pass = 0;
        // ---- P: count every key of the sub-range into its counter
        for (u64 jb = 0; jb < n; jb += SK2C_THREADS) {
          const u64 j = jb + threadIdx.x;
          Sk2Rec rec;
          rec.r0 = rec.r1 = rec.r2 = rec.nk = 0;
          if (j < n) rec = src[j];
          const int nk = (int)rec.nk;
          win_pass += counted ? 0 : (u64)nk;
          u64 x0 = rec.r0, x1 = rec.r1, x2 = rec.r2;
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            u64 khi = x0, klo = x1 & lomask;
            if constexpr (CANON) sk2_canon128(khi, klo, k);
            x0 = (x0 << 2) | (x1 >> 62);
            x1 = (x1 << 2) | (x2 >> 62);
            x2 <<= 2;
            const unsigned h = sk2p_hash(khi, klo);
            if (u < nk && (!s || (h >> sel_shift) == idx)) atomicAdd(&cnt32[h & (SK2P_CNT - 1)], 1u);
          }
        }
        __syncthreads();  // P done: the counters are final
        // ---- Q: the candidates into the exact table -- via the wave's stack (positions from ballots), 64 at a time with
        //      every lane busy: inserted one by one in the lane that found them, their LDS round trips run one after the other
        ulonglong2* const myq = cq[threadIdx.x >> 6];
        unsigned qcount = 0;
        for (u64 jb = 0; jb < n; jb += SK2C_THREADS) {
          const u64 j = jb + threadIdx.x;
          Sk2Rec rec;
          rec.r0 = rec.r1 = rec.r2 = rec.nk = 0;
          if (j < n) rec = src[j];
          const int nk = (int)rec.nk;
          u64 x0 = rec.r0, x1 = rec.r1, x2 = rec.r2;
          unsigned cv[SK2_NKMAX];
          u64 khi[SK2_NKMAX], klo[SK2_NKMAX];
          unsigned cand = 0;
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) {
            khi[u] = x0;
            klo[u] = x1 & lomask;
            if constexpr (CANON) sk2_canon128(khi[u], klo[u], k);
            x0 = (x0 << 2) | (x1 >> 62);
            x1 = (x1 << 2) | (x2 >> 62);
            x2 <<= 2;
            const unsigned h = sk2p_hash(khi[u], klo[u]);
            const bool mine = u < nk && (!s || (h >> sel_shift) == idx);
            cv[u] = mine ? cnt32[h & (SK2P_CNT - 1)] : 0u;
          }
#pragma unroll
          for (int u = 0; u < SK2_NKMAX; ++u) cand |= (cv[u] >= need && cv[u]) ? (1u << u) : 0u;
          if (__any(cand != 0)) {
#pragma unroll
            for (int u = 0; u < SK2_NKMAX; ++u) {
              const bool f = (cand >> u) & 1u;
              const u64 m = __ballot(f);
              if (m) {
                if (f) myq[qcount + sk2c_lane_rank(m)] = make_ulonglong2(khi[u], klo[u]);
                qcount += (unsigned)__popcll(m);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (qcount >= 64) {
                  qcount -= 64;
                  const ulonglong2 key = myq[qcount + lane];
                  sk2p_insert(tkey, tcnt, ovf, key.x, key.y, sk2c_hash(key.x, key.y));
                }
              }
            }
          }
        }
        if (qcount) {  // (< 64 left)
          if ((unsigned)lane < qcount) {
            const ulonglong2 key = myq[lane];
            sk2p_insert(tkey, tcnt, ovf, key.x, key.y, sk2c_hash(key.x, key.y));
          }
          qcount = 0;
        }
        __syncthreads();  // A: every insert of the pass is in the table
        if (threadIdx.x == 0) cursor[b] = lo_r;  // back to the region's start: the next chunk may inherit the regions
        const bool over = s_overflow[par] != 0;
        if (threadIdx.x == 0) { s_distinct[par ^ 1] = 0; s_overflow[par ^ 1] = 0; s_emit[par ^ 1] = 0; }
        {
          // counters: how many are in use (the distinct keys, less those that share one), and clear
          unsigned occ = 0;
#pragma unroll
          for (int q = 0; q < SK2P_CNT / SK2C_THREADS; q += 4) {
            const unsigned i = (q * SK2C_THREADS + 4 * threadIdx.x);
            const uint4 c4 = *reinterpret_cast<const uint4*>(&cnt32[i]);
            occ += (c4.x != 0) + (c4.y != 0) + (c4.z != 0) + (c4.w != 0);
            *reinterpret_cast<uint4*>(&cnt32[i]) = make_uint4(0u, 0u, 0u, 0u);
          }
          occ = mk_wave_sum(occ);
          if (lane == 0 && occ && !over) atomicAdd(&s_distinct[par], occ);
          // exact table: emit what reached min_count, clear
          constexpr int PER = SK2P_SLOTS / SK2C_THREADS;
          unsigned ec[PER];
          unsigned mine = 0;
#pragma unroll
          for (int q = 0; q < PER; ++q) {
            const unsigned i = q * SK2C_THREADS + threadIdx.x;
            ec[q] = tcnt[i];
            tcnt[i] = 0;
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
                const ulonglong2 key = tkey[q * SK2C_THREADS + threadIdx.x];
                my_hi[at + o] = key.x;
                my_lo[at + o] = key.y;
                my_cnt[at + o] = ec[q];
                ++o;
              }
            }
          }
        }
        __syncthreads();  // B
        emitted += s_emit[par];
        distinct_total += s_distinct[par];
        par ^= 1;
        if (over) {  // thousands of candidates: split the hash range and take the halves one after the other
          if (s >= 16) { ++nerr; break; }
          s += 1;
          idx <<= 1;
        } else {
          windows += win_pass;
          counted = true;
          while (s > 0 && (idx & 1u)) { idx >>= 1; --s; }
          if (s == 0) break;
          ++idx;
        }
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
}

// ------------------------------------------------------------------------------------ launcher
int mk_launch_count_superkmer2(mk_ctx* c, size_t seq_len, uint64_t min_count, bool exact) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const int k = c->k;
  int p1_log2 = 8;
  while (p1_log2 < SK2_MAX_P1_LOG2 && (seq_len >> p1_log2) > 8192) ++p1_log2;
  if (const char* e = getenv("MK_P1_LOG2")) { int v = atoi(e); if (v >= 4 && v <= SK2_MAX_P1_LOG2) p1_log2 = v; }
  c->p1_log2 = p1_log2;
  const size_t p1 = (size_t)1 << p1_log2;
  // bucket sizes from a 1-in-8 sample of the analysis threads for big chunks (see mk_skmer.hip)
  int sample_log2 = 0;
  float sigmas = 6.0f;
  {
    int want = 3;
    size_t min_len = (size_t)8 << 20;
    if (const char* e = getenv("MK_SAMPLE_LOG2")) { int v = atoi(e); if (v >= 0 && v <= 6) want = v; }
    if (const char* e = getenv("MK_SAMPLE_MIN")) min_len = (size_t)atoll(e);
    if (const char* e = getenv("MK_SAMPLE_SIGMAS")) sigmas = (float)atof(e);
    if (!exact && seq_len >= min_len) sample_log2 = want;
  }
  c->part_sampled = sample_log2 != 0;
  const bool reuse = mk_part_inherit(c, seq_len, p1_log2, min_count, sample_log2 != 0, exact);
  int rc;
  if ((rc = mk_buf_reserve(c, c->part_meta, (7 * p1 + 16) * sizeof(u64))) != MK_OK) return rc;
  const size_t part_cap = seq_len + 64;
  if ((rc = mk_buf_reserve(c, c->part, part_cap * sizeof(Sk2Rec))) != MK_OK) return rc;
  const u64 surv_div = min_count > 1 ? (u64)min_count : 1;  // <= ceil(m / min_count) survivors among m k-mers
  size_t surv_cap = seq_len / surv_div + p1 + 64;
  if (sample_log2) {  // room for the sampling error of every bucket (bound as in mk_skmer.hip)
    const double L = 1.25 * (double)seq_len, S = (double)(1u << sample_log2), w = (double)SK2_R;
    surv_cap = (size_t)((L + 6.0 * sqrt((double)p1 * S * w * L) + 16.0 * w * (double)p1) / (double)surv_div) + 2 * p1 + 64;
  }
  if ((rc = mk_buf_reserve(c, c->surv_keys, surv_cap * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_keys2, surv_cap * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->surv_cnts, surv_cap * sizeof(u64))) != MK_OK) return rc;
  u64* hist = (u64*)c->part_meta.p;
  u64* start = hist + p1;
  SkCursor* cursor = (SkCursor*)(start + p1 + 1);  // (packed 32-bit, in the space of p1 64-bit words)
  u64* khist = start + p1 + 1 + p1;
  u64* kstart = khist + p1;
  u64* kcursor = kstart + p1 + 1;
  u64* nsurv = kcursor + p1;
  if (!reuse) MK_HIP(hipMemsetAsync(hist, 0, (7 * p1 + 8) * sizeof(u64), c->stream));
  const size_t threads = div_up(seq_len, SK2_R);
  const size_t tiles = div_up(div_up(threads, (size_t)1 << sample_log2), SK2_HIST_THREADS);
  const size_t stiles = div_up(threads, SK2_SCAT_THREADS);
  const size_t hist_grid = sample_log2 ? 128 : 256;
  (void)kcursor;
  mk_prof_begin(c, MK_K_PART);
  const dim3 hgrid((unsigned)(tiles < hist_grid ? tiles : hist_grid)), sgrid((unsigned)(stiles < 4096 ? stiles : 4096));
  if (reuse) {
    // the regions of the chunk before stand as they are, every cursor back at its start
  } else if (c->canonical)
    hipLaunchKernelGGL(mk_sk2_hist_k<true>, hgrid, dim3(SK2_HIST_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, khist, p1_log2, k, threads, sample_log2);
  else
    hipLaunchKernelGGL(mk_sk2_hist_k<false>, hgrid, dim3(SK2_HIST_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, hist, khist, p1_log2, k, threads, sample_log2);
  if (!reuse)
    mk_launch_sk_scan(c, hist, khist, start, cursor, kstart, p1_log2, sample_log2, SK2_NKMAX, surv_div, (u64)part_cap,
                      (u64)surv_cap, sigmas, 1);
  if (c->canonical)
    hipLaunchKernelGGL(mk_sk2_scatter_k<true>, sgrid, dim3(SK2_SCAT_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, (const u64*)start, cursor, (Sk2Rec*)c->part.p, p1_log2, k, stiles);
  else if (getenv("MK_SCATTER_WALK") || part_cap >= SK2_NOFIT)
    hipLaunchKernelGGL(mk_sk2_scatter_k<false>, sgrid, dim3(SK2_SCAT_THREADS), 0, c->stream, (const u64*)c->codes.p,
                       (const u64*)c->bad.p, info, (const u64*)start, cursor, (Sk2Rec*)c->part.p, p1_log2, k, stiles);
  else {
    const int force_subt = getenv("MK_SKQ_SUBT") ? atoi(getenv("MK_SKQ_SUBT")) : 0;
    const bool three = force_subt == 3 || (force_subt != 2 && c->items_hint > 0 && c->items_hint * 64.0 + 48.0 < 376.0);
    unsigned qcap = three ? 376u : 512u;
    if (const char* e = getenv("MK_SKQ_CAP")) { const int v = atoi(e); if (v >= 0 && (unsigned)v < qcap) qcap = (unsigned)v; }
    const size_t qtiles = div_up(threads, (size_t)SK2Q_THREADS * (three ? 3 : 2));
    const dim3 qgrid((unsigned)(qtiles < 8192 ? qtiles : 8192));
    if (three)
      hipLaunchKernelGGL((mk_sk2_scatterq_k<3, 376>), qgrid, dim3(SK2Q_THREADS), 0, c->stream, (const u64*)c->codes.p,
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (Sk2Rec*)c->part.p, p1_log2, k, qtiles, qcap);
    else
      hipLaunchKernelGGL((mk_sk2_scatterq_k<2, 512>), qgrid, dim3(SK2Q_THREADS), 0, c->stream, (const u64*)c->codes.p,
                         (const u64*)c->bad.p, info, (const u64*)start, cursor, (Sk2Rec*)c->part.p, p1_log2, k, qtiles, qcap);
  }
  mk_prof_end(c);
  mk_prof_begin(c, MK_K_COUNT);
  {
    int ncu = 256;
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, c->device);
    const unsigned grid = (unsigned)((size_t)ncu < p1 ? (size_t)ncu : p1);
    // the counting pre-filter pays when few keys can reach min_count: min_count well above the mean count of a key,
    // which the chunk before has measured (windows / distinct keys); a sample's first chunk takes the exact kernel
    static const bool no_pre = getenv("MK_NO_PREFILTER") != nullptr;
    static const bool force_pre = getenv("MK_FORCE_PREFILTER") != nullptr;
    const bool pre = !no_pre && min_count >= 2 && (force_pre || (min_count >= 4 && c->dup_known && c->dup_hint * 2.5 < (double)min_count));
    if (pre && c->canonical)
      hipLaunchKernelGGL(mk_sk2_countp_k<true>, dim3(grid), dim3(SK2C_THREADS), 0, c->stream, (const Sk2Rec*)c->part.p,
                         (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count, (u64*)c->surv_keys.p,
                         (u64*)c->surv_keys2.p, (u64*)c->surv_cnts.p, k, (unsigned)p1);
    else if (pre)
      hipLaunchKernelGGL(mk_sk2_countp_k<false>, dim3(grid), dim3(SK2C_THREADS), 0, c->stream, (const Sk2Rec*)c->part.p,
                         (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count, (u64*)c->surv_keys.p,
                         (u64*)c->surv_keys2.p, (u64*)c->surv_cnts.p, k, (unsigned)p1);
    else if (c->canonical)
      hipLaunchKernelGGL(mk_sk2_count_k<true>, dim3(grid), dim3(SK2C_THREADS), 0, c->stream, (const Sk2Rec*)c->part.p,
                         (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count, (u64*)c->surv_keys.p,
                         (u64*)c->surv_keys2.p, (u64*)c->surv_cnts.p, k, (unsigned)p1, c->dup_hint, c->nk_hint);
    else
      hipLaunchKernelGGL(mk_sk2_count_k<false>, dim3(grid), dim3(SK2C_THREADS), 0, c->stream, (const Sk2Rec*)c->part.p,
                         (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count, (u64*)c->surv_keys.p,
                         (u64*)c->surv_keys2.p, (u64*)c->surv_cnts.p, k, (unsigned)p1, c->dup_hint, c->nk_hint);
  }
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  c->surv_regions = 2;  // {hi, lo, count} triples per bucket
  return MK_OK;
}
