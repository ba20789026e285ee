// mk_skmer_dev.h -- device code shared by the super-k-mer partitioners (mk_skmer.hip: one-word keys, 18 <= k <= 32;
// mk_skmer2.hip: two-word keys, 33 <= k <= 64): the minimizer order, the analysis of a thread's 32 windows into runs
// that share their minimizer, and the walk over those runs.
#pragma once
#include "mk_common.h"
#include "mk_device.h"

#define SK_M 11                 // minimizer length
#define SK_MASK ((1u << (2 * SK_M)) - 1)
#define SK_R 32                 // windows per thread

// Ordering hash: a BIJECTION on the 22-bit 11-mers (odd multiplies and xor-shifts modulo 2^22), so
// "smallest hash" is a strict total order on 11-mer values: two positions tie only when they hold
// the same 11-mer, and then either choice names the same minimizer. (The canonical mode relies on
// this: a window and its reverse complement see the candidates in opposite order.)
__device__ __forceinline__ unsigned sk_order_raw(unsigned mm) {
  // One 24-bit multiply-add (full rate on CDNA; a 32-bit v_mul_lo_u32 issues at a quarter of it): the low 22 bits
  // of mm * odd + c are a bijection of the 11-mer, and an order compares from the top bit down, where every bit of
  // mm has been multiplied in -- no xor-shift needed (measured: the same number of records per chunk).
  return __umul24(mm, 0x9277B5u) + 0x2C5A3Du;
}
__device__ __forceinline__ unsigned sk_order_hash(unsigned mm) { return sk_order_raw(mm) & SK_MASK; }
// 11-mer under which a window is filed: itself, or min(itself, reverse complement) in canonical mode.
__device__ __forceinline__ unsigned sk_canon_mmer(unsigned mm, bool canon) {
  if (!canon) return mm;
  const unsigned rc = (unsigned)mk_revcomp2((u64)mm, SK_M);
  return rc < mm ? rc : mm;
}
__device__ __forceinline__ unsigned sk_bucket(unsigned mm, int p1_log2) { return (mm * 0xC2B2AE3Du) >> (32 - p1_log2); }

// 11-mer starting at base q of the 64-base pair (w0,w1).
__device__ __forceinline__ unsigned sk_mmer(u64 w0, u64 w1, int q) {
  // (selects and one funnel, no branches: q differs from lane to lane; q + SK_M <= 64, so from q = 32 on the 11-mer
  // lies inside w1)
  const u64 a = q < 32 ? w0 : w1, b = q < 32 ? w1 : 0ull;
  const int s = 2 * (q & 31);
  const u64 x = (a << s) | ((b >> 1) >> (63 - s));
  return (unsigned)(x >> (64 - 2 * SK_M));
}

// Result of analysing a thread's 32 windows: which are valid, where runs start, and the 6-bit
// minimizer position of every window (packed 10 per word).
// Windows j = 0..31 whose k bases are all clean, from the 64 bad bits that start at the thread's
// first base (k <= 33: j + k - 1 <= 63).
__device__ __forceinline__ unsigned sk_valid32(u64 badw, int k) {
  if (badw == 0) return ~0u;
  // window j is clean iff bits j .. j+k-1 are: smear every bad bit down over the k-1 positions below it
  u64 d = badw;
  int cover = 1;
  while (2 * cover <= k) { d |= d >> cover; cover *= 2; }
  if (cover < k) d |= d >> (k - cover);
  return ~(unsigned)d;
}

struct SkRuns {
  unsigned valid, starts;
  u64 pos[4];
};

// W = k - SK_M + 1 minimizer candidates per window (compile time: the sliding minimum is a
// doubling network with static indices).
template <int W>
__device__ __forceinline__ SkRuns sk_analyse(u64 w0, u64 w1, unsigned valid, bool canon) {
  constexpr int NQ = SK_R + W - 1;  // candidate positions 0 .. NQ-1
  unsigned ord[NQ];
  {
    unsigned mm = sk_mmer(w0, w1, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      if (q) {
        const int pos = q + SK_M - 1;  // new base index
        const unsigned base = (unsigned)((pos < 32 ? (w0 >> (62 - 2 * pos)) : (w1 >> (62 - 2 * (pos - 32)))) & 3u);
        mm = ((mm << 2) | base) & SK_MASK;
      }
      // (canonical mode: one bit reversal per candidate.  Rolling the reverse complement along -- the base that enters
      // on the right enters it, complemented, on the left -- was measured slower, 361-371 against 335 us per S2 chunk:
      // it chains the 52 candidates of a thread one behind the other, the reversals are independent of each other)
      ord[q] = (sk_order_raw(sk_canon_mmer(mm, canon)) << 10) | (unsigned)q;  // (the hash's low 22 bits on top, bits 6..9 zero)
    }
  }
  constexpr int P = (W >= 16) ? 16 : (W >= 8) ? 8 : (W >= 4) ? 4 : (W >= 2) ? 2 : 1;
#pragma unroll
  for (int step = 1; step < P; step <<= 1) {
#pragma unroll
    for (int q = 0; q + step < NQ; ++q) ord[q] = min(ord[q], ord[q + step]);
  }
  // static, branch-free: minimizer positions and run starts (first valid window, minimizer
  // moved, or previous window invalid)
  SkRuns r;
  r.valid = valid;
  r.starts = 0;
  r.pos[0] = r.pos[1] = r.pos[2] = r.pos[3] = 0;
  unsigned prev_pos = 64;  // impossible position: window 0 always starts a run
#pragma unroll
  for (int j = 0; j < SK_R; ++j) {
    const bool ok = (valid >> j) & 1u;
    const unsigned best = min(ord[j], ord[j + W - P]) & 63u;
    r.starts |= (ok && best != prev_pos) ? (1u << j) : 0u;
    prev_pos = ok ? best : 64u;
    r.pos[j / 10] |= (u64)best << (6 * (j % 10));
  }
  return r;
}

// One iteration per run: a run ends at the next start, the next invalid window or the end of the
// thread's span; runs longer than nkmax windows are cut (same minimizer, same bucket).
template <class F>
__device__ __forceinline__ void sk_walk(const SkRuns& r, u64 w0, u64 w1, int nkmax, bool canon, F&& emit) {
  unsigned todo = r.starts;
  while (todo) {
    const int j = __ffs(todo) - 1;
    todo &= todo - 1;
    const unsigned stop = (r.starts | ~r.valid) & ~((2u << j) - 1);  // bits above j that end the run
    int nk = (stop ? (__ffs(stop) - 1) : SK_R) - j;
    const u64 pw = j < 10 ? r.pos[0] : (j < 20 ? r.pos[1] : (j < 30 ? r.pos[2] : r.pos[3]));
    const unsigned best = (unsigned)(pw >> (6 * (j % 10))) & 63u;
    const unsigned mm = sk_canon_mmer(sk_mmer(w0, w1, (int)best), canon);
    int at = j;
    while (nk > 0) {
      const int take = nk < nkmax ? nk : nkmax;
      emit(at, take, mm);
      at += take;
      nk -= take;
    }
  }
}

// Run starts with the cuts put in: a run longer than nkmax windows is cut every nkmax windows (what sk_walk does on
// the fly), so that every set bit is one record.
__device__ __forceinline__ unsigned sk_cut_starts(unsigned starts, unsigned valid, int nkmax) {
  if (nkmax >= SK_R) return starts;
  const unsigned cont = valid & ~starts;  // windows that continue their run
  unsigned a = cont;                      // a[p] = cont[p - nkmax + 1 .. p] all set
  int len = 1;
  while (2 * len <= nkmax) { a &= a << len; len *= 2; }
  if (len < nkmax) a &= a << (nkmax - len);
  unsigned s2 = starts;
  for (unsigned c = (starts << nkmax) & a; c; c = (c << nkmax) & a) s2 |= c;
  return s2;
}

