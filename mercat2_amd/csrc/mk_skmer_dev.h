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

// Bucket cursors are PACKED 32-BIT record indices (a chunk on these paths has fewer than 2^32 - 2^24 records: the
// callers check): the L2's atomic rate goes by the lines a wave's 64 lanes touch, and 13 M reservations per S2 chunk --
// 1 610 tiles x 8 192 buckets -- are a third of the scatter (tools/cursor_atomic_probe.hip: 64-bit cursors 88-92 us,
// 32-bit at an 8-byte stride 79-83, 32-bit packed 44-46, private to an XCD 88, without return 84).
typedef unsigned SkCursor;

// A tile's reservations, eight per thread: thread t of THREADS reserves count[i] records in bucket t + THREADS i
// (i = 0..7; `cursor` and `start` point at the first of these 8 x THREADS buckets, all of which exist) and learns whether
// the run fits the bucket's region.  The 8 cursor adds (lanes without records
// masked off) and the 8 region ends are in flight together, one wait.  Written out: under the scatter kernels' register
// pressure the compiler spilled the 8 addresses and put a full wait in front of every atomic and every load (in-kernel
// stamps, mk_sk_scatterq_k: 34 K of a tile's 104 K cycles went into this block).  Base pointers are uniform (scalar
// registers), one 32-bit offset per thread.  base[i] = the run's first record, or `nofit` (0xFF000000) + the free records
// the run found at the region's end (fewer than its own, so below 2^24); returns 1 when a run does not fit.
// (s_nop 4 in front of every access: the base pointers may have just come out of a v_readlane -- the compiler parks scalar
// registers in vector lanes here -- and a vector-memory instruction must not read a scalar register within 5 cycles of a
// vector instruction writing it; inside inline assembly the compiler does not insert those wait states itself.)
#ifndef SK_PLAIN_CURSORS
template <int THREADS>
__device__ __forceinline__ unsigned sk_reserve8(const unsigned (&count)[8], SkCursor* cursor, const u64* start, unsigned nofit,
                                                unsigned (&base)[8]) {
  unsigned r[8];
  u64 lim[8];
  const unsigned voff4 = threadIdx.x * 4u, voff8 = threadIdx.x * 8u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    u64 saved;  // exec while the lanes without records sit the add out
    r[i] = 0;
    asm volatile(
        "v_cmp_ne_u32 vcc, 0, %3\n\t"
        "s_and_saveexec_b64 %1, vcc\n\t"
        "s_nop 4\n\t"
        "global_atomic_add %0, %2, %3, %4 sc0\n\t"
        "s_mov_b64 exec, %1"
        : "+v"(r[i]), "=&s"(saved)
        : "v"(voff4), "v"(count[i]), "s"(cursor + i * THREADS)
        : "memory", "vcc", "scc");  // s_and_saveexec writes SCC
  }
#pragma unroll
  for (int i = 0; i < 8; ++i)
    asm volatile("s_nop 4\n\tglobal_load_dwordx2 %0, %1, %2 offset:8" : "=&v"(lim[i]) : "v"(voff8), "s"(start + i * THREADS) : "memory");
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(lim[0]),
                 "+v"(lim[1]), "+v"(lim[2]), "+v"(lim[3]), "+v"(lim[4]), "+v"(lim[5]), "+v"(lim[6]), "+v"(lim[7])
               :
               : "memory");
  unsigned spilled = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const bool fits = count[i] == 0 || (u64)r[i] + count[i] <= lim[i];
    spilled |= fits ? 0u : 1u;
    // (a run that does not fit: `nofit` plus, in its low 24 bits, the records still free at the region's end when it came --
    // nothing else will ever be stored there, every later run of the bucket starts past the end: see the callers)
    base[i] = fits ? r[i] : nofit | ((u64)r[i] < lim[i] ? (unsigned)(lim[i] - r[i]) : 0u);
  }
  return spilled;
}
#endif

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
// network of minima with static indices).
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
      // (the hash's low 22 bits on top, bits 6..9 zero.  The register constraint keeps the shift out of the multiply:
      // folded, the multiplier no longer fits 24 bits and the multiply-add becomes a quarter-rate v_mul_lo_u32 + add)
      unsigned h = sk_order_raw(sk_canon_mmer(mm, canon));
#ifndef SK_MIN_DOUBLING
      asm("" : "+v"(h));
#endif
      ord[q] = (h << 10) | (unsigned)q;
    }
  }
#ifdef SK_MIN_DOUBLING  // (A/B: the round-2 form -- two-input minima, widths 1, 2, 4, 8, 16 -- and the folded multiply)
  constexpr int WB = (W >= 16) ? 16 : (W >= 8) ? 8 : (W >= 4) ? 4 : (W >= 2) ? 2 : 1;
#pragma unroll
  for (int step = 1; step < WB; step <<= 1) {
#pragma unroll
    for (int q = 0; q + step < NQ; ++q) ord[q] = min(ord[q], ord[q + step]);
  }
#else
  // Sliding minimum over W candidates with three-input minima: the width covered triples (offsets 0, w, 2w) while
  // 3w <= W; the last step, in the loop below, closes to exactly W (W <= 22: two rounds at most).
  constexpr int WA = W >= 3 ? 3 : 1, WB = W >= 9 ? 9 : WA;
  if constexpr (W >= 3) {
#pragma unroll
    for (int q = 0; q + 2 < NQ; ++q) ord[q] = min(min(ord[q], ord[q + 1]), ord[q + 2]);
  }
  if constexpr (W >= 9) {
#pragma unroll
    for (int q = 0; q + 8 < NQ; ++q) ord[q] = min(min(ord[q], ord[q + 3]), ord[q + 6]);
  }
  static_assert(W <= 3 * WB, "a third round of minima would be needed");
#endif
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
    unsigned best;  // ord[q] covers candidates q .. q + WB - 1
    if constexpr (W == WB) best = ord[j];
    else if constexpr (W <= 2 * WB) best = min(ord[j], ord[j + W - WB]);
    else best = min(min(ord[j], ord[j + WB]), ord[j + W - WB]);
    best &= 63u;
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

// ---- regions of a bucket, as the count kernels see them (mk_skcount.hip, mk_skmer2.hip)
// A bucket's records lie in nseg regions (1, or 9: one per XCD and a shared one, see mk_sk_scatterq_k / mk_sk_scan_k) --
// region x of bucket b is [start[x * p1 + b], cursor[x * p1 + b]) -- and the kernel numbers them through, region after
// region.  The table of a bucket: words 0..8 the number of records before region x, word 9 their total, words 10..18 region
// x's first record minus the records before it (so that record j of the bucket is part[tab[10 + x] + j], x = the regions
// with tab[x] <= j, less one).  Lanes 0..15 of the workgroup's last wave build the table of the NEXT bucket while the
// current one is counted: they ask for its bounds at the top of the bucket and write the table just before barrier A, by
// when the answers have long arrived; nothing else in the kernel waits for a bucket's bounds.
#define SKC_SEG_WORDS 20
#define SKC_SEG_MAX 9
__device__ __forceinline__ unsigned skc_seg_at(const unsigned* tab, unsigned j, int nseg) {
  if (nseg == 1) return tab[10] + j;
  const uint4 a = *reinterpret_cast<const uint4*>(tab), b = *reinterpret_cast<const uint4*>(tab + 4);
  const unsigned c = tab[8];
  // (compare + add-with-carry, two instructions per region; the compiler made a compare, a select, a shift and an add of it)
  unsigned x = 0;
#define SKC_SEG_STEP(P) asm("v_cmp_le_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x) : "v"(P), "v"(j) : "vcc")
  SKC_SEG_STEP(a.y); SKC_SEG_STEP(a.z); SKC_SEG_STEP(a.w); SKC_SEG_STEP(b.x); SKC_SEG_STEP(b.y); SKC_SEG_STEP(b.z); SKC_SEG_STEP(b.w);
  SKC_SEG_STEP(c);
#undef SKC_SEG_STEP
  return tab[10 + x] + j;
}
// (lanes 0..15 of one wave, all sixteen active; seg_lo / seg_hi: the bounds of region x, or of region 0 for x >= nseg)
// A cursor past its region's end (runs that did not fit and went to the shared region, mk_sk_scatterq_k) counts to the end.
__device__ __forceinline__ void skc_seg_publish(unsigned* tab, unsigned seg_lo, unsigned seg_hi, unsigned seg_end, int nseg) {
  asm volatile("" : "+v"(seg_lo), "+v"(seg_hi), "+v"(seg_end));  // (nothing computed from the bounds before this point: they are waited for HERE)
  const unsigned x = threadIdx.x & 15u, cnt = x < (unsigned)nseg ? (seg_hi < seg_end ? seg_hi : seg_end) - seg_lo : 0u;
  unsigned inc = cnt;  // inclusive scan over the sixteen lanes: one DPP row (mk_wave_scan_incl's first four steps)
  inc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)inc, 0x111, 0xf, 0xf, false);
  inc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)inc, 0x112, 0xf, 0xf, false);
  inc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)inc, 0x114, 0xf, 0xf, false);
  inc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)inc, 0x118, 0xf, 0xf, false);
  if (x < SKC_SEG_MAX) {
    tab[x] = inc - cnt;
    tab[10 + x] = seg_lo - (inc - cnt);
  }
  if (x == 15) tab[9] = inc;
}

// in-kernel phase stamps of -DMK_STAMP builds (wave 0 of a workgroup; s_memtime ticks)
#ifdef MK_STAMP
#define STAMP(var) { __builtin_amdgcn_sched_barrier(0); unsigned long long t__; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__) :: "memory"); __builtin_amdgcn_sched_barrier(0); var = t__; }
#define STAMP_ADD(acc, t0) { unsigned long long t1__; STAMP(t1__); acc += t1__ - t0; t0 = t1__; }
#else
#define STAMP(var)
#define STAMP_ADD(acc, t0)
#endif

