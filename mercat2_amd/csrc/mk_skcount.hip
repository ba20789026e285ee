// mk_skcount.hip -- the count stage of the super-k-mer path for one-word keys (nucleotide 12 <= k <= 32): the kernels
// that expand a bucket's 16-byte records into k-mers and count them in an LDS table (lib/mercat2_kmers.py:56-60: every
// window +1), keep the entries with count >= min_count (lib/mercat2_kmers.py:73-76, per chunk) and hand them on to the
// running table (the dict sum of bin/mercat2.py:121-127).  Split from mk_skmer.hip (the partition: histogram, scan,
// scatter -- 168 kernel instances that take minutes to compile) so that this file builds in seconds.
#include "mk_skmer_dev.h"
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <type_traits>

#ifdef SKC_SMALL  // (mk_skcount_small.hip: this file once more with 512 threads / 4096 slots, under names of its own)
#define mk_sk_count_k mk_sk_count_small_k
#define mk_sk_countp_k mk_sk_countp_small_k
#define mk_launch_sk_count mk_launch_sk_count_small
#define mk_dbg_ptr mk_dbg_ptr_small
#define SKC_FCAP_A 256   // list entries per sweep of the fused upsert: the short and the long list
#define SKC_FCAP_B 512
#else
#define SKC_FCAP_A 512
#define SKC_FCAP_B 1024
#endif

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
// The next bucket's first records, asked for before the sweep of this one, are taken in (waited for) BEFORE the fused
// upsert's claims go out after the sweep: the loop head then has no load to wait for -- a wait there is vmcnt(0), which
// also waits for the claims just issued, a trip to the running table in HBM per bucket (7 % of the kernel by the stamps).
#ifdef SKC_NO_TAKE_IN
#define SKC_TAKE_IN(pre)
#else
#define SKC_TAKE_IN(pre)                                                                       \
  do {                                                                                         \
    _Pragma("unroll") for (int h_ = 0; h_ < SKC_PRE; ++h_) asm volatile("" ::"v"((pre)[h_].x), "v"((pre)[h_].y)); \
  } while (0)
#endif
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
#ifdef SKC_ABL_NODRAIN  // (timing ablation only: the deferred keys are pushed and then dropped)
  return;
#endif
  if (lane < n) {
    const u64 key = q[qcount + lane];
    unsigned slot = skc_step(skc_home(skc_hash(key)), 1);
    bool placed = false;
#pragma unroll 1
    for (int probe = 0; probe < SKC_MAX_PROBE; ++probe) {
#ifdef SKC_DRAIN_READ_FIRST  // (A/B: look before claiming -- two trips through the LDS for a free slot)
      u64 cur = tkey[slot];
      if (cur == MK_EMPTY) {
        cur = atomicCAS(&tkey[slot], MK_EMPTY, key);
        if (cur == MK_EMPTY) cur = key;
      }
#else
      u64 cur = atomicCAS(&tkey[slot], MK_EMPTY, key);  // claim-or-compare in one trip, as in the insert round
      if (cur == MK_EMPTY) cur = key;
#endif
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


// ---- survivors straight into the running table (round 4: FCAP > 0) --------------------------------------------------
// Up to round 3 a bucket's survivors went to a region of a buffer, the host waited for their number, grew the running
// table and launched an import kernel over the regions (32 us per S2 chunk alone, 106 us beside the other context's
// kernels; canonical keys: 144 us per chunk).  Now the count kernel upserts them itself -- and never waits for HBM doing
// so: the sweep of a bucket only LISTS its survivors {key, count, slot} in LDS; after the sweep's last barrier thread i
// issues the compare-and-swap that claims list entry i's slot in the running table and goes on to the next bucket's
// inserts; at the next sweep -- some 20 000 clocks later, the answer has long arrived -- it adds the count (an atomic add
// without return) when the slot was free or held the key, and otherwise lists the entry again for the next slot.  What
// is still in flight when the workgroup runs out of buckets is finished in a loop.  The host sizes the table BEFORE the
// launch from the previous chunk's survivors; an entry that has probed SKF_MAX_PROBE slots (a table filling up: the
// estimate was far off) is written to the spill list instead (the old survivor buffer) and imported by the host afterwards.
#define SKF_MAX_PROBE 96  // (default of the kernels' max_probe argument; MK_FUSE_MAX_PROBE lowers it: tests of the spill path)
__device__ __forceinline__ void skf_spill(MkChunkInfo* info, u64* sp_keys, u64* sp_cnts, u64 key, unsigned cnt) {
  const u64 at = atomicAdd(&info->spilled, 1ull);
  sp_keys[at] = key;
  sp_cnts[at] = (u64)cnt;
}
// `old` = what the compare-and-swap at `slot` returned; probes on (with waits) until the key has a slot
__device__ __forceinline__ void skf_finish(MkSlot* __restrict__ run, u64 mask, u64 key, unsigned cnt, u64 slot, u64 old, unsigned& fresh,
                                           MkChunkInfo* info, u64* sp_keys, u64* sp_cnts, unsigned max_probe) {
  for (unsigned probe = 0;; ++probe) {
    if (old == MK_EMPTY || old == key) {
      atomicAdd(&run[slot].cnt, (u64)cnt);
      fresh += old == MK_EMPTY ? 1 : 0;
      return;
    }
    if (probe >= max_probe) { skf_spill(info, sp_keys, sp_cnts, key, cnt); return; }
    slot = (slot + 1) & mask;
    old = atomicCAS(&run[slot].key, MK_EMPTY, key);
  }
}

// (the region table of a bucket -- skc_seg_at, skc_seg_publish -- is in mk_skmer_dev.h: the two-word kernels use it too)
// Persistent: gridDim.x workgroups (one per CU) walk the buckets b = blockIdx.x, +gridDim.x, ...
// The next bucket's bounds and its first two record batches are loaded while the current
// bucket is being emitted, so no global-memory latency sits on the critical path.
// K32: k == 32, the only k whose keys can equal the free-slot mark (32 x 'T'): that key is counted aside.
// FCAP: 0 = survivors into the bucket's region of out_keys / out_cnts (the host imports them); 512 or 1024 = list
// entries per sweep of the fused upsert into `run` (run_mask + 1 slots, at most 2^32), out_keys / out_cnts = spill list.
template <bool CANON, bool K32, int FCAP>
__global__ __launch_bounds__(SKC_LB) void mk_sk_count_k(const ulonglong2* __restrict__ part, const u64* __restrict__ start,
                                                             SkCursor* __restrict__ cursor,
                                                             const u64* __restrict__ kstart, u64* __restrict__ nsurv,
                                                             MkChunkInfo* __restrict__ info, u64 min_count,
                                                             u64* __restrict__ out_keys, u64* __restrict__ out_cnts,
                                                             int k, unsigned p1, double dup_hint, double nk_hint, u64* __restrict__ dbg,
                                                             int dflags, MkSlot* __restrict__ run, u64 run_mask, unsigned max_probe,
                                                             int nseg) {
  constexpr bool FUSED = FCAP > 0;
  // (the longer list takes its LDS from the deferred-key stacks: two slots pushed at a time instead of four, measured 1 % slower)
  constexpr int PUSH = FCAP > SKC_FCAP_A ? 2 : SKC_PUSH;
  constexpr int QCAP = 64 + 64 * PUSH;
#ifdef SKC_UNCOND
  __shared__ __attribute__((aligned(16))) u64 tkey[SKC_SLOTS + 64];  // (+ a word per lane for the swaps of lanes without a key)
#else
  __shared__ __attribute__((aligned(16))) u64 tkey[SKC_SLOTS];
#endif
  __shared__ __attribute__((aligned(16))) unsigned tcnt[SKC_SLOTS];
  __shared__ __attribute__((aligned(16))) u64 wq[SKC_WAVES][QCAP];  // deferred keys, one stack per wave
  // survivors on their way into the running table: two lists (one being resolved, one being filled)
  __shared__ __attribute__((aligned(16))) u64 pend_key[FUSED ? 2 : 1][FUSED ? FCAP : 1];
  __shared__ unsigned pend_cnt[FUSED ? 2 : 1][FUSED ? FCAP : 1], pend_slot[FUSED ? 2 : 1][FUSED ? FCAP : 1];
  __shared__ unsigned s_npend[2];
  __shared__ unsigned long long s_fresh;
  // per-pass flags, double-buffered by pass parity so that resetting them needs no extra barrier
  __shared__ unsigned s_distinct[2], s_overflow[2], s_emit[2];
  __shared__ unsigned long long s_windows;
  __shared__ unsigned long long s_tot_distinct, s_tot_survivors;  // (kept by thread 0: chunk totals cost no registers)
  __shared__ __attribute__((aligned(16))) unsigned s_seg[2][SKC_SEG_WORDS];  // the region tables of this bucket and the next
  __shared__ unsigned s_abort;  // (read once per workgroup: other workgroups of this launch may set the flag meanwhile)
  // (fused: this launch is speculative -- the host looks at the parser's verdict only after it -- and what it merges into
  // the running table cannot be taken back: a chunk the host will refuse or parse again must not get that far)
  if (threadIdx.x == 0) s_abort = info->part_overflow != 0 || (FUSED && (info->parse_fallback | info->non_ascii) != 0);
  __syncthreads();
  if (s_abort) return;  // the scatter did not fit its (sampled) regions: the chunk is partitioned again
  for (unsigned i = threadIdx.x; i < SKC_SLOTS; i += blockDim.x) { tkey[i] = MK_EMPTY; tcnt[i] = 0; }
  if (threadIdx.x < 2) { s_distinct[threadIdx.x] = 0; s_overflow[threadIdx.x] = 0; s_emit[threadIdx.x] = 0; s_npend[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { s_windows = 0; s_fresh = 0; s_tot_distinct = 0; s_tot_survivors = 0; }
  __syncthreads();
  // fused upsert: list f_cur holds the f_n entries whose compare-and-swaps are in flight (entry i: thread i, answer in f_old)
  unsigned f_cur = 0, f_n = 0;
  u64 f_old = 0;
  unsigned fresh = 0;  // (per lane; a chunk has fewer than 2^32 windows)
  const unsigned run_mask32 = (unsigned)run_mask;
  unsigned par = 0;
  const int kshift = 64 - 2 * k;
  const int lane = threadIdx.x & 63;
  u64* const myq = wq[threadIdx.x >> 6];
  u64 side = 0, nerr = 0;
  unsigned windows = 0;    // (per lane)
  u64 records_total = 0;   // what the chunk held (every record is expanded at least once)
  u64 tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, tF = 0, t0 = 0, npass = 0;
  STAMP(t0);

  // prefetched state of the bucket about to be processed
  unsigned bn = blockIdx.x;
  u64 ks_n = 0, ke_n = 0;  // survivor region [ks_n, ke_n)
  unsigned n_n = 0;        // its records
  unsigned bp = 0;         // s_seg[bp]: the table of the current bucket
  const bool seg_lane = threadIdx.x >= SKC_THREADS - 64 && threadIdx.x < SKC_THREADS - 48;  // the sixteen lanes that build the tables
  // (index of this lane's region in start / cursor: recomputed at every use -- the empty asm keeps the compiler from
  // hoisting 64-bit addresses out of the bucket loop into registers the kernel does not have)
  auto seg_index = [&](unsigned bucket) {
    unsigned x = threadIdx.x & 15u;
    asm volatile("" : "+v"(x));
    return (x < (unsigned)nseg ? x * p1 : 0u) + bucket;
  };
  unsigned seg_lo = 0, seg_hi = 0, seg_end = 0;  // (seg_lane) this lane's region of bucket bn -- first record, cursor, end -- on their way
  ulonglong2 pre[SKC_PRE];
#pragma unroll
  for (int h = 0; h < SKC_PRE; ++h) pre[h] = make_ulonglong2(0, 0);
  if (bn < p1) {
    ks_n = kstart[bn];
    ke_n = kstart[bn + 1];
    if (seg_lane) {
      seg_lo = (unsigned)start[seg_index(bn)];
      seg_end = (unsigned)start[seg_index(bn) + 1];
      seg_hi = cursor[seg_index(bn)];
      skc_seg_publish(s_seg[0], seg_lo, seg_hi, seg_end, nseg);
    }
  }
  __syncthreads();
  if (bn < p1) {
    n_n = s_seg[0][9];
#pragma unroll
    for (int h = 0; h < SKC_PRE; ++h) {
      const unsigned j = (unsigned)SKC_JMAP(h);
      if (j < n_n) pre[h] = skc_ldrec(part + skc_seg_at(s_seg[0], j, nseg));
    }
    SKC_TAKE_IN(pre);
  }
  for (unsigned b = blockIdx.x; b < p1; b += gridDim.x) {
    const u64 n = n_n;  // records of this bucket
    const unsigned* const seg = s_seg[bp];
    u64* __restrict__ my_keys = out_keys + ks_n;
    u64* __restrict__ my_cnts = out_cnts + ks_n;
    const u64 region = ke_n - ks_n;
    // bounds of the next bucket: in flight while this one is counted
    bn = b + gridDim.x;
    if (bn < p1) {
      ks_n = kstart[bn];
      ke_n = kstart[bn + 1];
      if (seg_lane) {  // (asked for here, looked at before barrier A)
        seg_lo = (unsigned)start[seg_index(bn)];
        seg_end = (unsigned)start[seg_index(bn) + 1];
        seg_hi = cursor[seg_index(bn)];  // (lanes past nseg read region 0's bounds and count nothing: no select here, which would wait for seg_lo)
      }
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
      u64 side_pass = 0;
      unsigned win_pass = 0;
      bool side_done = false;
      bool first_pass = true;
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
            for (int h = 0; h < SKC_PRE; ++h) recs2[h] = pre[h];  // (asked for during the bucket before)
          } else {
#pragma unroll
            for (int h = 0; h < SKC_PRE; ++h) {
              const u64 j = rb2 + SKC_JMAP(h);
              recs2[h] = j < n ? skc_ldrec(part + skc_seg_at(seg, (unsigned)j, nseg)) : make_ulonglong2(0, 0);
            }
            SKC_TAKE_IN(recs2);  // (waited for on this path, so that the join with the path above has nothing to wait for)
          }
          STAMP_ADD(tF, t0);
#pragma unroll
          for (int h = 0; h < SKC_PRE; ++h) {
            // ---- one record per thread, expanded 8 k-mers at a time: the 8 compare-and-swaps on the home
            //      slots are issued together, then the adds of the keys that found (or claimed) their slot;
            //      the others are deferred
            const ulonglong2 rec = recs2[h];
            const int nk = (int)(rec.y & 63);
            win_pass += side_done ? 0u : (unsigned)nk;
            u64 x = rec.x, y = rec.y;
            // (the whole wave walks the loop together -- lanes without a record or with a short one just have
            // no live slots -- because the deferred-key stack below is the wave's: every lane takes part)
            // One round = up to NB consecutive k-mers of every lane's record.  NB is a compile-time constant of the
            // body; with SKC_DYN the wave picks the body that fits its LONGEST record (4, 6 or 8 slots: scalar
            // branch, no per-slot tests), which pays when the records a wave holds are about equally long.
#ifdef SKC_OLD_ROUND
            auto round = [&](auto nb_tag, auto, int base) {  // (A/B: the round as it was up to round 3)
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
              for (int half = 0; half < NB; half += PUSH) {
#pragma unroll
                for (int u = half; u < half + PUSH && u < NB; ++u) {
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
#else
            auto round = [&](auto nb_tag, auto sub_tag, int base) {
              constexpr int NB = decltype(nb_tag)::value;
              constexpr bool SUB = decltype(sub_tag)::value;  // a sub-range pass (s > 0): keys are selected by hash bits
              // Straight-line, no branch between the first compare-and-swap and the last add (round 4, from the ISA and
              // the SQ counters: the kernel issued 0.76 scalar instructions per vector instruction -- exec-mask
              // bookkeeping of nested ifs -- and the compiler put a full lgkmcnt(0) in front of the FIRST answer it
              // looked at): which slots are live is kept as wave masks (bools), a slot's add is one masked LDS
              // instruction, the slots that failed are those masks again, and the sub-range selection (three vector +
              // two scalar instructions per key) is compiled in only for the sub-range passes.
              u64 kk[NB], cur[NB];
              unsigned sl[NB];
              bool on[NB];
              const int rem = nk - base;  // slots of this round that hold a k-mer: u < rem
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
                const unsigned hv = skc_hash(kk[u]);
                bool o = u < rem;
                if (K32 && o && kk[u] == MK_EMPTY) {
                  side_pass += side_done ? 0 : 1;
                  o = false;
                }
                if (SUB && ((hv & ((1u << SKC_SUB_BITS) - 1)) >> sel_shift) != idx) o = false;
                on[u] = o;
                sl[u] = skc_home(hv);
#ifdef SKC_UNCOND  // (A/B: no exec masking at all -- lanes without a key swap on a word of their own behind the table)
                cur[u] = atomicCAS(&tkey[o ? sl[u] : SKC_SLOTS + (unsigned)lane], MK_EMPTY, kk[u]);
#else
                if (o) cur[u] = atomicCAS(&tkey[sl[u]], MK_EMPTY, kk[u]);
#endif
#ifdef SKC_SCHED_FENCE
                __builtin_amdgcn_sched_barrier(0);
#endif
              }
#pragma unroll
              for (int u = 0; u < NB; ++u) {
                const bool ok = on[u] && (cur[u] == MK_EMPTY || cur[u] == kk[u]);
#ifdef SKC_UNCOND
                atomicAdd(&tcnt[sl[u]], ok ? 1u : 0u);
#else
#ifndef SKC_ABL_NOADD  // (timing ablation only, counts are wrong: 306 -> 274 us per S2 chunk)
                if (ok) atomicAdd(&tcnt[sl[u]], 1u);
#endif
#endif
#ifdef SKC_ABL_NODEFER  // (timing ablation only, keys that miss their home slot are dropped: 306 -> 212 us -- the stack
                on[u] = false;   // and the probing loop are a third of the kernel; trying the next slot inside the round as well
#else                    // -- a second batch of eight compare-and-swaps and adds -- measured 337 us: the round got dearer by more)
                on[u] = on[u] && !ok;  // from here on: the key did not settle at its home slot
#endif
              }
              // deferred keys -> the wave's stack (positions from ballots: no atomic), four slots at a time
              // so that the stack never holds more than SKC_QCAP; full groups of 64 are probed right away
#pragma unroll
              for (int half = 0; half < NB; half += PUSH) {
#pragma unroll
                for (int u = half; u < half + PUSH && u < NB; ++u) {
                  const bool f = on[u];
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
#endif
            auto round_any = [&](auto nb_tag, int base) {
              if (s) round(nb_tag, std::true_type{}, base);
              else round(nb_tag, std::false_type{}, base);
            };
#ifdef SKC_DYN
            int wmax = nk;  // the wave's longest record (wave-uniform)
            for (int d = 32; d > 0; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d));
            wmax = __builtin_amdgcn_readfirstlane(wmax);
            for (int base = 0; base < wmax;) {
              const int left = wmax - base;
              if (left <= 4) { round_any(std::integral_constant<int, 4>{}, base); base += 4; }
              else if (left <= 6) { round_any(std::integral_constant<int, 6>{}, base); base += 6; }
              else { round_any(std::integral_constant<int, SKC_B>{}, base); base += SKC_B; }
            }
#else
            for (int base = 0; __any(base < nk); base += SKC_B) round_any(std::integral_constant<int, SKC_B>{}, base);
#endif
          }
          STAMP_ADD(tC, t0);
          // hint only; decided after the barrier below (an LDS read: a volatile access through the generic pointer was a
          // flat load at system scope with a vmcnt(0) wait behind it)
          if (__hip_atomic_load(&s_overflow[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) over = true;
        }
        STAMP_ADD(tA, t0);
        if (qcount) skc_drain(tkey, tcnt, myq, qcount, qcount, ovf);  // (< 64 left)
        if (first_pass && seg_lane && bn < p1) skc_seg_publish(s_seg[bp ^ 1], seg_lo, seg_hi, seg_end, nseg);  // the next bucket's table
        first_pass = false;
        STAMP_ADD(tB, t0);
        __syncthreads();  // A: every insert of the pass is in the table
        // (every wave has long read this bucket's bounds: put its cursors back to the regions' starts, so that the next
        // chunk can inherit the regions without a histogram and a scan -- see the launcher)
        if (threadIdx.x < (unsigned)nseg) cursor[seg_index(b)] = seg[10 + threadIdx.x] + seg[threadIdx.x];
        if (bn < p1) n_n = __builtin_amdgcn_readfirstlane(s_seg[bp ^ 1][9]);
        STAMP_ADD(tF, t0);
        ++npass;
        over = s_overflow[par] != 0;
        if (threadIdx.x == 0) { s_distinct[par ^ 1] = 0; s_overflow[par ^ 1] = 0; s_emit[par ^ 1] = 0; }  // next pass's set
        if (FUSED && threadIdx.x < f_n) {
          // the claim issued after the sweep before this one has long been answered: add the count, or move on a slot
          const u64 key = pend_key[f_cur][threadIdx.x];
          const unsigned cnt = pend_cnt[f_cur][threadIdx.x], slot = pend_slot[f_cur][threadIdx.x];
          if (f_old == MK_EMPTY || f_old == key) {
            atomicAdd(&run[slot].cnt, (u64)cnt);
            fresh += f_old == MK_EMPTY ? 1 : 0;
          } else {
            const unsigned nslot = (slot + 1) & run_mask32;
            if (((nslot - (unsigned)mk_mix64(key)) & run_mask32) > max_probe) {
              skf_spill(info, out_keys, out_cnts, key, cnt);
            } else {
              const unsigned pos = atomicAdd(&s_npend[f_cur ^ 1], 1u);
              if (pos < (unsigned)FCAP) {
                pend_key[f_cur ^ 1][pos] = key;
                pend_cnt[f_cur ^ 1][pos] = cnt;
                pend_slot[f_cur ^ 1][pos] = nslot;
              } else {  // (the list is full: this one waits for its answers)
                skf_finish(run, run_mask, key, cnt, nslot, atomicCAS(&run[nslot].key, MK_EMPTY, key), fresh, info, out_keys, out_cnts, max_probe);
              }
            }
          }
        }
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
            const unsigned j = (unsigned)SKC_JMAP(h);
            pre[h] = j < n_n ? skc_ldrec(part + skc_seg_at(s_seg[bp ^ 1], j, nseg)) : make_ulonglong2(0, 0);
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
          unsigned occ = 0;   // occupied slots of the wave (wave-uniform: counted with ballots, not summed over the lanes
                              // by six dependent shuffles -- the sweep is short and nothing hides their round trips)
#pragma unroll
          for (int q = 0; q < PER; q += 2) {
            const unsigned i = (q * SKC_THREADS + 2 * threadIdx.x);
            const uint2 cp = *reinterpret_cast<const uint2*>(&tcnt[i]);
            ec[q] = cp.x;
            ec[q + 1] = cp.y;
            occ += (unsigned)__popcll(__ballot(cp.x != 0)) + (unsigned)__popcll(__ballot(cp.y != 0));
            keep |= (!over && cp.x && (u64)cp.x >= min_count) ? (1u << q) : 0u;
            keep |= (!over && cp.y && (u64)cp.y >= min_count) ? (2u << q) : 0u;
          }
          if (lane == 0 && occ && !over) atomicAdd(&s_distinct[par], occ);
          // (places in the list: one LDS add per WAVE -- its lanes' counts scanned with DPP -- instead of two adds to the
          // same two words from every thread that holds a survivor, which the LDS works off one after the other)
          unsigned pos0 = 0;
          if (FUSED) {
            const unsigned mine = (unsigned)__popc(keep);
            const unsigned inc = mk_wave_scan_incl(mine), total = mk_wave_last(inc);
            if (total) {  // (wave-uniform)
              unsigned base = 0;
              if (lane == 0) {
                atomicAdd(&s_emit[par], total);  // (the chunk's survivor count: a statistic here)
                base = atomicAdd(&s_npend[f_cur ^ 1], total);
              }
              pos0 = (unsigned)__builtin_amdgcn_readfirstlane((int)base) + inc - mine;
            }
          }
          if (FUSED && keep) {
            unsigned o = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              if ((keep >> q) & 1u) {
                const u64 key = tkey[(q & ~1) * SKC_THREADS + 2 * threadIdx.x + (q & 1)];
                const unsigned home = (unsigned)mk_mix64(key) & run_mask32;
                if (pos0 + o < (unsigned)FCAP) {
                  pend_key[f_cur ^ 1][pos0 + o] = key;
                  pend_cnt[f_cur ^ 1][pos0 + o] = ec[q];
                  pend_slot[f_cur ^ 1][pos0 + o] = home;
                } else {  // (more survivors than the list takes: rare by the host's choice of FCAP; exact, with waits)
                  skf_finish(run, run_mask, key, ec[q], home, atomicCAS(&run[home].key, MK_EMPTY, key), fresh, info, out_keys, out_cnts, max_probe);
                }
                ++o;
              }
            }
          }
          if (!FUSED && keep) {
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
        if (threadIdx.x == 0) s_tot_distinct += s_distinct[par];
        par ^= 1;
        if (FUSED) {
          // the list just filled is complete: thread i claims entry i's slot and does NOT wait (the answer is looked at
          // in the next sweep); the list just resolved is free again
          const unsigned filled = s_npend[f_cur ^ 1];
          if (threadIdx.x == 0) s_npend[f_cur] = 0;
          f_cur ^= 1;
          f_n = filled < (unsigned)FCAP ? filled : (unsigned)FCAP;
          SKC_TAKE_IN(pre);  // (every pass: a wait that depends on `last` would leave the loop head its own)
          if (threadIdx.x < f_n) f_old = atomicCAS(&run[pend_slot[f_cur][threadIdx.x]].key, MK_EMPTY, pend_key[f_cur][threadIdx.x]);
        }
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
      // no pass, so no table and nothing prefetched for the next bucket: do it here (the whole workgroup comes this way)
      if (seg_lane && bn < p1) skc_seg_publish(s_seg[bp ^ 1], seg_lo, seg_hi, seg_end, nseg);
      __syncthreads();
      if (n && threadIdx.x < (unsigned)nseg) cursor[seg_index(b)] = seg[10 + threadIdx.x] + seg[threadIdx.x];
      if (bn < p1) {
        n_n = __builtin_amdgcn_readfirstlane(s_seg[bp ^ 1][9]);
#pragma unroll
        for (int h = 0; h < SKC_PRE; ++h) {
          const unsigned j = (unsigned)SKC_JMAP(h);
          pre[h] = j < n_n ? skc_ldrec(part + skc_seg_at(s_seg[bp ^ 1], j, nseg)) : make_ulonglong2(0, 0);
        }
        SKC_TAKE_IN(pre);
      }
    }
    bp ^= 1;
    if (threadIdx.x == 0) nsurv[b] = emitted;
    if (threadIdx.x == 0) s_tot_survivors += emitted;
    STAMP_ADD(tE, t0);
  }
  if (FUSED) {  // what is still in flight: finished here, with waits
    if (threadIdx.x < f_n)
      skf_finish(run, run_mask, pend_key[f_cur][threadIdx.x], pend_cnt[f_cur][threadIdx.x], (u64)pend_slot[f_cur][threadIdx.x], f_old, fresh,
                 info, out_keys, out_cnts, max_probe);
    u64 fr = fresh;
    for (int d = 32; d > 0; d >>= 1) fr += __shfl_down(fr, d);
    if (lane == 0 && fr) atomicAdd(&s_fresh, (unsigned long long)fr);
  }
  {  // one global add per workgroup (adds to one address are serialised by the L2: ~4 ns each)
    u64 w = windows;
    for (int d = 32; d > 0; d >>= 1) w += __shfl_down(w, d);
    if (lane == 0 && w) atomicAdd(&s_windows, (unsigned long long)w);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (FUSED && s_fresh) atomicAdd(&info->new_rows, (u64)s_fresh);
    if (s_windows) atomicAdd(&info->windows, (u64)s_windows);
    if (records_total) atomicAdd(&info->records, records_total);
    if (s_tot_distinct) atomicAdd(&info->distinct, (u64)s_tot_distinct);
    if (s_tot_survivors) atomicAdd(&info->survivors, (u64)s_tot_survivors);
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
          occ = mk_wave_sum(occ);
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

// ------------------------------------------------------------------------------ launcher
#ifdef MK_STAMP
u64* mk_dbg_ptr = nullptr;
#endif

// The count kernel over the p1 bucket regions the scatter has filled (start / cursor: records, kstart: survivor regions,
// nsurv: survivors per bucket, written here).  Called by mk_launch_count_superkmer (mk_skmer.hip).
int mk_launch_sk_count(mk_ctx* c, const u64* start, SkCursor* cursor, const u64* kstart, u64* nsurv, uint64_t min_count, int nkmax,
                       size_t p1, bool exact, int nseg) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const int k = c->k;
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
    if (pre) c->fused_last = false;
    if (pre && nseg != 1) { c->err = "mk_launch_sk_count: the pre-filter kernel reads one region per bucket"; return MK_ERR_ARG; }
#define SKP_LAUNCH(CANON, K32)                                                                                          \
  hipLaunchKernelGGL((mk_sk_countp_k<CANON, K32>), dim3(grid), dim3(SKC_THREADS), 0, c->stream, (const ulonglong2*)c->part.p, \
                     (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count,                        \
                     (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, k, (unsigned)p1)
    if (pre) {
      if (c->canonical) { if (k == 32) SKP_LAUNCH(true, true); else SKP_LAUNCH(true, false); }
      else { if (k == 32) SKP_LAUNCH(false, true); else SKP_LAUNCH(false, false); }
    } else {
#define SKC_LAUNCH(CANON, K32, FCAP)                                                                                    \
  hipLaunchKernelGGL((mk_sk_count_k<CANON, K32, FCAP>), dim3(grid), dim3(SKC_THREADS), 0, c->stream, (const ulonglong2*)c->part.p, \
                     (const u64*)start, cursor, (const u64*)kstart, nsurv, info, (u64)min_count,                        \
                     (u64*)c->surv_keys.p, (u64*)c->surv_cnts.p, k, (unsigned)p1, c->dup_hint, c->nk_hint, dbgbuf, dflags, \
                     (MkSlot*)tab->run.p, (u64)(tab->run_slots ? tab->run_slots - 1 : 0), max_probe, nseg)
#define SKC_LAUNCH2(CANON, K32)                                                                                         \
  do {                                                                                                                  \
    if (fcap > 512) SKC_LAUNCH(CANON, K32, SKC_FCAP_B);                                                                 \
    else if (fcap > 0) SKC_LAUNCH(CANON, K32, SKC_FCAP_A);                                                              \
    else SKC_LAUNCH(CANON, K32, 0);                                                                                     \
  } while (0)
    // (fused: the survivors go straight into the running table, which process_chunk_fast has sized for them)
    const unsigned max_probe = getenv("MK_FUSE_MAX_PROBE") ? (unsigned)atoi(getenv("MK_FUSE_MAX_PROBE")) : (unsigned)SKF_MAX_PROBE;
    mk_ctx* tab = (c->fuse_cap > 0 && c->fuse_target) ? c->fuse_target : c;  // (mk_share_table: the caller holds that table's lock)
    const int fcap = (c->fuse_cap > 0 && tab->run_slots >= 1024 && tab->run_slots <= ((size_t)1 << 32)) ? c->fuse_cap : 0;
    c->fused_last = fcap > 0;
    if (c->canonical) { if (k == 32) SKC_LAUNCH2(true, true); else SKC_LAUNCH2(true, false); }
    else { if (k == 32) SKC_LAUNCH2(false, true); else SKC_LAUNCH2(false, false); }
#undef SKC_LAUNCH2
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
  return MK_OK;
}
