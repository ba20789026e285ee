// mk_count.hip -- the counting kernels: every length-k window of every record, +1 in its bin.
//
// Replaces the reference's inner loop (lib/mercat2_kmers.py:56-60, 65-69)
//     for i in range(len(seq)-k+1): kmerlist[seq[i:i+k]] += 1
// with three GPU forms that together count *every* window exactly:
//   dense   k*BITS <= 15: direct-index histogram, private to the workgroup in LDS
//           (replicated per lane group when the bin count is small), one global add per
//           non-zero bin per workgroup;
//   hash64  k*BITS <= 64: packed key, open-addressed table of 16-byte {key,count} slots in
//           HBM, linear probing, claim by atomicCAS, count by no-return atomicAdd;
//   byref   any k, any character: the table stores the *position* of the first occurrence
//           (plus a hash tag); equality is a k-byte compare in the parsed stream. It takes
//           (a) every window when there is no packed form for (alphabet,k) and (b) in the
//           packed modes exactly the windows that hold a character outside the alphabet, so
//           the union is the reference's "any character" semantics.
// A window is counted iff it lies inside one record (no separator) -- lib/mercat2_kmers.py:52-61.
#include "mk_common.h"
#include "mk_device.h"

__device__ __forceinline__ void add_windows(MkChunkInfo* info, unsigned mine, bool exotic) {
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(exotic ? &info->exotic : &info->windows, (u64)mine);
}

// ------------------------------------------------------------------------------ hash64
__device__ __forceinline__ void insert64(MkSlot* __restrict__ table, u64 mask, u64 key, u64 add) {
  u64 slot = mk_mix64(key) & mask;
  for (;;) {
    u64 cur = table[slot].key;
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&table[slot].key, MK_EMPTY, key);
      if (cur == MK_EMPTY) cur = key;
    }
    if (cur == key) {
      atomicAdd(&table[slot].cnt, add);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(256) void mk_count_hash64_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                         MkChunkInfo* __restrict__ info, MkSlot* __restrict__ table,
                                                         u64 mask, int k, int canon) {
  constexpr int R = SPW * WPT;
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t p0 = t * R;
  unsigned mine = 0;
  if (p0 < info->seq_len) {
    u64 w[WPT + 1];
#pragma unroll
    for (int i = 0; i <= WPT; ++i) w[i] = codes[t * WPT + i];
    const u64 badw = bad_window(bad, p0);
    const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
    if ((badw & ((R + k - 1 >= 64) ? ~0ull : ((1ull << (R + k - 1)) - 1))) == 0) {
      // fast path: no bad symbol anywhere in this thread's span
#pragma unroll
      for (int i = 0; i < WPT; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
          u64 key = mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
          if (key == MK_EMPTY) atomicAdd(&info->side, 1ull); else insert64(table, mask, key, 1);
        }
      }
      mine = R;
    } else {
#pragma unroll
      for (int i = 0; i < WPT; ++i) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
          if (((badw >> (i * SPW + s)) & kmask) == 0) {
            u64 key = mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
            if (key == MK_EMPTY) atomicAdd(&info->side, 1ull); else insert64(table, mask, key, 1);
            ++mine;
          }
        }
      }
    }
  }
  add_windows(info, mine, false);
}

// ------------------------------------------------------------------------------- dense
// LDS histogram hist[bin * copies + (lane & (copies-1))]: with few bins the copies put the
// lanes of a wave on different banks/addresses; flushed with one atomicAdd per non-zero bin.
template <int BITS, int SPW, int WPT>
__global__ __launch_bounds__(256) void mk_count_dense_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                        MkChunkInfo* __restrict__ info, u64* __restrict__ bins,
                                                        unsigned nbins, unsigned copies_log2, int k, size_t nthreads_total,
                                                        int canon) {
  extern __shared__ unsigned hist[];
  constexpr int R = SPW * WPT;
  const unsigned copies = 1u << copies_log2;
  const unsigned my_copy = threadIdx.x & (copies - 1);
  for (unsigned i = threadIdx.x; i < nbins * copies; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const size_t seq_len = info->seq_len;
  const u64 kmask = (1ull << k) - 1;
  unsigned mine = 0;
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
          unsigned key = (unsigned)mk_canon2(window_key<BITS, SPW>(w[i], w[i + 1], s, k), k, BITS == 2 && canon);
          atomicAdd(&hist[(key << copies_log2) | my_copy], 1u);
          ++mine;
        }
      }
    }
  }
  __syncthreads();
  for (unsigned b = threadIdx.x; b < nbins; b += blockDim.x) {
    u64 sum = 0;
    for (unsigned cpy = 0; cpy < copies; ++cpy) sum += hist[(b << copies_log2) | cpy];
    if (sum) atomicAdd(&bins[b], sum);
  }
  add_windows(info, mine, false);
}

// ------------------------------------------------------------------------------- byref
// Slot key = (tag << 40) | position; tag = 23 hash bits (bit 63 stays 0, so MK_EMPTY is free).
#define REF_POS_BITS 40
#define REF_POS_MASK ((1ull << REF_POS_BITS) - 1)

__device__ __forceinline__ bool same_bytes(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int k) {
  for (int i = 0; i < k; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

__device__ __forceinline__ void insert_ref(MkSlot* __restrict__ table, u64 mask, const uint8_t* __restrict__ seq,
                                           u64 pos, u64 h, int k, u64 add) {
  const u64 tag = (h >> 41) << REF_POS_BITS;  // 23 bits
  const u64 mine = tag | pos;
  u64 slot = h & mask;
  for (;;) {
    u64 cur = table[slot].key;
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&table[slot].key, MK_EMPTY, mine);
      if (cur == MK_EMPTY) {
        atomicAdd(&table[slot].cnt, add);
        return;
      }
    }
    if ((cur & ~REF_POS_MASK) == tag && same_bytes(seq + (cur & REF_POS_MASK), seq + pos, k)) {
      atomicAdd(&table[slot].cnt, add);
      return;
    }
    slot = (slot + 1) & mask;
  }
}

__device__ __forceinline__ bool in_alphabet(int alphabet, unsigned c) {
  if (alphabet == MK_ALPHABET_NT2) return c == 'A' || c == 'C' || c == 'G' || c == 'T';
  return c >= 'A' && c <= 'Z';
}

// Each thread owns RB consecutive window starts. Rolling polynomial hash over the last k bytes.
// EXOTIC: count only windows with >= 1 character outside `alphabet` (the packed kernel has the rest).
#define RB 32
template <bool EXOTIC>
__global__ __launch_bounds__(256) void mk_count_byref_k(const uint8_t* __restrict__ seq, const u64* __restrict__ bad,
                                                        MkChunkInfo* __restrict__ info, MkSlot* __restrict__ table,
                                                        u64 mask, int k, int alphabet, u64 bpow /* B^k */) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t p0 = t * RB;
  const size_t n = info->seq_len;
  unsigned mine = 0;
  bool work = p0 < n && p0 + k <= n;
  if (EXOTIC && work) {
    // skip quickly when no bad symbol lies in [p0, p0+RB+k-1): check the bitmap word by word
    const size_t lo = p0, hi = p0 + RB + k - 1;  // exclusive
    bool any = false;
    for (size_t b = lo >> 6; b <= ((hi - 1) >> 6) && !any; ++b) {
      u64 wv = bad[b];
      size_t base = b << 6;
      if (base < lo) wv &= ~0ull << (lo - base);
      if (base + 64 > hi) wv &= ~0ull >> (base + 64 - hi);
      any = wv != 0;
    }
    work = any;
  }
  if (work) {
    u64 h = 0;
    long long since_sep = 0;  // consecutive non-separator bytes ending at the current byte
    long long since_bad = 1ll << 40;  // bytes since the last out-of-alphabet character (0 = current)
    const size_t last = (p0 + RB - 1 + k <= n) ? p0 + RB - 1 + k : n;  // exclusive end of bytes to read
    // Runs of one character (the N gaps of an assembly: megabases) make every window the same k-mer, and
    // adds to ONE slot are serialised by the L2: consecutive windows that lie inside such a run are
    // counted together -- one insert with their number -- instead of one by one.
    long long same = 0;       // consecutive equal bytes ending at the current byte
    unsigned prev = 0x100;
    u64 held_pos = 0, held_h = 0, held = 0;  // a run of identical windows not yet inserted
    for (size_t e = p0; e < last; ++e) {
      const unsigned ch = seq[e];
      h = h * MK_POLY_B + ch;
      if (e >= p0 + k) h -= bpow * (u64)seq[e - k];
      if (ch == MK_SEP) { since_sep = 0; } else { ++since_sep; }
      if (EXOTIC) { if (ch != MK_SEP && !in_alphabet(alphabet, ch)) since_bad = 0; else ++since_bad; }
      same = ch == prev ? same + 1 : 1;
      prev = ch;
      if (e + 1 >= p0 + k) {  // window [e-k+1, e] starts at >= p0
        if (since_sep >= k && (!EXOTIC || since_bad < k)) {
          ++mine;
          if (same > k && held) {
            ++held;  // the same k-mer as the window before (k + 1 equal bytes end here)
          } else {
            if (held) insert_ref(table, mask, seq, held_pos, held_h, k, held);
            held = 0;
            if (same >= k) { held = 1; held_pos = (u64)(e + 1 - k); held_h = mk_mix64(h); }
            else insert_ref(table, mask, seq, (u64)(e + 1 - k), mk_mix64(h), k, 1);
          }
        } else if (held) {
          insert_ref(table, mask, seq, held_pos, held_h, k, held);
          held = 0;
        }
      }
    }
    if (held) insert_ref(table, mask, seq, held_pos, held_h, k, held);
  }
  add_windows(info, mine, true);
}

// ----------------------------------------------------------------- packed by-reference
// 33 <= k <= 64 nucleotide windows without a bad symbol: the slot still stores the POSITION of
// the first occurrence (no 128-bit compare-and-swap exists), but hashing and equality run on the
// 2-bit packed stream: a window is its left-aligned 128-bit key, fetched from 3 packed words.
struct Key128 {
  u64 hi, lo;
};

// Left-aligned 2k-bit key of the window starting at base `pos` (k <= 64), low bits zeroed.
__device__ __forceinline__ Key128 key128_at(const u64* __restrict__ codes, u64 pos, int k) {
  const u64 w = pos >> 5;
  const int sh = (int)(pos & 31) * 2;
  const u64 a = codes[w], b = codes[w + 1], c = codes[w + 2];
  Key128 r;
  r.hi = sh ? ((a << sh) | (b >> (64 - sh))) : a;
  r.lo = sh ? ((b << sh) | (c >> (64 - sh))) : b;
  if (k <= 32) { r.hi &= (k == 32) ? ~0ull : (~0ull << (64 - 2 * k)); r.lo = 0; }
  else if (k < 64) r.lo &= ~0ull << (128 - 2 * k);
  return r;
}

// Amino acids, 13 <= k <= 25: 5 bits per symbol, 12 symbols in bits 63..4 of a packed word.  The key is the NUMBER
// sum(symbol_i * 32^(k-1-i)) in 128 bits (hi = its upper word): fixed length and codes in ASCII order, so (hi, lo)
// order is the byte order of the k-mer text.  Three packed words hold any such window (11 + 25 = 36 symbols).
__device__ __forceinline__ Key128 key128aa_from(u64 c0, u64 c1, u64 c2, int o, int k) {
  const unsigned __int128 p0 = c0 >> 4, p1 = c1 >> 4, p2 = c2 >> 4;  // 60-bit payloads, first symbol on top
  const unsigned __int128 top = (p0 << 60) | p1;                      // symbols 0 .. 23 of the three words
  const int r = 180 - 5 * o - 5 * k;                                  // bits to the right of the window, 0 .. 120
  unsigned __int128 v = r >= 60 ? (top >> (r - 60)) : ((top << (60 - r)) | (p2 >> r));
  v &= (((unsigned __int128)1 << (5 * k)) - 1);                       // (5 k <= 125)
  Key128 key;
  key.hi = (u64)(v >> 64);
  key.lo = (u64)v;
  return key;
}
__device__ __forceinline__ Key128 key128aa_at(const u64* __restrict__ codes, u64 pos, int k) {
  const u64 w = pos / 12;
  return key128aa_from(codes[w], codes[w + 1], codes[w + 2], (int)(pos - w * 12), k);
}

template <bool AA>
__device__ __forceinline__ void insert_ref128(MkSlot* __restrict__ table, u64 mask, const u64* __restrict__ codes,
                                              u64 pos, Key128 key, int k) {
  const u64 h = mk_mix64(key.hi ^ mk_mix64(key.lo + 0x9E3779B97F4A7C15ull));
  const u64 tag = (h >> 41) << REF_POS_BITS;
  const u64 mine = tag | pos;
  u64 slot = h & mask;
  for (;;) {
    u64 cur = table[slot].key;
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&table[slot].key, MK_EMPTY, mine);
      if (cur == MK_EMPTY) {
        atomicAdd(&table[slot].cnt, 1ull);
        return;
      }
    }
    if ((cur & ~REF_POS_MASK) == tag) {
      const Key128 other = AA ? key128aa_at(codes, cur & REF_POS_MASK, k) : key128_at(codes, cur & REF_POS_MASK, k);
      if (other.hi == key.hi && other.lo == key.lo) {
        atomicAdd(&table[slot].cnt, 1ull);
        return;
      }
    }
    slot = (slot + 1) & mask;
  }
}

// One thread per 32 window starts; 4 packed words cover 32 + 63 bases.
__global__ __launch_bounds__(256) void mk_count_ref128_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                         MkChunkInfo* __restrict__ info, MkSlot* __restrict__ table,
                                                         u64 mask, int k) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t p0 = t * 32;
  unsigned mine = 0;
  if (p0 < info->seq_len) {
    const u64 w0 = codes[t], w1 = codes[t + 1], w2 = codes[t + 2], w3 = codes[t + 3];
    // 128 bad bits for bases p0 .. p0+127 (p0 is a multiple of 32)
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
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      // window j is clean iff bad bits j .. j+k-1 are zero
      const u64 lo_part = b_lo >> j;                       // bits j..63 of b_lo -> 64-j bits
      const u64 hi_part = j ? (b_hi << (64 - j)) : 0ull;   // next bits
      const u64 win_lo = lo_part | hi_part;                // bad bits j .. j+63
      const u64 win_hi = b_hi >> j;                        // bad bits j+64 .. (only needed for k = 64 & j > 0: none)
      const u64 kmask = (k >= 64) ? ~0ull : ((1ull << k) - 1);
      (void)win_hi;
      if ((win_lo & kmask) == 0) {
        const int sh = 2 * j;
        Key128 key;
        key.hi = sh ? ((w0 << sh) | (w1 >> (64 - sh))) : w0;
        key.lo = sh ? ((w1 << sh) | (w2 >> (64 - sh))) : w1;
        if (k < 64) key.lo &= ~0ull << (128 - 2 * k);
        (void)w3;
        insert_ref128<false>(table, mask, codes, (u64)(p0 + j), key, k);
        ++mine;
      }
    }
  }
  add_windows(info, mine, true);
}

// The same for amino acids (13 <= k <= 25): one thread per packed word = 12 window starts.
__global__ __launch_bounds__(256) void mk_count_ref128aa_k(const u64* __restrict__ codes, const u64* __restrict__ bad,
                                                           MkChunkInfo* __restrict__ info, MkSlot* __restrict__ table,
                                                           u64 mask, int k) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t p0 = t * 12;
  unsigned mine = 0;
  if (p0 < info->seq_len) {
    const u64 c0 = codes[t], c1 = codes[t + 1], c2 = codes[t + 2];
    const u64 kmask = (1ull << k) - 1;  // (k <= 25)
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
      if ((bad_window(bad, p0 + j) & kmask) != 0) continue;  // (a separator, a character outside 'A'..'Z', or past the end)
      insert_ref128<true>(table, mask, codes, (u64)(p0 + j), key128aa_from(c0, c1, c2, j, k), k);
      ++mine;
    }
  }
  add_windows(info, mine, true);
}

int mk_launch_count_ref128(mk_ctx* c, size_t seq_len) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  if (c->alphabet == MK_ALPHABET_AA5) {
    const size_t words = (seq_len + 11) / 12;
    mk_prof_begin(c, MK_K_COUNT);
    hipLaunchKernelGGL(mk_count_ref128aa_k, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, (MkSlot*)c->rtab_chunk.p,
                       (u64)(c->rtab_chunk_slots - 1), c->k);
    mk_prof_end(c);
    MK_HIP(hipGetLastError());
    return MK_OK;
  }
  const size_t threads = (seq_len + 31) / 32;
  mk_prof_begin(c, MK_K_COUNT);
  hipLaunchKernelGGL(mk_count_ref128_k, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream,
                     (const u64*)c->codes.p, (const u64*)c->bad.p, info, (MkSlot*)c->rtab_chunk.p,
                     (u64)(c->rtab_chunk_slots - 1), c->k);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------------------ launchers
static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }

int mk_launch_count_hash64(mk_ctx* c, size_t seq_len) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const u64 mask = c->ctab_slots - 1;
  mk_prof_begin(c, MK_K_COUNT);
  if (c->alphabet == MK_ALPHABET_NT2) {
    const size_t threads = div_up(seq_len, 32);
    hipLaunchKernelGGL((mk_count_hash64_k<2, 32, 1>), dim3((unsigned)div_up(threads, 256)), dim3(256), 0, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, (MkSlot*)c->ctab.p, mask, c->k, c->canonical);
  } else {
    const size_t threads = div_up(seq_len, 36);
    hipLaunchKernelGGL((mk_count_hash64_k<5, 12, 3>), dim3((unsigned)div_up(threads, 256)), dim3(256), 0, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, (MkSlot*)c->ctab.p, mask, c->k, c->canonical);
  }
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_count_dense(mk_ctx* c, size_t seq_len) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const unsigned nbins = 1u << (c->bits * c->k);
  unsigned copies_log2 = 0;
  while ((nbins << (copies_log2 + 1)) <= 8192u && copies_log2 < 5) ++copies_log2;
  const size_t lds = (size_t)(nbins << copies_log2) * 4;
  mk_prof_begin(c, MK_K_COUNT);
  if (c->alphabet == MK_ALPHABET_NT2) {
    const size_t threads = div_up(seq_len, 32);
    size_t blocks = div_up(threads, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((mk_count_dense_k<2, 32, 1>), dim3((unsigned)blocks), dim3(256), lds, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, (u64*)c->ctab.p, nbins, copies_log2, c->k,
                       threads, c->canonical);
  } else {
    const size_t threads = div_up(seq_len, 36);
    size_t blocks = div_up(threads, 256);
    const size_t cap = (lds > 65536) ? 256 : 2048;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((mk_count_dense_k<5, 12, 3>), dim3((unsigned)blocks), dim3(256), lds, c->stream,
                       (const u64*)c->codes.p, (const u64*)c->bad.p, info, (u64*)c->ctab.p, nbins, copies_log2, c->k,
                       threads, c->canonical);
  }
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

static u64 pow_u64(u64 b, int e) {
  u64 r = 1;
  while (e > 0) {
    if (e & 1) r *= b;
    b *= b;
    e >>= 1;
  }
  return r;
}

int mk_launch_count_byref(mk_ctx* c, size_t seq_len, bool exotic_only) {
  if (seq_len == 0) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const u64 mask = c->rtab_chunk_slots - 1;
  const size_t threads = div_up(seq_len, RB);
  const u64 bpow = pow_u64(MK_POLY_B, c->k);
  mk_prof_begin(c, MK_K_EXOTIC);
  if (exotic_only)
    hipLaunchKernelGGL((mk_count_byref_k<true>), dim3((unsigned)div_up(threads, 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->seq.p, (const u64*)c->bad.p, info, (MkSlot*)c->rtab_chunk.p, mask, c->k,
                       c->alphabet, bpow);
  else
    hipLaunchKernelGGL((mk_count_byref_k<false>), dim3((unsigned)div_up(threads, 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->seq.p, (const u64*)nullptr, info, (MkSlot*)c->rtab_chunk.p, mask, c->k,
                       c->alphabet, bpow);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
