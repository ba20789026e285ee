// mk_fparse.hip -- fast FASTA parser for the common shape of input; the general transducer of
// mk_parse.hip stays the reference for everything else.
//
// Same contract as mk_parse.hip (reference loop lib/mercat2_kmers.py:49-69): raw bytes -> seq
// (kept characters, one MK_SEP where a header line starts).  The fast form assumes ONE thing:
// no byte <= 0x20 other than '\n' / '\r' occurs in a line that does not start with '>'.  Under
// that assumption strip() has nothing to trim in sequence lines and a line is a header iff its
// first byte is '>'.  The kernels check the assumption on every byte; when it fails they raise
// info->parse_fallback and the host re-parses the chunk with the general kernels (exact for any
// input).  Blanks inside header lines (">r1 some description") are the normal case and fine.
//
// Bit-parallel formulation.  Every lane owns 16 consecutive bytes (one 16-byte load) and turns
// them into 16-bit class masks (NL, '>', '*', low).  "Inside a header line" is the recurrence
//      H[j] = start[j] | (~NL[j] & H[j-1])
// which is exactly a carry chain: with a = start|~NL and b = start, the carries of a+b are H.
// One 32-bit add resolves it inside a lane, one 64-bit add over the wave's ballots resolves it
// across the 64 lanes (1 KiB), and the state between waves is a 2-state map scanned by
// mk_fparse_scan -- no per-byte branching anywhere.
//   pass 1 mk_fparse_summ  per wave (4 KiB): {has newline, exit state, bytes emitted for entry 0 / 1}
//   pass 2 mk_fparse_scan  exclusive scan of those maps -> entry state + output offset per wave
//   pass 3 mk_fparse_emit  recompute masks, compact the kept bytes into LDS at the alignment of
//                          their destination, write them out with aligned 16-byte stores
#include "mk_common.h"
#include "mk_device.h"

typedef unsigned long long u64;

#define FP_THREADS 256
#ifndef FP_SUB
#define FP_SUB 8
#endif
// 1 KiB sub-steps per wave (8: a quarter of the entries for the single-workgroup scan than with 2, measured best)
#define FP_WAVE_BYTES (FP_SUB * 1024)  // input bytes per wave
#define FP_WAVES (FP_THREADS / 64)

struct FpEntry {  // per wave summary / scan result
  unsigned a;     // summ: bit0 has_nl, bit1 exit0, bits 2.. separators (header lines that start in the wave)
  unsigned b;     // summ: cnt0 | cnt1 << 16
};

// Class masks of 16 bytes. Bytes outside [begin, n) behave like newlines (they emit nothing): `raw` is
// the 16-byte-aligned address at or below the chunk's first byte, `begin` the chunk's offset in it.
__device__ __forceinline__ uint4 classify16(const uint8_t* __restrict__ raw, size_t pos, size_t begin, size_t n, unsigned& nl,
                                            unsigned& gt, unsigned& st, unsigned& low, unsigned& hi) {
  uint4 v;
  if (pos + 16 <= n && pos >= begin) {
    v = *reinterpret_cast<const uint4*>(raw + pos);
  } else {
    unsigned w0 = 0, w1 = 0, w2 = 0, w3 = 0;  // static indexing only (no scratch): bytes past n read as '\n'
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const unsigned c = (pos + j < n && pos + j >= begin) ? (unsigned)raw[pos + j] : 10u;
      const unsigned sh = 8 * (j & 3);
      if (j < 4) w0 |= c << sh; else if (j < 8) w1 |= c << sh; else if (j < 12) w2 |= c << sh; else w3 |= c << sh;
    }
    v = make_uint4(w0, w1, w2, w3);
  }
  // Four bytes per 32-bit operation.  FP_NZ7(t): bit 7 of every byte of t that is not zero (the other bits are
  // garbage); so ~FP_NZ7(x ^ pattern) & 0x80808080 flags the bytes equal to the pattern.  A v_dot4_u32_u8 with the
  // weights 1,2,4,8 (16,..,128 for the odd words) then lines the four flags of a word up as a nibble: two words
  // accumulate into 128 x (8 mask bits).  (Byte by byte this took twice the instructions.)
#define FP_NZ7(t) ((((t) & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | (t))
  unsigned a_nl[2] = {0, 0}, a_gt[2] = {0, 0}, a_st[2] = {0, 0}, a_lo[2] = {0, 0}, a_hi[2] = {0, 0};
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const unsigned xd = d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));  // no array: stays in registers
    const unsigned K = (d & 1) ? 0x80402010u : 0x08040201u;
    const unsigned notnl = FP_NZ7(xd ^ 0x0A0A0A0Au) & FP_NZ7(xd ^ 0x0D0D0D0Du);
    const unsigned ge21 = ((xd & 0x7F7F7F7Fu) + 0x5F5F5F5Fu) | xd;  // bit 7: byte >= 0x21
    a_nl[d >> 1] = __builtin_amdgcn_udot4(~notnl & 0x80808080u, K, a_nl[d >> 1], false);
    a_gt[d >> 1] = __builtin_amdgcn_udot4(~FP_NZ7(xd ^ 0x3E3E3E3Eu) & 0x80808080u, K, a_gt[d >> 1], false);
    a_st[d >> 1] = __builtin_amdgcn_udot4(~FP_NZ7(xd ^ 0x2A2A2A2Au) & 0x80808080u, K, a_st[d >> 1], false);
    a_lo[d >> 1] = __builtin_amdgcn_udot4(~ge21 & notnl & 0x80808080u, K, a_lo[d >> 1], false);  // <= 0x20, not a newline
    a_hi[d >> 1] = __builtin_amdgcn_udot4(xd & 0x80808080u, K, a_hi[d >> 1], false);
  }
#undef FP_NZ7
  nl = (a_nl[0] >> 7) | ((a_nl[1] >> 7) << 8);
  gt = (a_gt[0] >> 7) | ((a_gt[1] >> 7) << 8);
  st = (a_st[0] >> 7) | ((a_st[1] >> 7) << 8);
  low = (a_lo[0] >> 7) | ((a_lo[1] >> 7) << 8);
  hi = (a_hi[0] >> 7) | ((a_hi[1] >> 7) << 8);  // bytes >= 0x80 (only the emit pass uses it: there it is known which bytes are kept)
  return v;
}

// Header mask of one lane given its entry state: carries of (start|~NL) + start + hin.
__device__ __forceinline__ unsigned lane_header(unsigned start, unsigned nl, unsigned hin, unsigned& hout) {
  const unsigned p = ~nl & 0xFFFFu;
  const unsigned a = start | p, b = start;
  const unsigned sum = a + b + hin;
  const unsigned ci = sum ^ a ^ b;  // bit j = carry into bit j ; bit j+1 = H[j]
  hout = (ci >> 16) & 1u;
  return (ci >> 1) & 0xFFFFu;
}

// One 1 KiB sub-step of a wave. In: class masks of this lane, prev_nl (previous byte of the
// stream is a newline, or start of chunk) and cin (stream is inside a header line).
// Out: `out` mask of emitted bytes, `sep` mask (emitted bytes that are separators), flags;
// returns the wave's exit header state; *last_nl = last byte of the sub-step is a newline.
__device__ __forceinline__ unsigned wave_step(unsigned nl, unsigned gt, unsigned st, unsigned low, unsigned prev_nl,
                                              unsigned cin, unsigned& out, unsigned& sep, unsigned& bad_low,
                                              unsigned& last_nl) {
  const int lane = threadIdx.x & 63;
  const unsigned my_last_nl = (nl >> 15) & 1u;
  // (the lane before this one: a DPP wave shift right by one; lane 0 keeps `old` = prev_nl)
  const unsigned prev = (unsigned)__builtin_amdgcn_update_dpp((int)prev_nl, (int)my_last_nl, 0x138, 0xf, 0xf, false);  // wave_shr:1
  const unsigned ls = ((nl << 1) | prev) & 0xFFFFu;  // line-start bits
  const unsigned start = ls & gt;                    // '>' at a line start
  unsigned hout0;
  (void)lane_header(start, nl, 0u, hout0);
  const u64 G = __ballot(hout0 != 0);        // lane ends inside a header whatever its entry
  const u64 P = __ballot(nl == 0);           // lane has no newline: passes its entry state on
  const u64 A = G | P, B = G;
  const u64 sum = A + B + (u64)cin;
  const u64 CI = sum ^ A ^ B;                // bit l = header state entering lane l
  const unsigned hin = (unsigned)(CI >> lane) & 1u;
  unsigned hout;
  const unsigned H = lane_header(start, nl, hin, hout);
  sep = start;
  out = ((~H & ~nl & ~st) | start) & 0xFFFFu;
  bad_low = low & ~H;
  last_nl = mk_wave_last(my_last_nl);
  return mk_wave_last(hout);
}

// zero_codes / zero_bad (fused nucleotide pack only): the emit pass ORs partial words in, so the packed words and the
// bad bitmap must start from zero.  This pass is over before the emit pass starts, so every wave clears the slice
// that corresponds to its input bytes (two memset launches of 39 MB per 100 MiB chunk less); the last wave takes
// the padding words as well.
__global__ __launch_bounds__(FP_THREADS) void mk_fparse_summ(const uint8_t* __restrict__ raw, size_t begin, size_t n, size_t nwaves,
                                                             FpEntry* __restrict__ entries, MkChunkInfo* __restrict__ info,
                                                             u64* __restrict__ zero_codes, u64* __restrict__ zero_bad,
                                                             size_t code_words, size_t bad_words) {
  const size_t wave = (size_t)blockIdx.x * FP_WAVES + (threadIdx.x >> 6);
  if (wave >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * FP_WAVE_BYTES;
  if (zero_codes) {
    constexpr size_t CW = FP_WAVE_BYTES / 32, BW = FP_WAVE_BYTES / 64;  // words of this wave's slice
    const size_t c_end = wave + 1 == nwaves ? code_words : (wave + 1) * CW, b_end = wave + 1 == nwaves ? bad_words : (wave + 1) * BW;
    for (size_t w = wave * CW + (size_t)lane; w < c_end && w < code_words; w += 64) zero_codes[w] = 0ull;
    for (size_t w = wave * BW + (size_t)lane; w < b_end && w < bad_words; w += 64) zero_bad[w] = 0ull;
  }
  unsigned prev_nl = (base <= begin) ? 1u : ((raw[base - 1] == 10 || raw[base - 1] == 13) ? 1u : 0u);
  unsigned s0 = 0, s1 = 1, c0 = 0, c1 = 0, seps = 0, any_nl = 0, flag_low = 0;
#pragma unroll 1
  for (int sub = 0; sub < FP_SUB; ++sub) {
    const size_t pos = base + (size_t)sub * 1024 + (size_t)lane * 16;
    unsigned nl, gt, st, low, hi;
    (void)classify16(raw, pos, begin, n, nl, gt, st, low, hi);
    unsigned out, sep, bl, last_nl;
    const unsigned e0 = wave_step(nl, gt, st, low, prev_nl, s0, out, sep, bl, last_nl);
    c0 += __popc(out);
    seps += __popc(sep);  // '>' at a line start: the same whatever the entry state
    // (a blank inside a sequence line is looked for by the EMIT pass, which knows the wave's true entry state: under
    // the wrong one of the two assumed here every blank of a header line that the wave starts inside looked like one,
    // and read files whose header lines hold blanks -- most do -- fell back to the general parser, chunk after chunk)
    (void)bl;
    if (s1 != s0) {
      unsigned out1, sep1, bl1, ln1;
      const unsigned e1 = wave_step(nl, gt, st, low, prev_nl, s1, out1, sep1, bl1, ln1);
      c1 += __popc(out1);
      s1 = e1;
    } else {
      c1 += __popc(out);
      s1 = e0;
    }
    s0 = e0;
    any_nl |= (__ballot(nl != 0) != 0) ? 1u : 0u;
    prev_nl = last_nl;
  }
  c0 = mk_wave_sum(c0);  // (DPP: mk_device.h)
  c1 = mk_wave_sum(c1);
  seps = mk_wave_sum(seps);
  (void)flag_low;
  if (lane == 0) {
    entries[wave].a = any_nl | (s0 << 1) | (seps << 2);
    entries[wave].b = c0 | (c1 << 16);
  }
}

struct FpScan {
  u64 off;
  unsigned st;
  unsigned pad;
};

// One workgroup: thread t owns waves [t*per, (t+1)*per). A wave without a newline passes its
// entry state through (or-ed with its own exit0); one with a newline resets it to exit0.
__global__ __launch_bounds__(1024) void mk_fparse_scan(const FpEntry* __restrict__ entries, size_t nwaves,
                                                       FpScan* __restrict__ scan, MkChunkInfo* __restrict__ info,
                                                       u64* __restrict__ bad) {
  const size_t per = (nwaves + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = (lo + per < nwaves) ? lo + per : nwaves;
  unsigned e0 = 0, e1 = 1, has = 0;  // running map of this thread's range
  u64 c0 = 0, c1 = 0, nsep = 0;
  for (size_t w0 = lo; w0 < hi; w0 += 8) {
    FpEntry en[8];  // 8 independent loads in flight, then the (serial) state update
#pragma unroll
    for (int i = 0; i < 8; ++i) en[i] = (w0 + i < hi) ? entries[w0 + i] : FpEntry{0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (w0 + i >= hi) break;
      const unsigned wnl = en[i].a & 1u, wx0 = (en[i].a >> 1) & 1u;
      const unsigned k0 = en[i].b & 0xFFFFu, k1 = en[i].b >> 16;
      nsep += en[i].a >> 2;
      c0 += e0 ? k1 : k0;
      c1 += e1 ? k1 : k0;
      e0 = wnl ? wx0 : (e0 | wx0);
      e1 = wnl ? wx0 : (e1 | wx0);
      has |= wnl;
    }
  }
  // ---- entry state of every thread's range: the same carry chain as inside the parser
  //      (generate = range ends in a header whatever its entry, propagate = range has no newline)
  __shared__ unsigned long long w_G[16], w_P[16];
  __shared__ unsigned w_cin[16];
  __shared__ u64 w_cnt[16], w_sep[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const u64 G = __ballot(e0 != 0), P = __ballot(has == 0);
  if (lane == 0) { w_G[wv] = G; w_P[wv] = P; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned cin = 0;  // the chunk starts outside any header
    for (int w = 0; w < 16; ++w) {
      w_cin[w] = cin;
      const u64 A = w_G[w] | w_P[w], B = w_G[w];
      const u64 sum = A + B + cin;
      const u64 CI = sum ^ A ^ B;
      cin = (unsigned)((w_G[w] >> 63) | ((w_P[w] >> 63) & (CI >> 63))) & 1u;  // carry out of lane 63
    }
  }
  __syncthreads();
  unsigned q;
  {
    const u64 A = G | P, B = G;
    const u64 CI = (A + B + (u64)w_cin[wv]) ^ A ^ B;
    q = (unsigned)(CI >> lane) & 1u;
  }
  // ---- output offsets: prefix sum of the count that matches each range's entry state
  const u64 mine = q ? c1 : c0;
  u64 inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const u64 up = __shfl_up(inc, d);
    if (lane >= d) inc += up;
  }
  for (int d = 32; d > 0; d >>= 1) nsep += __shfl_down(nsep, d);
  if (lane == 63) w_cnt[wv] = inc;
  if (lane == 0) w_sep[wv] = nsep;
  __syncthreads();
  u64 off = inc - mine;
  u64 total = 0, seps = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    const u64 v = w_cnt[w];
    if (w < wv) off += v;
    total += v;
    seps += w_sep[w];
  }
  if (threadIdx.x == 0) {
    info->seq_len = total;
    info->symbols = total - seps;  // kept bytes that are not separators
    if (bad) {  // symbols past the end of seq are "bad" (fused pack): tail of the last word + 3 more words.  The summary
                // pass has cleared the bitmap; the emit pass only ORs into the (partial) last word and never touches the
                // words behind it
      if (total & 63) atomicOr(&bad[total >> 6], ~0ull << (total & 63));
      const u64 w = (total + 63) >> 6;
      bad[w] = ~0ull;
      bad[w + 1] = ~0ull;
      bad[w + 2] = ~0ull;
    }
  }
  for (size_t w0 = lo; w0 < hi; w0 += 8) {
    FpEntry en[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) en[i] = (w0 + i < hi) ? entries[w0 + i] : FpEntry{0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (w0 + i >= hi) break;
      scan[w0 + i].off = off;
      scan[w0 + i].st = q;
      const unsigned wnl = en[i].a & 1u, wx0 = (en[i].a >> 1) & 1u;
      off += q ? (en[i].b >> 16) : (en[i].b & 0xFFFFu);
      q = wnl ? wx0 : (q | wx0);
    }
  }
}

__global__ __launch_bounds__(FP_THREADS) void mk_fparse_emit(const uint8_t* __restrict__ raw, size_t begin, size_t n, size_t nwaves,
                                                             FpScan* __restrict__ scan, uint8_t* __restrict__ seq,
                                                             MkChunkInfo* __restrict__ info, u64* __restrict__ codes,
                                                             u64* __restrict__ bad, int write_seq) {
  // 64 bytes of slack on both sides: the fused pack reads whole 64-symbol words around the ends
  // (+ 256: one dump word per lane for the bytes a lane drops, see the compaction below)
  __shared__ __attribute__((aligned(16))) uint8_t stage[FP_WAVES][64 + FP_WAVE_BYTES + 32 + 64 + 256];
  const int wv = threadIdx.x >> 6;
  const size_t wave = (size_t)blockIdx.x * FP_WAVES + wv;
  if (wave >= nwaves) return;
  const int lane = threadIdx.x & 63;
  const size_t base = wave * FP_WAVE_BYTES;
  const FpScan sc = scan[wave];
  const unsigned shift = (unsigned)(sc.off & 15);
  uint8_t* __restrict__ lds = stage[wv] + 64;
  unsigned prev_nl = (base <= begin) ? 1u : ((raw[base - 1] == 10 || raw[base - 1] == 13) ? 1u : 0u);
  unsigned state = sc.st;
  unsigned filled = 0;   // bytes emitted so far by this wave
  int nbad = 0;          // kept characters outside the alphabet (separators subtracted: they are marked bad, not counted)
  unsigned flag_low = 0;
  unsigned nhi = 0;      // KEPT bytes >= 0x80: sequence characters the reference would decode as multi-byte text
                         // (bytes of header lines never enter a k-mer: lib/mercat2_kmers.py:52-53)
#pragma unroll 1
  for (int sub = 0; sub < FP_SUB; ++sub) {
    const size_t pos = base + (size_t)sub * 1024 + (size_t)lane * 16;
    unsigned nl, gt, st, low, hi;
    const uint4 v = classify16(raw, pos, begin, n, nl, gt, st, low, hi);
    unsigned out, sep, bl, last_nl;
    state = wave_step(nl, gt, st, low, prev_nl, state, out, sep, bl, last_nl);
    prev_nl = last_nl;
    flag_low |= bl;  // a blank (or control byte) outside header lines: this parser's assumption does not hold
    const unsigned cnt = __popc(out);
    nbad -= (int)__popc(sep);
    nhi += __popc(hi & out & ~sep);
    const unsigned inc = mk_wave_scan_incl(cnt);  // inclusive scan over the wave
    unsigned at = shift + filled + inc - cnt;
#define FP_BYTE(j) ((uint8_t)(((j) < 4 ? v.x : ((j) < 8 ? v.y : ((j) < 12 ? v.z : v.w))) >> (8 * ((j) & 3))))
    if (out == 0xFFFFu && sep == 0) {
#ifdef FP_BYTE_STORES
#pragma unroll
      for (int j = 0; j < 16; ++j) lds[at + j] = FP_BYTE(j);
#else
      // all 16 bytes kept: one unaligned 16-byte LDS store (gfx950 takes unaligned DS accesses) instead of 16 byte stores
      __builtin_memcpy(lds + at, &v, 16);
#endif
    } else if (out) {
      // Some bytes dropped (a line end, a header): 16 byte stores without a branch -- a kept byte goes to the running
      // position, a dropped one to the lane's dump word (a branch per byte cost twice the instructions, and every wave
      // of a read file holds such lanes).  Separators are then written over the '>' bytes they stand for.
      const unsigned dump = FP_WAVE_BYTES + 32 + 64 + ((unsigned)lane << 2);
      unsigned a = at;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const unsigned kept = (out >> j) & 1u;
        lds[kept ? a : dump] = FP_BYTE(j);
        a += kept;
      }
      for (unsigned sp = sep; sp; sp &= sp - 1) lds[at + __popc(out & ((1u << (__ffs(sp) - 1)) - 1u))] = (uint8_t)MK_SEP;
    }
#undef FP_BYTE
    filled += mk_wave_last(inc);
  }
  // ---- write out: LDS offset == destination address (mod 16), so full 16-byte pieces are aligned
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  uint8_t* __restrict__ dst = seq + (sc.off - shift);
  const unsigned end = shift + filled;
  // (write_seq == 0: the caller only wants the packed words and the bad bitmap -- the parsed stream is read again by the
  // by-reference kernel alone, i.e. only when the chunk holds characters outside the alphabet, and then the chunk is
  // parsed once more with the stream written: 94 MB per 100 MiB chunk of reads less to write)
  for (unsigned c = (unsigned)lane * 16; write_seq && c < end; c += 64 * 16) {
    if (c >= shift && c + 16 <= end) {
      *reinterpret_cast<uint4*>(dst + c) = *reinterpret_cast<const uint4*>(lds + c);
    } else {
      const unsigned lo = c < shift ? shift : c;
      const unsigned hi = c + 16 < end ? c + 16 : end;
      for (unsigned j = lo; j < hi; ++j) dst[j] = lds[j];
    }
  }
  // ---- fused 2-bit pack (nt): the wave's symbols are still in LDS, so the packed words and the
  //      bad bitmap are produced here instead of re-reading seq (mk_pack.hip layout). Words that
  //      the wave covers completely are stored; the partial word at either end is OR-ed into the
  //      zero-initialised arrays (the neighbouring wave ORs its part).
  if (codes) {
    const u64 g0 = sc.off, g1 = sc.off + filled;
    for (u64 b = (g0 >> 6) + lane; b < ((g1 + 63) >> 6); b += 64) {
      const u64 s_lo = b << 6, s_hi = s_lo + 64;
      const u64 lo = s_lo > g0 ? s_lo : g0, hi = s_hi < g1 ? s_hi : g1;
      u64 w0 = 0, w1 = 0, bd = 0;
      unsigned b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      // always the 64-byte vector path, four characters per 32-bit operation:
      //   code      = ((ch >> 1) & 3) ^ ((ch >> 2) & 1)      A 0, C 1, G 2, T 3 (any other byte: something in 0..3)
      //   expected  = "ACGT"[code]   (one v_perm_b32 with the codes as byte selectors)
      //   bad       = expected != ch (non-zero byte test), its code cleared
      //   8 bits of codes, first character on top: a dot product with the weights 64, 16, 4, 1
      const uint4* p = reinterpret_cast<const uint4*>(lds + (long)shift + (long)(s_lo - g0));  // 16-byte aligned
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 v = p[q];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const unsigned xd = d == 0 ? v.x : (d == 1 ? v.y : (d == 2 ? v.z : v.w));
          const unsigned c = ((xd >> 1) & 0x03030303u) ^ ((xd >> 2) & 0x01010101u);
          const unsigned df = __builtin_amdgcn_perm(0u, 0x54474341u, c) ^ xd;
          const unsigned nz = (((df & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | df) & 0x80808080u;  // 0x80 in every bad byte
          const unsigned m1 = nz >> 7;
          const unsigned ck = c & ~(m1 | (m1 << 1));                                     // (a bad character's code is 0)
          // v_dot4_u32_u8 as the gather: four codes -> one byte (first character on top), four flags -> one nibble
          const unsigned g = __builtin_amdgcn_udot4(ck, 0x01041040u, 0u, false);
          const int idx = q * 4 + d;  // characters 4 idx .. 4 idx + 3 of the word
          if (idx < 8) w0 |= (u64)g << (56 - 8 * idx); else w1 |= (u64)g << (56 - 8 * (idx - 8));
          b8[idx >> 1] = __builtin_amdgcn_udot4(nz, (idx & 1) ? 0x80402010u : 0x08040201u, b8[idx >> 1], false);  // 128 x flags
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) bd |= (u64)(b8[i] >> 7) << (8 * i);
      if (hi - lo == 64) {
        codes[2 * b] = w0;
        codes[2 * b + 1] = w1;
        bad[b] = bd;
      } else {
        // symbols of the word outside [lo, hi) (another wave's, or past the end: whatever the LDS slack holds) are
        // masked off; the neighbouring wave ORs its part in
        const unsigned a = (unsigned)(lo - s_lo), e = (unsigned)(hi - s_lo);  // valid symbols a .. e-1, a < e
        const u64 inmask = ((e - a) >= 64 ? ~0ull : ((1ull << (e - a)) - 1)) << a;
        const unsigned a0 = a < 32 ? a : 32, e0 = e < 32 ? e : 32, a1 = a > 32 ? a - 32 : 0, e1 = e > 32 ? e - 32 : 0;
        // symbol j of a word sits at bits 63-2j, 62-2j: symbols x .. y-1 are the bit range [64-2y, 64-2x)
        const u64 m0 = a0 < e0 ? ((~0ull >> (2 * a0)) & ~(e0 == 32 ? 0ull : (~0ull >> (2 * e0)))) : 0ull;
        const u64 m1 = a1 < e1 ? ((~0ull >> (2 * a1)) & ~(e1 == 32 ? 0ull : (~0ull >> (2 * e1)))) : 0ull;
        w0 &= m0;
        w1 &= m1;
        bd &= inmask;
        if (w0) atomicOr(&codes[2 * b], w0);
        if (w1) atomicOr(&codes[2 * b + 1], w1);
        if (bd) atomicOr(&bad[b], bd);
      }
      nbad += (int)__popcll(bd);
    }
  }
  // (without the fused pack nobody reads the sum: the pack kernel counts the bad symbols)
  for (int d = 32; d > 0; d >>= 1) nbad += __shfl_down(nbad, d);
  if (codes && lane == 0 && nbad > 0) atomicAdd(&info->bad_symbols, (u64)nbad);
  if (__ballot(flag_low != 0) && lane == 0) atomicOr(&info->parse_fallback, 1ull);  // the host re-parses the chunk generally
  if (__ballot(nhi != 0)) {  // (never, for ASCII input)
    for (int d = 32; d > 0; d >>= 1) nhi += __shfl_down(nhi, d);
    if (lane == 0) atomicAdd(&info->non_ascii, (u64)nhi);
  }
}

// Returns MK_OK after enqueueing; info->parse_fallback != 0 afterwards means "re-parse with the general kernels".
// d_raw is 16-byte aligned; the chunk is its bytes [begin, begin + len) (begin < 16).
int mk_launch_fparse(mk_ctx* c, const uint8_t* d_raw, size_t begin, size_t len, bool fuse_pack_nt, bool write_seq) {
  if (!fuse_pack_nt) write_seq = true;
  const size_t n = begin + len;
  u64* codes = fuse_pack_nt ? (u64*)c->codes.p : nullptr;
  u64* bad = fuse_pack_nt ? (u64*)c->bad.p : nullptr;
  const size_t bad_words = n / 64 + 4;  // partial words are OR-ed in: they start from zero (padding words included)
  if (fuse_pack_nt && len == 0) {
    MK_HIP(hipMemsetAsync(c->codes.p, 0, 2 * bad_words * sizeof(u64), c->stream));
    MK_HIP(hipMemsetAsync(c->bad.p, 0, bad_words * sizeof(u64), c->stream));
    if (n == 0) MK_HIP(hipMemsetAsync(c->bad.p, 0xFF, 4 * sizeof(u64), c->stream));
  }
  if (len == 0) return MK_OK;
  const size_t nwaves = (n + FP_WAVE_BYTES - 1) / FP_WAVE_BYTES;
  const size_t e_bytes = (nwaves * sizeof(FpEntry) + 15) & ~(size_t)15;
  int rc = mk_buf_reserve(c, c->tile_maps, e_bytes + nwaves * sizeof(FpScan));
  if (rc) return rc;
  FpEntry* entries = (FpEntry*)c->tile_maps.p;
  FpScan* scan = (FpScan*)((char*)c->tile_maps.p + e_bytes);
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const unsigned blocks = (unsigned)((nwaves + FP_WAVES - 1) / FP_WAVES);
  mk_prof_begin(c, MK_K_PARSE);
  hipLaunchKernelGGL(mk_fparse_summ, dim3(blocks), dim3(FP_THREADS), 0, c->stream, d_raw, begin, n, nwaves, entries, info, codes, bad,
                     2 * bad_words, bad_words);  // (the summary pass clears the packed words and the bad bitmap)
  hipLaunchKernelGGL(mk_fparse_scan, dim3(1), dim3(1024), 0, c->stream, (const FpEntry*)entries, nwaves, scan, info, bad);
  hipLaunchKernelGGL(mk_fparse_emit, dim3(blocks), dim3(FP_THREADS), 0, c->stream, d_raw, begin, n, nwaves, scan,
                     (uint8_t*)c->seq.p, info, codes, bad, write_seq ? 1 : 0);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
