// mk_pack.hip -- symbol packing: seq (1 byte/symbol) -> codes (2-bit nt / 5-bit aa) + bad bitmap.
//
// The reference keys its dict by the substring itself (lib/mercat2_kmers.py:56-60); the packed
// key is the same string in a fixed-width integer: symbols MSB first, codes increasing in ASCII
// order (A<C<G<T, 'A'..'Z'), so integer order of keys == Python sorted() order of the strings
// (bin/mercat2.py:132).  Symbols outside the alphabet (and separators, and the padding past the
// end) get a 1 in the `bad` bitmap; a window with a bad bit never enters the packed path -- if
// it holds no separator it is counted by the by-reference kernel instead.
//
// Layout: codes word w holds symbols [w*SPW, (w+1)*SPW), symbol j of the word at bits
// (63 - BITS*j - (BITS-1)) .. (63 - BITS*j); SPW = 32 (nt) or 12 (aa, low 4 bits unused).
// bad word b holds symbols [64b, 64b+64), LSB first.
#include "mk_common.h"

__device__ __forceinline__ unsigned nt_code(unsigned c) { return ((c >> 1) & 3u) ^ ((c >> 2) & 1u); }  // A0 C1 G2 T3
__device__ __forceinline__ bool nt_ok(unsigned c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }
__device__ __forceinline__ bool aa_ok(unsigned c) { return c >= 'A' && c <= 'Z'; }

// nt: one thread per 64 symbols -> 2 code words + 1 bad word. `words` = code words to write
// (covers seq_len plus padding so that every reader stays in bounds and sees bad = 1 there).
__global__ __launch_bounds__(256) void mk_pack_nt(const uint8_t* __restrict__ seq, const MkChunkInfo* __restrict__ info,
                                                  unsigned long long* __restrict__ codes,
                                                  unsigned long long* __restrict__ bad, size_t bad_words,
                                                  MkChunkInfo* __restrict__ info_out) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bad_words) return;
  const size_t n = info->seq_len;
  const size_t s0 = b * 64;
  unsigned long long w0 = 0, w1 = 0, bd = 0;
  unsigned nbad = 0;
  if (s0 + 64 <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(seq + s0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint4 v = p[q];
      unsigned x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          unsigned c = (x[d] >> (8 * e)) & 0xFFu;
          int j = q * 16 + d * 4 + e;
          bool ok = nt_ok(c);
          unsigned long long code = ok ? nt_code(c) : 0u;
          if (j < 32) w0 |= code << (62 - 2 * j); else w1 |= code << (62 - 2 * (j - 32));
          if (!ok) { bd |= 1ull << j; nbad += (c != MK_SEP); }
        }
      }
    }
  } else {
    for (int j = 0; j < 64; ++j) {
      size_t i = s0 + j;
      unsigned c = (i < n) ? seq[i] : MK_SEP;
      bool ok = nt_ok(c);
      unsigned long long code = ok ? nt_code(c) : 0u;
      if (j < 32) w0 |= code << (62 - 2 * j); else w1 |= code << (62 - 2 * (j - 32));
      if (!ok) { bd |= 1ull << j; nbad += (i < n && c != MK_SEP); }
    }
  }
  codes[2 * b] = w0;
  codes[2 * b + 1] = w1;
  bad[b] = bd;
  if (nbad) atomicAdd(&info_out->bad_symbols, (unsigned long long)nbad);
}

// aa: bad bitmap, one thread per 64 symbols.
__global__ __launch_bounds__(256) void mk_pack_aa_bad(const uint8_t* __restrict__ seq, const MkChunkInfo* __restrict__ info,
                                                      unsigned long long* __restrict__ bad, size_t bad_words,
                                                      MkChunkInfo* __restrict__ info_out) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= bad_words) return;
  const size_t n = info->seq_len;
  const size_t s0 = b * 64;
  unsigned long long bd = 0;
  unsigned nbad = 0;
  if (s0 + 64 <= n) {
    // the 64 bytes as four 16-byte loads (byte loads at a 64-byte stride across the lanes were one memory transaction
    // per symbol: 111 us per 60 M residues, as long as the parser)
    const uint4* p = reinterpret_cast<const uint4*>(seq + s0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = p[q];
      const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned c = (w4[d] >> (8 * e)) & 0xFFu;
          if (!aa_ok(c)) { bd |= 1ull << (q * 16 + d * 4 + e); nbad += c != MK_SEP; }
        }
    }
  } else {
    for (int j = 0; j < 64; ++j) {
      size_t i = s0 + j;
      unsigned c = (i < n) ? seq[i] : MK_SEP;
      if (!aa_ok(c)) { bd |= 1ull << j; nbad += (i < n && c != MK_SEP); }
    }
  }
  bad[b] = bd;
  if (nbad) atomicAdd(&info_out->bad_symbols, (unsigned long long)nbad);
}

// aa: codes, one thread per word of 12 symbols.
__global__ __launch_bounds__(256) void mk_pack_aa_codes(const uint8_t* __restrict__ seq, const MkChunkInfo* __restrict__ info,
                                                        unsigned long long* __restrict__ codes, size_t words) {
  const size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= words) return;
  const size_t n = info->seq_len;
  const size_t s0 = w * 12;
  unsigned long long v = 0;
  if (s0 + 12 <= n) {  // three 32-bit loads (12 w is a multiple of four)
    const unsigned* p = reinterpret_cast<const unsigned*>(seq + s0);
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const unsigned x = p[d];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned c = (x >> (8 * e)) & 0xFFu;
        const unsigned long long code = aa_ok(c) ? (c - 'A') : 0u;
        v |= code << (59 - 5 * (4 * d + e));
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      size_t i = s0 + j;
      unsigned c = (i < n) ? seq[i] : MK_SEP;
      unsigned long long code = aa_ok(c) ? (c - 'A') : 0u;
      v |= code << (59 - 5 * j);
    }
  }
  codes[w] = v;
}

// seq_cap: upper bound of seq_len known on the host (the raw byte count). Buffers were
// reserved by the caller for mk_pack_words(seq_cap).
int mk_launch_pack(mk_ctx* c, size_t seq_cap) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  const size_t bad_words = seq_cap / 64 + 4;  // padding words are written as all-bad
  mk_prof_begin(c, MK_K_PACK);
  if (c->alphabet == MK_ALPHABET_NT2) {
    hipLaunchKernelGGL(mk_pack_nt, dim3((unsigned)((bad_words + 255) / 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->seq.p, info, (unsigned long long*)c->codes.p,
                       (unsigned long long*)c->bad.p, bad_words, info);
  } else {
    const size_t words = (bad_words * 64 + 11) / 12;
    hipLaunchKernelGGL(mk_pack_aa_bad, dim3((unsigned)((bad_words + 255) / 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->seq.p, info, (unsigned long long*)c->bad.p, bad_words, info);
    hipLaunchKernelGGL(mk_pack_aa_codes, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->stream,
                       (const uint8_t*)c->seq.p, info, (unsigned long long*)c->codes.p, words);
  }
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
