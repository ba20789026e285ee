// mk_tsv.hip -- the rows of a sorted table as TSV text, formatted on the GPU.
//
// Replaces the print loop of run_mercat2 (bin/mercat2.py:130-133: one "{kmer}\t{count}\n" per row, 13 s for 7.7 M rows in
// the reference, and a third of this engine's file-to-TSV window while one host thread did it: 25 ms for the 2 M rows of
// S2).  The sorted packed keys and counts are already on the device: a row's length is k + 2 + the count's decimal
// digits, a scan of the lengths gives every row its byte offset, one thread writes one row, and the host only copies
// the text out (pinned memory, piece by piece) and writes it to the file.
#include "mk_common.h"
#include <algorithm>
#include <rocprim/device/device_scan.hpp>

typedef unsigned long long u64;

__device__ __forceinline__ int tsv_digits(u64 v) {
  int d = 1;
  while (v >= 10ull) { v /= 10ull; ++d; }
  return d;
}

__global__ void mk_tsv_len_k(const u64* __restrict__ cnts, size_t rows, int k, u64* __restrict__ lens) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= rows; i += (size_t)gridDim.x * blockDim.x)
    lens[i] = i < rows ? (u64)(k + 2 + tsv_digits(cnts[i])) : 0ull;  // (one element more: its offset is the total)
}

// WORDS 64-bit words per key; BITS per symbol (2: A C G T, MSB-first; 5: 'A' + code).  One-word keys are numbers of
// BITS * k bits; two-word nucleotide keys hold bases 0..31 in hi and the rest in lo, both left-aligned; two-word
// protein keys are one number of 5 k bits in (hi, lo) (mk_common.h / DESIGN.md section 3).
template <int WORDS, int BITS>
__global__ void mk_tsv_fill_k(const u64* __restrict__ keys, const u64* __restrict__ cnts, const u64* __restrict__ off, size_t rows,
                              int k, char* __restrict__ out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    char* p = out + off[i];
    if (WORDS == 1) {
      const u64 key = keys[i];
      for (int j = 0; j < k; ++j) {
        const unsigned code = (unsigned)(key >> (BITS * (k - 1 - j))) & ((1u << BITS) - 1);
        p[j] = BITS == 2 ? (char)(0x54474341u >> (8 * code)) : (char)('A' + code);
      }
    } else {
      const u64 hi = keys[2 * i], lo = keys[2 * i + 1];
      if (BITS == 2) {
        for (int j = 0; j < k; ++j) {
          const unsigned code = (unsigned)((j < 32 ? hi >> (62 - 2 * j) : lo >> (62 - 2 * (j - 32))) & 3ull);
          p[j] = (char)(0x54474341u >> (8 * code));
        }
      } else {
        for (int j = 0; j < k; ++j) {
          const int s = 5 * (k - 1 - j);  // bits below this symbol in the 128-bit number
          u64 v;
          if (s >= 64) v = hi >> (s - 64);
          else if (s == 0) v = lo;
          else v = (lo >> s) | (hi << (64 - s));
          p[j] = (char)('A' + (unsigned)(v & 31ull));
        }
      }
    }
    p[k] = '\t';
    u64 n = cnts[i];
    const int d = tsv_digits(n);
    for (int q = d - 1; q >= 0; --q) { p[k + 1 + q] = (char)('0' + (unsigned)(n % 10ull)); n /= 10ull; }
    p[k + 1 + d] = '\n';
  }
}

// d_keys / d_cnts: the sorted rows on the device (words 64-bit words per key).  Writes the text of all rows into c->seq
// (grown as needed) and returns its length; d_off must hold rows + 1 words (scratch: offsets), d_len rows + 1 more.
int mk_launch_tsv_format(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_cnts, size_t rows, int words, uint64_t* d_len,
                         uint64_t* d_off, size_t* text_bytes) {
  *text_bytes = 0;
  if (!rows) return MK_OK;
  const unsigned grid = (unsigned)std::min<size_t>((rows + 256) / 256, 16384);
  hipLaunchKernelGGL(mk_tsv_len_k, dim3(grid), dim3(256), 0, c->stream, (const u64*)d_cnts, rows, c->k, (u64*)d_len);
  size_t tmp = 0;
  MK_HIP(rocprim::exclusive_scan((void*)nullptr, tmp, (const u64*)d_len, (u64*)d_off, 0ull, rows + 1, rocprim::plus<u64>(), c->stream));
  int rc = mk_buf_reserve(c, c->ex_tmp, tmp ? tmp : 16);
  if (rc) return rc;
  MK_HIP(rocprim::exclusive_scan(c->ex_tmp.p, tmp, (const u64*)d_len, (u64*)d_off, 0ull, rows + 1, rocprim::plus<u64>(), c->stream));
  u64 total = 0;
  MK_HIP(hipMemcpyAsync(&total, d_off + rows, 8, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  if ((rc = mk_buf_reserve(c, c->seq, (size_t)total + 64)) != MK_OK) return rc;
  char* out = (char*)c->seq.p;
  const int bits = c->alphabet == MK_ALPHABET_NT2 ? 2 : 5;
#define TSV_FILL(W, B) hipLaunchKernelGGL((mk_tsv_fill_k<W, B>), dim3(grid), dim3(256), 0, c->stream, (const u64*)d_keys, (const u64*)d_cnts, (const u64*)d_off, rows, c->k, out)
  if (words == 1) { if (bits == 2) TSV_FILL(1, 2); else TSV_FILL(1, 5); }
  else { if (bits == 2) TSV_FILL(2, 2); else TSV_FILL(2, 5); }
#undef TSV_FILL
  MK_HIP(hipGetLastError());
  *text_bytes = (size_t)total;
  return MK_OK;
}
