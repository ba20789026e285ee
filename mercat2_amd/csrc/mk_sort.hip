// mk_sort.hip -- key order of the exported table.
//
// The reference writes rows in Python sorted() order of the k-mer strings (bin/mercat2.py:132).
// Packed keys are MSB-first with codes in ASCII order, so that order is the unsigned integer
// order of the keys: one device radix sort of (key, count) pairs over the key's used bits.
// rocPRIM's device radix sort is the ROCm library primitive used for this step (it is not on
// the counting hot path: it runs once per sample over the surviving rows).
#include "mk_common.h"
#include <cstring>
#include <string.h>
#include <algorithm>
#include <rocprim/device/device_radix_sort.hpp>

int mk_sort_pairs(mk_ctx* c, const uint64_t* keys_in, const uint64_t* vals_in, uint64_t* keys_out, uint64_t* vals_out,
                  size_t n, int key_bits) {
  if (n == 0) return MK_OK;
  if (key_bits < 1) key_bits = 1;
  if (key_bits > 64) key_bits = 64;
  size_t tmp_bytes = 0;
  MK_HIP(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, (const unsigned long long*)keys_in,
                                   (unsigned long long*)keys_out, (const unsigned long long*)vals_in,
                                   (unsigned long long*)vals_out, n, 0u, (unsigned)key_bits, c->stream));
  int rc = mk_buf_reserve(c, c->ex_tmp, tmp_bytes ? tmp_bytes : 16);
  if (rc) return rc;
  MK_HIP(rocprim::radix_sort_pairs(c->ex_tmp.p, tmp_bytes, (const unsigned long long*)keys_in,
                                   (unsigned long long*)keys_out, (const unsigned long long*)vals_in,
                                   (unsigned long long*)vals_out, n, 0u, (unsigned)key_bits, c->stream));
  return MK_OK;
}

// ------------------------------------------------------------------ rows kept as text
// By-reference rows (k bytes each in the arena) in byte order == Python's sorted(str) for ASCII:
// least-significant-word-first radix sort over ceil(k / 8) big-endian 8-byte words of the rows, each pass
// a stable device sort of (word of the row at the current position, row).  d_order receives the row
// indices in sorted order.  (A std::sort with memcmp on the host took 4 s for 10 M 63-mers.)
__global__ void mk_row_word_k(const uint8_t* __restrict__ arena, const unsigned long long* __restrict__ order, size_t rows,
                              int k, int word, unsigned long long* __restrict__ keys, unsigned long long* __restrict__ vals) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long r = order ? order[i] : (unsigned long long)i;
    const uint8_t* p = arena + (size_t)r * (size_t)k + 8 * (size_t)word;
    const int nb = k - 8 * word < 8 ? k - 8 * word : 8;
    unsigned long long key = 0;
    for (int b = 0; b < 8; ++b) key = (key << 8) | (b < nb ? (unsigned long long)p[b] : 0ull);
    keys[i] = key;
    vals[i] = r;
  }
}

int mk_sort_rows(mk_ctx* c, const uint8_t* d_arena, size_t rows, int k, uint64_t** d_order) {
  *d_order = nullptr;
  if (rows == 0) return MK_OK;
  int rc;
  if ((rc = mk_buf_reserve(c, c->ex_keys, rows * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_cnts, rows * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_keys2, rows * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_cnts2, rows * 8 + 64)) != MK_OK) return rc;
  unsigned long long* keys_a = (unsigned long long*)c->ex_keys.p;
  unsigned long long* vals_a = (unsigned long long*)c->ex_cnts.p;
  unsigned long long* keys_b = (unsigned long long*)c->ex_keys2.p;
  unsigned long long* vals_b = (unsigned long long*)c->ex_cnts2.p;
  const int words = (k + 7) / 8;
  const unsigned grid = (unsigned)std::min<size_t>((rows + 255) / 256, 8192);
  const unsigned long long* order = nullptr;
  for (int w = words - 1; w >= 0; --w) {
    hipLaunchKernelGGL(mk_row_word_k, dim3(grid), dim3(256), 0, c->stream, d_arena, order, rows, k, w, keys_a, vals_a);
    if ((rc = mk_sort_pairs(c, (const uint64_t*)keys_a, (const uint64_t*)vals_a, (uint64_t*)keys_b, (uint64_t*)vals_b, rows, 64)) != MK_OK)
      return rc;
    order = vals_b;
  }
  MK_HIP(hipGetLastError());
  *d_order = (uint64_t*)vals_b;
  return MK_OK;
}

// Rows and their counts in sorted order, written by the device: the host then walks them front to back
// (following the order through a 600 MB array of rows cost a second of cache misses for 10 M rows).
__global__ void mk_rows_by_slot_k(const unsigned long long* __restrict__ slot_keys, const unsigned long long* __restrict__ slot_cnts,
                                  size_t rows, unsigned long long* __restrict__ cnt_by_row, unsigned long long* __restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long row = slot_keys[i] & ((1ull << 40) - 1);
    if (row < rows) cnt_by_row[row] = slot_cnts[i];
    else atomicAdd(bad, 1ull);
  }
}
__global__ void mk_rows_gather_k(const uint8_t* __restrict__ arena, const unsigned long long* __restrict__ order,
                                 const unsigned long long* __restrict__ cnt_by_row, size_t rows, int k,
                                 uint8_t* __restrict__ out_rows, unsigned long long* __restrict__ out_cnts) {
  // one wave per row: lanes copy its k bytes side by side
  const int lane = threadIdx.x & 63;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t i = wave; i < rows; i += nwaves) {
    const unsigned long long r = order[i];
    const uint8_t* src = arena + (size_t)r * (size_t)k;
    uint8_t* dst = out_rows + i * (size_t)k;
    for (int b = lane; b < k; b += 64) dst[b] = src[b];
    if (lane == 0) out_cnts[i] = cnt_by_row[r];
  }
}

int mk_launch_rows_by_slot(mk_ctx* c, const uint64_t* slot_keys, const uint64_t* slot_cnts, size_t rows, uint64_t* cnt_by_row,
                           uint64_t* d_bad) {
  const unsigned grid = (unsigned)std::min<size_t>((rows + 255) / 256, 8192);
  hipLaunchKernelGGL(mk_rows_by_slot_k, dim3(grid), dim3(256), 0, c->stream, (const unsigned long long*)slot_keys,
                     (const unsigned long long*)slot_cnts, rows, (unsigned long long*)cnt_by_row, (unsigned long long*)d_bad);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
int mk_launch_rows_gather(mk_ctx* c, const uint8_t* arena, const uint64_t* order, const uint64_t* cnt_by_row, size_t rows, int k,
                          uint8_t* out_rows, uint64_t* out_cnts) {
  const unsigned grid = (unsigned)std::min<size_t>((rows * 64 + 255) / 256, 16384);
  hipLaunchKernelGGL(mk_rows_gather_k, dim3(grid), dim3(256), 0, c->stream, arena, (const unsigned long long*)order,
                     (const unsigned long long*)cnt_by_row, rows, k, out_rows, (unsigned long long*)out_cnts);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------ two-word packed rows
// Rows {hi, lo, count} of the 33..64-mer table in key order (hi, then lo == byte order of the k-mer text):
// least-significant word first -- a stable sort of the row numbers by lo (only its used bits), then by hi --
// and one gather that writes the rows {hi, lo} interleaved.  scratch: 4 * n words.
__global__ void mk_iota_k(unsigned long long* __restrict__ v, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) v[i] = i;
}
__global__ void mk_gather_u64_k(const unsigned long long* __restrict__ src, const unsigned long long* __restrict__ idx, size_t n,
                                unsigned long long* __restrict__ dst) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[idx[i]];
}
__global__ void mk_gather_rows128_k(const unsigned long long* __restrict__ hi, const unsigned long long* __restrict__ lo,
                                    const unsigned long long* __restrict__ cnt, const unsigned long long* __restrict__ idx, size_t n,
                                    ulonglong2* __restrict__ keys2, unsigned long long* __restrict__ cnts) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long r = idx[i];
    keys2[i] = make_ulonglong2(hi[r], lo[r]);
    cnts[i] = cnt[r];
  }
}

static int sort_pairs_bits(mk_ctx* c, const uint64_t* keys_in, const uint64_t* vals_in, uint64_t* keys_out, uint64_t* vals_out,
                           size_t n, unsigned begin_bit, unsigned end_bit) {
  size_t tmp_bytes = 0;
  MK_HIP(rocprim::radix_sort_pairs((void*)nullptr, tmp_bytes, (const unsigned long long*)keys_in, (unsigned long long*)keys_out,
                                   (const unsigned long long*)vals_in, (unsigned long long*)vals_out, n, begin_bit, end_bit, c->stream));
  int rc = mk_buf_reserve(c, c->ex_tmp, tmp_bytes ? tmp_bytes : 16);
  if (rc) return rc;
  MK_HIP(rocprim::radix_sort_pairs(c->ex_tmp.p, tmp_bytes, (const unsigned long long*)keys_in, (unsigned long long*)keys_out,
                                   (const unsigned long long*)vals_in, (unsigned long long*)vals_out, n, begin_bit, end_bit, c->stream));
  return MK_OK;
}

int mk_sort_pairs128(mk_ctx* c, const uint64_t* hi, const uint64_t* lo, const uint64_t* cnts, size_t n, int lo_bits,
                     uint64_t* scratch, uint64_t* keys2_out, uint64_t* cnts_out) {
  if (n == 0) return MK_OK;
  int rc;
  unsigned long long* idx_a = (unsigned long long*)scratch;
  unsigned long long* idx_b = idx_a + n;
  unsigned long long* key_a = idx_b + n;
  unsigned long long* key_b = key_a + n;
  const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  hipLaunchKernelGGL(mk_iota_k, dim3(grid), dim3(256), 0, c->stream, idx_a, n);
  // (all 64 bits of lo: its unused low bits are zero.  A sub-range [64 - lo_bits, 64) left rows with equal
  // hi out of lo order on gfx950 / rocPRIM 7.2 at a few 100 k rows -- measured, tools/dbg128.py)
  (void)lo_bits;
  if ((rc = sort_pairs_bits(c, lo, (const uint64_t*)idx_a, (uint64_t*)key_a, (uint64_t*)idx_b, n, 0u, 64u)) != MK_OK) return rc;
  hipLaunchKernelGGL(mk_gather_u64_k, dim3(grid), dim3(256), 0, c->stream, (const unsigned long long*)hi, (const unsigned long long*)idx_b, n, key_a);
  if ((rc = sort_pairs_bits(c, (const uint64_t*)key_a, (const uint64_t*)idx_b, (uint64_t*)key_b, (uint64_t*)idx_a, n, 0u, 64u)) != MK_OK) return rc;
  hipLaunchKernelGGL(mk_gather_rows128_k, dim3(grid), dim3(256), 0, c->stream, (const unsigned long long*)hi, (const unsigned long long*)lo,
                     (const unsigned long long*)cnts, (const unsigned long long*)idx_a, n, (ulonglong2*)keys2_out, (unsigned long long*)cnts_out);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
