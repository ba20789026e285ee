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
