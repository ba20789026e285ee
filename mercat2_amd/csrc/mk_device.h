// mk_device.h -- device helpers shared by the counting kernels.
#pragma once
#include "mk_common.h"

typedef unsigned long long u64;

// Key of the k symbols that start at symbol s of `cur`, continuing into `nxt` (packed layout of
// mk_pack.hip: symbols MSB first, BITS each, SPW per word).
template <int BITS, int SPW>
__device__ __forceinline__ u64 window_key(u64 cur, u64 nxt, int s, int k) {
  u64 key = (cur << (BITS * s)) >> (64 - BITS * k);
  const int over = s + k - SPW;  // symbols taken from the next word
  if (over > 0) key |= nxt >> (64 - BITS * over);
  return key;
}

// 64 bad-bits for symbols p0 .. p0+63.
__device__ __forceinline__ u64 bad_window(const u64* __restrict__ bad, size_t p0) {
  const size_t bi = p0 >> 6;
  const int bo = (int)(p0 & 63);
  const u64 b0 = bad[bi];
  if (bo == 0) return b0;
  return (b0 >> bo) | (bad[bi + 1] << (64 - bo));
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP row shifts and row broadcasts -- vector instructions only.
// (A scan written with __shfl_up is six DEPENDENT ds_bpermute round trips through the LDS, ~120 clocks each, plus the
// address arithmetic and selects around them: tools/valu_probe.hip.)  Lanes that a step has no source for add 0.
__device__ __forceinline__ unsigned mk_wave_scan_incl(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);  // row_shr:8: every row of 16 is scanned
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
  return v;
}
// sum over the wave, in every lane
__device__ __forceinline__ unsigned mk_wave_sum(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)mk_wave_scan_incl(v), 63); }
// the value lane 63 holds, in every lane (v_readlane: no trip through the LDS)
__device__ __forceinline__ unsigned mk_wave_last(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }

__device__ __forceinline__ void wave_add(u64* target, u64 mine) {
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(target, mine);
}

// One global add per workgroup (adds to one address are serialised by the L2, ~4 ns each: per-wave
// adds from thousands of waves cost more than the kernels they end). Every thread must call it.
__device__ __forceinline__ void block_add(u64* target, u64 mine) {
  __shared__ unsigned long long s_block_sum;
  if (threadIdx.x == 0) s_block_sum = 0;
  __syncthreads();
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_block_sum, (unsigned long long)mine);
  __syncthreads();
  if (threadIdx.x == 0 && s_block_sum) atomicAdd(target, (u64)s_block_sum);
}


// Reverse complement of a packed nucleotide key (k symbols, 2 bits each, A0 C1 G2 T3: the
// complement of code c is 3-c, i.e. ~c). Bit-reverse the complemented word, put the two bits
// of every symbol back in order, and shift the k symbols down.
__device__ __forceinline__ u64 mk_revcomp2(u64 key, int k) {
  u64 x = __brevll(~key);
  x = ((x & 0xAAAAAAAAAAAAAAAAull) >> 1) | ((x & 0x5555555555555555ull) << 1);
  return x >> (64 - 2 * k);
}
__device__ __forceinline__ u64 mk_canon2(u64 key, int k, bool canonical) {
  if (!canonical) return key;
  const u64 rc = mk_revcomp2(key, k);
  return rc < key ? rc : key;
}
