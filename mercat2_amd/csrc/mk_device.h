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

__device__ __forceinline__ void wave_add(u64* target, u64 mine) {
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(target, mine);
}
