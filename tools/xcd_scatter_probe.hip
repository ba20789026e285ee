// Would bucket regions private to an XCD pay in the scatter?  The product's pattern: 1610 workgroups x 1024 threads, every
// thread reserves one record in 8 of 8192 buckets (a 32-bit add with return on the bucket's cursor) and stores 16 bytes
// at the position it got.  Against: cursors and regions private to the XCD the workgroup runs on (HW_REG_XCC_ID), with
// adds at agent scope and at workgroup scope (no sc1: the add is done in the XCD's own L2).
// Checks that no add is lost (every bucket must end with 1610 records over its 8 copies).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_scatter_probe.hip -o build/xcd_scatter_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;

// -DRANDOM_BUCKETS: a record's bucket is a hash of (workgroup, thread, i) -- Poisson numbers of records per (workgroup, bucket)
// as in the product -- instead of exactly one record per workgroup and bucket
#ifdef RANDOM_BUCKETS
#define BUCKET(t, i) ((((blockIdx.x * 1024u + (t)) * 8u + (i)) * 2654435761u >> 13) & (NB - 1))
#else
#define BUCKET(t, i) (((t) * 8 + (i) * 1031 + blockIdx.x * 77) & (NB - 1))  // (8 different buckets per thread, spread over the cursor lines)
#endif
#define NB 8192
#define CAP 2048      // records a bucket's shared region holds
#define CAPX 512      // records a bucket's region of one XCD holds

// MODE 0: shared cursors, agent scope   1: per-XCD, agent scope   2: per-XCD, workgroup scope;  STORE: also write the records
template <int MODE, bool STORE>
__global__ __launch_bounds__(1024) void probe(unsigned* __restrict__ cur, ulonglong2* __restrict__ rec, u64* __restrict__ sink) {
  const unsigned t = threadIdx.x;
  unsigned x = 0;
  if (MODE != 0) {
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    x &= 7u;
  }
  unsigned r[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned b = BUCKET(t, i);
    if (MODE == 0) r[i] = __hip_atomic_fetch_add(&cur[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) r[i] = __hip_atomic_fetch_add(&cur[x * NB + b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else r[i] = __hip_atomic_fetch_add(&cur[x * NB + b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  u64 acc = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned b = BUCKET(t, i);
    if (STORE) {
      const size_t at = MODE == 0 ? (size_t)b * CAP + (r[i] & (CAP - 1)) : ((size_t)x * NB + b) * CAPX + (r[i] % CAPX);
      rec[at] = make_ulonglong2(((u64)blockIdx.x << 32) | t, r[i]);
    }
    acc += r[i];
  }
  if (acc == 0x123456789ull) sink[0] = acc;
}

int main() {
  const int G = 1610;
  unsigned* cur;
  ulonglong2* rec;
  u64* sink;
  hipMalloc(&cur, 8 * NB * 4);
  hipMalloc(&rec, (size_t)NB * (CAP > 8 * CAPX ? CAP : 8 * CAPX) * 16);
  hipMalloc(&sink, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[6] = {"shared cursors, agent scope (product)", "per-XCD cursors, agent scope", "per-XCD cursors, workgroup scope",
                          "shared + 16-byte stores", "per-XCD, agent scope + stores", "per-XCD, workgroup scope + stores"};
  std::vector<unsigned> h(8 * NB);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 6; ++mode) {
      hipMemset(cur, 0, 8 * NB * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL((probe<0, false>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
        case 1: hipLaunchKernelGGL((probe<1, false>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
        case 2: hipLaunchKernelGGL((probe<2, false>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
        case 3: hipLaunchKernelGGL((probe<0, true>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
        case 4: hipLaunchKernelGGL((probe<1, true>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
        default: hipLaunchKernelGGL((probe<2, true>), dim3(G), dim3(1024), 0, 0, cur, rec, sink); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cur, 8 * NB * 4, hipMemcpyDeviceToHost);
      size_t bad = 0;
      unsigned mx = 0;
      for (int b = 0; b < NB; ++b) {
        unsigned s = 0;
        for (int x = 0; x < 8; ++x) { s += h[x * NB + b]; if (h[x * NB + b] > mx) mx = h[x * NB + b]; }
#ifdef RANDOM_BUCKETS
        (void)s;
#else
        bad += s != (unsigned)G;
#endif
      }
      const double n = (double)G * 1024 * 8;
      printf("rep %d  %-40s %7.1f us  %6.1f G adds/s  buckets with lost adds: %zu  largest copy: %u\n", rep, names[mode], ms * 1e3,
             n / (ms * 1e-3) / 1e9, bad, mx);
    }
  return 0;
}
