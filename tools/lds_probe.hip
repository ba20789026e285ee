// lds_probe.hip -- what the LDS of one CU sustains for the access shapes the count kernels use.
// A probe, not part of the product:  hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o build/lds_probe
// One 1024-thread workgroup per CU; every thread issues ITER x 8 independent operations on LDS
// addresses that are either random (xorshift per lane) or conflict-free (lane-linear).
// Prints lane-operations per clock per CU for each (operation, pattern).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define SLOTS 8192
#define EMPTY 0xFFFFFFFFFFFFFFFFull

__device__ __forceinline__ unsigned xs(unsigned& x) {
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}

enum Op { OP_NONE, OP_RD64, OP_WR64, OP_ADD32, OP_ADD32_RTN, OP_CAS64, OP_ADD64_RTN, OP_ADD64, OP_CAS64_ADD32, OP_RD32,
          OP_MAX64_RTN, OP_RD64_ADD32, OP_CAS32, OP_RD128, OP_NUM };
static const char* kNames[OP_NUM] = {"valu only", "ds_read_b64", "ds_write_b64", "ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b64",
                                     "ds_add_rtn_u64", "ds_add_u64", "cmpst_b64 + add_u32", "ds_read_b32", "ds_max_rtn_u64",
                                     "read_b64 + add_u32", "ds_cmpst_rtn_b32", "ds_read_b128"};

template <int OP, bool RANDOM, int ACTIVE>
__global__ __launch_bounds__(1024) void probe(u64* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(16))) u64 tkey[SLOTS];
  __shared__ __attribute__((aligned(16))) unsigned tcnt[SLOTS];
  for (unsigned i = threadIdx.x; i < SLOTS; i += blockDim.x) { tkey[i] = (OP == OP_CAS64 || OP == OP_CAS64_ADD32) ? (u64)i * 77 : 0; tcnt[i] = 0; }
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  const unsigned lane = threadIdx.x & 63;
  u64 sink = 0;
  const bool on = (int)lane < ACTIVE;
  for (int it = 0; it < iters; ++it) {
    unsigned a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned r = xs(x);
      a[u] = RANDOM ? (r >> 19) : ((threadIdx.x + (unsigned)(it * 8 + u) * 1024u) & (SLOTS - 1));
    }
    if (!on) continue;
    if (OP == OP_NONE) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += a[u];
    } else if (OP == OP_RD64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += tkey[a[u]];
    } else if (OP == OP_RD32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += tcnt[a[u]];
    } else if (OP == OP_RD128) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(&tkey[a[u] & ~1u]); sink += v.x ^ v.y; }
    } else if (OP == OP_WR64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) tkey[a[u]] = x + u;
    } else if (OP == OP_ADD32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) atomicAdd(&tcnt[a[u]], 1u);
    } else if (OP == OP_ADD32_RTN) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += atomicAdd(&tcnt[a[u]], 1u);
    } else if (OP == OP_CAS32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += atomicCAS(&tcnt[a[u]], 0xFFFFFFFFu, x);
    } else if (OP == OP_CAS64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += atomicCAS(&tkey[a[u]], EMPTY, (u64)x);
    } else if (OP == OP_ADD64_RTN) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += atomicAdd(&tkey[a[u]], 1ull);
    } else if (OP == OP_MAX64_RTN) {
#pragma unroll
      for (int u = 0; u < 8; ++u) sink += atomicMax(&tkey[a[u]], (u64)x);
    } else if (OP == OP_ADD64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) atomicAdd(&tkey[a[u]], 1ull);
    } else if (OP == OP_CAS64_ADD32) {
      u64 cur[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = atomicCAS(&tkey[a[u]], EMPTY, (u64)x);
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (cur[u] == (u64)a[u] * 77) atomicAdd(&tcnt[a[u]], 1u);
    } else if (OP == OP_RD64_ADD32) {
      u64 cur[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = tkey[a[u]];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (cur[u] == 0) atomicAdd(&tcnt[a[u]], 1u);
    }
  }
  __syncthreads();
  if (sink == 0x123456789ull || tcnt[threadIdx.x] == 0xdeadbeefu) out[blockIdx.x * 1024 + threadIdx.x] = sink + tkey[threadIdx.x];
}

template <int OP, bool RANDOM, int ACTIVE>
static void run(int ncu, double ghz, u64* d_out, int iters) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((probe<OP, RANDOM, ACTIVE>), dim3(ncu), dim3(1024), 0, 0, d_out, iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((probe<OP, RANDOM, ACTIVE>), dim3(ncu), dim3(1024), 0, 0, d_out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  const double ops = (double)iters * 8 * 16 * ACTIVE;  // lane-operations per CU (16 waves)
  const double clk = ms * 1e-3 * ghz * 1e9;
  const int per = (OP == OP_CAS64_ADD32 || OP == OP_RD64_ADD32) ? 2 : 1;
  printf("%-22s %-8s active=%2d  %8.3f ms  %6.2f lane-ops/clk/CU  (%5.1f clk per wave instruction)\n", kNames[OP], RANDOM ? "random" : "linear",
         ACTIVE, ms, ops / clk, clk / ((double)iters * 8 * 16 * per) * 1.0 * 1.0);
  hipEventDestroy(a);
  hipEventDestroy(b);
}

#define BOTH(OP) run<OP, true, 64>(ncu, ghz, d_out, iters); run<OP, false, 64>(ncu, ghz, d_out, iters);

int main(int argc, char** argv) {
  int iters = argc > 1 ? atoi(argv[1]) : 4096;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  const double ghz = p.clockRate * 1e-6;
  printf("%s: %d CUs, %.2f GHz, LDS %zu KB per workgroup\n", p.name, ncu, ghz, p.sharedMemPerBlock / 1024);
  u64* d_out = nullptr;
  hipMalloc(&d_out, (size_t)ncu * 1024 * 8);
  BOTH(OP_NONE)
  BOTH(OP_RD32)
  BOTH(OP_RD64)
  BOTH(OP_RD128)
  BOTH(OP_WR64)
  BOTH(OP_ADD32)
  BOTH(OP_ADD32_RTN)
  BOTH(OP_CAS32)
  BOTH(OP_CAS64)
  BOTH(OP_ADD64)
  BOTH(OP_ADD64_RTN)
  BOTH(OP_MAX64_RTN)
  BOTH(OP_CAS64_ADD32)
  BOTH(OP_RD64_ADD32)
  run<OP_CAS64, true, 32>(ncu, ghz, d_out, iters);
  run<OP_CAS64, true, 16>(ncu, ghz, d_out, iters);
  run<OP_ADD32, true, 32>(ncu, ghz, d_out, iters);
  run<OP_ADD32, true, 16>(ncu, ghz, d_out, iters);
  run<OP_RD64, true, 16>(ncu, ghz, d_out, iters);
  hipFree(d_out);
  return 0;
}
