// h2d_probe.hip -- what the host-to-device path gives: 1.5 GiB out of pinned memory in 4 MiB pieces (what mk_count_file's
// ring does), on 1, 2, 4 streams at once, and one 1.5 GiB copy; hipHostMalloc'ed and hipHostRegister'ed memory.
//   hipcc --offload-arch=gfx950 -O3 tools/h2d_probe.hip -o build/h2d_probe && build/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const size_t total = (size_t)1536 << 20, piece = (size_t)4 << 20;
  char* dev; CK(hipMalloc(&dev, total));
  char* pin; CK(hipHostMalloc(&pin, total, hipHostMallocDefault));
  memset(pin, 1, total);
  char* reg = (char*)aligned_alloc(4096, total); memset(reg, 2, total);
  CK(hipHostRegister(reg, total, hipHostRegisterDefault));
  hipStream_t st[8]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto run = [&](const char* name, char* src, int ns, size_t pc) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      for (int i = 0; i < ns; ++i) (void)hipStreamSynchronize(st[i]);
      auto t0 = std::chrono::steady_clock::now();
      size_t i = 0;
      for (size_t off = 0; off < total; off += pc, ++i) (void)hipMemcpyAsync(dev + off, src + off, pc < total - off ? pc : total - off, hipMemcpyHostToDevice, st[i % ns]);
      for (int k = 0; k < ns; ++k) (void)hipStreamSynchronize(st[k]);
      double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (s < best) best = s;
    }
    printf("%-34s streams=%d piece=%4zu MiB: %6.1f ms  %5.1f GB/s\n", name, ns, pc >> 20, best * 1e3, total / best / 1e9);
    return 0;
  };
  for (int ns : {1, 2, 4, 8}) run("hipHostMalloc", pin, ns, piece);
  for (int ns : {1, 2, 4}) run("hipHostMalloc", pin, ns, (size_t)32 << 20);
  run("hipHostMalloc one copy", pin, 1, total);
  for (int ns : {1, 2, 4}) run("hipHostRegister", reg, ns, piece);
  // device to host, the TSV's way
  auto back = [&](int ns, size_t pc) {
    auto t0 = std::chrono::steady_clock::now();
    size_t i = 0;
    for (size_t off = 0; off < total; off += pc, ++i) (void)hipMemcpyAsync(pin + off, dev + off, pc, hipMemcpyDeviceToHost, st[i % ns]);
    for (int k = 0; k < ns; ++k) (void)hipStreamSynchronize(st[k]);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("D2H pinned streams=%d piece=%zu MiB: %6.1f ms  %5.1f GB/s\n", ns, pc >> 20, s * 1e3, total / s / 1e9);
  };
  back(1, piece); back(2, piece); back(1, (size_t)32 << 20);
  std::vector<char> pageable((size_t)64 << 20);
  auto t0 = std::chrono::steady_clock::now();
  (void)hipMemcpy(pageable.data(), dev, pageable.size(), hipMemcpyDeviceToHost);
  double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("D2H pageable 64 MiB: %6.1f ms  %5.1f GB/s\n", s * 1e3, pageable.size() / s / 1e9);
  return 0;
}
