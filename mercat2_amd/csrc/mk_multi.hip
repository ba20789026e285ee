// mk_multi.hip -- one process, several GPUs: the merge of the per-GPU count tables.
//
// Replaces the reference's cross-worker merge -- ray.get of every chunk's dict and the dict sum in run_mercat2
// (bin/mercat2.py:121-127), the workers being the Ray tasks of bin/mercat2.py:119-120,336-339 -- for the GPUs of one
// node driven by ONE process.  Chunks are the shard unit (mk_count_file deals chunk i to ctxs[i mod nctx], every chunk
// filtered on its own GPU: the per-chunk min_count rule needs no exchange); the one exchange step is this merge:
//   1. every context groups the rows of its running table by OWNER (owner = key range of the first key word):
//      one histogram pass and one scatter pass over the table, no sort;
//   2. the segments are copied straight to their owners' GPUs with peer copies -- each pair of GPUs has its own xGMI
//      link, so in round s source i sends to owner (i + s) mod n and all links carry rows at once;
//   3. every owner insert-adds its own segment and what it received into its (emptied) running table.
// Rows kept as text (characters outside the alphabet, k > 64: rare) go through the host into ctxs[0].
// The same two device primitives serve one-process-per-GPU callers (mercat2_amd/dist.py over RCCL) through
// mk_bucket_rows_device / mk_import_rows_device.
#include "mk_common.h"
#include "mk_device.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#define MK_MAX_OWNERS 64

namespace {

using Clock = std::chrono::steady_clock;
static double secs(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// ---- the three packed running tables seen as "slot i -> (occupied, key word(s), count)" ------------------------
struct View64 {
  const MkSlot* t;
  static constexpr int W = 1;
  __device__ __forceinline__ bool get(size_t i, u64& a, u64& b, u64& c) const {
    const ulonglong2 s = reinterpret_cast<const ulonglong2*>(t)[i];
    a = s.x; b = 0; c = s.y;
    return s.x != MK_EMPTY && s.y != 0;
  }
};
struct View128 {
  const MkSlot128* t;
  static constexpr int W = 2;
  __device__ __forceinline__ bool get(size_t i, u64& a, u64& b, u64& c) const {
    const ulonglong4 s = reinterpret_cast<const ulonglong4*>(t)[i];
    a = s.x; b = s.y; c = s.z;
    return s.z != 0;
  }
};
struct ViewDense {
  const u64* bins;
  static constexpr int W = 1;
  __device__ __forceinline__ bool get(size_t i, u64& a, u64& b, u64& c) const {
    a = (u64)i; b = 0; c = bins[i];
    return c != 0;
  }
};

__device__ __forceinline__ int owner_of(const u64* __restrict__ s_bounds, int n, u64 key) {
  int o = 0;
  for (int j = 0; j + 1 < n; ++j) o += key >= s_bounds[j] ? 1 : 0;
  return o;
}

// rows per owner
template <class V>
__global__ __launch_bounds__(256) void mk_owner_hist_k(V v, size_t slots, const u64* __restrict__ bounds, int n, u64* __restrict__ hist) {
  __shared__ u64 s_bounds[MK_MAX_OWNERS];
  __shared__ unsigned s_cnt[MK_MAX_OWNERS];
  if (threadIdx.x < MK_MAX_OWNERS) {
    s_cnt[threadIdx.x] = 0;
    s_bounds[threadIdx.x] = (int)threadIdx.x + 1 < n ? bounds[threadIdx.x] : ~0ull;
  }
  __syncthreads();
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    u64 a, b, c;
    if (v.get(i, a, b, c)) atomicAdd(&s_cnt[owner_of(s_bounds, n, a)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < n && s_cnt[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (u64)s_cnt[threadIdx.x]);
}

// rows grouped by owner: every workgroup owns a contiguous slice of the table, counts its rows per owner, reserves
// its part of every owner's segment with one atomic per owner (cursor[j] starts at the segment's first row), and
// places its rows (the slice is L2-hot on the second pass).  Rows are W + 1 words: {key word(s), count}.
template <class V>
__global__ __launch_bounds__(256) void mk_owner_scatter_k(V v, size_t slots, const u64* __restrict__ bounds, int n,
                                                           u64* __restrict__ cursor, u64* __restrict__ out, u64 cap_rows) {
  __shared__ u64 s_bounds[MK_MAX_OWNERS];
  __shared__ unsigned s_cnt[MK_MAX_OWNERS];
  __shared__ u64 s_base[MK_MAX_OWNERS];
  if (threadIdx.x < MK_MAX_OWNERS) {
    s_cnt[threadIdx.x] = 0;
    s_bounds[threadIdx.x] = (int)threadIdx.x + 1 < n ? bounds[threadIdx.x] : ~0ull;
  }
  __syncthreads();
  const size_t per = (slots + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < slots ? lo + per : slots;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    u64 a, b, c;
    if (v.get(i, a, b, c)) atomicAdd(&s_cnt[owner_of(s_bounds, n, a)], 1u);
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (u64)s_cnt[threadIdx.x]) : 0ull;
    s_cnt[threadIdx.x] = 0;
  }
  __syncthreads();
  constexpr int RW = V::W + 1;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    u64 a, b, c;
    if (!v.get(i, a, b, c)) continue;
    const int o = owner_of(s_bounds, n, a);
    const u64 pos = s_base[o] + atomicAdd(&s_cnt[o], 1u);
    if (pos >= cap_rows) continue;  // (cannot happen: the histogram sized the segments; never write past the buffer)
    if (V::W == 1) {
      reinterpret_cast<ulonglong2*>(out)[pos] = make_ulonglong2(a, c);
    } else {
      out[pos * RW] = a;
      out[pos * RW + 1] = b;
      out[pos * RW + 2] = c;
    }
  }
}

// cursor[j] = first row of owner j's segment (exclusive prefix of the histogram): on the device, so that the host
// waits once for histogram, prefix and scatter together
__global__ void mk_owner_prefix_k(const u64* __restrict__ hist, u64* __restrict__ cursor, int n) {
  if (threadIdx.x == 0) {
    u64 at = 0;
    for (int j = 0; j < MK_MAX_OWNERS; ++j) {
      cursor[j] = at;
      if (j < n) at += hist[j];
    }
  }
}

// first key words of the rows in every stride-th slot (the table is hashed: a uniform sample of its rows)
template <class V>
__global__ void mk_sample_keys_k(V v, size_t slots, size_t stride, u64* __restrict__ out, u64 cap, u64* __restrict__ cursor) {
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j * stride < slots; j += (size_t)gridDim.x * blockDim.x) {
    u64 a, b, c;
    if (v.get(j * stride, a, b, c)) {
      const u64 pos = atomicAdd(cursor, 1ull);
      if (pos < cap) out[pos] = a;
    }
  }
}

static unsigned grid_for(size_t items, unsigned per_block, unsigned cap) {
  size_t g = (items + per_block - 1) / per_block;
  if (g > cap) g = cap;
  if (g == 0) g = 1;
  return (unsigned)g;
}

struct TableRef {
  int kind = 0;  // 0 none, 1 one-word, 2 two-word, 3 dense
  const void* p = nullptr;
  size_t slots = 0;
  size_t rows = 0;  // rows the host knows of (dense: unknown, bins)
};
static TableRef table_of(const mk_ctx* c) {
  TableRef t;
  if (c->mode == MK_MODE_HASH64 && c->run_slots) { t.kind = 1; t.p = c->run.p; t.slots = c->run_slots; t.rows = c->run_rows; }
  else if (c->mode == MK_MODE_HASH128 && c->run128_slots) { t.kind = 2; t.p = c->run128.p; t.slots = c->run128_slots; t.rows = c->run128_rows; }
  else if (c->mode == MK_MODE_DENSE) { t.kind = 3; t.p = c->run.p; t.slots = c->run_slots; t.rows = c->run_slots; }
  return t;
}

// meta buffer: bounds[64] | hist[64] | cursor[64]
static int reserve_meta(mk_ctx* c) { return mk_buf_reserve(c, c->xfer_meta, 3 * MK_MAX_OWNERS * sizeof(u64)); }

// counts[j] = rows of owner j; with d_rows the rows themselves, owner after owner.  The one key that lives beside
// the one-word table (the all-T 32-mer) goes to owner side_owner as an ordinary row (the import sets it aside again).
static int bucket_rows(mk_ctx* c, const u64* bounds, int n, u64* d_rows, size_t cap_rows, u64* counts, int side_owner) {
  int rc;
  for (int j = 0; j < n; ++j) counts[j] = 0;
  MK_HIP(hipSetDevice(c->device));
  if ((rc = mk_settle(c)) != MK_OK) return rc;
  const TableRef t = table_of(c);
  const bool side = c->mode == MK_MODE_HASH64 && c->run_side != 0;
  const int rw = mk_words_per_key(c) + 1;
  if (t.kind && t.rows) {
    if ((rc = reserve_meta(c)) != MK_OK) return rc;
    u64* d_bounds = (u64*)c->xfer_meta.p;
    u64* d_hist = d_bounds + MK_MAX_OWNERS;
    u64* d_cursor = d_hist + MK_MAX_OWNERS;
    u64 hb[MK_MAX_OWNERS];
    for (int j = 0; j < MK_MAX_OWNERS; ++j) hb[j] = j + 1 < n ? bounds[j] : ~0ull;
    MK_HIP(hipMemcpyAsync(d_bounds, hb, sizeof hb, hipMemcpyHostToDevice, c->stream));
    MK_HIP(hipMemsetAsync(d_hist, 0, MK_MAX_OWNERS * sizeof(u64), c->stream));
    const unsigned grid = grid_for(t.slots, 256 * 16, 2048);
    if (t.kind == 1) hipLaunchKernelGGL(mk_owner_hist_k<View64>, dim3(grid), dim3(256), 0, c->stream, View64{(const MkSlot*)t.p}, t.slots, (const u64*)d_bounds, n, d_hist);
    else if (t.kind == 2) hipLaunchKernelGGL(mk_owner_hist_k<View128>, dim3(grid), dim3(256), 0, c->stream, View128{(const MkSlot128*)t.p}, t.slots, (const u64*)d_bounds, n, d_hist);
    else hipLaunchKernelGGL(mk_owner_hist_k<ViewDense>, dim3(grid), dim3(256), 0, c->stream, ViewDense{(const u64*)t.p}, t.slots, (const u64*)d_bounds, n, d_hist);
    if (d_rows) {  // (the scatter never writes past cap_rows; whether everything fitted is checked below)
      hipLaunchKernelGGL(mk_owner_prefix_k, dim3(1), dim3(64), 0, c->stream, (const u64*)d_hist, d_cursor, n);
      if (t.kind == 1) hipLaunchKernelGGL(mk_owner_scatter_k<View64>, dim3(grid), dim3(256), 0, c->stream, View64{(const MkSlot*)t.p}, t.slots, (const u64*)d_bounds, n, d_cursor, d_rows, (u64)cap_rows);
      else if (t.kind == 2) hipLaunchKernelGGL(mk_owner_scatter_k<View128>, dim3(grid), dim3(256), 0, c->stream, View128{(const MkSlot128*)t.p}, t.slots, (const u64*)d_bounds, n, d_cursor, d_rows, (u64)cap_rows);
      else hipLaunchKernelGGL(mk_owner_scatter_k<ViewDense>, dim3(grid), dim3(256), 0, c->stream, ViewDense{(const u64*)t.p}, t.slots, (const u64*)d_bounds, n, d_cursor, d_rows, (u64)cap_rows);
    }
    MK_HIP(hipGetLastError());
    u64 hh[MK_MAX_OWNERS];
    MK_HIP(hipMemcpyAsync(hh, d_hist, sizeof hh, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    for (int j = 0; j < n; ++j) counts[j] = hh[j];
  }
  u64 total = side ? 1 : 0;
  for (int j = 0; j < n; ++j) total += counts[j];
  if (d_rows && total > cap_rows) {
    c->err = "rows by owner: buffer of " + std::to_string(cap_rows) + " rows is too small for " + std::to_string(total);
    return MK_ERR_RANGE;
  }
  if (side) {
    // the one key kept beside the table: an ordinary row {all-ones key, count} at the END of the buffer's rows, counted
    // for side_owner.  (With side_owner = the last owner -- the key is the largest there is -- that is the end of its
    // segment; mk_merge_devices gathering to one owner has only that owner.)
    if (d_rows) {
      u64 at = 0;
      for (int j = 0; j < n; ++j) at += counts[j];
      const u64 row[2] = {MK_EMPTY, c->run_side};
      MK_HIP(hipMemcpyAsync(d_rows + at * (u64)rw, row, sizeof row, hipMemcpyHostToDevice, c->stream));
      MK_HIP(hipStreamSynchronize(c->stream));
    }
    counts[side_owner] += 1;
  }
  return MK_OK;
}

// about `want` first key words of the context's rows (for MK_MERGE_BALANCED), every stride-th slot
static int sample_keys(mk_ctx* c, size_t stride, std::vector<u64>& out) {
  int rc;
  MK_HIP(hipSetDevice(c->device));
  if ((rc = mk_settle(c)) != MK_OK) return rc;
  const TableRef t = table_of(c);
  if (!(t.kind == 1 || t.kind == 2) || !t.rows) return MK_OK;
  const size_t cap = 2 * (t.rows / stride) + 1024;
  if ((rc = mk_buf_reserve(c, c->xfer_in, (cap + 8) * sizeof(u64))) != MK_OK) return rc;
  u64* d_cursor = (u64*)c->xfer_in.p;
  u64* d_out = d_cursor + 8;
  MK_HIP(hipMemsetAsync(d_cursor, 0, 8, c->stream));
  const unsigned grid = grid_for(t.slots / stride + 1, 256, 1024);
  if (t.kind == 1) hipLaunchKernelGGL(mk_sample_keys_k<View64>, dim3(grid), dim3(256), 0, c->stream, View64{(const MkSlot*)t.p}, t.slots, stride, d_out, (u64)cap, d_cursor);
  else hipLaunchKernelGGL(mk_sample_keys_k<View128>, dim3(grid), dim3(256), 0, c->stream, View128{(const MkSlot128*)t.p}, t.slots, stride, d_out, (u64)cap, d_cursor);
  MK_HIP(hipGetLastError());
  u64 got = 0;
  MK_HIP(hipMemcpyAsync(&got, d_cursor, 8, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  if (got > cap) got = cap;
  const size_t at = out.size();
  out.resize(at + (size_t)got);
  if (got) {
    MK_HIP(hipMemcpyAsync(out.data() + at, d_out, (size_t)got * 8, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
  }
  return MK_OK;
}

// ---- RCCL as the transport of the exchange (MK_MERGE_RCCL): grouped ncclSend / ncclRecv of the owner segments, straight
// between peers over xGMI -- the collective SURVEY 8(e) names, behind the C ABI.  The library is opened when the flag is
// first used (dlopen: a box without RCCL still loads this library and keeps the peer-copy transport); one communicator
// per distinct device, made once per device list with ncclCommInitAll (one process, all GPUs) and kept.
struct Rccl {
  typedef void* comm_t;
  int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string why;  // why it is not available
  std::vector<std::pair<std::vector<int>, std::vector<comm_t>>> comms;  // per device list
  bool ok() const { return CommInitAll != nullptr; }
  static Rccl& get() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void* h = nullptr;
    const char* env = getenv("MK_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names)
      if (nm && (h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!h) { r.why = std::string("librccl.so not found (") + (dlerror() ? dlerror() : "dlopen failed") + ")"; return r; }
    auto sym = [&](const char* nm) { void* p = dlsym(h, nm); if (!p) r.why = std::string("librccl.so lacks ") + nm; return p; };
    void* f[6] = {sym("ncclCommInitAll"), sym("ncclGroupStart"), sym("ncclGroupEnd"), sym("ncclSend"), sym("ncclRecv"), sym("ncclGetErrorString")};
    for (void* p : f) if (!p) return r;
    r.GroupStart = (int (*)())f[1];
    r.GroupEnd = (int (*)())f[2];
    r.Send = (int (*)(const void*, size_t, int, int, comm_t, hipStream_t))f[3];
    r.Recv = (int (*)(void*, size_t, int, int, comm_t, hipStream_t))f[4];
    r.GetErrorString = (const char* (*)(int))f[5];
    r.CommInitAll = (int (*)(comm_t*, int, const int*))f[0];
    return r;
  }
  // communicators for these devices (rank r = devs[r]); nullptr + why on failure
  const std::vector<comm_t>* comms_for(const std::vector<int>& devs) {
    for (auto& e : comms) if (e.first == devs) return &e.second;
    std::vector<comm_t> cs(devs.size(), nullptr);
    const int rc = CommInitAll(cs.data(), (int)devs.size(), devs.data());
    if (rc != 0) { why = std::string("ncclCommInitAll: ") + GetErrorString(rc); return nullptr; }
    comms.emplace_back(devs, cs);
    return &comms.back().second;
  }
};
#define MK_NCCL_UINT64 5  /* ncclUint64 (rccl.h: ncclDataType_t) */

// direct access between two devices, both ways; 1 = direct (xGMI / PCIe P2P), 0 = hipMemcpyPeer stages the bytes itself
static int enable_peer_pair(int a, int b) {
  if (a == b) return 1;
  int ok = 1;
  const int pair[2][2] = {{a, b}, {b, a}};
  for (auto& p : pair) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, p[0], p[1]) != hipSuccess || !can) { ok = 0; continue; }
    if (hipSetDevice(p[0]) != hipSuccess) { ok = 0; continue; }
    const hipError_t e = hipDeviceEnablePeerAccess(p[1], 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = 0;
    (void)hipGetLastError();  // (already-enabled is not an error to carry along)
  }
  return ok;
}

// f(i) for every context at once (each has host waits of its own: the GPUs work side by side); first error wins.
// The helper threads are kept for the life of the process: a fresh thread pays ~100 us for its first HIP call, which
// is a third of a whole merge at S2 size.
class Helpers {
 public:
  template <class F>
  int run(int n, F&& f) {
    std::vector<int> rcs((size_t)n, MK_OK);
    if (n > 1) {
      std::unique_lock<std::mutex> g(mu_);
      while ((int)threads_.size() < n - 1) threads_.emplace_back([this] { work(); });
      job_ = [&](int i) { rcs[(size_t)i] = f(i); };
      next_ = 1;
      jobs_ = n;
      pending_ = n - 1;
      cv_.notify_all();
    }
    rcs[0] = f(0);
    if (n > 1) {
      std::unique_lock<std::mutex> g(mu_);
      done_.wait(g, [&] { return pending_ == 0; });
      jobs_ = 0;
      job_ = nullptr;
    }
    for (int i = 0; i < n; ++i)
      if (rcs[(size_t)i]) return rcs[(size_t)i];
    return MK_OK;
  }

 private:
  void work() {
    std::unique_lock<std::mutex> g(mu_);
    for (;;) {
      cv_.wait(g, [&] { return next_ < jobs_; });
      const int i = next_++;
      auto job = job_;
      g.unlock();
      job(i);
      g.lock();
      if (--pending_ == 0) done_.notify_all();
    }
  }
  std::mutex mu_;
  std::condition_variable cv_, done_;
  std::vector<std::thread> threads_;  // (never joined: they sleep on cv_ until the process ends)
  std::function<void(int)> job_;
  int next_ = 0, jobs_ = 0, pending_ = 0;
};
static std::mutex g_merge_mu;  // one merge at a time uses the helpers
static Helpers* g_helpers = nullptr;
template <class F>
static int on_every(int n, F&& f) {
  if (!g_helpers) g_helpers = new Helpers();
  return g_helpers->run(n, f);
}

}  // namespace

extern "C" int mk_bucket_rows_device(mk_ctx* c, const uint64_t* bounds, int n, uint64_t* d_rows, size_t cap_rows, uint64_t* counts) {
  if (!c || !counts || n < 1 || (n > 1 && !bounds)) return MK_ERR_ARG;
  if (n > MK_MAX_OWNERS) { c->err = "mk_bucket_rows_device: at most 64 owners"; return MK_ERR_ARG; }
  if (c->in_chunk) { c->err = "mk_bucket_rows_device: a chunk is open"; return MK_ERR_STATE; }
  for (int j = 1; j + 1 < n; ++j)
    if (bounds[j] < bounds[j - 1]) { c->err = "mk_bucket_rows_device: bounds must ascend"; return MK_ERR_ARG; }
  return bucket_rows(c, (const u64*)bounds, n, (u64*)d_rows, cap_rows, (u64*)counts, n - 1);
}

extern "C" int mk_import_rows_device(mk_ctx* c, const uint64_t* d_rows, size_t rows) {
  if (!c) return MK_ERR_ARG;
  if (!rows) return MK_OK;
  if (!d_rows) return MK_ERR_ARG;
  if (c->mode == MK_MODE_BYREF) { c->err = "mk_import_rows_device: context has no packed table"; return MK_ERR_STATE; }
  int rc;
  if ((rc = mk_settle(c)) != MK_OK) return rc;
  MK_HIP(hipSetDevice(c->device));
  MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  if ((rc = mk_grow_run(c, rows)) != MK_OK) return rc;
  if ((rc = mk_launch_import_rows(c, d_rows, rows)) != MK_OK) return rc;
  if ((rc = mk_pull_info(c)) != MK_OK) return rc;
  if (c->mode == MK_MODE_HASH128) c->run128_rows += (size_t)c->h_info->new_rows;
  else if (c->mode == MK_MODE_HASH64) { c->run_rows += (size_t)c->h_info->new_rows; c->run_side += c->h_info->side; }
  return MK_OK;
}

extern "C" int mk_sample_keys(mk_ctx* c, size_t stride, uint64_t* out, size_t cap, size_t* n) {
  if (!c || !n || (cap && !out)) return MK_ERR_ARG;
  if (c->in_chunk) { c->err = "mk_sample_keys: a chunk is open"; return MK_ERR_STATE; }
  std::vector<u64> got;
  int rc = sample_keys(c, stride ? stride : 1, got);
  if (rc) return rc;
  *n = got.size() < cap ? got.size() : cap;
  if (*n) memcpy(out, got.data(), *n * sizeof(u64));
  return MK_OK;
}

// The dense histogram (small k) as one array: out to / in from a caller's DEVICE buffer, so that several GPUs can
// sum their bins with one reduce ("dense bins: ncclAllReduce / ncclReduce of u64 bins").
extern "C" int mk_dense_bins_device(mk_ctx* c, uint64_t* d_bins, size_t nbins, int store) {
  if (!c || !d_bins) return MK_ERR_ARG;
  if (c->mode != MK_MODE_DENSE) { c->err = "mk_dense_bins_device: the context does not count into dense bins"; return MK_ERR_STATE; }
  if (c->in_chunk) { c->err = "mk_dense_bins_device: a chunk is open"; return MK_ERR_STATE; }
  if (nbins != c->run_slots) { c->err = "mk_dense_bins_device: the table has " + std::to_string(c->run_slots) + " bins"; return MK_ERR_RANGE; }
  int rc;
  if ((rc = mk_settle(c)) != MK_OK) return rc;
  MK_HIP(hipSetDevice(c->device));
  if (store) MK_HIP(hipMemcpyAsync(c->run.p, d_bins, nbins * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  else MK_HIP(hipMemcpyAsync(d_bins, c->run.p, nbins * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  return MK_OK;
}

extern "C" int mk_merge_devices(mk_ctx* const* ctxs, int n, int flags, mk_merge_stats_t* st) {
  if (!ctxs || n < 1 || !ctxs[0]) return MK_ERR_ARG;
  mk_ctx* c0 = ctxs[0];
  if (n > MK_MAX_OWNERS) { c0->err = "mk_merge_devices: at most 64 contexts"; return MK_ERR_ARG; }
  for (int i = 0; i < n; ++i) {
    mk_ctx* c = ctxs[i];
    if (!c) { c0->err = "mk_merge_devices: a context is NULL"; return MK_ERR_ARG; }
    if (c->alphabet != c0->alphabet || c->k != c0->k || c->canonical != c0->canonical || c->mode != c0->mode) {
      c0->err = "mk_merge_devices: contexts differ in alphabet, k or canonical mode";
      return MK_ERR_ARG;
    }
    if (c->in_chunk) { c0->err = "mk_merge_devices: a chunk is open"; return MK_ERR_STATE; }
    for (int j = 0; j < i; ++j)
      if (ctxs[j] == c) { c0->err = "mk_merge_devices: the same context twice"; return MK_ERR_ARG; }
  }
  std::lock_guard<std::mutex> merge_lock(g_merge_mu);
  // (the phases below set the calling thread's device context by context: put the caller's back on every way out --
  // the same process may drive torch on another device)
  struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = -1; } }
    ~DeviceGuard() { if (dev >= 0) (void)hipSetDevice(dev); }
  } device_guard;
  const auto t_begin = Clock::now();
  const bool gather = (flags & MK_MERGE_GATHER) != 0;
  const int m = gather ? 1 : n;  // owners
  const int rw = mk_words_per_key(c0) + 1;
  const bool packed = c0->mode != MK_MODE_BYREF;
  int rc;
  mk_merge_stats_t S;
  memset(&S, 0, sizeof S);
  S.contexts = n;
  {
    std::vector<int> devs;
    for (int i = 0; i < n; ++i)
      if (std::find(devs.begin(), devs.end(), ctxs[i]->device) == devs.end()) devs.push_back(ctxs[i]->device);
    S.devices = (int)devs.size();
  }
  auto fail = [&](int i, int code) {
    if (i > 0 && !ctxs[i]->err.empty()) c0->err = ctxs[i]->err;
    return code;
  };

  // ---- owner bounds
  u64 bounds[MK_MAX_OWNERS];
  for (auto& b : bounds) b = ~0ull;
  if (m > 1 && packed) {
    // (first word of a two-word key: nucleotides fill it -- left-aligned; an amino-acid key is a number of 5 k bits)
    const int key_bits = c0->mode == MK_MODE_HASH128 ? (c0->alphabet == MK_ALPHABET_AA5 ? std::max(1, 5 * c0->k - 64) : 64) : c0->bits * c0->k;
    if ((rc = mk_owner_bounds(key_bits, m, (uint64_t*)bounds)) != MK_OK) return rc;
    if ((flags & MK_MERGE_BALANCED) && (c0->mode == MK_MODE_HASH64 || c0->mode == MK_MODE_HASH128)) {
      size_t total = 0;
      for (int i = 0; i < n; ++i) {
        if ((rc = mk_settle(ctxs[i])) != MK_OK) return fail(i, rc);
        total += table_of(ctxs[i]).rows;
      }
      // ~512 keys per owner in all, each context at the same rate: owners within a few per cent of each other, and the
      // host sorts a few thousand values only
      const size_t stride = std::max<size_t>(1, total / (512 * (size_t)m));
      std::vector<std::vector<u64>> parts((size_t)n);
      std::atomic<int> bad{-1};
      rc = on_every(n, [&](int i) { int r = sample_keys(ctxs[i], stride, parts[i]); if (r) bad = i; return r; });
      if (rc) return fail(bad, rc);
      std::vector<u64> all;
      for (auto& p : parts) all.insert(all.end(), p.begin(), p.end());
      if (all.size() >= (size_t)(8 * m)) {  // (fewer: keep the equal ranges)
        std::sort(all.begin(), all.end());
        for (int j = 1; j < m; ++j) bounds[j - 1] = all[all.size() * (size_t)j / (size_t)m];
      }
    }
  }

  // ---- phase A: text rows to the host, packed rows grouped by owner (the gathering context keeps its own table)
  std::vector<std::vector<u64>> counts((size_t)n, std::vector<u64>((size_t)m, 0));
  std::vector<std::vector<uint8_t>> ex_k((size_t)n);
  std::vector<std::vector<uint64_t>> ex_c((size_t)n);
  std::vector<u64> rows_before((size_t)n, 0);
  const double s_bounds = secs(t_begin);
  const auto t_a = Clock::now();
  {
    std::atomic<int> bad{-1};
    rc = on_every(n, [&](int i) {
      mk_ctx* c = ctxs[i];
      int r;
      if ((r = mk_settle(c)) != MK_OK) { bad = i; return r; }
      if (table_of(c).kind == 3) {  // dense bins: the rows are the bins that are not zero (tiny: counted on the host)
        size_t r0 = 0;
        if ((r = mk_export_size(c, &r0)) != MK_OK) { bad = i; return r; }
        rows_before[i] = r0 - c->run_ref_rows;
      } else {
        rows_before[i] = table_of(c).rows;
      }
      if (gather && i == 0) return MK_OK;
      if (c->run_ref_rows) {
        size_t nr = 0;
        if ((r = mk_export_exotic(c, nullptr, nullptr, 0, &nr)) != MK_OK) { bad = i; return r; }
        ex_k[i].resize(nr * (size_t)c->k + 1);
        ex_c[i].resize(nr + 1);
        if ((r = mk_export_exotic(c, ex_k[i].data(), ex_c[i].data(), nr, &nr)) != MK_OK) { bad = i; return r; }
        ex_c[i].resize(nr);
      }
      if (!packed) return MK_OK;
      const TableRef t = table_of(c);
      const size_t cap = (t.kind == 3 ? t.slots : t.rows) + 1;  // (the rows the host knows of, + the one key kept beside the table)
      if (hipSetDevice(c->device) != hipSuccess) { bad = i; c->err = "hipSetDevice failed"; return MK_ERR_HIP; }
      if ((r = mk_buf_reserve(c, c->xfer_out, cap * rw * sizeof(u64) + 64)) != MK_OK) { bad = i; return r; }
      if ((r = bucket_rows(c, bounds, m, (u64*)c->xfer_out.p, cap, counts[i].data(), m - 1)) != MK_OK) { bad = i; return r; }
      return MK_OK;
    });
    if (rc) return fail(bad, rc);
  }
  S.s_bucket = secs(t_a);
  for (int i = 0; i < n; ++i) S.rows_in += rows_before[i] + (ctxs[i]->mode == MK_MODE_HASH64 && ctxs[i]->run_side ? 1 : 0);

  // ---- phase B: empty the tables that are rebuilt, room for what arrives
  std::vector<u64> recv((size_t)m, 0), own((size_t)n, 0);
  std::vector<std::vector<u64>> recv_off((size_t)n, std::vector<u64>((size_t)m, 0));  // row offset of source i in owner j's receive buffer
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i) {
      if (i == j && !(gather && i == 0)) { own[i] = counts[i][j]; continue; }
      recv_off[i][j] = recv[j];
      recv[j] += counts[i][j];
    }
  {
    std::atomic<int> bad{-1};
    rc = on_every(n, [&](int i) {
      mk_ctx* c = ctxs[i];
      int r;
      // (an owner's table is sized for what it will hold: its own segment and what arrives, duplicates included)
      if (!(gather && i == 0) && (r = (i < m ? mk_reset_for(c, own[i] + recv[i] + 1) : mk_reset(c))) != MK_OK) { bad = i; return r; }
      if (i >= m) return MK_OK;
      if (hipSetDevice(c->device) != hipSuccess) { bad = i; c->err = "hipSetDevice failed"; return MK_ERR_HIP; }
      if (recv[i] && (r = mk_buf_reserve(c, c->xfer_in, (size_t)recv[i] * rw * sizeof(u64) + 64)) != MK_OK) { bad = i; return r; }
      if ((own[i] + recv[i]) && (r = mk_grow_run(c, (size_t)(own[i] + recv[i]))) != MK_OK) { bad = i; return r; }
      return MK_OK;
    });
    if (rc) return fail(bad, rc);
  }

  // ---- phase C: the segments travel (source i -> owner (i + s) mod n in round s: every pair of GPUs at once)
  const double s_prepare = secs(t_begin) - s_bounds - S.s_bucket;
  const auto t_c = Clock::now();
  std::vector<hipEvent_t> sent((size_t)n, nullptr);
  auto drop_events = [&] { for (auto& e : sent) if (e) { (void)hipEventDestroy(e); e = nullptr; } };
  std::vector<char> sends((size_t)n, 0);
  const std::vector<Rccl::comm_t>* comms = nullptr;
  std::vector<int> dev_rank((size_t)n, 0);  // rank of a context's device among the distinct devices
  if (flags & MK_MERGE_RCCL) {
    std::vector<int> devs;
    for (int i = 0; i < n; ++i) {
      auto it = std::find(devs.begin(), devs.end(), ctxs[i]->device);
      dev_rank[i] = (int)(it - devs.begin());
      if (it == devs.end()) devs.push_back(ctxs[i]->device);
    }
    Rccl& R = Rccl::get();
    if (R.ok()) comms = R.comms_for(devs);
    if (!comms) { c0->err = "mk_merge_devices(MK_MERGE_RCCL): " + R.why; return MK_ERR_UNSUPPORTED; }
    S.rccl = 1;
  }
  if (comms) {
    // Round s: source i sends its segment for owner (i + s) mod n -- with one context per GPU every rank has one send and
    // one receive per round, the usual all-to-all schedule, all of a round's in ONE group: ncclSend on the source's
    // stream, ncclRecv on the owner's (the owner's import, queued on that stream behind it, needs no event).  Segments
    // between two contexts of ONE device (the one-GPU rehearsal; the product sums a GPU's contexts before it merges
    // GPUs) are a send of that device's rank to itself, each in a group of its own: several self-sends in one group
    // lost rows on RCCL 2.x here.
    Rccl& R = Rccl::get();
    int nrc = 0;
    auto post = [&](int i, int j) {
      mk_ctx* src = ctxs[i];
      mk_ctx* dst = ctxs[j];
      u64 seg = 0;
      for (int q = 0; q < j; ++q) seg += counts[i][q];
      const size_t words = (size_t)counts[i][j] * rw;
      const u64* from = (const u64*)src->xfer_out.p + seg * (u64)rw;
      u64* to = (u64*)dst->xfer_in.p + recv_off[i][j] * (u64)rw;
      if (hipSetDevice(src->device) != hipSuccess) return -1;
      int r = R.Send(from, words, MK_NCCL_UINT64, dev_rank[j], (*comms)[(size_t)dev_rank[i]], src->stream);
      if (r) return r;
      if (hipSetDevice(dst->device) != hipSuccess) return -1;
      r = R.Recv(to, words, MK_NCCL_UINT64, dev_rank[i], (*comms)[(size_t)dev_rank[j]], dst->stream);
      sends[i] = 1;
      S.rows_moved += counts[i][j];
      S.bytes_moved += words * sizeof(u64);
      return r;
    };
    for (int s = 1; s < n && nrc == 0; ++s) {
      bool any = false;
      for (int i = 0; i < n; ++i) {
        const int j = (i + s) % n;
        any = any || (j < m && counts[i][j] && ctxs[i]->device != ctxs[j]->device);
      }
      if (any) {
        nrc = R.GroupStart();
        for (int i = 0; i < n && nrc == 0; ++i) {
          const int j = (i + s) % n;
          if (j < m && counts[i][j] && ctxs[i]->device != ctxs[j]->device) nrc = post(i, j);
        }
        const int erc = R.GroupEnd();
        if (nrc == 0) nrc = erc;
      }
      for (int i = 0; i < n && nrc == 0; ++i) {
        const int j = (i + s) % n;
        if (!(j < m && counts[i][j] && ctxs[i]->device == ctxs[j]->device)) continue;
        nrc = R.GroupStart();
        if (nrc == 0) nrc = post(i, j);
        const int erc = R.GroupEnd();
        if (nrc == 0) nrc = erc;
      }
    }
    if (nrc != 0) { c0->err = std::string("RCCL exchange of table rows: ") + (nrc > 0 ? R.GetErrorString(nrc) : "hipSetDevice failed"); rc = MK_ERR_HIP; }
  } else
  {
    std::vector<std::pair<int, int>> pairs_seen;
    for (int s = 1; s < n && rc == MK_OK; ++s)
      for (int i = 0; i < n && rc == MK_OK; ++i) {
        const int j = (i + s) % n;
        if (j >= m || !counts[i][j]) continue;
        mk_ctx* src = ctxs[i];
        mk_ctx* dst = ctxs[j];
        u64 seg = 0;
        for (int q = 0; q < j; ++q) seg += counts[i][q];
        const size_t bytes = (size_t)counts[i][j] * rw * sizeof(u64);
        const u64* from = (const u64*)src->xfer_out.p + seg * (u64)rw;
        u64* to = (u64*)dst->xfer_in.p + recv_off[i][j] * (u64)rw;
        hipError_t e;
        if (src->device != dst->device) {
          const std::pair<int, int> pr(std::min(src->device, dst->device), std::max(src->device, dst->device));
          if (std::find(pairs_seen.begin(), pairs_seen.end(), pr) == pairs_seen.end()) {
            pairs_seen.push_back(pr);
            S.peer_direct += enable_peer_pair(pr.first, pr.second);
          }
        }
        if ((e = hipSetDevice(src->device)) == hipSuccess)
          e = src->device == dst->device ? hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, src->stream)
                                         : hipMemcpyPeerAsync(to, dst->device, from, src->device, bytes, src->stream);
        if (e != hipSuccess) { c0->err = std::string("peer copy of table rows: ") + hipGetErrorString(e); rc = MK_ERR_HIP; break; }
        sends[i] = 1;
        S.rows_moved += counts[i][j];
        S.bytes_moved += bytes;
      }
    for (int i = 0; i < n && rc == MK_OK; ++i) {
      if (!sends[i]) continue;
      hipError_t e = hipSetDevice(ctxs[i]->device);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&sent[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(sent[i], ctxs[i]->stream);
      if (e != hipSuccess) { c0->err = std::string("event after the peer copies: ") + hipGetErrorString(e); rc = MK_ERR_HIP; }
    }
    // an owner's stream goes on only when every source that sent to it has
    for (int j = 0; j < m && rc == MK_OK; ++j) {
      if (hipSetDevice(ctxs[j]->device) != hipSuccess) { rc = MK_ERR_HIP; c0->err = "hipSetDevice failed"; break; }
      for (int i = 0; i < n && rc == MK_OK; ++i) {
        if (i == j || !counts[i][j] || !sent[i]) continue;
        if (hipStreamWaitEvent(ctxs[j]->stream, sent[i], 0) != hipSuccess) { rc = MK_ERR_HIP; c0->err = "hipStreamWaitEvent failed"; }
      }
    }
  }
  if (rc) {
    for (int i = 0; i < n; ++i) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
    drop_events();
    return rc;
  }

  // ---- phase D: every owner insert-adds its own segment and what arrived; text rows into ctxs[0]
  {
    std::atomic<int> bad{-1};
    rc = on_every(n, [&](int j) {
      mk_ctx* c = ctxs[j];
      int r;
      if (hipSetDevice(c->device) != hipSuccess) { bad = j; c->err = "hipSetDevice failed"; return MK_ERR_HIP; }
      if (j < m && packed && (own[j] + recv[j])) {
        if (hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream) != hipSuccess) { bad = j; c->err = "hipMemsetAsync failed"; return MK_ERR_HIP; }
        if (own[j]) {
          u64 seg = 0;
          for (int q = 0; q < j; ++q) seg += counts[j][q];
          if ((r = mk_launch_import_rows(c, (const uint64_t*)c->xfer_out.p + seg * (u64)rw, (size_t)own[j])) != MK_OK) { bad = j; return r; }
        }
        if (recv[j] && (r = mk_launch_import_rows(c, (const uint64_t*)c->xfer_in.p, (size_t)recv[j])) != MK_OK) { bad = j; return r; }
        if ((r = mk_pull_info(c)) != MK_OK) { bad = j; return r; }
        if (c->mode == MK_MODE_HASH128) c->run128_rows += (size_t)c->h_info->new_rows;
        else if (c->mode == MK_MODE_HASH64) { c->run_rows += (size_t)c->h_info->new_rows; c->run_side += c->h_info->side; }
      } else {
        if (hipStreamSynchronize(c->stream) != hipSuccess) { bad = j; c->err = "hipStreamSynchronize failed"; return MK_ERR_HIP; }
      }
      return MK_OK;
    });
    if (rc) {
      for (int i = 0; i < n; ++i) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
      drop_events();
      return fail(bad, rc);
    }
  }
  // (the sources' copies are done: every owner has waited for them; a source that only sent waits here)
  for (int i = 0; i < n; ++i) { (void)hipSetDevice(ctxs[i]->device); (void)hipStreamSynchronize(ctxs[i]->stream); }
  drop_events();
  S.s_copy = secs(t_c);
  const auto t_d = Clock::now();
  for (int i = 0; i < n; ++i)
    if (!ex_c[i].empty() && (rc = mk_import_exotic(c0, ex_k[i].data(), ex_c[i].data(), ex_c[i].size())) != MK_OK) return rc;
  S.s_import = secs(t_d);
  for (int i = 0; i < n; ++i) {
    size_t r = 0;
    if ((rc = mk_export_size(ctxs[i], &r)) != MK_OK) return fail(i, rc);
    S.rows_out += r;
    S.max_owned = std::max<uint64_t>(S.max_owned, r);
  }
  S.s_total = secs(t_begin);
  if (getenv("MK_VERBOSE"))
    fprintf(stderr, "[mk] merge_devices: bounds %.0f us, bucket %.0f, reset+reserve %.0f, copies+import %.0f, text rows %.0f, total %.0f us\n",
            s_bounds * 1e6, S.s_bucket * 1e6, s_prepare * 1e6, S.s_copy * 1e6, S.s_import * 1e6, S.s_total * 1e6);
  if (st) *st = S;
  return MK_OK;
}
