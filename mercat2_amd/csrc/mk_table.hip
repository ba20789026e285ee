// mk_table.hip -- chunk-table -> running-table kernels.
//
// Replaces (a) the per-file min_count filter of find_kmers (lib/mercat2_kmers.py:73-76: keep a
// key iff its count IN THIS FILE/CHUNK is >= min_count) and (b) run_mercat2's merge of the
// surviving dicts (bin/mercat2.py:121-127: kmers[k] += v).  The filter is applied per chunk,
// before the merge -- there is no post-merge filter in the reference (SURVEY.md trap T2).
#include "mk_common.h"
#include "mk_device.h"
#include "mk_skmer_dev.h"

typedef unsigned long long u64;
#define REF_POS_BITS 40
#define REF_POS_MASK ((1ull << REF_POS_BITS) - 1)

static size_t div_up(size_t a, size_t b) { return (a + b - 1) / b; }
static unsigned grid_for(size_t items, unsigned per_block = 256, unsigned cap = 1u << 20) {
  size_t g = div_up(items, per_block);
  if (g > cap) g = cap;
  if (g == 0) g = 1;
  return (unsigned)g;
}

// ----------------------------------------------------------------------------------- clear
__global__ void mk_clear_slots_k(MkSlot* __restrict__ t, size_t slots) {
  const ulonglong2 e = make_ulonglong2(MK_EMPTY, 0ull);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x)
    reinterpret_cast<ulonglong2*>(t)[i] = e;
}

int mk_launch_clear_slots(mk_ctx* c, MkSlot* t, size_t slots) {
  if (!slots) return MK_OK;
  hipLaunchKernelGGL(mk_clear_slots_k, dim3(grid_for(slots, 256, 16384)), dim3(256), 0, c->stream, t, slots);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------------------ survivors
__global__ void mk_count_survivors_k(const MkSlot* __restrict__ t, size_t slots, u64 min_count, u64* __restrict__ out) {
  u64 mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(t)[i];
    mine += (s.x != MK_EMPTY && s.y >= min_count) ? 1 : 0;
  }
  block_add(out, mine);
}

int mk_launch_count_survivors(mk_ctx* c, uint64_t min_count) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  if (c->mode == MK_MODE_HASH64 && c->ctab_slots)
    hipLaunchKernelGGL(mk_count_survivors_k, dim3(grid_for(c->ctab_slots, 256, 8192)), dim3(256), 0, c->stream,
                       (const MkSlot*)c->ctab.p, c->ctab_slots, (u64)min_count, &info->survivors);
  if (c->rtab_chunk_slots)
    hipLaunchKernelGGL(mk_count_survivors_k, dim3(grid_for(c->rtab_chunk_slots, 256, 8192)), dim3(256), 0, c->stream,
                       (const MkSlot*)c->rtab_chunk.p, c->rtab_chunk_slots, (u64)min_count, &info->survivors_ref);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------- running table: hash64
// Where a key's probe sequence starts.  Plain: a mixing hash of the key.  BUCKET-MAJOR (nucleotide keys of the
// super-k-mer path, tables of at least 2^13 slots; experiment switch MK_BUCKET_MAJOR): the table is 2^13 slices, one per
// minimizer bucket, and a key starts in the slice of ITS bucket -- the bucket its windows were filed under when they
// were counted, recomputed here from the key alone (the smallest 11-mer of the key in the partition's order) -- at a
// hashed offset.  The survivors a chunk's count kernel emits for one bucket then all land in one slice of the table
// (a few tens of KB) instead of anywhere in it.  Probing is linear over the whole table either way: a full slice spills
// into the next.
struct RunAddr {
  u64 mask;      // slots - 1 (a power of two)
  int k;         // > 0: bucket-major addressing for k-mers of this length
  int canon;
};
__device__ __forceinline__ unsigned run_bucket13(u64 key, int k, bool canon) {
  unsigned best = ~0u;
  for (int q = 0; q + SK_M <= k; ++q) {
    const unsigned mm = (unsigned)(key >> (2 * (k - SK_M - q))) & SK_MASK;
    const unsigned o = (sk_order_raw(sk_canon_mmer(mm, canon)) << 10) | (unsigned)q;  // (as sk_analyse: leftmost on ties)
    best = o < best ? o : best;
  }
  const int q = (int)(best & 63u);
  const unsigned mm = (unsigned)(key >> (2 * (k - SK_M - q))) & SK_MASK;
  return sk_bucket(sk_canon_mmer(mm, canon), 13);
}
__device__ __forceinline__ u64 run_home(const RunAddr& a, u64 key) {
  const u64 h = mk_mix64(key);
  if (a.k <= 0 || a.mask < 8191) return h & a.mask;
  const u64 per = (a.mask + 1) >> 13;  // slots per slice
  return (u64)run_bucket13(key, a.k, a.canon != 0) * per + (h & (per - 1));
}

static RunAddr run_addr(const mk_ctx* c, size_t slots) {
  RunAddr a;
  a.mask = (u64)(slots - 1);
  a.k = c->run_bucket_major ? c->k : 0;
  a.canon = c->canonical;
  return a;
}

// Returns true when the key was new to the table.
__device__ __forceinline__ bool upsert64(MkSlot* __restrict__ table, const RunAddr& addr, u64 key, u64 add) {
  const u64 mask = addr.mask;
  u64 slot = run_home(addr, key);
  for (;;) {
    u64 cur = table[slot].key;
    bool fresh = false;
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&table[slot].key, MK_EMPTY, key);
      if (cur == MK_EMPTY) { cur = key; fresh = true; }
    }
    if (cur == key) {
      atomicAdd(&table[slot].cnt, add);
      return fresh;
    }
    slot = (slot + 1) & mask;
  }
}

__global__ void mk_accumulate64_k(const MkSlot* __restrict__ from, size_t slots, u64 min_count, MkSlot* __restrict__ run,
                                  RunAddr run_mask, u64* __restrict__ new_rows) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(from)[i];
    if (s.x != MK_EMPTY && s.y >= min_count && s.y != 0) fresh += upsert64(run, run_mask, s.x, s.y) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}

// The all-ones key (32 x 'T') cannot live in the table (it is the free-slot mark): wherever it stands in
// the rows -- rows received from several peers are a concatenation of sorted segments -- its count goes
// to *side (the context keeps that one key beside the table).
__global__ void mk_import_pairs_k(const u64* __restrict__ keys, const u64* __restrict__ cnts, size_t rows,
                                  MkSlot* __restrict__ run, RunAddr run_mask, u64* __restrict__ new_rows, u64* __restrict__ side) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const u64 key = keys[i], cnt = cnts[i];
    if (!cnt) continue;
    if (key == MK_EMPTY) atomicAdd(side, cnt);
    else fresh += upsert64(run, run_mask, key, cnt) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}

// The same for keys that are DISTINCT within the launch and that no other launch adds to at the same time -- the
// survivors of one chunk: every key sits in exactly one bucket and is emitted once.  Then only the CLAIM of a free slot
// races (two new keys may want it: compare-and-swap); the count of a slot that holds the key is this lane's alone and
// is read and written with plain accesses -- one 16-byte load and one 8-byte store per key instead of a load and an
// atomic add that the L2 has to serialise (canonical S2: 2.2 M survivors per chunk; the merge was a third of the step).
__device__ __forceinline__ bool upsert64_distinct(MkSlot* __restrict__ table, const RunAddr& addr, u64 key, u64 add) {
  const u64 mask = addr.mask;
  u64 slot = run_home(addr, key);
  for (;;) {
    const ulonglong2 s = *reinterpret_cast<const ulonglong2*>(&table[slot]);
    u64 cur = s.x;
    if (cur == key) {
      table[slot].cnt = s.y + add;
      return false;
    }
    if (cur == MK_EMPTY) {
      cur = atomicCAS(&table[slot].key, MK_EMPTY, key);
      if (cur == MK_EMPTY) {
        table[slot].cnt = add;  // (the slot was cleared: its count is zero, and it is this key's from now on)
        return true;
      }
      if (cur == key) {  // (cannot happen for distinct keys; kept exact anyway)
        atomicAdd(&table[slot].cnt, add);
        return false;
      }
    }
    slot = (slot + 1) & mask;
  }
}

// Survivors laid out per bucket: bucket b holds nsurv[b] pairs from kstart[b] on. One wave per bucket.
// ATOMIC: the counts are added with atomics -- for a table that other contexts' count kernels upsert into at the same
// time (mk_share_table); otherwise the plain-store form above.
template <bool ATOMIC>
__global__ void mk_import_regions_k(const u64* __restrict__ keys, const u64* __restrict__ cnts, const u64* __restrict__ kstart,
                                    const u64* __restrict__ nsurv, size_t p1, MkSlot* __restrict__ run, RunAddr run_mask,
                                    u64* __restrict__ new_rows) {
  u64 fresh = 0;
  const int lane = threadIdx.x & 63;
  for (size_t b = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < p1; b += (size_t)gridDim.x * (blockDim.x >> 6)) {
    const u64 base = kstart[b], n = nsurv[b];
    if (ATOMIC) { for (u64 i = lane; i < n; i += 64) fresh += upsert64(run, run_mask, keys[base + i], cnts[base + i]) ? 1 : 0; }
    else { for (u64 i = lane; i < n; i += 64) fresh += upsert64_distinct(run, run_mask, keys[base + i], cnts[base + i]) ? 1 : 0; }
  }
  block_add(new_rows, fresh);
}

// Pairs that are DISTINCT within the launch (a chunk's survivors from the direct-index and the 8-byte-key paths: every
// key is counted in one bucket) into a table nobody else writes meanwhile: the plain-store upsert above.
__global__ void mk_import_pairs_distinct_k(const u64* __restrict__ keys, const u64* __restrict__ cnts, size_t rows,
                                           MkSlot* __restrict__ run, RunAddr run_mask, u64* __restrict__ new_rows, u64* __restrict__ side) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const u64 key = keys[i], cnt = cnts[i];
    if (!cnt) continue;
    if (key == MK_EMPTY) atomicAdd(side, cnt);
    else fresh += upsert64_distinct(run, run_mask, key, cnt) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}

int mk_launch_import_regions(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_counts, const uint64_t* kstart,
                             const uint64_t* nsurv, size_t p1, size_t survivors) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  // (few workgroups, each wave walking several buckets: every workgroup ends with one add to the same counter, and
  // adds to one address are serialised by the L2 at ~4 ns each)
  // Many survivors per bucket (-c 1, canonical keys: a bucket keeps a thousand keys, each upsert is two dependent
  // round trips to HBM): every bucket gets its own wave at once -- the kernel is bound by requests in flight.
  const unsigned cap = survivors > 64 * p1 ? 4096u : 512u;
  static const bool always_atomic = getenv("MK_IMPORT_ATOMIC") != nullptr;  // (A/B: the add as an atomic, as before round 3)
  if (always_atomic || !c->sharers.empty())
    hipLaunchKernelGGL(mk_import_regions_k<true>, dim3(grid_for(p1 * 64, 256, cap)), dim3(256), 0, c->stream, (const u64*)d_keys,
                       (const u64*)d_counts, (const u64*)kstart, (const u64*)nsurv, p1, (MkSlot*)c->run.p,
                       run_addr(c, c->run_slots), &info->new_rows);
  else
    hipLaunchKernelGGL(mk_import_regions_k<false>, dim3(grid_for(p1 * 64, 256, cap)), dim3(256), 0, c->stream, (const u64*)d_keys,
                       (const u64*)d_counts, (const u64*)kstart, (const u64*)nsurv, p1, (MkSlot*)c->run.p,
                       run_addr(c, c->run_slots), &info->new_rows);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

__global__ void mk_import_bins_k(const u64* __restrict__ keys, const u64* __restrict__ cnts, size_t rows,
                                 u64* __restrict__ bins, size_t nbins) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x)
    if (keys[i] < nbins) atomicAdd(&bins[keys[i]], cnts[i]);
}

// ------------------------------------------------------------- running table: two-word keys
// Insert-add of one {hi, lo} key (protocol: MkSlot128 in mk_common.h).  A lane that claims a slot writes the
// key words and publishes the count inside the loop iteration in which it won, so lanes of the same wave that
// meet MK_LOCK128 and look again cannot starve it.  Returns true when the key was new.
__device__ __forceinline__ u64 home128(u64 hi, u64 lo, u64 mask) { return mk_mix64(hi ^ mk_mix64(lo + MK_POLY_B)) & mask; }

// Ordering without cache maintenance: every access to a slot's words is an agent-scope atomic (sc1: performed at the
// device's point of coherence, per-location coherent across the XCDs' L2s by themselves).  The claimer's two key stores
// are write-through; `s_waitcnt vmcnt(0)` holds the publishing store back until both have been acknowledged.  A C++
// release store / acquire fence at agent scope would do the same job with `buffer_wbl2 sc1` / `buffer_inv sc1` -- a
// write-back and an invalidation of the XCD's whole L2 -- per NEW ROW and per probe: measured on 2.1 M new rows
// (protein 13-mers, tools/aa128_probe.py) 5.3 ms against 0.6 ms for the merge kernel.
// The fast form leans on two gfx9-family facts: stores are counted in vmcnt (gfx10+ counts them in vscnt, which this wait
// would not cover) and sc1 accesses are served at the device's point of coherence.  Any other target takes the C++ form.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx942__) && !defined(__gfx950__) && !defined(MK_UPSERT128_FENCES)
#define MK_UPSERT128_FENCES 1
#endif
#ifdef MK_UPSERT128_FENCES  // (A/B builds: the C++ memory-order form)
#define MK_PUBLISH128(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT)
#define MK_ACQUIRE128() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define MK_PUBLISH128(p, v)                                                        \
  do {                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                               \
    __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      \
  } while (0)
#define MK_ACQUIRE128() asm volatile("" ::: "memory")
#endif
__device__ __forceinline__ bool upsert128(MkSlot128* __restrict__ t, u64 mask, u64 hi, u64 lo, u64 add) {
  u64 slot = home128(hi, lo, mask);
  bool done = false, fresh = false;
  while (!done) {
    u64 st = __hip_atomic_load(&t[slot].cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (st == 0) {
      st = atomicCAS(&t[slot].cnt, 0ull, MK_LOCK128);
      if (st == 0) {
        __hip_atomic_store(&t[slot].hi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&t[slot].lo, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        MK_PUBLISH128(&t[slot].cnt, add);
        done = true;
        fresh = true;
      }
    }
    if (!done && st != MK_LOCK128) {  // a published slot: its key words are final
      MK_ACQUIRE128();
      const u64 h2 = __hip_atomic_load(&t[slot].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u64 l2 = __hip_atomic_load(&t[slot].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (h2 == hi && l2 == lo) {
        atomicAdd(&t[slot].cnt, add);
        done = true;
      } else {
        slot = (slot + 1) & mask;
      }
    }
  }
  return fresh;
}

// Survivors of the partitioned 33..64-mer path: {hi, lo, count} per bucket region. One wave per bucket.
__global__ void mk_import128_regions_k(const u64* __restrict__ hi, const u64* __restrict__ lo, const u64* __restrict__ cnts,
                                       const u64* __restrict__ kstart, const u64* __restrict__ nsurv, size_t p1,
                                       MkSlot128* __restrict__ run, u64 run_mask, u64* __restrict__ new_rows) {
  u64 fresh = 0;
  const int lane = threadIdx.x & 63;
  for (size_t b = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); b < p1; b += (size_t)gridDim.x * (blockDim.x >> 6)) {
    const u64 base = kstart[b], n = nsurv[b];
    for (u64 i = lane; i < n; i += 64) fresh += upsert128(run, run_mask, hi[base + i], lo[base + i], cnts[base + i]) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}

int mk_launch_import128_regions(mk_ctx* c, const uint64_t* hi, const uint64_t* lo, const uint64_t* cnts, const uint64_t* kstart,
                                const uint64_t* nsurv, size_t p1) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  hipLaunchKernelGGL(mk_import128_regions_k, dim3(grid_for(p1 * 64, 256, 512)), dim3(256), 0, c->stream, (const u64*)hi,
                     (const u64*)lo, (const u64*)cnts, (const u64*)kstart, (const u64*)nsurv, p1, (MkSlot128*)c->run128.p,
                     (u64)(c->run128_slots - 1), &info->new_rows);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// Rows {hi, lo} interleaved + counts (another context's or rank's table; the same key may come more than once).
__global__ void mk_import128_pairs_k(const u64* __restrict__ keys2, const u64* __restrict__ cnts, size_t rows,
                                     MkSlot128* __restrict__ run, u64 run_mask, u64* __restrict__ new_rows) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x)
    if (cnts[i]) fresh += upsert128(run, run_mask, keys2[2 * i], keys2[2 * i + 1], cnts[i]) ? 1 : 0;
  block_add(new_rows, fresh);
}

int mk_launch_import128_pairs(mk_ctx* c, const uint64_t* d_keys2, const uint64_t* d_counts, size_t rows) {
  if (!rows) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  hipLaunchKernelGGL(mk_import128_pairs_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, (const u64*)d_keys2,
                     (const u64*)d_counts, rows, (MkSlot128*)c->run128.p, (u64)(c->run128_slots - 1), &info->new_rows);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// Re-insert after growth: rows are distinct and final -> claim the first free slot.
__global__ void mk_rehash128_k(const MkSlot128* __restrict__ from, size_t slots, MkSlot128* __restrict__ to, u64 to_mask) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong4 s = reinterpret_cast<const ulonglong4*>(from)[i];
    if (s.z == 0) continue;
    u64 slot = home128(s.x, s.y, to_mask);
    for (;;) {
      if (atomicCAS(&to[slot].cnt, 0ull, s.z) == 0ull) {
        to[slot].hi = s.x;
        to[slot].lo = s.y;
        break;
      }
      slot = (slot + 1) & to_mask;
    }
  }
}

int mk_launch_rehash128(mk_ctx* c, const MkSlot128* from, size_t from_slots, MkSlot128* to, size_t to_slots) {
  if (!from_slots) return MK_OK;
  hipLaunchKernelGGL(mk_rehash128_k, dim3(grid_for(from_slots, 256, 8192)), dim3(256), 0, c->stream, from, from_slots, to,
                     (u64)(to_slots - 1));
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// Occupied slots -> {hi, lo, count} in arbitrary order; *cursor counts them (wave-aggregated cursor).
__global__ __launch_bounds__(256) void mk_compact128_k(const MkSlot128* __restrict__ t, size_t slots, u64* __restrict__ hi,
                                                       u64* __restrict__ lo, u64* __restrict__ cnts, size_t cap,
                                                       u64* __restrict__ cursor) {
  const int lane = threadIdx.x & 63;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t rounds = (slots + stride - 1) / stride;
  for (size_t r = 0; r < rounds; ++r) {
    const size_t i = r * stride + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    ulonglong4 s = make_ulonglong4(0, 0, 0, 0);
    if (i < slots) s = reinterpret_cast<const ulonglong4*>(t)[i];
    const bool keep = s.z != 0;
    const u64 m = __ballot(keep);
    if (m) {
      u64 at = 0;
      if (lane == 0) at = atomicAdd(cursor, (u64)__popcll(m));
      const u64 pos = __shfl(at, 0) + __popcll(m & ((1ull << lane) - 1));
      if (keep && pos < cap) { hi[pos] = s.x; lo[pos] = s.y; cnts[pos] = s.z; }
    }
  }
}

int mk_launch_compact128(mk_ctx* c, const MkSlot128* t, size_t slots, uint64_t* hi, uint64_t* lo, uint64_t* cnts, size_t cap,
                         uint64_t* d_cursor) {
  if (!slots) return MK_OK;
  hipLaunchKernelGGL(mk_compact128_k, dim3(grid_for(slots, 256, 4096)), dim3(256), 0, c->stream, t, slots, (u64*)hi, (u64*)lo,
                     (u64*)cnts, cap, (u64*)d_cursor);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------- running table: dense
__global__ void mk_accumulate_dense_k(u64* __restrict__ chunk, size_t nbins, u64 min_count, u64* __restrict__ run) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbins; i += (size_t)gridDim.x * blockDim.x) {
    u64 v = chunk[i];
    if (v >= min_count && v) run[i] += v;
    chunk[i] = 0;
  }
}

// --------------------------------------------------------------- running table: by reference
// Running slot key = (tag << 40) | arena row; the k bytes of row r live at arena[r*k .. r*k+k).
// Arena bytes written in this launch are read by other workgroups of the same launch, so they
// are read with agent-scope loads (never from a stale L1 line).
__device__ __forceinline__ unsigned ld_u8_agent(const uint8_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The k-mer text comes through an accessor get(i) -> byte i, so the same code serves text held in
// memory (chunk stream, staging buffer) and text decoded on the fly from packed keys.
struct BytesAt {
  const uint8_t* p;
  __device__ __forceinline__ unsigned operator()(int i) const { return p[i]; }
};
struct Key128Text {  // 2-bit packed, left-aligned {hi, lo}: base i of the k-mer
  u64 hi, lo;
  __device__ __forceinline__ unsigned operator()(int i) const {
    const unsigned code = (unsigned)((i < 32 ? hi >> (62 - 2 * i) : lo >> (62 - 2 * (i - 32))) & 3u);
    return (unsigned)"ACGT"[code];
  }
};

template <class Get>
__device__ __forceinline__ u64 poly_hash_of(const Get& get, int k) {
  u64 h = 0;
  for (int i = 0; i < k; ++i) h = h * MK_POLY_B + get(i);
  return mk_mix64(h);
}

__device__ __forceinline__ u64 poly_hash(const uint8_t* __restrict__ s, int k) { return poly_hash_of(BytesAt{s}, k); }

template <class Get>
__device__ __forceinline__ bool upsert_ref_of(MkSlot* __restrict__ run, u64 mask, uint8_t* __restrict__ arena, const Get& get,
                                              int k, u64 add, u64 arena_base, u64* __restrict__ new_rows) {
  const u64 h = poly_hash_of(get, k);
  const u64 tag = (h >> 41) << REF_POS_BITS;
  u64 slot = h & mask;
  u64 my_row = MK_EMPTY;
  for (;;) {
    u64 cur = __hip_atomic_load(&run[slot].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == MK_EMPTY) {
      if (my_row == MK_EMPTY) {
        my_row = arena_base + atomicAdd(new_rows, 1ull);
        uint8_t* dst = arena + my_row * (u64)k;
        for (int i = 0; i < k; ++i)
          __hip_atomic_store(dst + i, (uint8_t)get(i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the row's bytes are write-through stores; they are acknowledged before the slot that names the row is claimed.
        // A __threadfence() here is a write-back and an invalidation of the XCD's L2 per NEW ROW: see upsert128)
#ifdef MK_UPSERT128_FENCES
        __threadfence();
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      }
      cur = atomicCAS(&run[slot].key, MK_EMPTY, tag | my_row);
      if (cur == MK_EMPTY) {
        atomicAdd(&run[slot].cnt, add);
        return true;
      }
    }
    if ((cur & ~REF_POS_MASK) == tag && my_row == MK_EMPTY) {
      const uint8_t* other = arena + (cur & REF_POS_MASK) * (u64)k;
      bool same = true;
      for (int i = 0; i < k && same; ++i) same = ld_u8_agent(other + i) == get(i);
      if (same) {
        atomicAdd(&run[slot].cnt, add);
        return false;
      }
    }
    slot = (slot + 1) & mask;
  }
}

__device__ __forceinline__ bool upsert_ref(MkSlot* __restrict__ run, u64 mask, uint8_t* __restrict__ arena,
                                           const uint8_t* __restrict__ str, int k, u64 add, u64 arena_base,
                                           u64* __restrict__ new_rows) {
  return upsert_ref_of(run, mask, arena, BytesAt{str}, k, add, arena_base, new_rows);
}

// Survivors of the by-reference chunk table -> running by-reference table. Within one launch
// every inserted string is distinct (they come from distinct chunk slots), so once a thread
// has reserved an arena row (my_row) its string is known to be new and it only looks for a
// free slot.
// run128 != nullptr (contexts of nucleotide 33..64-mers): a surviving row whose k bytes are all ACGT is a
// packed two-word key and goes to the packed table (every such key lives there and only there); rows
// holding other characters stay text.
__global__ void mk_accumulate_ref_k(const MkSlot* __restrict__ from, size_t slots, u64 min_count,
                                    const uint8_t* __restrict__ seq, int k, MkSlot* __restrict__ run, u64 run_mask,
                                    uint8_t* __restrict__ arena, u64 arena_base, u64* __restrict__ new_rows,
                                    MkSlot128* __restrict__ run128, u64 run128_mask, u64* __restrict__ new_rows128, int aa) {
  u64 fresh128 = 0;  // (one add per workgroup at the end: a counter every new row adds to is serialised by the L2, ~4 ns a row)
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(from)[i];
    if (s.x != MK_EMPTY && s.y >= min_count && s.y != 0) {
      const uint8_t* str = seq + (s.x & REF_POS_MASK);
      bool packed = run128 != nullptr;
      u64 hi = 0, lo = 0;
      if (packed && aa) {  // amino acids: the key is the number sum(code_j * 32^(k-1-j)), code = letter - 'A' (mk_count.hip)
        unsigned __int128 v = 0;
        for (int j = 0; j < k && packed; ++j) {
          const unsigned ch = str[j];
          if (ch < 'A' || ch > 'Z') packed = false;
          else v = (v << 5) | (unsigned __int128)(ch - 'A');
        }
        hi = (u64)(v >> 64);
        lo = (u64)v;
      } else if (packed) {
        for (int j = 0; j < k && packed; ++j) {
          const unsigned ch = str[j];
          const unsigned code = ch == 'A' ? 0u : ch == 'C' ? 1u : ch == 'G' ? 2u : ch == 'T' ? 3u : 4u;
          if (code > 3u) packed = false;
          else if (j < 32) hi |= (u64)code << (62 - 2 * j);
          else lo |= (u64)code << (62 - 2 * (j - 32));
        }
      }
      if (packed) fresh128 += upsert128(run128, run128_mask, hi, lo, s.y) ? 1 : 0;
      else upsert_ref(run, run_mask, arena, str, k, s.y, arena_base, new_rows);
    }
  }
  if (run128) block_add(new_rows128, fresh128);
}

// Strings from a staging buffer (rows*k bytes), e.g. rows received from another GPU. Distinct
// among themselves as well (they are rows of one table).
__global__ void mk_import_ref_k(const uint8_t* __restrict__ strs, const u64* __restrict__ cnts, size_t rows, int k,
                                MkSlot* __restrict__ run, u64 run_mask, uint8_t* __restrict__ arena, u64 arena_base,
                                u64* __restrict__ new_rows) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x)
    if (cnts[i]) upsert_ref(run, run_mask, arena, strs + i * (size_t)k, k, cnts[i], arena_base, new_rows);
}

// Re-insert after growth: rows are distinct and their bytes are final -> only find a free slot.
__global__ void mk_rehash_ref_k(const MkSlot* __restrict__ from, size_t slots, MkSlot* __restrict__ to, u64 to_mask,
                                const uint8_t* __restrict__ arena, int k) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(from)[i];
    if (s.x == MK_EMPTY) continue;
    const u64 h = poly_hash(arena + (s.x & REF_POS_MASK) * (u64)k, k);
    u64 slot = h & to_mask;
    for (;;) {
      if (atomicCAS(&to[slot].key, MK_EMPTY, s.x) == MK_EMPTY) {
        to[slot].cnt = s.y;
        break;
      }
      slot = (slot + 1) & to_mask;
    }
  }
}

__global__ void mk_rehash64_k(const MkSlot* __restrict__ from, size_t slots, MkSlot* __restrict__ to, RunAddr to_addr) {
  const u64 to_mask = to_addr.mask;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(from)[i];
    if (s.x == MK_EMPTY) continue;
    u64 slot = run_home(to_addr, s.x);
    for (;;) {
      if (atomicCAS(&to[slot].key, MK_EMPTY, s.x) == MK_EMPTY) {
        to[slot].cnt = s.y;
        break;
      }
      slot = (slot + 1) & to_mask;
    }
  }
}

// Rebuild keeping only rows with count >= min_count (the post-merge filter of a single-chunk sample split
// over several ranks); *kept counts them.
__global__ void mk_refilter64_k(const MkSlot* __restrict__ from, size_t slots, MkSlot* __restrict__ to, RunAddr to_addr, u64 min_count,
                                u64* __restrict__ kept) {
  const u64 to_mask = to_addr.mask;
  u64 mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    ulonglong2 s = reinterpret_cast<const ulonglong2*>(from)[i];
    if (s.x == MK_EMPTY || s.y < min_count || s.y == 0) continue;
    ++mine;
    u64 slot = run_home(to_addr, s.x);
    for (;;) {
      if (atomicCAS(&to[slot].key, MK_EMPTY, s.x) == MK_EMPTY) {
        to[slot].cnt = s.y;
        break;
      }
      slot = (slot + 1) & to_mask;
    }
  }
  block_add(kept, mine);
}
__global__ void mk_refilter128_k(const MkSlot128* __restrict__ from, size_t slots, MkSlot128* __restrict__ to, u64 to_mask,
                                 u64 min_count, u64* __restrict__ kept) {
  u64 mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong4 s = reinterpret_cast<const ulonglong4*>(from)[i];
    if (s.z == 0 || s.z < min_count) continue;
    ++mine;
    u64 slot = home128(s.x, s.y, to_mask);
    for (;;) {
      if (atomicCAS(&to[slot].cnt, 0ull, s.z) == 0ull) {
        to[slot].hi = s.x;
        to[slot].lo = s.y;
        break;
      }
      slot = (slot + 1) & to_mask;
    }
  }
  block_add(kept, mine);
}
__global__ void mk_refilter_dense_k(u64* __restrict__ bins, size_t nbins, u64 min_count) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nbins; i += (size_t)gridDim.x * blockDim.x)
    if (bins[i] < min_count) bins[i] = 0;
}
int mk_launch_refilter64(mk_ctx* c, const MkSlot* from, MkSlot* to, size_t slots, uint64_t min_count, uint64_t* d_kept) {
  hipLaunchKernelGGL(mk_refilter64_k, dim3(grid_for(slots, 256, 8192)), dim3(256), 0, c->stream, from, slots, to, run_addr(c, slots),
                     (u64)min_count, (u64*)d_kept);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
int mk_launch_refilter128(mk_ctx* c, const MkSlot128* from, MkSlot128* to, size_t slots, uint64_t min_count, uint64_t* d_kept) {
  hipLaunchKernelGGL(mk_refilter128_k, dim3(grid_for(slots, 256, 8192)), dim3(256), 0, c->stream, from, slots, to, (u64)(slots - 1),
                     (u64)min_count, (u64*)d_kept);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
int mk_launch_refilter_dense(mk_ctx* c, uint64_t* bins, size_t nbins, uint64_t min_count) {
  hipLaunchKernelGGL(mk_refilter_dense_k, dim3(grid_for(nbins)), dim3(256), 0, c->stream, (u64*)bins, nbins, (u64)min_count);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_merge_table64(mk_ctx* c, const MkSlot* from, size_t from_slots) {
  if (!from_slots) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  hipLaunchKernelGGL(mk_accumulate64_k, dim3(grid_for(from_slots, 256, 2048)), dim3(256), 0, c->stream, from, from_slots, (u64)0,
                     (MkSlot*)c->run.p, run_addr(c, c->run_slots), &info->new_rows);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_rehash64(mk_ctx* c, const MkSlot* from, size_t from_slots, MkSlot* to, size_t to_slots) {
  if (!from_slots) return MK_OK;
  hipLaunchKernelGGL(mk_rehash64_k, dim3(grid_for(from_slots, 256, 8192)), dim3(256), 0, c->stream, from, from_slots, to,
                     run_addr(c, to_slots));
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_rehash_ref(mk_ctx* c, const MkSlot* from, size_t from_slots, MkSlot* to, size_t to_slots) {
  if (!from_slots) return MK_OK;
  hipLaunchKernelGGL(mk_rehash_ref_k, dim3(grid_for(from_slots, 256, 8192)), dim3(256), 0, c->stream, from, from_slots,
                     to, (u64)(to_slots - 1), (const uint8_t*)c->arena.p, c->k);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// Filter + merge of the chunk tables into the running tables (capacities already ensured).
int mk_launch_accumulate(mk_ctx* c, uint64_t min_count) {
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  mk_prof_begin(c, MK_K_FILTER);
  if (c->mode == MK_MODE_DENSE) {
    const size_t nbins = (size_t)1 << (c->bits * c->k);
    hipLaunchKernelGGL(mk_accumulate_dense_k, dim3(grid_for(nbins)), dim3(256), 0, c->stream, (u64*)c->ctab.p, nbins,
                       (u64)min_count, (u64*)c->run.p);
  } else if (c->mode == MK_MODE_HASH64 && c->ctab_slots && c->h_info->survivors) {
    hipLaunchKernelGGL(mk_accumulate64_k, dim3(grid_for(c->ctab_slots, 256, 8192)), dim3(256), 0, c->stream,
                       (const MkSlot*)c->ctab.p, c->ctab_slots, (u64)min_count, (MkSlot*)c->run.p,
                       run_addr(c, c->run_slots), &info->new_rows);
  }
  if (c->rtab_chunk_slots && c->h_info->survivors_ref) {
    hipLaunchKernelGGL(mk_accumulate_ref_k, dim3(grid_for(c->rtab_chunk_slots, 256, 8192)), dim3(256), 0, c->stream,
                       (const MkSlot*)c->rtab_chunk.p, c->rtab_chunk_slots, (u64)min_count, (const uint8_t*)c->seq.p,
                       c->k, (MkSlot*)c->run_ref.p, (u64)(c->run_ref_slots - 1), (uint8_t*)c->arena.p,
                       (u64)c->run_ref_rows, &info->new_rows_ref,
                       c->mode == MK_MODE_HASH128 ? (MkSlot128*)c->run128.p : (MkSlot128*)nullptr,
                       (u64)(c->run128_slots ? c->run128_slots - 1 : 0), &info->new_rows, c->alphabet == MK_ALPHABET_AA5 ? 1 : 0);
  }
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_import_pairs(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_counts, size_t rows, bool distinct) {
  if (!rows) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  static const bool always_atomic = getenv("MK_IMPORT_ATOMIC") != nullptr;
  if (distinct && c->mode == MK_MODE_HASH64 && !always_atomic && c->sharers.empty()) {
    hipLaunchKernelGGL(mk_import_pairs_distinct_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, (const u64*)d_keys,
                       (const u64*)d_counts, rows, (MkSlot*)c->run.p, run_addr(c, c->run_slots), &info->new_rows, &info->side);
    MK_HIP(hipGetLastError());
    return MK_OK;
  }
  if (c->mode == MK_MODE_DENSE) {
    const size_t nbins = (size_t)1 << (c->bits * c->k);
    hipLaunchKernelGGL(mk_import_bins_k, dim3(grid_for(rows)), dim3(256), 0, c->stream, (const u64*)d_keys,
                       (const u64*)d_counts, rows, (u64*)c->run.p, nbins);
  } else {
    hipLaunchKernelGGL(mk_import_pairs_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, (const u64*)d_keys,
                       (const u64*)d_counts, rows, (MkSlot*)c->run.p, run_addr(c, c->run_slots), &info->new_rows, &info->side);
  }
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// Interleaved rows {key, count} / {hi, lo, count} / {bin, count}: the layout rows travel in between GPUs
// (mk_multi.hip, mercat2_amd/dist.py), so that a row's words move side by side and are read with one access.
__global__ void mk_import_rows64_k(const ulonglong2* __restrict__ rows2, size_t rows, MkSlot* __restrict__ run, RunAddr run_mask,
                                   u64* __restrict__ new_rows, u64* __restrict__ side) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2 r = rows2[i];
    if (!r.y) continue;
    if (r.x == MK_EMPTY) atomicAdd(side, r.y);
    else fresh += upsert64(run, run_mask, r.x, r.y) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}
__global__ void mk_import_rows128_k(const u64* __restrict__ rows3, size_t rows, MkSlot128* __restrict__ run, u64 run_mask,
                                    u64* __restrict__ new_rows) {
  u64 fresh = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const u64 hi = rows3[3 * i], lo = rows3[3 * i + 1], cnt = rows3[3 * i + 2];
    if (cnt) fresh += upsert128(run, run_mask, hi, lo, cnt) ? 1 : 0;
  }
  block_add(new_rows, fresh);
}
__global__ void mk_import_rows_bins_k(const ulonglong2* __restrict__ rows2, size_t rows, u64* __restrict__ bins, size_t nbins) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (size_t)gridDim.x * blockDim.x) {
    const ulonglong2 r = rows2[i];
    if (r.x < nbins && r.y) atomicAdd(&bins[r.x], r.y);
  }
}

int mk_launch_import_rows(mk_ctx* c, const uint64_t* d_rows, size_t rows) {
  if (!rows) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  if (c->mode == MK_MODE_DENSE) {
    hipLaunchKernelGGL(mk_import_rows_bins_k, dim3(grid_for(rows)), dim3(256), 0, c->stream, (const ulonglong2*)d_rows, rows,
                       (u64*)c->run.p, (size_t)1 << (c->bits * c->k));
  } else if (c->mode == MK_MODE_HASH64) {
    hipLaunchKernelGGL(mk_import_rows64_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, (const ulonglong2*)d_rows,
                       rows, (MkSlot*)c->run.p, run_addr(c, c->run_slots), &info->new_rows, &info->side);
  } else if (c->mode == MK_MODE_HASH128) {
    hipLaunchKernelGGL(mk_import_rows128_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, (const u64*)d_rows, rows,
                       (MkSlot128*)c->run128.p, (u64)(c->run128_slots - 1), &info->new_rows);
  } else {
    c->err = "import of packed rows: the context has no packed table";
    return MK_ERR_STATE;
  }
  MK_HIP(hipGetLastError());
  return MK_OK;
}

int mk_launch_import_ref(mk_ctx* c, const uint8_t* d_kmers, const uint64_t* d_counts, size_t rows) {
  if (!rows) return MK_OK;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  hipLaunchKernelGGL(mk_import_ref_k, dim3(grid_for(rows, 256, 8192)), dim3(256), 0, c->stream, d_kmers,
                     (const u64*)d_counts, rows, c->k, (MkSlot*)c->run_ref.p, (u64)(c->run_ref_slots - 1),
                     (uint8_t*)c->arena.p, (u64)c->run_ref_rows, &info->new_rows_ref);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// --------------------------------------------------------------------------------- compact
// Occupied slots of a table -> (keys, counts) in arbitrary order; *cursor counts them.
// Each workgroup owns a contiguous slice: pass 1 counts its rows, ONE cursor atomic reserves the
// output range, pass 2 (slice is L2-hot) places the rows with a wave-aggregated LDS cursor.
__global__ __launch_bounds__(256) void mk_compact_k(const MkSlot* __restrict__ t, size_t slots, u64* __restrict__ keys,
                                                    u64* __restrict__ cnts, size_t cap, u64* __restrict__ cursor) {
  __shared__ unsigned s_n;
  __shared__ u64 s_base;
  const size_t per = (slots + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < slots ? lo + per : slots;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  unsigned mine = 0;
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    const ulonglong2 s = reinterpret_cast<const ulonglong2*>(t)[i];
    mine += (s.x != MK_EMPTY && s.y != 0) ? 1u : 0u;
  }
  for (int d = 32; d > 0; d >>= 1) mine += __shfl_down(mine, d);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&s_n, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    s_base = s_n ? atomicAdd(cursor, (u64)s_n) : 0ull;
    s_n = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const size_t rounds = (hi > lo ? hi - lo + blockDim.x - 1 : 0) / blockDim.x;
  for (size_t r = 0; r < rounds; ++r) {
    const size_t i = lo + r * blockDim.x + threadIdx.x;
    ulonglong2 s = make_ulonglong2(MK_EMPTY, 0);
    if (i < hi) s = reinterpret_cast<const ulonglong2*>(t)[i];
    const bool keep = s.x != MK_EMPTY && s.y != 0;
    const u64 m = __ballot(keep);
    if (m) {
      unsigned at = 0;
      if (lane == 0) at = atomicAdd(&s_n, (unsigned)__popcll(m));
      const u64 pos = s_base + __shfl(at, 0) + __popcll(m & ((1ull << lane) - 1));
      if (keep && pos < cap) { keys[pos] = s.x; cnts[pos] = s.y; }
    }
  }
}

int mk_launch_compact(mk_ctx* c, const MkSlot* t, size_t slots, uint64_t* d_keys, uint64_t* d_counts, size_t cap,
                      uint64_t* d_cursor) {
  if (!slots) return MK_OK;
  hipLaunchKernelGGL(mk_compact_k, dim3(grid_for(slots, 256, 8192)), dim3(256), 0, c->stream, t, slots, (u64*)d_keys,
                     (u64*)d_counts, cap, (u64*)d_cursor);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// ------------------------------------------------------------------- alpha-diversity moments
// One pass over the running table(s): everything the nine alpha metrics of lib/mercat2_diversity.py:13-53
// need from the count column (out[0] rows, out[1] sum c, out[2..12] rows with count 0..10 (slot 2 unused),
// then as doubles out[13] sum c^2, out[14] sum c ln c).
__device__ __forceinline__ void alpha_take(u64 c, u64& rows, u64& total, double& sq, double& clnc, unsigned* s_freq) {
  if (!c) return;
  rows += 1;
  total += c;
  const double d = (double)c;
  sq += d * d;
  clnc += d * log(d);
  if (c <= 10) atomicAdd(&s_freq[c], 1u);
}
__global__ __launch_bounds__(256) void mk_alpha_k(const MkSlot* __restrict__ slots, size_t nslots, const u64* __restrict__ bins,
                                                  size_t nbins, u64* __restrict__ out) {
  __shared__ unsigned s_freq[11];
  __shared__ unsigned long long s_rows, s_total;
  __shared__ double s_sq, s_clnc;
  if (threadIdx.x < 11) s_freq[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_rows = 0; s_total = 0; s_sq = 0; s_clnc = 0; }
  __syncthreads();
  u64 rows = 0, total = 0;
  double sq = 0, clnc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = first; i < nslots; i += stride) {
    const ulonglong2 s = reinterpret_cast<const ulonglong2*>(slots)[i];
    if (s.x != MK_EMPTY) alpha_take(s.y, rows, total, sq, clnc, s_freq);
  }
  for (size_t i = first; i < nbins; i += stride) alpha_take(bins[i], rows, total, sq, clnc, s_freq);
  for (int d = 32; d > 0; d >>= 1) {
    rows += __shfl_down(rows, d);
    total += __shfl_down(total, d);
    sq += __shfl_down(sq, d);
    clnc += __shfl_down(clnc, d);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_rows, (unsigned long long)rows);
    atomicAdd(&s_total, (unsigned long long)total);
    atomicAdd(&s_sq, sq);
    atomicAdd(&s_clnc, clnc);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_rows) atomicAdd(&out[0], (u64)s_rows);
    if (s_total) atomicAdd(&out[1], (u64)s_total);
    if (s_sq != 0) atomicAdd(reinterpret_cast<double*>(&out[13]), s_sq);
    if (s_clnc != 0) atomicAdd(reinterpret_cast<double*>(&out[14]), s_clnc);
  }
  if (threadIdx.x < 11 && s_freq[threadIdx.x]) atomicAdd(&out[2 + threadIdx.x], (u64)s_freq[threadIdx.x]);
}

__global__ __launch_bounds__(256) void mk_alpha128_k(const MkSlot128* __restrict__ slots, size_t nslots, u64* __restrict__ out) {
  __shared__ unsigned s_freq[11];
  __shared__ unsigned long long s_rows, s_total;
  __shared__ double s_sq, s_clnc;
  if (threadIdx.x < 11) s_freq[threadIdx.x] = 0;
  if (threadIdx.x == 0) { s_rows = 0; s_total = 0; s_sq = 0; s_clnc = 0; }
  __syncthreads();
  u64 rows = 0, total = 0;
  double sq = 0, clnc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nslots; i += (size_t)gridDim.x * blockDim.x)
    alpha_take(slots[i].cnt, rows, total, sq, clnc, s_freq);
  for (int d = 32; d > 0; d >>= 1) {
    rows += __shfl_down(rows, d);
    total += __shfl_down(total, d);
    sq += __shfl_down(sq, d);
    clnc += __shfl_down(clnc, d);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_rows, (unsigned long long)rows);
    atomicAdd(&s_total, (unsigned long long)total);
    atomicAdd(&s_sq, sq);
    atomicAdd(&s_clnc, clnc);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_rows) atomicAdd(&out[0], (u64)s_rows);
    if (s_total) atomicAdd(&out[1], (u64)s_total);
    if (s_sq != 0) atomicAdd(reinterpret_cast<double*>(&out[13]), s_sq);
    if (s_clnc != 0) atomicAdd(reinterpret_cast<double*>(&out[14]), s_clnc);
  }
  if (threadIdx.x < 11 && s_freq[threadIdx.x]) atomicAdd(&out[2 + threadIdx.x], (u64)s_freq[threadIdx.x]);
}

int mk_launch_alpha(mk_ctx* c, u64* d_out) {
  MK_HIP(hipMemsetAsync(d_out, 0, 16 * sizeof(u64), c->stream));
  if (c->mode == MK_MODE_DENSE) {
    hipLaunchKernelGGL(mk_alpha_k, dim3(grid_for(c->run_slots, 256, 1024)), dim3(256), 0, c->stream, (const MkSlot*)nullptr,
                       (size_t)0, (const u64*)c->run.p, c->run_slots, d_out);
  } else if (c->run_slots) {
    hipLaunchKernelGGL(mk_alpha_k, dim3(grid_for(c->run_slots, 256, 1024)), dim3(256), 0, c->stream, (const MkSlot*)c->run.p,
                       c->run_slots, (const u64*)nullptr, (size_t)0, d_out);
  }
  if (c->run_ref_slots)
    hipLaunchKernelGGL(mk_alpha_k, dim3(grid_for(c->run_ref_slots, 256, 1024)), dim3(256), 0, c->stream,
                       (const MkSlot*)c->run_ref.p, c->run_ref_slots, (const u64*)nullptr, (size_t)0, d_out);
  if (c->run128_slots)
    hipLaunchKernelGGL(mk_alpha128_k, dim3(grid_for(c->run128_slots, 256, 1024)), dim3(256), 0, c->stream,
                       (const MkSlot128*)c->run128.p, c->run128_slots, d_out);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
