// mk_parse.hip -- FASTA record parsing on the GPU.
//
// Replaces the reference's per-line Python loop (lib/mercat2_kmers.py:49-69):
//   line = line.strip(); header iff it starts with '>'; otherwise seq += line.replace("*","")
// Text-mode iteration there splits lines at '\n', '\r\n' and lone '\r' (universal newlines);
// str.strip() removes the ASCII blanks \t \n \v \f \r ' ' and \x1c..\x1f at both ends only.
//
// Here the same language is a 3-state byte transducer
//     L0  line start, only blanks so far      H  inside a header line      S  inside a sequence line
// run as a parallel scan over state maps (each thread summarises 32 bytes as a map
// {entry state -> exit state, bytes emitted}); maps compose associatively, so
//   pass 1  per-tile map            (mk_parse_tiles)
//   pass 2  exclusive scan of tiles (mk_parse_scan, one workgroup)
//   pass 3  re-walk every tile with its entry state/offset and emit  (mk_parse_emit)
// Output "seq": kept characters in order, with one MK_SEP byte wherever a header line starts,
// so that no window can span two records.  A blank inside a sequence line is kept iff a
// non-blank follows before the line ends (strip() only trims the ends) -- decided by a short
// forward look-ahead, which only blanks pay for.
#include "mk_common.h"

#define PT 256   // threads per workgroup
#define PB 32    // bytes per thread
#define PTILE (PT * PB)

enum { Q_L0 = 0, Q_H = 1, Q_S = 2 };

struct PMap {
  unsigned st;  // exit state for entry L0 | H<<2 | S<<4
  unsigned c0, c1, c2;
};
struct PMap64 {
  unsigned st;
  unsigned pad;
  unsigned long long c0, c1, c2;
};

__device__ __forceinline__ PMap pm_identity() { return PMap{0x24u, 0u, 0u, 0u}; }
__device__ __forceinline__ unsigned pm_st(unsigned st, unsigned q) { return (st >> (2 * q)) & 3u; }
__device__ __forceinline__ unsigned pm_cnt(const PMap& m, unsigned q) { return q == 0 ? m.c0 : (q == 1 ? m.c1 : m.c2); }
__device__ __forceinline__ PMap pm_compose(const PMap& a, const PMap& b) {  // a, then b
  unsigned s0 = pm_st(a.st, 0), s1 = pm_st(a.st, 1), s2 = pm_st(a.st, 2);
  PMap r;
  r.st = pm_st(b.st, s0) | (pm_st(b.st, s1) << 2) | (pm_st(b.st, s2) << 4);
  r.c0 = a.c0 + pm_cnt(b, s0);
  r.c1 = a.c1 + pm_cnt(b, s1);
  r.c2 = a.c2 + pm_cnt(b, s2);
  return r;
}

__device__ __forceinline__ bool is_nl(unsigned c) { return c == 10u || c == 13u; }
__device__ __forceinline__ bool is_blank(unsigned c) {  // str.strip() set minus the newlines
  return c == 32u || c == 9u || c == 11u || c == 12u || (c >= 28u && c <= 31u);
}

// Does a non-blank character follow position i before the line ends?
__device__ __forceinline__ bool blank_is_inner(const uint8_t* __restrict__ raw, size_t i, size_t n) {
  size_t j = i + 1;
  while (j < n && is_blank(raw[j])) ++j;
  return j < n && !is_nl(raw[j]);
}

// One transducer step from state q on byte ch. Returns the emitted byte in `out` (valid iff true).
__device__ __forceinline__ bool pstep(unsigned& q, unsigned ch, const uint8_t* __restrict__ raw, size_t i, size_t n,
                                      unsigned& out) {
  if (is_nl(ch)) { q = Q_L0; return false; }
  if (q == Q_H) return false;
  if (q == Q_L0) {
    if (is_blank(ch)) return false;
    if (ch == '>') { q = Q_H; out = MK_SEP; return true; }
    q = Q_S;
    out = ch;
    return ch != '*';
  }
  // Q_S
  out = ch;
  if (is_blank(ch)) return blank_is_inner(raw, i, n);
  return ch != '*';
}

// Summarise this thread's PB bytes as a map (all three entry states advanced together).
__device__ __forceinline__ PMap thread_map(const uint8_t* __restrict__ raw, size_t base, size_t n, const unsigned char* b) {
  unsigned q0 = Q_L0, q1 = Q_H, q2 = Q_S, c0 = 0, c1 = 0, c2 = 0;
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    size_t i = base + j;
    if (i >= n) break;
    unsigned ch = b[j], out;
    if (is_nl(ch)) { q0 = q1 = q2 = Q_L0; continue; }
    if (q0 == q1 && q1 == q2) {  // converged (the common case after the first newline)
      unsigned e = pstep(q0, ch, raw, i, n, out) ? 1u : 0u;
      q1 = q2 = q0;
      c0 += e; c1 += e; c2 += e;
    } else {
      c0 += pstep(q0, ch, raw, i, n, out) ? 1u : 0u;
      c1 += pstep(q1, ch, raw, i, n, out) ? 1u : 0u;
      c2 += pstep(q2, ch, raw, i, n, out) ? 1u : 0u;
    }
  }
  return PMap{q0 | (q1 << 2) | (q2 << 4), c0, c1, c2};
}

__device__ __forceinline__ void load_bytes(const uint8_t* __restrict__ raw, size_t base, size_t n, unsigned char* b) {
  if (base + PB <= n) {
    const uint4* p = reinterpret_cast<const uint4*>(raw + base);  // raw is 16-byte aligned, base % 32 == 0
    uint4 v0 = p[0], v1 = p[1];
    *reinterpret_cast<uint4*>(b) = v0;
    *reinterpret_cast<uint4*>(b + 16) = v1;
  } else {
    for (int j = 0; j < PB; ++j) b[j] = (base + j < n) ? raw[base + j] : 10;
  }
}

__device__ __forceinline__ PMap shfl_up_map(const PMap& m, int d) {
  PMap r;
  r.st = __shfl_up(m.st, d);
  r.c0 = __shfl_up(m.c0, d);
  r.c1 = __shfl_up(m.c1, d);
  r.c2 = __shfl_up(m.c2, d);
  return r;
}

// Inclusive scan of the workgroup's thread maps. wave_tot[] is LDS scratch (PT/64 entries).
// Returns the inclusive map of this thread; *excl is the map of everything before it.
__device__ __forceinline__ PMap block_scan(PMap mine, PMap* wave_tot, PMap* excl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PMap inc = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    PMap up = shfl_up_map(inc, d);
    if (lane >= d) inc = pm_compose(up, inc);
  }
  if (lane == 63) wave_tot[wave] = inc;
  __syncthreads();
  PMap pre = pm_identity();
  for (int w = 0; w < wave; ++w) pre = pm_compose(pre, wave_tot[w]);
  PMap up1 = shfl_up_map(inc, 1);
  PMap ex = (lane == 0) ? pre : pm_compose(pre, up1);
  *excl = ex;
  return pm_compose(pre, inc);
}

__global__ __launch_bounds__(PT) void mk_parse_tiles(const uint8_t* __restrict__ raw, size_t n, PMap* __restrict__ tile_maps) {
  __shared__ PMap wave_tot[PT / 64];
  const size_t base = (size_t)blockIdx.x * PTILE + (size_t)threadIdx.x * PB;
  unsigned char b[PB];
  load_bytes(raw, base, n, b);
  PMap m = (base < n) ? thread_map(raw, base, n, b) : pm_identity();
  PMap ex;
  PMap inc = block_scan(m, wave_tot, &ex);
  if (threadIdx.x == PT - 1) tile_maps[blockIdx.x] = inc;
}

// One workgroup: thread t owns tiles [t*per, (t+1)*per); exclusive scan of maps with 64-bit counts.
// Writes per tile {entry state, output offset}; the chunk starts in L0 at offset 0.
struct TileEntry {
  unsigned long long off;
  unsigned st;
  unsigned pad;
};

__global__ __launch_bounds__(1024) void mk_parse_scan(const PMap* __restrict__ tile_maps, size_t ntiles,
                                                      TileEntry* __restrict__ entries, MkChunkInfo* __restrict__ info) {
  __shared__ unsigned s_st[1024];
  __shared__ unsigned long long s_c[1024][3];
  const size_t per = (ntiles + 1023) / 1024;
  const size_t lo = (size_t)threadIdx.x * per, hi = (lo + per < ntiles) ? lo + per : ntiles;
  unsigned st = 0x24u;
  unsigned long long c[3] = {0, 0, 0};
  for (size_t t = lo; t < hi; ++t) {
    PMap b = tile_maps[t];
    unsigned s0 = pm_st(st, 0), s1 = pm_st(st, 1), s2 = pm_st(st, 2);
    st = pm_st(b.st, s0) | (pm_st(b.st, s1) << 2) | (pm_st(b.st, s2) << 4);
    c[0] += pm_cnt(b, s0);
    c[1] += pm_cnt(b, s1);
    c[2] += pm_cnt(b, s2);
  }
  s_st[threadIdx.x] = st;
  s_c[threadIdx.x][0] = c[0];
  s_c[threadIdx.x][1] = c[1];
  s_c[threadIdx.x][2] = c[2];
  __syncthreads();
  // The entry of the whole chunk is L0, so only the L0 row of each prefix is needed: walk it.
  __shared__ unsigned e_st[1025];
  __shared__ unsigned long long e_off[1025];
  if (threadIdx.x == 0) {
    unsigned q = Q_L0;
    unsigned long long off = 0;
    for (int t = 0; t < 1024; ++t) {
      e_st[t] = q;
      e_off[t] = off;
      off += s_c[t][q];
      q = pm_st(s_st[t], q);
    }
    e_st[1024] = q;
    e_off[1024] = off;
    info->seq_len = off;
  }
  __syncthreads();
  unsigned q = e_st[threadIdx.x];
  unsigned long long off = e_off[threadIdx.x];
  for (size_t t = lo; t < hi; ++t) {
    entries[t].off = off;
    entries[t].st = q;
    PMap b = tile_maps[t];
    off += pm_cnt(b, q);
    q = pm_st(b.st, q);
  }
}

__global__ __launch_bounds__(PT) void mk_parse_emit(const uint8_t* __restrict__ raw, size_t n,
                                                    const TileEntry* __restrict__ entries, uint8_t* __restrict__ seq,
                                                    MkChunkInfo* __restrict__ info) {
  __shared__ PMap wave_tot[PT / 64];
  __shared__ unsigned s_sym, s_hi;
  if (threadIdx.x == 0) { s_sym = 0; s_hi = 0; }
  const size_t base = (size_t)blockIdx.x * PTILE + (size_t)threadIdx.x * PB;
  unsigned char b[PB];
  load_bytes(raw, base, n, b);
  PMap m = (base < n) ? thread_map(raw, base, n, b) : pm_identity();
  PMap ex;
  block_scan(m, wave_tot, &ex);
  const TileEntry te = entries[blockIdx.x];
  unsigned q = pm_st(ex.st, te.st);
  unsigned long long off = te.off + pm_cnt(ex, te.st);
  unsigned syms = 0, hi = 0;
  if (base < n) {
#pragma unroll 4
    for (int j = 0; j < PB; ++j) {
      size_t i = base + j;
      if (i >= n) break;
      unsigned ch = b[j], out;
      if (pstep(q, ch, raw, i, n, out)) {
        seq[off++] = (uint8_t)out;
        syms += (out != MK_SEP);
        hi += (out != MK_SEP) & (ch >> 7);  // kept characters only: header bytes never enter a k-mer
      }
    }
  }
  // workgroup totals -> one atomic each
  for (int d = 32; d > 0; d >>= 1) {
    syms += __shfl_down(syms, d);
    hi += __shfl_down(hi, d);
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s_sym, syms);
    atomicAdd(&s_hi, hi);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_sym) atomicAdd(&info->symbols, (unsigned long long)s_sym);
    if (s_hi) atomicAdd(&info->non_ascii, (unsigned long long)s_hi);
  }
}

int mk_launch_parse(mk_ctx* c, const uint8_t* d_raw, size_t n) {
  const size_t ntiles = (n + PTILE - 1) / PTILE;
  MkChunkInfo* info = (MkChunkInfo*)c->info.p;
  if (ntiles == 0) return MK_OK;  // info already zeroed: seq_len = 0
  const size_t maps_bytes = (ntiles * sizeof(PMap) + 15) & ~(size_t)15;
  int rc = mk_buf_reserve(c, c->tile_maps, maps_bytes + ntiles * sizeof(TileEntry));
  if (rc) return rc;
  PMap* maps = (PMap*)c->tile_maps.p;
  TileEntry* entries = (TileEntry*)((char*)c->tile_maps.p + maps_bytes);
  mk_prof_begin(c, MK_K_PARSE);
  hipLaunchKernelGGL(mk_parse_tiles, dim3((unsigned)ntiles), dim3(PT), 0, c->stream, d_raw, n, maps);
  hipLaunchKernelGGL(mk_parse_scan, dim3(1), dim3(1024), 0, c->stream, maps, ntiles, entries, info);
  hipLaunchKernelGGL(mk_parse_emit, dim3((unsigned)ntiles), dim3(PT), 0, c->stream, d_raw, n, entries,
                     (uint8_t*)c->seq.p, info);
  mk_prof_end(c);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
