// mk_clean.hip -- removeN's effect on the COUNT, on the GPU.
//
// MerCat2 rewrites every nucleotide FASTA before counting (removeN, lib/mercat2_fasta.py:53-119; bin/mercat2.py:239-244):
// records are cut at every run of upper-case 'N' into records of their own, text in front of the first header is
// dropped, -toupper upper-cases the sequence afterwards.  For the k-mer table that rewrite means exactly three things:
// a run of N separates records (no window spans it, none holds it), the leading text is not counted, and -- with
// -toupper -- lower-case letters count as their upper-case forms (a lower-case 'n' is no cut: it becomes an 'N'
// that IS counted, lib/mercat2_fasta.py:92-113).  A context in clean mode (mk_set_clean) counts the RAW file that way,
// sharing the parser's and packer's pass over the text, so that the table does not wait for the host's rewrite and the
// level-9 gzip of <base>_clean.fna.gz (both still produce the file, in the background):
//   1 mk_clean_pre_k   on the raw bytes, in place, before the parser: 'N' -> a marker byte (0x7F: kept by the parser,
//                      outside every alphabet, so the packer flags it), a..z -> A..Z with -toupper; counts '>' bytes and
//                      marker bytes already in the input; finds the first header line
//   2 mk_clean_head_k  the bytes in front of that header become line ends (nothing of them is kept)
//   3 mk_clean_post_k  on the parsed stream: marker -> separator (the by-reference kernel, which takes the windows the
//                      packer flagged, drops windows that hold a separator); N bytes, run starts / ends, G + C counted
// What the GPU cannot reproduce is reported, not guessed (mk_clean_stats_gpu_t.exact = 0; the host layer then counts
// the text the host rewrite produced): a blank inside a sequence line (textwrap drops blanks at its line breaks when
// the record is split), a '>' that does not start a header line (a wrapped line may begin with it: the reference then
// takes it for a header), a 0x7F byte in the input.
#include "mk_common.h"

typedef unsigned long long u64;
#define MK_CLEAN_MARK 0x7Fu

// meta words: [0] first header offset (init: n)  [1] '>' bytes  [2] marker bytes in the input  [3] N bytes
//             [4] N runs  [5] G + C  [6] run starts written  [7] run ends written
__global__ __launch_bounds__(256) void mk_clean_pre_k(uint8_t* __restrict__ raw, size_t n, int toupper, u64* __restrict__ meta) {
  u64 gts = 0, clash = 0, first = ~0ull;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * blockDim.x * 16) {
    const unsigned prev = i ? raw[i - 1] : 10u;
    unsigned p = prev;
    const size_t e = i + 16 < n ? i + 16 : n;
    for (size_t j = i; j < e; ++j) {
      unsigned c = raw[j];
      if (c == '>') {
        ++gts;
        if ((p == 10u || p == 13u) && first == ~0ull) first = j;
      }
      clash += c == MK_CLEAN_MARK;
      p = c;
      if (c == 'N') c = MK_CLEAN_MARK;
      else if (toupper && c >= 'a' && c <= 'z') c -= 32;
      raw[j] = (uint8_t)c;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    gts += __shfl_down(gts, d);
    clash += __shfl_down(clash, d);
    const u64 o = __shfl_down(first, d);
    first = o < first ? o : first;
  }
  if ((threadIdx.x & 63) == 0) {
    if (gts) atomicAdd(&meta[1], gts);
    if (clash) atomicAdd(&meta[2], clash);
    if (first != ~0ull) atomicMin(&meta[0], first);
  }
}

__global__ __launch_bounds__(256) void mk_clean_head_k(uint8_t* __restrict__ raw, size_t n, u64* __restrict__ meta) {
  const u64 first = meta[0] < (u64)n ? meta[0] : (u64)n;
  u64 odd = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < first; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned ch = raw[i];
    // a blank in the dropped text may be the indentation of a header line that str.strip() would find ("  >name"):
    // not taken for one here, so the chunk is reported as not reproducible rather than cut short
    odd += ch <= 0x20 && ch != 10 && ch != 13;
    raw[i] = 10;
  }
  if (odd) atomicAdd(&meta[2], odd);
}

__global__ __launch_bounds__(256) void mk_clean_post_k(uint8_t* __restrict__ seq, const MkChunkInfo* __restrict__ info,
                                                        u64* __restrict__ meta, u64* __restrict__ run_starts,
                                                        u64* __restrict__ run_ends, u64 cap) {
  const size_t n = info->seq_len;
  u64 nn = 0, runs = 0, gc = 0;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * blockDim.x * 16) {
    unsigned p = i ? seq[i - 1] : 10u;  // (nobody rewrites the stream during this pass: mk_clean_mark_k does, afterwards)
    const size_t e = i + 16 < n ? i + 16 : n;
    for (size_t j = i; j < e; ++j) {
      const unsigned c = seq[j];
      if (c == MK_CLEAN_MARK) {
        ++nn;
        if (p != MK_CLEAN_MARK) {
          const u64 at = atomicAdd(&meta[6], 1ull);
          if (at < cap) run_starts[at] = j;
          ++runs;
        }
        const unsigned nx = j + 1 < n ? seq[j + 1] : 10u;
        if (nx != MK_CLEAN_MARK) {
          const u64 at = atomicAdd(&meta[7], 1ull);
          if (at < cap) run_ends[at] = j + 1;
        }
      }
      gc += (c == 'G') | (c == 'C');
      p = c;
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    nn += __shfl_down(nn, d);
    runs += __shfl_down(runs, d);
    gc += __shfl_down(gc, d);
  }
  if ((threadIdx.x & 63) == 0) {
    if (nn) atomicAdd(&meta[3], nn);
    if (runs) atomicAdd(&meta[4], runs);
    if (gc) atomicAdd(&meta[5], gc);
  }
}

// markers -> separators (its own launch: the pass above reads its neighbours' bytes)
__global__ __launch_bounds__(256) void mk_clean_mark_k(uint8_t* __restrict__ seq, const MkChunkInfo* __restrict__ info) {
  const size_t n = info->seq_len;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += (size_t)gridDim.x * blockDim.x * 16) {
    const size_t e = i + 16 < n ? i + 16 : n;
    for (size_t j = i; j < e; ++j)
      if (seq[j] == MK_CLEAN_MARK) seq[j] = (uint8_t)MK_SEP;
  }
}

static unsigned grid16(size_t n) {
  size_t g = (n / 16 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g == 0) g = 1;
  return (unsigned)g;
}

// before the parser: raw bytes [0, n) of the chunk (the context's own buffer), in place
int mk_launch_clean_pre(mk_ctx* c, uint8_t* d_raw, size_t n) {
  int rc = mk_buf_reserve(c, c->clean_meta, 8 * sizeof(u64));
  if (rc) return rc;
  u64 init[8] = {(u64)n, 0, 0, 0, 0, 0, 0, 0};
  MK_HIP(hipMemcpyAsync(c->clean_meta.p, init, sizeof init, hipMemcpyHostToDevice, c->stream));
  if (!n) return MK_OK;
  hipLaunchKernelGGL(mk_clean_pre_k, dim3(grid16(n)), dim3(256), 0, c->stream, d_raw, n, c->clean_upper, (u64*)c->clean_meta.p);
  hipLaunchKernelGGL(mk_clean_head_k, dim3(64), dim3(256), 0, c->stream, d_raw, n, (u64*)c->clean_meta.p);
  MK_HIP(hipGetLastError());
  return MK_OK;
}

// after the parser (and its fused pack): seq_cap bounds the parsed length (the kernels read the true one on the device)
int mk_launch_clean_post(mk_ctx* c, size_t seq_cap) {
  if (!seq_cap) return MK_OK;
  const size_t cap = 1 << 16;  // runs listed per chunk (more: counted, the list is cut -- mk_clean_stats_gpu_t.runs_listed)
  int rc = mk_buf_reserve(c, c->clean_runs, 2 * cap * sizeof(u64));
  if (rc) return rc;
  u64* starts = (u64*)c->clean_runs.p;
  hipLaunchKernelGGL(mk_clean_post_k, dim3(grid16(seq_cap)), dim3(256), 0, c->stream, (uint8_t*)c->seq.p, (const MkChunkInfo*)c->info.p,
                     (u64*)c->clean_meta.p, starts, starts + cap, (u64)cap);
  hipLaunchKernelGGL(mk_clean_mark_k, dim3(grid16(seq_cap)), dim3(256), 0, c->stream, (uint8_t*)c->seq.p, (const MkChunkInfo*)c->info.p);
  MK_HIP(hipGetLastError());
  return MK_OK;
}
