// mk_host.cpp -- host-side helpers behind the C ABI that need no GPU: the virtual Chunker and
// the deterministic synthetic-read generator.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include <vector>
#include "../../include/mercat_hip.h"
#include "mk_cutscan.h"
#include "mk_inflate.h"
#include "mk_crc32.h"
#include "mk_pgunzip.h"

// ---------------------------------------------------------------------------- virtual Chunker
// Restates Chunker.stream_delim (lib/mercat2_Chunker.py:39-59) without writing chunk files:
// the reference iterates text-mode lines (universal newlines: "\n", "\r\n" and lone "\r" each
// end a line and are written back as one "\n"), and when a line CONTAINS '>' and the bytes
// written to the current chunk so far are >= chunksize, that line opens the next chunk.
// We return the byte offsets (into the original, un-normalised text) where chunks 1.. begin.
extern "C" int mk_chunk_cuts(const uint8_t* text, size_t n, uint64_t chunksize, uint64_t* cuts, size_t cap,
                             size_t* ncuts) {
  if (!ncuts || (n && !text)) return MK_ERR_ARG;
  size_t found = 0;
  uint64_t written = 0;
  size_t pos = 0;
  while (pos < n) {
    // find the end of this line: first '\n' or '\r' at or after pos
    const uint8_t* p = text + pos;
    const size_t left = n - pos;
    const uint8_t* nl = (const uint8_t*)memchr(p, '\n', left);
    size_t line_len = nl ? (size_t)(nl - p) : left;       // content length if terminated by \n
    const uint8_t* cr = (const uint8_t*)memchr(p, '\r', line_len);
    size_t term = 0;                                       // bytes of terminator in the raw text
    if (cr) {
      line_len = (size_t)(cr - p);
      term = (pos + line_len + 1 < n && p[line_len + 1] == '\n') ? 2 : 1;
    } else if (nl) {
      term = 1;
    }
    const bool has_delim = memchr(p, '>', line_len) != nullptr;
    if (has_delim && written >= chunksize) {
      // (the very first line can only cut when chunksize == 0: chunk 0 is then empty)
      if (found < cap && cuts) cuts[found] = (uint64_t)pos;
      ++found;
      written = 0;
    }
    written += line_len + (term ? 1 : 0);
    pos += line_len + term;
  }
  *ncuts = found;
  return (found > cap && cuts) ? MK_ERR_RANGE : MK_OK;
}

// The same rule through the streaming scanner the file reader uses (mk_cutscan.h), fed `block`
// bytes at a time.  Besides the cuts it checks that the scanner hands every byte on exactly
// once, in order, and that each cut falls where the bytes fed so far end.
namespace {
struct RecordingSink : MkCutSink {
  const uint8_t* text;
  size_t fed = 0;
  bool ok = true;
  std::vector<uint64_t> cuts;
  int feed(const uint8_t* p, size_t n) override {
    if (memcmp(p, text + fed, n) != 0) ok = false;
    fed += n;
    return 0;
  }
  int cut(uint64_t abs) override {
    if (abs != fed) ok = false;
    cuts.push_back(abs);
    return 0;
  }
};
}  // namespace

static int stream_cuts(const uint8_t* text, size_t n, uint64_t chunksize, size_t block, uint64_t* cuts, size_t cap,
                       size_t* ncuts, bool record_starts_only);
extern "C" int mk_stream_cuts(const uint8_t* text, size_t n, uint64_t chunksize, size_t block, uint64_t* cuts,
                              size_t cap, size_t* ncuts) {
  return stream_cuts(text, n, chunksize, block, cuts, cap, ncuts, false);
}
// The cuts mk_count_file makes when it splits ONE filter unit (a file below the chunk size) over several GPUs:
// pieces of at least `piece` bytes that end where a record starts (first non-blank byte of the line is '>').
extern "C" int mk_record_cuts(const uint8_t* text, size_t n, uint64_t piece, size_t block, uint64_t* cuts, size_t cap,
                              size_t* ncuts) {
  return stream_cuts(text, n, piece, block, cuts, cap, ncuts, true);
}
static int stream_cuts(const uint8_t* text, size_t n, uint64_t chunksize, size_t block, uint64_t* cuts, size_t cap,
                       size_t* ncuts, bool record_starts_only) {
  if (!ncuts || (n && !text) || block == 0) return MK_ERR_ARG;
  RecordingSink sink;
  sink.text = text;
  MkCutScanner scan(chunksize, &sink, record_starts_only);
  for (size_t off = 0; off < n; off += block) {
    const size_t m = n - off < block ? n - off : block;
    scan.block(text + off, m, memchr(text + off, '\r', m) != nullptr);
  }
  scan.finish();
  if (!sink.ok || sink.fed != n) return MK_ERR_STATE;
  *ncuts = sink.cuts.size();
  for (size_t i = 0; i < sink.cuts.size() && i < cap && cuts; ++i) cuts[i] = sink.cuts[i];
  return (sink.cuts.size() > cap && cuts) ? MK_ERR_RANGE : MK_OK;
}

// ------------------------------------------------------------------------------- gunzip
// The reader's own gzip decoder (mk_inflate.h) over a whole file in memory, producing `block` bytes
// per call as the file reader does, with every member's CRC-32 and length checked (zlib's crc32).
// A self-check for tests (against zlib / gzip.py); `out` must hold the whole text.
extern "C" int mk_gunzip(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, size_t block, size_t* written, int* members) {
  if (!written || (n && !gz) || block == 0) return MK_ERR_ARG;
  MkGzReader rd(gz, n);
  size_t at = 0, member_start = 0;
  for (;;) {
    size_t got = 0;
    const size_t room = cap - at < block ? cap - at : block;
    const MkGzReader::Status s = rd.fill(out + at, out + at + room, out, &got);
    at += got;
    if (s == MkGzReader::END) break;
    if (s != MkGzReader::MORE) return s == MkGzReader::TRUNCATED ? MK_ERR_RANGE : MK_ERR_IO;
    if (rd.member_ended()) {
      const uint32_t crc = mk_crc32(0, out + member_start, at - member_start);
      if (crc != rd.member_crc() || (uint32_t)(at - member_start) != rd.member_isize()) return MK_ERR_IO;
      member_start = at;
    } else if (got == 0 && room == 0) {
      return MK_ERR_NOMEM;  // cap too small
    }
  }
  *written = at;
  if (members) *members = rd.members();
  return MK_OK;
}

// The same text through the parallel decoder (mk_pgunzip.h): `threads` pieces of `piece_bytes` compressed
// bytes per round.  A self-check for tests.
extern "C" int mk_gunzip_parallel(const uint8_t* gz, size_t n, uint8_t* out, size_t cap, int threads, size_t piece_bytes,
                                  size_t* written, int* members) {
  if (!written || (n && !gz)) return MK_ERR_ARG;
  MkParallelGunzip rd(gz, n, threads, piece_bytes);
  size_t at = 0;
  MkRawBuf<uint8_t> buf;
  for (;;) {
    const uint8_t* part = nullptr;
    size_t len = 0;
    const MkParallelGunzip::Status s = rd.next(buf, &part, &len);
    if (s == MkParallelGunzip::END) break;
    if (s != MkParallelGunzip::MORE) return s == MkParallelGunzip::TRUNCATED ? MK_ERR_RANGE : MK_ERR_IO;
    if (len > cap - at) return MK_ERR_NOMEM;
    memcpy(out + at, part, len);
    at += len;
  }
  *written = at;
  if (members) *members = rd.members();
  return MK_OK;
}

// CRC-32 as the gzip reader computes it (mk_crc32.h), for tests against zlib.crc32.
extern "C" uint32_t mk_crc32_of(const uint8_t* p, size_t n, uint32_t seed) { return mk_crc32(seed, p, n); }

// ------------------------------------------------------------------------- synthetic reads
static inline uint64_t splitmix64(uint64_t& s) {
  s += 0x9E3779B97F4A7C15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static size_t dec_len(uint64_t v) {
  size_t l = 1;
  while (v >= 10) { v /= 10; ++l; }
  return l;
}

extern "C" int mk_synth_reads(uint64_t genome_len, uint64_t genome_seed, uint64_t reads, uint32_t read_len,
                              uint64_t read_seed, uint32_t sub_ppm, uint64_t first_index, uint8_t* out, size_t cap,
                              size_t* written) {
  if (!written || read_len == 0 || genome_len < read_len) return MK_ERR_ARG;
  // size: ">r" + digits + "\n" + read_len + "\n"
  size_t total = 0;
  {
    uint64_t i = first_index, end = first_index + reads;
    while (i < end) {  // group indices by decimal length
      size_t l = dec_len(i);
      uint64_t lim = 1;
      for (size_t d = 0; d < l; ++d) lim *= 10;  // first index with one more digit
      uint64_t hi = end < lim ? end : lim;
      total += (size_t)(hi - i) * (2 + l + 1 + read_len + 1);
      i = hi;
    }
  }
  *written = total;
  if (!out) return MK_OK;
  if (cap < total) return MK_ERR_RANGE;
  // genome as ASCII, forward and reverse-complement (rc[i] = complement(fwd[G-1-i])), so that a
  // read is one memcpy. 32 bases per splitmix64 draw (base j of a draw = bits 2j..2j+1).
  static const char L[4] = {'A', 'C', 'G', 'T'};
  std::vector<uint8_t> fwd(genome_len), rc(genome_len);
  {
    uint64_t s = genome_seed;
    for (uint64_t i = 0; i < genome_len; i += 32) {
      uint64_t r = splitmix64(s);
      uint64_t m = genome_len - i < 32 ? genome_len - i : 32;
      for (uint64_t j = 0; j < m; ++j) {
        unsigned code = (unsigned)((r >> (2 * j)) & 3);
        fwd[i + j] = (uint8_t)L[code];
        rc[genome_len - 1 - (i + j)] = (uint8_t)L[3 - code];
      }
    }
  }
  uint8_t* w = out;
  const uint64_t span = genome_len - read_len + 1;
  for (uint64_t r = 0; r < reads; ++r) {
    const uint64_t idx = first_index + r;
    uint64_t s = read_seed + idx * 0x632BE59BD9B4E019ull;
    const uint64_t start = splitmix64(s) % span;
    const bool is_rc = (splitmix64(s) >> 63) != 0;
    *w++ = '>';
    *w++ = 'r';
    char num[24];
    int len = 0;
    uint64_t v = idx;
    do { num[len++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (len) *w++ = (uint8_t)num[--len];
    *w++ = '\n';
    // reverse strand: revcomp(fwd[start .. start+L)) == rc[G-start-L .. G-start)
    const uint8_t* src = is_rc ? rc.data() + (genome_len - start - read_len) : fwd.data() + start;
    memcpy(w, src, read_len);
    if (sub_ppm) {
      for (uint32_t j = 0; j < read_len; ++j) {
        uint64_t d = splitmix64(s);
        if (d % 1000000ull < sub_ppm) {
          unsigned b = w[j] == 'A' ? 0 : (w[j] == 'C' ? 1 : (w[j] == 'G' ? 2 : 3));
          w[j] = (uint8_t)L[(b + 1 + (unsigned)((d >> 32) % 3)) & 3u];
        }
      }
    }
    w += read_len;
    *w++ = '\n';
  }
  return MK_OK;
}

// ------------------------------------------------------------------------------ removeN
// Restates removeN / split_sequenceN (lib/mercat2_fasta.py:53-119, 21-49): the text rewrite MerCat2 runs on
// every nucleotide FASTA before counting (bin/mercat2.py:239-244, 276).  Reference behaviour kept:
//  * text-mode lines (universal newlines), each line str.strip()ped; a header is a stripped line that starts
//    with '>'; lines in front of the first header are dropped;
//  * a record whose concatenated sequence holds an upper-case 'N' is cut at every run of N: piece i (from 1,
//    empty pieces included) gets the header ">{first word}_{i} {other words joined by one blank}" and its
//    sequence re-wrapped at 80 columns (textwrap.wrap of a string without blanks or hyphens);
//  * any other record is written as ">" + header text and its stripped lines one by one (empty lines too);
//  * -toupper upper-cases the sequence lines on output only (a lower-case 'n' is not a cut);
//  * GC content counts 'G' + 'C' and the length over the concatenated sequence of an unsplit record, and over
//    EVERY emitted line of a split one, header lines included, before upper-casing (lib/mercat2_fasta.py:99-100).
// What it does not restate: textwrap's handling of blanks and hyphens INSIDE a sequence that is being split
// (st->unsupported_record is set and nothing is produced; the Python host layer handles such a file).
namespace {
inline bool py_space(unsigned c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

struct LineReader {  // text-mode readline + strip over a byte buffer
  const uint8_t* p;
  size_t n, pos = 0;
  LineReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  // false at EOF; [a, b) = the stripped line
  bool next(size_t& a, size_t& b) {
    if (pos >= n) return false;
    const uint8_t* s = p + pos;
    const size_t left = n - pos;
    const uint8_t* nl = (const uint8_t*)memchr(s, '\n', left);
    size_t len = nl ? (size_t)(nl - s) : left;
    const uint8_t* cr = (const uint8_t*)memchr(s, '\r', len);
    size_t term = nl ? 1 : 0;
    if (cr) {
      len = (size_t)(cr - s);
      term = (pos + len + 1 < n && s[len + 1] == '\n') ? 2 : 1;
    }
    a = pos;
    b = pos + len;
    pos += len + term;
    while (a < b && py_space(p[a])) ++a;
    while (b > a && py_space(p[b - 1])) --b;
    return true;
  }
};
}  // namespace

extern "C" int mk_remove_n(const uint8_t* text, size_t n, int toupper, uint8_t** out, size_t* out_len, mk_clean_stats_t* st) {
  if (!out || !out_len || !st || (n && !text)) return MK_ERR_ARG;
  *out = nullptr;
  *out_len = 0;
  memset(st, 0, sizeof *st);
  st->unsupported_record = -1;
  for (size_t i = 0; i < n; ++i)
    if (text[i] >= 0x80) {  // the reference works on decoded characters (lengths, wrapping): left to the Python layer
      st->unsupported_record = 0;
      return MK_OK;
    }
  std::vector<uint8_t> o;
  o.reserve(n + n / 64 + 4096);
  std::vector<uint8_t> seq;
  std::vector<std::pair<size_t, size_t>> lines;
  LineReader rd(text, n);
  size_t a = 0, b = 0;
  bool have = rd.next(a, b);
  // (upper = false for a wrapped line of a split record that starts with '>': the reference takes every such line
  // of its piece list for a header and prints it as it stands, lib/mercat2_fasta.py:103-104)
  auto put_seq = [&](const uint8_t* s, size_t len, bool upper = true) {
    const size_t at = o.size();
    o.insert(o.end(), s, s + len);
    if (toupper && upper)
      for (size_t i = at; i < o.size(); ++i)
        if (o[i] >= 'a' && o[i] <= 'z') o[i] = (uint8_t)(o[i] - 32);
    o.push_back('\n');
  };
  auto count_gc = [&](const uint8_t* s, size_t len) {
    uint64_t g = 0;
    for (size_t i = 0; i < len; ++i) g += (s[i] == 'G') | (s[i] == 'C');
    st->gc_count += g;
    st->total_length += len;
  };
  while (have) {
    if (!(a < b && text[a] == '>')) {  // not a header: skipped (only happens in front of the first one)
      have = rd.next(a, b);
      continue;
    }
    const size_t name_a = a + 1, name_b = b;
    st->records += 1;
    seq.clear();
    lines.clear();
    bool has_n = false;
    while ((have = rd.next(a, b))) {
      if (a < b && text[a] == '>') break;
      lines.emplace_back(a, b);
      if (!has_n && a < b && memchr(text + a, 'N', b - a)) has_n = true;
    }
    if (!has_n) {
      o.push_back('>');
      o.insert(o.end(), text + name_a, text + name_b);
      o.push_back('\n');
      for (auto& ln : lines) {
        put_seq(text + ln.first, ln.second - ln.first);
        count_gc(text + ln.first, ln.second - ln.first);
      }
      continue;
    }
    // split at runs of N
    for (auto& ln : lines) seq.insert(seq.end(), text + ln.first, text + ln.second);
    for (uint8_t ch : seq)
      if (ch == ' ' || ch == '\t' || ch == 0x0b || ch == 0x0c || ch == '-' || (ch >= 0x1c && ch <= 0x1f)) {
        st->unsupported_record = (int64_t)st->records - 1;
        return MK_OK;
      }
    // header words
    std::vector<std::pair<size_t, size_t>> words;
    for (size_t i = name_a; i < name_b;) {
      while (i < name_b && py_space(text[i])) ++i;
      size_t j = i;
      while (j < name_b && !py_space(text[j])) ++j;
      if (j > i) words.emplace_back(i, j);
      i = j;
    }
    if (words.empty()) return MK_ERR_RANGE;  // the reference raises IndexError (header.split()[0])
    st->split_records += 1;
    size_t i = 0, piece = 0;
    const size_t L = seq.size();
    for (;;) {  // pieces = seq.split at N+ (leading / trailing runs give empty pieces)
      size_t j = i;
      while (j < L && seq[j] != 'N') ++j;
      ++piece;
      const size_t h0 = o.size();
      o.push_back('>');
      o.insert(o.end(), text + words[0].first, text + words[0].second);
      o.push_back('_');
      char num[24];
      const int nd = snprintf(num, sizeof num, "%zu", piece);
      o.insert(o.end(), num, num + nd);
      o.push_back(' ');
      for (size_t w = 1; w < words.size(); ++w) {
        if (w > 1) o.push_back(' ');
        o.insert(o.end(), text + words[w].first, text + words[w].second);
      }
      count_gc(o.data() + h0, o.size() - h0);  // (the reference counts the header line of a split record too)
      o.push_back('\n');
      for (size_t q = i; q < j; q += 80) {
        const size_t len = j - q < 80 ? j - q : 80;
        count_gc(seq.data() + q, len);
        put_seq(seq.data() + q, len, seq[q] != '>');
      }
      st->pieces += 1;
      if (j >= L) break;
      while (j < L && seq[j] == 'N') ++j;  // the run
      st->n_runs += 1;
      i = j;
    }
  }
  uint8_t* mem = (uint8_t*)malloc(o.size() ? o.size() : 1);
  if (!mem) return MK_ERR_NOMEM;
  memcpy(mem, o.data(), o.size());
  *out = mem;
  *out_len = o.size();
  return MK_OK;
}

extern "C" void mk_free(void* p) { free(p); }

// ---- planning helpers of the multi-GPU path (no GPU needed) -------------------------------------------------
// First key of owner 1..n-1 when [0, 2^key_bits) is cut into n equal ranges (the same cut as
// mercat2_amd.dist.range_bounds): owner of a key = number of bounds <= key.
extern "C" int mk_owner_bounds(int key_bits, int n, uint64_t* bounds) {
  if (key_bits < 1 || key_bits > 64 || n < 1 || (n > 1 && !bounds)) return MK_ERR_ARG;
  for (int i = 1; i < n; ++i) {
    const unsigned __int128 span = (unsigned __int128)1 << key_bits;
    bounds[i - 1] = (uint64_t)((span * (unsigned)i + (unsigned)n - 1) / (unsigned)n);
  }
  return MK_OK;
}

// Context creation order for mk_count_file over several devices: chunk i -> ctxs[i mod nctx] must mean
// device devices[i mod ndev], and the chunks of one device must take turns on its streams.
extern "C" int mk_plan_contexts(const int* devices, int ndev, int streams, int* ctx_device) {
  if (!devices || !ctx_device || ndev < 1 || streams < 1) return MK_ERR_ARG;
  for (int j = 0; j < ndev * streams; ++j) ctx_device[j] = devices[j % ndev];
  return MK_OK;
}
