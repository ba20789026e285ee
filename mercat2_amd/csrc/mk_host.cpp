// mk_host.cpp -- host-side helpers behind the C ABI that need no GPU: the virtual Chunker and
// the deterministic synthetic-read generator.
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include <string>
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
//  * a sequence that is being split and holds blanks, tabs or hyphens goes through textwrap's word rules (tw_wrap
//    below restates them);
//  * bytes >= 0x80: header lines may hold any (they are copied as they stand; where the reference works on decoded
//    text -- str.strip() of a header, header.split() and len() of a split record's header -- UTF-8 sequences of
//    Unicode white space are blanks and a character counts once); in a SEQUENCE line they are refused
//    (MK_ERR_NON_ASCII), as the counting engine refuses them.
namespace {
inline bool py_space(unsigned c) { return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31); }

// Length of the UTF-8 sequence at s (n bytes left) if it encodes a character str.isspace() holds for besides the
// ASCII ones: U+0085, U+00A0, U+1680, U+2000..U+200A, U+2028, U+2029, U+202F, U+205F, U+3000; else 0.
inline size_t uni_space(const uint8_t* s, size_t n) {
  if (n >= 2 && s[0] == 0xC2 && (s[1] == 0x85 || s[1] == 0xA0)) return 2;
  if (n >= 3 && s[0] == 0xE1 && s[1] == 0x9A && s[2] == 0x80) return 3;
  if (n >= 3 && s[0] == 0xE2 && s[1] == 0x80 && ((s[2] >= 0x80 && s[2] <= 0x8A) || s[2] == 0xA8 || s[2] == 0xA9 || s[2] == 0xAF)) return 3;
  if (n >= 3 && s[0] == 0xE2 && s[1] == 0x81 && s[2] == 0x9F) return 3;
  if (n >= 3 && s[0] == 0xE3 && s[1] == 0x80 && s[2] == 0x80) return 3;
  return 0;
}
// the same, for a sequence that ENDS at s + n
inline size_t uni_space_before(const uint8_t* s, size_t n) {
  if (n >= 2 && uni_space(s + n - 2, 2) == 2) return 2;
  if (n >= 3 && uni_space(s + n - 3, 3) == 3) return 3;
  return 0;
}

struct LineReader {  // text-mode readline + strip over a byte buffer
  const uint8_t* p;
  size_t n, pos = 0;
  // where the next '\n' / '\r' at or after pos is (n = none): found once and kept until pos has passed it, so that a
  // file with one kind of line end only -- CR-only files have no '\n' at all -- is not searched to its end per line
  size_t nl_at = 0, cr_at = 0;
  bool nl_known = false, cr_known = false;
  LineReader(const uint8_t* p_, size_t n_) : p(p_), n(n_) {}
  // false at EOF; [a, b) = the stripped line
  bool next(size_t& a, size_t& b) {
    if (pos >= n) return false;
    if (!nl_known || nl_at < pos) {
      const uint8_t* q = (const uint8_t*)memchr(p + pos, '\n', n - pos);
      nl_at = q ? (size_t)(q - p) : n;
      nl_known = true;
    }
    if (!cr_known || cr_at < pos) {
      const uint8_t* q = (const uint8_t*)memchr(p + pos, '\r', n - pos);
      cr_at = q ? (size_t)(q - p) : n;
      cr_known = true;
    }
    const size_t end = nl_at < cr_at ? nl_at : cr_at;  // first line end, or n
    size_t term = 0;
    if (end < n) term = (p[end] == '\r' && end + 1 < n && p[end + 1] == '\n') ? 2 : 1;
    a = pos;
    b = end;
    pos = end + term;
    for (;;) {  // str.strip()
      if (a < b && py_space(p[a])) { ++a; continue; }
      const size_t u = a < b && p[a] >= 0x80 ? uni_space(p + a, b - a) : 0;
      if (!u) break;
      a += u;
    }
    for (;;) {
      if (b > a && py_space(p[b - 1])) { --b; continue; }
      const size_t u = b > a && p[b - 1] >= 0x80 ? uni_space_before(p + a, b - a) : 0;
      if (!u) break;
      b -= u;
    }
    return true;
  }
};

// ---- textwrap.wrap(text, 80) for ASCII text (CPython 3.10 Lib/textwrap.py: TextWrapper with its defaults:
//      expand_tabs, replace_whitespace, drop_whitespace, break_long_words, break_on_hyphens, tabsize 8) ----------
// split_sequenceN wraps every piece of a split sequence with it (lib/mercat2_fasta.py:47).  For a piece without
// blanks and hyphens that is "80 characters per line"; with them the standard library's word rules apply, and they
// are restated here so that no Python copy of removeN is needed: _munge_whitespace, wordsep_re, _wrap_chunks,
// _handle_long_word.
inline bool tw_ws(unsigned c) { return c == ' ' || (c >= 9 && c <= 13); }                  // textwrap._whitespace
inline bool tw_word(unsigned c) { return (c >= '0' && c <= '9') || (c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z') || c == '_'; }  // \w
inline bool tw_letter(unsigned c) { return tw_word(c) && !(c >= '0' && c <= '9'); }        // [^\d\W]
inline bool tw_wp(unsigned c) { return tw_word(c) || c == '!' || c == '"' || c == '\'' || c == '&' || c == '.' || c == ',' || c == '?'; }

// chunks of wordsep_re.split(t): every position of t is matched by one of the pattern's three alternatives, so the
// chunks tile the text; returned as end offsets
static void tw_split(const std::string& t, std::vector<size_t>& ends) {
  const size_t n = t.size();
  auto at = [&](size_t i) -> unsigned { return i < n ? (unsigned char)t[i] : 0u; };
  size_t i = 0;
  while (i < n) {
    size_t e;
    if (tw_ws(at(i))) {  // any whitespace: ws+
      e = i;
      while (e < n && tw_ws(at(e))) ++e;
    } else {
      e = 0;
      // em-dash between words: (?<=wp) -{2,} (?=\w)
      if (at(i) == '-' && i > 0 && tw_wp(at(i - 1))) {
        size_t h = i;
        while (h < n && at(h) == '-') ++h;
        if (h - i >= 2 && h < n && tw_word(at(h))) e = h;
      }
      if (!e) {
        // word, possibly hyphenated: nws+? then the first of (hyphen with letters around | end of word | before an em-dash)
        for (size_t p = i + 1;; ++p) {  // p = end of the lazily grown nws+?  (t[i .. p) holds no whitespace)
          if (p < n && at(p) == '-') {
            const bool behind = (p >= 2 && tw_letter(at(p - 1)) && tw_letter(at(p - 2))) ||
                                (p >= 3 && tw_letter(at(p - 1)) && at(p - 2) == '-' && tw_letter(at(p - 3)));
            const bool ahead = tw_letter(at(p + 1)) && (tw_letter(at(p + 2)) || (at(p + 2) == '-' && tw_letter(at(p + 3))));
            if (behind && ahead) { e = p + 1; break; }
          }
          if (p >= n || tw_ws(at(p))) { e = p; break; }  // end of word
          if (tw_wp(at(p - 1)) && at(p) == '-') {        // before an em-dash: (?<=wp)(?=-{2,}\w)
            size_t h = p;
            while (h < n && at(h) == '-') ++h;
            if (h - p >= 2 && h < n && tw_word(at(h))) { e = p; break; }
          }
        }
      }
    }
    ends.push_back(e);
    i = e;
  }
}

// the lines of textwrap.wrap(piece, width): f(const char* line, size_t len) per line
template <class F>
static void tw_wrap(const uint8_t* piece, size_t len, size_t width, F&& emit) {
  // _munge_whitespace: expandtabs(8), then every whitespace character becomes a blank
  std::string t;
  t.reserve(len + 16);
  size_t col = 0;
  for (size_t i = 0; i < len; ++i) {
    const unsigned c = piece[i];
    if (c == '\t') {
      const size_t pad = 8 - col % 8;
      t.append(pad, ' ');
      col += pad;
    } else if (c == '\n' || c == '\r') {  // (cannot occur inside a piece; str.expandtabs restarts its column there)
      t.push_back(' ');
      col = 0;
    } else {
      t.push_back(tw_ws(c) ? ' ' : (char)c);
      col += 1;
    }
  }
  std::vector<size_t> ends;
  tw_split(t, ends);
  // chunks as (begin, end) in t; _wrap_chunks works on the reversed list: `next` walks forward instead
  std::vector<std::pair<size_t, size_t>> chunks;
  {
    size_t b = 0;
    for (size_t e : ends) { if (e > b) chunks.emplace_back(b, e); b = e; }
  }
  auto blank = [&](const std::pair<size_t, size_t>& c) {  // chunk.strip() == '' (str.strip: also \x1c..\x1f)
    for (size_t i = c.first; i < c.second; ++i)
      if (!py_space((unsigned char)t[i])) return false;
    return true;
  };
  size_t next = 0;
  bool any_line = false;
  std::string line;
  while (next < chunks.size()) {
    line.clear();
    size_t cur_len = 0;
    size_t last_b = 0, last_e = 0;  // the last chunk put on the line (for the trailing-whitespace drop)
    bool have_last = false;
    if (blank(chunks[next]) && any_line) { ++next; if (next >= chunks.size()) break; }
    while (next < chunks.size()) {
      const size_t l = chunks[next].second - chunks[next].first;
      if (cur_len + l <= width) {
        line.append(t, chunks[next].first, l);
        last_b = line.size() - l; last_e = line.size(); have_last = true;
        cur_len += l;
        ++next;
      } else break;
    }
    if (next < chunks.size() && chunks[next].second - chunks[next].first > width) {  // _handle_long_word
      const size_t space_left = width < 1 ? 1 : width - cur_len;
      auto& c = chunks[next];
      size_t end = space_left;
      if (c.second - c.first > space_left) {  // break after the last hyphen that has a non-hyphen before it
        size_t hy = std::string::npos;
        for (size_t q = space_left; q-- > 0;)
          if (t[c.first + q] == '-') { hy = q; break; }
        if (hy != std::string::npos && hy > 0) {
          bool other = false;
          for (size_t q = 0; q < hy; ++q)
            if (t[c.first + q] != '-') { other = true; break; }
          if (other) end = hy + 1;
        }
      }
      if (end > c.second - c.first) end = c.second - c.first;
      line.append(t, c.first, end);
      last_b = line.size() - end; last_e = line.size(); have_last = true;
      c.first += end;  // (the rest of the chunk stays at the head of the list; an emptied chunk has length 0 and is taken next)
    }
    if (have_last) {  // drop a trailing all-whitespace chunk
      bool ws = true;
      for (size_t q = last_b; q < last_e; ++q)
        if (!py_space((unsigned char)line[q])) { ws = false; break; }
      if (ws) line.resize(last_b);
    }
    if (!line.empty()) { emit(line.data(), line.size()); any_line = true; }
    else if (!have_last && next < chunks.size() && chunks[next].second == chunks[next].first) ++next;  // (an emptied chunk)
  }
}
}  // namespace

extern "C" int mk_remove_n(const uint8_t* text, size_t n, int toupper, uint8_t** out, size_t* out_len, mk_clean_stats_t* st) {
  if (!out || !out_len || !st || (n && !text)) return MK_ERR_ARG;
  *out = nullptr;
  *out_len = 0;
  memset(st, 0, sizeof *st);
  st->unsupported_record = -1;
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
  // G + C and the length in CHARACTERS (a header line of a split record may hold multi-byte characters)
  auto count_gc = [&](const uint8_t* s, size_t len) {
    uint64_t g = 0, cont = 0;
    for (size_t i = 0; i < len; ++i) {
      g += (s[i] == 'G') | (s[i] == 'C');
      cont += (s[i] & 0xC0) == 0x80;  // UTF-8 continuation bytes
    }
    st->gc_count += g;
    st->total_length += len - cont;
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
      for (size_t i = a; i < b; ++i)
        if (text[i] >= 0x80) {  // non-ASCII sequence text: refused here as the counting engine refuses it
          st->unsupported_record = (int64_t)st->records - 1;
          return MK_ERR_NON_ASCII;
        }
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
    // header words: header.split() (white space of any kind, Unicode's included)
    std::vector<std::pair<size_t, size_t>> words;
    for (size_t i = name_a; i < name_b;) {
      for (;;) {
        if (i < name_b && py_space(text[i])) { ++i; continue; }
        const size_t u = i < name_b && text[i] >= 0x80 ? uni_space(text + i, name_b - i) : 0;
        if (!u) break;
        i += u;
      }
      size_t j = i;
      while (j < name_b && !py_space(text[j]) && !(text[j] >= 0x80 && uni_space(text + j, name_b - j))) ++j;
      if (j > i) words.emplace_back(i, j);
      i = j;
    }
    if (words.empty()) return MK_ERR_RANGE;  // the reference raises IndexError (header.split()[0])
    st->split_records += 1;
    size_t i = 0, piece = 0;
    const size_t L = seq.size();
    for (;;) {  // pieces = seq.split at N+ (leading / trailing runs give empty pieces)
      size_t j = i;
      bool words_matter = false;  // blanks or hyphens: textwrap's word rules instead of plain 80-column cuts
      for (; j < L && seq[j] != 'N'; ++j) {
        const uint8_t ch = seq[j];
        words_matter |= ch == '-' || py_space(ch);
      }
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
      if (!words_matter) {
        for (size_t q = i; q < j; q += 80) {
          const size_t len = j - q < 80 ? j - q : 80;
          count_gc(seq.data() + q, len);
          put_seq(seq.data() + q, len, seq[q] != '>');
        }
      } else {
        tw_wrap(seq.data() + i, j - i, 80, [&](const char* line, size_t len) {
          count_gc((const uint8_t*)line, len);
          put_seq((const uint8_t*)line, len, line[0] != '>');
        });
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

// textwrap.wrap(text, width) as restated above, for tests against the standard library: lines joined by '\n'.
extern "C" int mk_textwrap(const uint8_t* text, size_t n, size_t width, uint8_t** out, size_t* out_len) {
  if (!out || !out_len || (n && !text) || width < 1) return MK_ERR_ARG;
  std::string o;
  tw_wrap(text, n, width, [&](const char* line, size_t len) { o.append(line, len); o.push_back('\n'); });
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
