// mk_cutscan.h -- the virtual Chunker as a streaming scanner (host code, no GPU).
//
// Restates Chunker.stream_delim (lib/mercat2_Chunker.py:39-59) over a text that arrives in
// blocks of any size: text-mode lines ("\n", "\r\n" and a lone "\r" end a line, each written
// back as one "\n"); a line that CONTAINS '>' opens the next chunk when the bytes written to
// the current chunk before it are >= chunksize.  mk_chunk_cuts (mk_host.cpp) is the
// whole-buffer statement of the same rule and the two are tested against each other.
//
// Far from the threshold the scanner does no per-line work: the bytes a chunk has received are
// (raw bytes - "\r\n" pairs), so it jumps to the last line start that can still be below the
// threshold and only walks lines from there to the next header.  A line that may open a chunk
// and is cut by a block boundary is held back (copied) until its end is seen, so that bytes
// are only ever handed to the chunk they belong to.
#ifndef MK_CUTSCAN_H
#define MK_CUTSCAN_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <vector>

struct MkCutSink {
  virtual int feed(const uint8_t* p, size_t n) = 0;  // bytes of the current chunk, in order
  virtual int cut(uint64_t abs_offset) = 0;          // the current chunk ends; the next byte fed opens a new one
  virtual ~MkCutSink() {}
};

class MkCutScanner {
 public:
  // record_starts_only: a line opens the next chunk only if it STARTS a record -- its first non-blank byte is '>'
  // (what find_kmers takes for a header after line.strip(), lib/mercat2_kmers.py:51-52) -- instead of merely
  // containing '>' (the Chunker's rule).  For pieces of ONE filter unit counted on several GPUs: no window may span
  // a cut, and a sequence line such as "AC>GT" is not a record boundary.
  MkCutScanner(uint64_t chunksize, MkCutSink* sink, bool record_starts_only = false)
      : chunksize_(chunksize), sink_(sink), strict_(record_starts_only) {}

  // Next block of the text. has_cr: the block holds at least one '\r' (pass true when unknown).
  int block(const uint8_t* p, size_t n, bool has_cr) {
    int rc = 0;
    size_t pos = 0, run = 0;  // [run, pos) = bytes of this block not yet fed and not held
    if (n == 0) return 0;
    if (prev_cr_) {  // "\r" ended the previous block: a leading "\n" belongs to that terminator
      prev_cr_ = false;
      if (p[0] == '\n') pos = 1;
    }
    while (pos < n) {
      if (!mid_line_) {
        if (written_ < chunksize_) {
          // ---- bulk: skip to the last line start that is certainly not past the threshold
          const uint64_t need = chunksize_ - written_;
          const size_t avail = n - pos;
          size_t stop;  // complete lines [pos, stop) are consumed without looking at them
          if ((uint64_t)avail < need) {
            stop = last_terminator(p, pos, n, has_cr);  // index after the last terminator, or pos
            if (stop == n && p[n - 1] == '\r') prev_cr_ = true;
          } else {
            const uint8_t* nl = (const uint8_t*)memrchr(p + pos, '\n', (size_t)need);
            stop = nl ? (size_t)(nl - p) + 1 : pos;
          }
          if (stop > pos) {
            written_ += (uint64_t)(stop - pos) - (has_cr ? crlf_pairs(p + pos, stop - pos) : 0);
            pos = stop;
            continue;
          }
        }
        // ---- a line starts at pos
        mid_line_ = true;
        armed_ = written_ >= chunksize_;
        has_delim_ = false;
        decided_ = false;
        line_len_ = 0;
        line_abs_ = base_ + pos;
        lp_ = pos;
      }
      // ---- the line in progress: find its end
      const uint8_t* q = p + pos;
      const size_t left = n - pos;
      const uint8_t* nl = (const uint8_t*)memchr(q, '\n', left);
      size_t len = nl ? (size_t)(nl - q) : left;
      bool term = nl != nullptr;
      if (has_cr) {
        const uint8_t* cr = (const uint8_t*)memchr(q, '\r', len);
        if (cr) { len = (size_t)(cr - q); term = true; }
      }
      if (armed_ && !strict_ && !has_delim_ && memchr(q, '>', len)) has_delim_ = true;
      if (armed_ && strict_ && !decided_) {
        for (size_t t = 0; t < len; ++t) {
          const uint8_t ch = q[t];
          const bool blank = ch == ' ' || (ch >= 0x09 && ch <= 0x0D) || (ch >= 0x1C && ch <= 0x1F);  // str.strip(), ASCII
          if (!blank) { has_delim_ = ch == '>'; decided_ = true; break; }
        }
      }
      line_len_ += len;
      if (!term) {  // the line goes on in the next block
        if (armed_) {
          if (lp_ > run && (rc = sink_->feed(p + run, lp_ - run))) return rc;
          hold_.insert(hold_.end(), p + lp_, p + n);
          run = n;
        }
        pos = n;
        lp_ = 0;
        break;
      }
      size_t after = pos + len + 1;
      if (p[pos + len] == '\r') {
        if (after < n) { if (p[after] == '\n') ++after; }
        else prev_cr_ = true;
      }
      if (armed_) {
        if (has_delim_) {
          if (lp_ > run && (rc = sink_->feed(p + run, lp_ - run))) return rc;
          run = lp_;
          if ((rc = sink_->cut(line_abs_))) return rc;
          written_ = 0;
        }
        if (!hold_.empty()) {
          if ((rc = sink_->feed(hold_.data(), hold_.size()))) return rc;
          hold_.clear();
        }
      }
      written_ += line_len_ + 1;
      mid_line_ = false;
      pos = after;
    }
    if (n > run && (rc = sink_->feed(p + run, n - run))) return rc;
    base_ += n;
    return 0;
  }

  // End of the text: an unterminated last line is a line too.
  int finish() {
    int rc = 0;
    if (mid_line_ && armed_) {
      if (has_delim_ && (rc = sink_->cut(line_abs_))) return rc;
      if (!hold_.empty()) {
        if ((rc = sink_->feed(hold_.data(), hold_.size()))) return rc;
        hold_.clear();
      }
    }
    mid_line_ = false;
    return 0;
  }

 private:
  static size_t last_terminator(const uint8_t* p, size_t pos, size_t n, bool has_cr) {
    const uint8_t* a = (const uint8_t*)memrchr(p + pos, '\n', n - pos);
    const uint8_t* b = has_cr ? (const uint8_t*)memrchr(p + pos, '\r', n - pos) : nullptr;
    const uint8_t* m = a > b ? a : b;
    return m ? (size_t)(m - p) + 1 : pos;
  }
  static uint64_t crlf_pairs(const uint8_t* p, size_t n) {
    uint64_t c = 0;
    const uint8_t* q = p;
    const uint8_t* end = p + n;
    while (q < end) {
      q = (const uint8_t*)memchr(q, '\r', (size_t)(end - q));
      if (!q) break;
      if (q + 1 < end && q[1] == '\n') ++c;
      ++q;
    }
    return c;
  }

  uint64_t chunksize_;
  MkCutSink* sink_;
  uint64_t written_ = 0;    // bytes the current chunk holds at the start of the line in progress
  uint64_t base_ = 0;       // absolute offset of the current block
  bool prev_cr_ = false;    // the previous block ended with a '\r' terminator
  bool mid_line_ = false;   // a line is in progress
  bool armed_ = false;      // ... and it started at or past the threshold (it cuts if it holds '>')
  bool has_delim_ = false;
  bool strict_ = false;     // cut at record starts only (see the constructor)
  bool decided_ = false;    // strict: the first non-blank byte of the line in progress has been seen
  uint64_t line_len_ = 0;   // content bytes of the line in progress seen so far
  uint64_t line_abs_ = 0;   // absolute offset of its first byte
  size_t lp_ = 0;           // where its bytes start in the current block
  std::vector<uint8_t> hold_;  // bytes of an armed line from earlier blocks
};

#endif
