// mk_pgunzip.h -- one DEFLATE stream decoded by several threads (host code).
//
// A plain gzip file is a single bit stream with no index: block k can only be located by decoding
// blocks 0..k-1, and its matches reach into the 32 KiB of text before it.  Both obstacles have a known
// way around (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to
// DNA sequences", 2019), restated here for the reader's decoder (mk_inflate.h):
//   1. The compressed bytes are cut at arbitrary places.  From each cut a thread tries bit offset after
//      bit offset until a dynamic-Huffman block header parses under strict rules (complete code sets, an
//      end-of-block code) and 16 K symbols decode after it without an error and with text-like literals.
//      For random bits that is practically impossible, for a true block start it always succeeds.
//   2. Each thread decodes from its start to the next thread's start into 16-bit elements, the 32 KiB
//      before its start filled with the values 256 + position: a match that reaches into the unknown
//      history just copies those place holders along.  When the text before a piece is known, every
//      element >= 256 is replaced by the byte it names.
// Nothing rests on the search having been right: the piece before (decoded from a verified start, hence
// the true stream) must arrive at a block header at exactly the bit the next piece started from; if it
// does not, that piece and everything after it is dropped and decoding resumes from the last verified
// block header.  The gzip CRC-32 is still checked by the caller.
#ifndef MK_PGUNZIP_H
#define MK_PGUNZIP_H
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <emmintrin.h>

#include "mk_crc32.h"
#include "mk_inflate.h"

// Big blocks of the decoder outlive the file they were used for: a block of 1 MiB or more goes to a process-wide
// pool instead of back to the allocator, and the next file's decoder takes it from there.  Decoding a 0.55 GB .gz takes
// ~1.3 GB of these buffers; giving them back was 90 ms of munmap at the end of every file (the reader thread's exit,
// inside the window that is timed) and taking fresh ones a page fault per 4 KiB in the decoding threads.  The pool
// keeps at most MK_POOL_BYTES (default 4 GiB) and 96 blocks; MK_NO_BUF_POOL=1 turns it off.
struct MkBlockPool {
  struct Block { void* p; size_t bytes; };
  std::mutex mu;
  std::vector<Block> free_;
  size_t held = 0, limit = (size_t)4 << 30;
  bool off = false;
  MkBlockPool() {
    off = getenv("MK_NO_BUF_POOL") != nullptr;
    if (const char* e = getenv("MK_POOL_BYTES")) limit = (size_t)strtoull(e, nullptr, 10);
  }
  // (never destroyed: buffers may be given back while the process winds down; its blocks go with the process)
  static MkBlockPool& get() { static MkBlockPool* g = new MkBlockPool; return *g; }
  // the smallest pooled block of at least `bytes` (not more than four times as much), or nullptr
  void* take(size_t bytes, size_t* got) {
    if (off || bytes < ((size_t)1 << 20)) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    size_t best = free_.size();
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].bytes >= bytes && free_[i].bytes <= 4 * bytes && (best == free_.size() || free_[i].bytes < free_[best].bytes)) best = i;
    if (best == free_.size()) return nullptr;
    void* p = free_[best].p;
    *got = free_[best].bytes;
    held -= free_[best].bytes;
    free_[best] = free_.back();
    free_.pop_back();
    return p;
  }
  void give(void* p, size_t bytes) {
    if (!p) return;
    {
      std::lock_guard<std::mutex> g(mu);
      if (!off && bytes >= ((size_t)1 << 20) && free_.size() < 96 && held + bytes <= limit) {
        free_.push_back({p, bytes});
        held += bytes;
        return;
      }
    }
    free(p);
  }
};

// Growable array without value-initialisation (std::vector would zero hundreds of megabytes per round).
template <class T>
struct MkRawBuf {
  T* p = nullptr;
  size_t cap = 0;
  MkRawBuf() = default;
  MkRawBuf(const MkRawBuf&) = delete;
  MkRawBuf& operator=(const MkRawBuf&) = delete;
  ~MkRawBuf() { MkBlockPool::get().give(p, cap * sizeof(T)); }
  bool reserve(size_t n) {  // keeps the content
    if (n <= cap) return true;
    if (!p) {  // (a first reservation: a block some earlier file left in the pool, pages already there)
      size_t got = 0;
      if (void* q = MkBlockPool::get().take(n * sizeof(T), &got)) {
        p = (T*)q;
        cap = got / sizeof(T);
        return true;
      }
    }
    T* q = (T*)realloc(p, n * sizeof(T));
    if (!q) return false;
    p = q;
    cap = n;
    return true;
  }
};

class MkParallelInflate {
 public:
  static constexpr size_t WINDOW = 32768;
  enum Result { MORE = 0, STREAM_END = 1, BAD_DATA = -1, TRUNCATED = -2 };

  double s_find = 0, s_decode = 0, s_stitch = 0;  // seconds per phase, summed over the rounds
  size_t pieces_started = 0, pieces_kept = 0;

  MkParallelInflate(int threads, size_t piece_bytes) : threads_(threads < 1 ? 1 : threads), piece_(piece_bytes < 4096 ? 4096 : piece_bytes) {
    if (const char* e = getenv("MK_PGUNZIP_FAIL_ROUND")) fail_round_ = atoi(e);  // (tests: make that round give up)
  }

  // Decode from bit `start_bit` of [base, end) -- a verified block header -- whose preceding text ends
  // with history[0..history_len) (up to 32 KiB).  One round: up to `threads` pieces of `piece_bytes`
  // compressed bytes.  The text is appended to out; *next_bit is where the next round starts (a block
  // header) or, at STREAM_END, the bit after the final block.
  Result round(const uint8_t* base, const uint8_t* end, uint64_t start_bit, const uint8_t* history, size_t history_len,
               MkRawBuf<uint8_t>& out_, const uint8_t** text, size_t* text_len, uint64_t* next_bit, uint32_t* text_crc) {
    *text = nullptr;
    *text_len = 0;
    *text_crc = mk_crc32(0, nullptr, 0);
    if (rounds_++ == fail_round_) return BAD_DATA;
    const uint64_t end_bit = (uint64_t)(end - base) * 8;
    const auto t_a = std::chrono::steady_clock::now();
    // ---- 1. piece starts: the verified one, then the first plausible block header after every cut
    std::vector<uint64_t> start(1, start_bit);
    {
      std::vector<uint64_t> found((size_t)threads_, NONE);
      std::vector<std::thread> th;
      for (int t = 1; t < threads_; ++t) {
        const uint64_t from = start_bit + (uint64_t)t * piece_ * 8, to = from + (uint64_t)piece_ * 8;
        if (from + 64 >= end_bit) break;
        th.emplace_back([=, &found] { found[(size_t)t] = find_block(base, end, from, to < end_bit ? to : end_bit); });
      }
      for (auto& x : th) x.join();
      for (int t = 1; t < threads_; ++t)
        if (found[(size_t)t] != NONE && found[(size_t)t] > start.back()) start.push_back(found[(size_t)t]);
    }
    const size_t n = start.size();
    const auto t_b = std::chrono::steady_clock::now();
    // ---- 2. decode every piece up to the start of the next one (buffers are kept from round to round)
    if (piece_buf_.size() < n) {
      const size_t old = piece_buf_.size();
      piece_buf_.resize(n);
      for (size_t t = old; t < n; ++t) piece_buf_[t].reset(new Piece);
    }
    auto piece = [&](size_t t) -> Piece& { return *piece_buf_[t]; };
    {
      std::vector<std::thread> th;
      for (size_t t = 0; t < n; ++t) {
        const uint64_t stop = t + 1 < n ? start[t + 1] : start_bit + (uint64_t)threads_ * piece_ * 8;  // (last: the round's nominal end)
        Piece* pc = piece_buf_[t].get();
        const uint64_t from = start[t];
        th.emplace_back([=] {
          if (t == 0) decode_known(base, end, from, stop, history, history_len, *pc);
          else decode_unknown(base, end, from, stop, *pc);
        });
      }
      for (auto& x : th) x.join();
    }
    const auto t_c = std::chrono::steady_clock::now();
    // ---- 3. keep the pieces that follow on from a verified one
    size_t good = 1;
    if (piece(0).status < 0) return (Result)piece(0).status;
    while (good < n && piece(good - 1).status == MkInflate::STOPPED && piece(good - 1).end_bit == start[good] && piece(good).status >= 0) ++good;
    // ---- 4. resolve the place holders: the window before each piece first (a chain), then the pieces at once
    std::vector<size_t> off(good + 1, 0);
    for (size_t t = 0; t < good; ++t) off[t + 1] = off[t] + piece(t).len;
    if (!out_.reserve(off[good] + 8)) return BAD_DATA;
    std::vector<std::vector<uint8_t>> win(good);  // win[t]: the WINDOW bytes before piece t (right-aligned)
    for (size_t t = 1; t < good; ++t) {
      win[t].assign(WINDOW, 0);
      // the last WINDOW bytes of (window before piece t-1) ++ (piece t-1)
      const size_t have = piece(t - 1).len;
      if (t == 1) {
        const uint8_t* p0 = piece(0).text8.p;  // [history window][text]
        memcpy(win[1].data(), p0 + have, WINDOW);   // == last WINDOW bytes of that buffer
      } else if (have >= WINDOW) {
        const uint16_t* s = piece(t - 1).text16.p + WINDOW + have - WINDOW;
        for (size_t i = 0; i < WINDOW; ++i) win[t][i] = s[i] < 256 ? (uint8_t)s[i] : win[t - 1][s[i] - 256];
      } else {
        memcpy(win[t].data(), win[t - 1].data() + have, WINDOW - have);
        const uint16_t* s = piece(t - 1).text16.p + WINDOW;
        for (size_t i = 0; i < have; ++i) win[t][WINDOW - have + i] = s[i] < 256 ? (uint8_t)s[i] : win[t - 1][s[i] - 256];
      }
    }
    {
      std::vector<std::thread> th;
      for (size_t t = 0; t < good; ++t) {
        Piece* pc = piece_buf_[t].get();
        uint8_t* d = out_.p + off[t];
        const uint8_t* w = t ? win[t].data() : nullptr;
        th.emplace_back([=] {
          const size_t m = pc->len;
          if (!w) {
            memcpy(d, pc->text8.p + WINDOW, m);
          } else {
            const uint16_t* s = pc->text16.p + WINDOW;
            resolve(d, s, m, w);
          }
          pc->crc = mk_crc32(0, d, m);  // (each piece's CRC here, in parallel; combined below)
        });
      }
      for (auto& x : th) x.join();
      for (size_t t = 0; t < good; ++t) *text_crc = (uint32_t)crc32_combine(*text_crc, piece(t).crc, (z_off_t)piece(t).len);
    }
    *text = out_.p;
    *text_len = off[good];
    {
      const auto t_d = std::chrono::steady_clock::now();
      auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
      s_find += sec(t_a, t_b);
      s_decode += sec(t_b, t_c);
      s_stitch += sec(t_c, t_d);
      pieces_started += n;
      pieces_kept += good;
    }
    const Piece& last = piece(good - 1);
    *next_bit = last.end_bit;
    if (last.status == MkInflate::STREAM_END) return STREAM_END;
    if (last.status == MkInflate::STOPPED) return MORE;
    return (Result)last.status;  // (an error in the true stream: corrupt or truncated data)
  }

 private:
  static constexpr uint64_t NONE = ~0ull;
  struct Piece {
    MkRawBuf<uint8_t> text8;    // piece 0: [WINDOW bytes of history][text]
    MkRawBuf<uint16_t> text16;  // others:  [WINDOW place holders][elements]
    size_t len = 0;
    int status = MkInflate::BAD_DATA;
    uint64_t end_bit = 0;
    uint32_t crc = 0;
  };

  // 16-bit elements -> text: an element below 256 is its byte, any other names a byte of the 32 KiB before the piece.
  // Sixteen elements at a time (SSE2, part of x86-64): nearly all of them are plain bytes, packed with one instruction;
  // a group that holds a place holder goes element by element.  (The scalar loop ran at 0.6 GB/s per thread and was
  // a fifth of a round.)
  static void resolve(uint8_t* d, const uint16_t* s, size_t m, const uint8_t* w) {
    size_t i = 0;
    const __m128i hi = _mm_set1_epi16((short)0xFF00), zero = _mm_setzero_si128();
    for (; i + 16 <= m; i += 16) {
      const __m128i a = _mm_loadu_si128((const __m128i*)(s + i)), b = _mm_loadu_si128((const __m128i*)(s + i + 8));
      if (_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_and_si128(_mm_or_si128(a, b), hi), zero)) == 0xFFFF) {
        _mm_storeu_si128((__m128i*)(d + i), _mm_packus_epi16(a, b));
      } else {
        for (size_t j = i; j < i + 16; ++j) d[j] = s[j] < 256 ? (uint8_t)s[j] : w[s[j] - 256];
      }
    }
    for (; i < m; ++i) d[i] = s[i] < 256 ? (uint8_t)s[i] : w[s[i] - 256];
  }

  template <class T>
  static void decode_loop(MkInflateT<T>& inf, MkRawBuf<T>& buf, Piece& p, size_t limit) {
    size_t len = 0;
    for (;;) {
      size_t got = 0;
      const auto st = inf.run(buf.p + WINDOW + len, buf.p + buf.cap, buf.p, &got);
      len += got;
      if (st != MkInflateT<T>::OUT_FULL) {
        p.status = (int)st;
        break;
      }
      // (a piece that inflates more than 256-fold is not text worth decoding 16 at a time: refuse rather than
      // let a crafted file take the host's memory; MK_GZ_SERIAL=1 streams such a file through fixed blocks)
      if (buf.cap > limit || !buf.reserve(buf.cap + buf.cap / 2 + 65536)) { p.status = MkInflate::BAD_DATA; break; }
    }
    p.len = len;
    p.end_bit = inf.bit_position();
  }
  void decode_known(const uint8_t* base, const uint8_t* end, uint64_t from, uint64_t stop, const uint8_t* history,
                    size_t history_len, Piece& p) const {
    MkInflateT<uint8_t> inf;
    inf.reset_at_bit(base, end, from);
    inf.set_stop_bit(stop);
    p.status = MkInflate::BAD_DATA;
    p.len = 0;
    if (!p.text8.reserve(WINDOW + piece_ * 5 + 65536)) return;
    memset(p.text8.p, 0, WINDOW);
    if (history_len > WINDOW) { history += history_len - WINDOW; history_len = WINDOW; }
    if (history_len) memcpy(p.text8.p + WINDOW - history_len, history, history_len);
    decode_loop(inf, p.text8, p, (piece_ * 256 > ((size_t)64 << 20) ? piece_ * 256 : ((size_t)64 << 20)) + WINDOW);
  }
  void decode_unknown(const uint8_t* base, const uint8_t* end, uint64_t from, uint64_t stop, Piece& p) const {
    MkInflateT<uint16_t> inf;
    inf.reset_at_bit(base, end, from);
    inf.set_stop_bit(stop);
    p.status = MkInflate::BAD_DATA;
    p.len = 0;
    if (!p.text16.reserve(WINDOW + piece_ * 5 + 65536)) return;
    for (size_t i = 0; i < WINDOW; ++i) p.text16.p[i] = (uint16_t)(256 + i);
    decode_loop(inf, p.text16, p, (piece_ * 256 > ((size_t)64 << 20) ? piece_ * 256 : ((size_t)64 << 20)) + WINDOW);
  }

  // First bit in [from, to) where a non-final dynamic block plausibly starts.
  static uint64_t find_block(const uint8_t* base, const uint8_t* end, uint64_t from, uint64_t to) {
    const uint64_t end_bit = (uint64_t)(end - base) * 8;
    if (to > end_bit) to = end_bit;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    constexpr size_t TRIAL = 16384;
    std::vector<uint16_t> scratch(WINDOW + TRIAL);
    for (size_t i = 0; i < WINDOW; ++i) scratch[i] = (uint16_t)(256 + i);
    MkInflateT<uint16_t> inf;
    for (uint64_t b = from; b + 17 + 4 * 3 < to; ++b) {
      // cheap filters on the raw bits: BFINAL = 0, BTYPE = 2, HLIT <= 29, HDIST <= 29, complete code-length code
      const uint8_t* p = base + (b >> 3);
      if (p + 12 > end) break;
      uint64_t w;
      memcpy(&w, p, 8);
      w >>= (b & 7);  // >= 57 bits
      if ((w & 7) != 4) continue;  // BFINAL 0, BTYPE 10b (read LSB first: bit0 = BFINAL, bits 1..2 = 2)
      const unsigned hlit = (unsigned)(w >> 3) & 31, hdist = (unsigned)(w >> 8) & 31, hclen = ((unsigned)(w >> 13) & 15) + 4;
      if (hlit > 29 || hdist > 29) continue;
      {
        // 17 header bits + up to 19 x 3 = 74 bits: take them from a second load
        uint64_t w2;
        memcpy(&w2, base + ((b + 17) >> 3), 8);
        w2 >>= ((b + 17) & 7);  // >= 57 bits = 19 lengths
        unsigned kraft = 0;
        bool any = false;
        for (unsigned i = 0; i < hclen; ++i) {
          const unsigned l = (unsigned)(w2 >> (3 * i)) & 7;
          if (l) { kraft += 128u >> l; any = true; }
        }
        (void)order;
        if (!any || kraft != 128u) continue;
      }
      inf.reset_at_bit(base, end, b);
      inf.set_strict(true);
      size_t got = 0;
      const auto st = inf.run(scratch.data() + WINDOW, scratch.data() + WINDOW + TRIAL, scratch.data(), &got);
      if (st != MkInflateT<uint16_t>::OUT_FULL && st != MkInflateT<uint16_t>::STREAM_END) continue;
      if (st == MkInflateT<uint16_t>::STREAM_END && got == 0) continue;
      bool text = true;
      for (size_t i = 0; i < got && text; ++i) {
        const unsigned c = scratch[WINDOW + i];
        text = c >= 256 || c == 9 || c == 10 || c == 13 || (c >= 32 && c < 127);
      }
      if (text) return b;
    }
    return NONE;
  }

  int threads_;
  size_t piece_;
  int rounds_ = 0, fail_round_ = -1;
  std::vector<std::unique_ptr<Piece>> piece_buf_;
};

// gzip framing around MkParallelInflate over a whole file in memory: header, rounds of parallel decoding,
// CRC-32 / length trailer, further members, zero padding (as MkGzReader, which it matches output for output).
class MkParallelGunzip {
 public:
  enum Status { MORE = 0, END = 1, BAD_DATA = -1, TRUNCATED = -2, BAD_HEADER = -3, BAD_CRC = -4 };
  MkParallelGunzip(const uint8_t* data, size_t n, int threads, size_t piece_bytes)
      : base_(data), end_(data + n), p_(data), par_(threads, piece_bytes) {}
  int members() const { return members_; }
  const MkParallelInflate& engine() const { return par_; }
  // The next stretch of text (one round of pieces) into `buf` (grown as needed); *text points into it.
  // END: nothing was produced, the file is done.  The caller may hand a different buffer to every call
  // (and so keep using the previous text while the next round runs).
  Status next(MkRawBuf<uint8_t>& buf, const uint8_t** text, size_t* len) {
    *text = nullptr;
    *len = 0;
    if (!in_member_) {
      while (p_ < end_ && *p_ == 0) ++p_;
      if (p_ == end_) return END;
      size_t hl = 0;
      const int h = MkGzReader::header_length(p_, end_, &hl);
      if (h) return h == 1 ? TRUNCATED : BAD_HEADER;
      p_ += hl;
      bit_ = (uint64_t)(p_ - base_) * 8;
      in_member_ = true;
      crc_ = mk_crc32(0, nullptr, 0);
      len_ = 0;
      hist_.clear();
    }
    uint64_t next_bit = 0;
    uint32_t round_crc = 0;
    MkParallelInflate::Result r = MkParallelInflate::BAD_DATA;
    if (!serial_) {
      const std::vector<uint8_t> hist(hist_);  // (the round may hand out the buffer the history points into)
      r = par_.round(base_, end_, bit_, hist.data(), hist.size(), buf, text, len, &next_bit, &round_crc);
      if (r == MkParallelInflate::BAD_DATA) {
        // Whatever the reason (an implausible ratio, damage, a stream the pieces cannot be cut from): go on
        // front to back from the last verified block header, with one decoder that keeps its state from call
        // to call.  If the data is damaged it is this decoder that says so.
        serial_ = true;
        ++serial_fallbacks;
        seq_.reset_at_bit(base_, end_, bit_);
        if (!sbuf_.reserve(MkParallelInflate::WINDOW + SERIAL_STEP)) return BAD_DATA;
        memset(sbuf_.p, 0, MkParallelInflate::WINDOW);
        if (!hist_.empty()) memcpy(sbuf_.p + MkParallelInflate::WINDOW - hist_.size(), hist_.data(), hist_.size());
        slen_ = 0;
      } else if (r < 0) {
        return TRUNCATED;
      }
    }
    if (serial_) {
      // slide: the last 32 KiB written become the window in front of the next stretch
      uint8_t* const w = sbuf_.p;
      if (slen_ >= MkParallelInflate::WINDOW) memcpy(w, w + slen_, MkParallelInflate::WINDOW);  // (regions cannot overlap: slen_ >= WINDOW)
      else if (slen_) memmove(w, w + slen_, MkParallelInflate::WINDOW);
      size_t got = 0;
      const MkInflate::Status st = seq_.run(w + MkParallelInflate::WINDOW, w + MkParallelInflate::WINDOW + SERIAL_STEP, w, &got);
      slen_ = got;
      if (st == MkInflate::BAD_DATA) return BAD_DATA;
      if (st == MkInflate::TRUNCATED) return TRUNCATED;
      if (!buf.reserve(got + 8)) return BAD_DATA;
      memcpy(buf.p, w + MkParallelInflate::WINDOW, got);
      *text = buf.p;
      *len = got;
      round_crc = mk_crc32(0, buf.p, got);
      r = st == MkInflate::STREAM_END ? MkParallelInflate::STREAM_END : MkParallelInflate::MORE;
      next_bit = st == MkInflate::STREAM_END ? (uint64_t)(seq_.input_pos() - base_) * 8 : bit_;
      if (st == MkInflate::STREAM_END) serial_ = false;  // the next member gets the pieces again
    }
    const uint8_t* out = *text;
    const size_t n = *len;
    crc_ = (uint32_t)crc32_combine(crc_, round_crc, (z_off_t)n);
    len_ += n;
    // the last 32 KiB of the member's text so far
    if (n >= MkParallelInflate::WINDOW) {
      hist_.assign(out + n - MkParallelInflate::WINDOW, out + n);
    } else {
      hist_.insert(hist_.end(), out, out + n);
      if (hist_.size() > MkParallelInflate::WINDOW) hist_.erase(hist_.begin(), hist_.end() - MkParallelInflate::WINDOW);
    }
    bit_ = next_bit;
    if (r == MkParallelInflate::STREAM_END) {
      p_ = base_ + ((bit_ + 7) >> 3);
      if (end_ - p_ < 8) return TRUNCATED;
      const uint32_t crc = (uint32_t)p_[0] | ((uint32_t)p_[1] << 8) | ((uint32_t)p_[2] << 16) | ((uint32_t)p_[3] << 24);
      const uint32_t isize = (uint32_t)p_[4] | ((uint32_t)p_[5] << 8) | ((uint32_t)p_[6] << 16) | ((uint32_t)p_[7] << 24);
      if (crc != crc_ || isize != (uint32_t)len_) return BAD_CRC;
      p_ += 8;
      in_member_ = false;
      ++members_;
    }
    return MORE;
  }

 private:
  const uint8_t* base_;
  const uint8_t* end_;
  const uint8_t* p_;
  MkParallelInflate par_;
  bool in_member_ = false;
  uint64_t bit_ = 0;
  uint32_t crc_ = 0;
  uint64_t len_ = 0;
  int members_ = 0;
  std::vector<uint8_t> hist_;
  // front-to-back continuation (see next())
  static constexpr size_t SERIAL_STEP = (size_t)16 << 20;
  bool serial_ = false;
  MkInflate seq_;
  MkRawBuf<uint8_t> sbuf_;
  size_t slen_ = 0;

 public:
  int serial_fallbacks = 0;
};

#endif
