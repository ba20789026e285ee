// mk_inflate.h -- a DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the file reader (host code).
//
// Why not zlib's inflate(): after the GPU offload a '.gz' sample is bound by its one inflating
// thread.  This decoder trades zlib's generality (any input/output chunking)
// for speed on the one shape the reader has -- the whole compressed file contiguous in memory
// (mmap), output into large blocks that follow each other in memory: a 64-bit bit buffer refilled
// with one unaligned load, one 11-bit table lookup per literal/length code (8-bit for distances,
// sub-tables for the longer codes), up to three literals per refill, matches copied eight bytes at a
// time.  Near the end of the input or of an output block it falls back to a careful byte-wise loop,
// so it never reads or writes outside the ranges it was given, and it can stop at the end of an
// output block in the middle of a match and go on in the next one.
// The caller verifies each member's CRC-32 and length (MkGzReader::Member) -- a corrupt file or a
// decoder bug cannot pass silently -- and mk_gunzip (mk_host.cpp) is tested against zlib.
#ifndef MK_INFLATE_H
#define MK_INFLATE_H
#include <stddef.h>
#include <stdint.h>
#include <string.h>

// OutT is the output element: uint8_t for text, uint16_t for the speculative decoding of a stream whose
// first 32 KiB of history are not known yet (MkParallelGunzip below: elements >= 256 name a position in
// that unknown window, and copying them around like any other element is all a match ever does).
template <class OutT>
class MkInflateT {
 public:
  enum Status { OUT_FULL = 0, STREAM_END = 1, STOPPED = 2, BAD_DATA = -1, TRUNCATED = -2 };

  // Start a raw DEFLATE stream at `in` (in_end: end of the readable input).
  void reset(const uint8_t* in, const uint8_t* in_end) { reset_at_bit(in, in_end, 0); }
  // Start at bit `bitpos` of [base, in_end): a block header is expected there.
  void reset_at_bit(const uint8_t* base, const uint8_t* in_end, uint64_t bitpos) {
    base_ = base;
    in_ = base + (bitpos >> 3);
    in_end_ = in_end;
    bitbuf_ = 0;
    bitcnt_ = 0;
    state_ = ST_BLOCK_HEADER;
    last_ = false;
    mlen_ = 0;
    stored_left_ = 0;
    stop_bit_ = ~0ull;
    if (bitpos & 7) {
      refill_safe();
      if (bitcnt_ >= (int)(bitpos & 7)) drop((int)(bitpos & 7));
    }
  }
  // Bit offset (from the base given to reset) of the next unread bit.
  uint64_t bit_position() const { return (uint64_t)(in_ - base_) * 8 - (uint64_t)bitcnt_; }
  // run() returns STOPPED instead of reading a block header at (or, having missed it, beyond) this bit.
  void set_stop_bit(uint64_t bit) { stop_bit_ = bit; }
  // Accept only what a compressor writes: complete Huffman codes (strict), for the search for a block start.
  void set_strict(bool on) { strict_ = on; }
  bool at_block_header() const { return state_ == ST_BLOCK_HEADER; }
  bool final_block_seen() const { return last_; }
  // First input byte not yet consumed once STREAM_END was returned (the stream's padding bits dropped).
  const uint8_t* input_pos() const { return in_ - (bitcnt_ >> 3); }

  // Decode into [out, out_end).  `window` is the lowest address a match may reach back to (the 32 KiB
  // before `out` belong to the same stream when they lie at or above it).  *produced = bytes written.
  Status run(OutT* out, OutT* out_end, const OutT* window, size_t* produced) {
    OutT* const out0 = out;
    Status rc = OUT_FULL;
    for (;;) {
      if (state_ == ST_MATCH) {  // a match cut by the end of the previous output block
        while (mlen_ && out < out_end) { *out = *(out - mdist_); ++out; --mlen_; }
        if (mlen_) break;
        state_ = ST_HUFF;
      }
      if (state_ == ST_BLOCK_HEADER) {
        if (last_) { rc = STREAM_END; break; }
        if (bit_position() >= stop_bit_) { rc = STOPPED; break; }
        if (!need(3)) { rc = TRUNCATED; break; }
        last_ = take(1) != 0;
        const unsigned type = take(2);
        if (type == 0) {
          drop(bitcnt_ & 7);  // to the byte boundary
          if (!need(32)) { rc = TRUNCATED; break; }
          const unsigned len = take(16), nlen = take(16);
          if ((len ^ nlen) != 0xFFFFu) { rc = BAD_DATA; break; }
          // hand the bytes still in the bit buffer back to the input: stored data is copied directly
          in_ -= bitcnt_ >> 3;
          bitbuf_ = 0;
          bitcnt_ = 0;
          stored_left_ = len;
          state_ = ST_STORED;
        } else if (type == 1) {
          build_fixed();
          state_ = ST_HUFF;
        } else if (type == 2) {
          const int e = read_dynamic();
          if (e) { rc = e < 0 ? BAD_DATA : TRUNCATED; break; }
          state_ = ST_HUFF;
        } else {
          rc = BAD_DATA;
          break;
        }
      }
      if (state_ == ST_STORED) {
        size_t n = stored_left_;
        if ((size_t)(in_end_ - in_) < n) { rc = TRUNCATED; break; }
        if ((size_t)(out_end - out) < n) n = (size_t)(out_end - out);
        if (sizeof(OutT) == 1) memcpy(out, in_, n);
        else for (size_t q = 0; q < n; ++q) out[q] = (OutT)in_[q];
        out += n;
        in_ += n;
        stored_left_ -= (unsigned)n;
        if (stored_left_) break;  // output block full
        state_ = ST_BLOCK_HEADER;
        continue;
      }
      if (state_ == ST_HUFF) {
        const int e = huff(out, out_end, window);
        if (e == H_BLOCK_END) { state_ = ST_BLOCK_HEADER; continue; }
        if (e == H_OUT_FULL) break;
        rc = e == H_TRUNCATED ? TRUNCATED : BAD_DATA;
        break;
      }
    }
    *produced = (size_t)(out - out0);
    return rc;
  }

 private:
  enum { ST_BLOCK_HEADER, ST_STORED, ST_HUFF, ST_MATCH };
  enum { H_BLOCK_END = 1, H_OUT_FULL = 2, H_BAD = -1, H_TRUNCATED = -2 };
  static constexpr int LBITS = 11, DBITS = 8;
  // table entry: bits 0..7 code length (or, for a sub-table pointer, the primary index width),
  // 8..12 extra bits / sub-table index width, 16..30 value (literal, base length, base distance,
  // sub-table offset), flags above.  An all-zero entry is an unused code: invalid data.
  static constexpr uint32_t F_LIT = 0x80000000u, F_EOB = 0x40000000u, F_SUB = 0x20000000u;
  // the distance table needs bits 16..30 for its base (up to 24577): its sub-table flag is bit 31
  static constexpr uint32_t F_DSUB = 0x80000000u, INVALID = 0xFFFFFFFFu;
  // literal/length table only: two literals in one entry (bits 16..23 first byte, 8..15 second byte,
  // 24..27 the first code's length, 0..7 both lengths together)
  static constexpr uint32_t F_DBL = 0x10000000u;

  // ---- bit input -------------------------------------------------------------------------
  static inline uint64_t load64(const uint8_t* p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;  // little-endian host (x86-64)
  }
  inline void refill_fast() {  // needs in_ + 8 <= in_end_
    bitbuf_ |= load64(in_) << bitcnt_;
    in_ += (63 - bitcnt_) >> 3;
    bitcnt_ |= 56;
  }
  inline void refill_safe() {
    while (bitcnt_ < 56 && in_ < in_end_) {  // (at most 63 bits: refill_fast shifts by the count)
      bitbuf_ |= (uint64_t)*in_++ << bitcnt_;
      bitcnt_ += 8;
    }
  }
  inline bool need(int n) {
    if (bitcnt_ < n) refill_safe();
    return bitcnt_ >= n;
  }
  inline unsigned take(int n) {
    const unsigned v = (unsigned)(bitbuf_ & ((1ull << n) - 1));
    bitbuf_ >>= n;
    bitcnt_ -= n;
    return v;
  }
  inline void drop(int n) {
    bitbuf_ >>= n;
    bitcnt_ -= n;
  }

  // ---- tables ----------------------------------------------------------------------------
  static inline unsigned reverse_bits(unsigned code, int len) {
    unsigned r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
  }
  // Canonical Huffman decode table from code lengths. `entry_of(sym)` gives the entry without its
  // length byte.  Returns false for an over-subscribed set; an incomplete set leaves zero entries.
  template <class EntryOf>
  static bool build(const uint8_t* lens, int nsyms, int primary_bits, uint32_t* table, int table_cap, uint32_t sub_flag,
                    EntryOf entry_of, bool* complete = nullptr, int* used_codes = nullptr) {
    int count[16] = {0};
    for (int s = 0; s < nsyms; ++s) count[lens[s]]++;
    count[0] = 0;
    int maxlen = 15;
    while (maxlen > 0 && !count[maxlen]) --maxlen;
    long left = 1;
    for (int l = 1; l <= 15; ++l) {
      left = (left << 1) - count[l];
      if (left < 0) return false;
    }
    if (complete) *complete = left == 0;
    if (used_codes) { *used_codes = 0; for (int l = 1; l <= 15; ++l) *used_codes += count[l]; }
    unsigned next[16];
    {
      unsigned code = 0;
      for (int l = 1; l <= 15; ++l) { code = (code + (unsigned)count[l - 1]) << 1; next[l] = code; }
    }
    const int psize = 1 << primary_bits;
    memset(table, 0, sizeof(uint32_t) * (size_t)psize);
    int used = psize;
    // sub-table width per primary index: the longest code that shares it
    uint8_t subw[1 << LBITS];
    if (maxlen > primary_bits) {
      memset(subw, 0, (size_t)psize);
      unsigned nx[16];
      memcpy(nx, next, sizeof nx);
      for (int s = 0; s < nsyms; ++s) {
        const int l = lens[s];
        if (l > primary_bits) {
          const unsigned rc = reverse_bits(nx[l], l);
          const unsigned idx = rc & (unsigned)(psize - 1);
          if (l - primary_bits > subw[idx]) subw[idx] = (uint8_t)(l - primary_bits);
        }
        if (l) nx[l]++;
      }
      for (int i = 0; i < psize; ++i) {
        if (!subw[i]) continue;
        const int size = 1 << subw[i];
        if (used + size > table_cap) return false;
        memset(table + used, 0, sizeof(uint32_t) * (size_t)size);
        table[i] = sub_flag | ((uint32_t)used << 16) | ((uint32_t)subw[i] << 8) | (uint32_t)primary_bits;
        used += size;
      }
    }
    for (int s = 0; s < nsyms; ++s) {
      const int l = lens[s];
      if (!l) continue;
      const unsigned rc = reverse_bits(next[l]++, l);
      const uint32_t body = entry_of(s);
      if (l <= primary_bits) {
        const uint32_t e = body == INVALID ? 0u : (body | (uint32_t)l);
        for (unsigned i = rc; i < (unsigned)psize; i += 1u << l) table[i] = e;
      } else {
        const unsigned idx = rc & (unsigned)(psize - 1);
        const uint32_t p = table[idx];
        const int w = (int)((p >> 8) & 31);
        uint32_t* sub = table + ((p >> 16) & 0x1FFF);
        const uint32_t e = body == INVALID ? 0u : (body | (uint32_t)(l - primary_bits));
        for (unsigned i = rc >> primary_bits; i < (1u << w); i += 1u << (l - primary_bits)) sub[i] = e;
      }
    }
    return true;
  }
  static uint32_t litlen_entry(int s) {
    static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    if (s < 256) return F_LIT | ((uint32_t)s << 16);
    if (s == 256) return F_EOB;
    if (s > 285) return INVALID;  // 286, 287: never valid in data
    return ((uint32_t)base[s - 257] << 16) | ((uint32_t)extra[s - 257] << 8);
  }
  static uint32_t dist_entry(int s) {
    static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    if (s > 29) return INVALID;
    return ((uint32_t)base[s] << 16) | ((uint32_t)extra[s] << 8);
  }
  // Where two literal codes fit into one primary index, let the entry deliver both (FASTA text is mostly
  // literals with 2..4-bit codes: half the lookups).
  void pair_literals() {
#ifdef MK_INFLATE_NO_PAIRS
    return;
#endif
    for (unsigned i = 0; i < (1u << LBITS); ++i) {
      const uint32_t e1 = lt_[i];
      if ((e1 & (F_LIT | F_SUB | F_DBL)) != F_LIT) continue;
      const unsigned l1 = e1 & 0xFF;
      if (l1 >= LBITS) continue;
      const uint32_t e2 = lt_[i >> l1];  // the bits after the first code (zero-filled above LBITS - l1)
      if ((e2 & (F_LIT | F_SUB)) != F_LIT) continue;
      const unsigned l2 = (e2 & F_DBL) ? ((e2 >> 24) & 15) : (e2 & 0xFF);
      if (l1 + l2 > LBITS) continue;  // the second code would need bits this index does not have
      pend_[i] = F_LIT | F_DBL | (l1 << 24) | (e1 & 0x00FF0000u) | (((e2 >> 16) & 0xFF) << 8) | (l1 + l2);
      mark_[i >> 5] |= 1u << (i & 31);
    }
    for (unsigned i = 0; i < (1u << LBITS); ++i)
      if (mark_[i >> 5] & (1u << (i & 31))) lt_[i] = pend_[i];
    memset(mark_, 0, sizeof mark_);
  }
  void build_fixed() {
    uint8_t lens[288 + 32];
    int i = 0;
    for (; i < 144; ++i) lens[i] = 8;
    for (; i < 256; ++i) lens[i] = 9;
    for (; i < 280; ++i) lens[i] = 7;
    for (; i < 288; ++i) lens[i] = 8;
    for (int d = 0; d < 32; ++d) lens[288 + d] = 5;
    build(lens, 288, LBITS, lt_, LT_CAP, F_SUB, litlen_entry);
    build(lens + 288, 32, DBITS, dt_, DT_CAP, F_DSUB, dist_entry);
    pair_literals();
  }
  // 0 ok, -1 bad data, 1 truncated
  int read_dynamic() {
    if (!need(14)) return 1;
    const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
    if (hlit > 286 || hdist > 30) return -1;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) {
      if (!need(3)) return 1;
      cl[order[i]] = (uint8_t)take(3);
    }
    uint32_t ct[128];
    bool full = false;
    if (!build(cl, 19, 7, ct, 128, 0u, [](int s) { return (uint32_t)s << 16; }, &full)) return -1;
    if (strict_ && !full) return -1;
    uint8_t lens[286 + 30 + 138];
    int n = 0;
    const int total = hlit + hdist;
    while (n < total) {
      if (!need(7 + 7)) {  // (a code of up to 7 bits and up to 7 extra bits; the stream may simply end here)
        if (bitcnt_ == 0) return 1;
      }
      const uint32_t e = ct[bitbuf_ & 127];
      const int l = (int)(e & 0xFF);
      if (!l) return -1;
      if (l > bitcnt_) return 1;
      drop(l);
      const int sym = (int)((e >> 16) & 0xFF);
      if (sym < 16) {
        lens[n++] = (uint8_t)sym;
      } else {
        int rep, val = 0, xb;
        if (sym == 16) {
          if (!n) return -1;
          val = lens[n - 1];
          xb = 2;
          rep = 3;
        } else if (sym == 17) {
          xb = 3;
          rep = 3;
        } else {
          xb = 7;
          rep = 11;
        }
        if (bitcnt_ < xb) return 1;
        rep += (int)take(xb);
        if (n + rep > total) return -1;
        while (rep--) lens[n++] = (uint8_t)val;
      }
    }
    if (!lens[256]) return -1;  // no end-of-block code
    int ndist = 0;
    if (!build(lens, hlit, LBITS, lt_, LT_CAP, F_SUB, litlen_entry, &full)) return -1;
    if (strict_ && !full) return -1;
    if (!build(lens + hlit, hdist, DBITS, dt_, DT_CAP, F_DSUB, dist_entry, &full, &ndist)) return -1;
    if (strict_ && !full && ndist > 1) return -1;  // (zlib writes a lone distance code as an incomplete set)
    pair_literals();
    return 0;
  }

  // ---- the symbol loop -------------------------------------------------------------------
  int huff(OutT*& out_ref, OutT* out_end, const OutT* window) {
    OutT* out = out_ref;
    constexpr int E = 8 / (int)sizeof(OutT);  // elements per 8-byte copy
    int rc = 0;
    // fast part: room for one whole step without checks (3 literals or a 258-byte match copied in
    // 8-byte pieces; 8 input bytes per refill, two refills per step at most)
    while (out_end - out >= 6 + 258 + 8 && in_end_ - in_ >= 16) {
      refill_fast();
      uint32_t e = lt_[bitbuf_ & ((1u << LBITS) - 1)];
      if (e & F_SUB) {
        drop(LBITS);
        e = lt_[((e >> 16) & 0x1FFF) + (bitbuf_ & ((1u << ((e >> 8) & 31)) - 1))];
      }
      if (e & F_LIT) {
        // up to three entries (one or two literals each) from one refill: <= 33 of the >= 56 bits
        drop((int)(e & 0xFF));
        out[0] = (OutT)(uint8_t)(e >> 16);
        out[1] = (OutT)(uint8_t)(e >> 8);
        out += 1 + ((e >> 28) & 1);
        e = lt_[bitbuf_ & ((1u << LBITS) - 1)];
        if ((e & (F_LIT | F_SUB)) == F_LIT) {
          drop((int)(e & 0xFF));
          out[0] = (OutT)(uint8_t)(e >> 16);
          out[1] = (OutT)(uint8_t)(e >> 8);
          out += 1 + ((e >> 28) & 1);
          e = lt_[bitbuf_ & ((1u << LBITS) - 1)];
          if ((e & (F_LIT | F_SUB)) == F_LIT) {
            drop((int)(e & 0xFF));
            out[0] = (OutT)(uint8_t)(e >> 16);
            out[1] = (OutT)(uint8_t)(e >> 8);
            out += 1 + ((e >> 28) & 1);
          }
        }
        continue;
      }
      const int cl = (int)(e & 0xFF);
      if (!cl) { rc = H_BAD; break; }
      drop(cl);
      if (e & F_EOB) { rc = H_BLOCK_END; break; }
      const int lx = (int)((e >> 8) & 31);
      unsigned len = ((e >> 16) & 0x1FF) + (unsigned)(bitbuf_ & ((1u << lx) - 1));
      drop(lx);
      if (bitcnt_ < 32) refill_fast();
      uint32_t d = dt_[bitbuf_ & ((1u << DBITS) - 1)];
      if (d & F_DSUB) {
        drop(DBITS);
        d = dt_[((d >> 16) & 0x3FF) + (bitbuf_ & ((1u << ((d >> 8) & 31)) - 1))];
      }
      const int dl = (int)(d & 0xFF);
      if (!dl) { rc = H_BAD; break; }
      drop(dl);
      const int dx = (int)((d >> 8) & 31);
      const unsigned dist = ((d >> 16) & 0x7FFF) + (unsigned)(bitbuf_ & ((1u << dx) - 1));
      drop(dx);
      if ((size_t)(out - window) < dist) { rc = H_BAD; break; }
      const OutT* src = out - dist;
      OutT* const end = out + len;
      if (dist >= (unsigned)E) {
        do {
          memcpy(out, src, 8);
          out += E;
          src += E;
        } while (out < end);
      } else {
        do { *out++ = *src++; } while (out < end);
      }
      out = end;
    }
    // careful part: one symbol at a time, every bound checked
    while (!rc) {
      refill_safe();
      uint32_t e = lt_[bitbuf_ & ((1u << LBITS) - 1)];
      int used = 0;
      if (e & F_SUB) {
        used = LBITS;
        e = lt_[((e >> 16) & 0x1FFF) + ((bitbuf_ >> LBITS) & ((1u << ((e >> 8) & 31)) - 1))];
      }
      const int cl = (e & F_DBL) ? (int)((e >> 24) & 15) : (int)(e & 0xFF);  // (of a pair, only the first literal)
      if (!cl) { rc = bitcnt_ < 15 && in_ >= in_end_ ? H_TRUNCATED : H_BAD; break; }
      if (used + cl > bitcnt_) { rc = H_TRUNCATED; break; }
      // (an end-of-block code needs no room: a text that fills its buffer exactly still ends cleanly)
      if (!(e & F_EOB) && out == out_end) { rc = H_OUT_FULL; break; }
      drop(used + cl);
      if (e & F_LIT) { *out++ = (OutT)(uint8_t)(e >> 16); continue; }
      if (e & F_EOB) { rc = H_BLOCK_END; break; }
      const int lx = (int)((e >> 8) & 31);
      if (bitcnt_ < lx) refill_safe();
      if (bitcnt_ < lx) { rc = H_TRUNCATED; break; }
      unsigned len = ((e >> 16) & 0x1FF) + take(lx);
      refill_safe();
      uint32_t d = dt_[bitbuf_ & ((1u << DBITS) - 1)];
      used = 0;
      if (d & F_DSUB) {
        used = DBITS;
        d = dt_[((d >> 16) & 0x3FF) + ((bitbuf_ >> DBITS) & ((1u << ((d >> 8) & 31)) - 1))];
      }
      const int dl = (int)(d & 0xFF);
      if (!dl) { rc = bitcnt_ < 15 && in_ >= in_end_ ? H_TRUNCATED : H_BAD; break; }
      const int dx = (int)((d >> 8) & 31);
      if (used + dl + dx > bitcnt_) { rc = H_TRUNCATED; break; }
      drop(used + dl);
      const unsigned dist = ((d >> 16) & 0x7FFF) + take(dx);
      if ((size_t)(out - window) < dist) { rc = H_BAD; break; }
      while (len && out < out_end) { *out = *(out - dist); ++out; --len; }
      if (len) {  // the output block ends inside the match
        mlen_ = len;
        mdist_ = dist;
        state_ = ST_MATCH;
        rc = H_OUT_FULL;
      }
    }
    out_ref = out;
    return rc;
  }

  static constexpr int LT_CAP = (1 << LBITS) + 1024, DT_CAP = (1 << DBITS) + 512;
  const uint8_t* base_ = nullptr;
  const uint8_t* in_ = nullptr;
  const uint8_t* in_end_ = nullptr;
  uint64_t stop_bit_ = ~0ull;
  bool strict_ = false;
  uint64_t bitbuf_ = 0;
  int bitcnt_ = 0;
  int state_ = ST_BLOCK_HEADER;
  bool last_ = false;
  unsigned mlen_ = 0, mdist_ = 0, stored_left_ = 0;
  uint32_t lt_[LT_CAP];
  uint32_t dt_[DT_CAP];
  uint32_t pend_[1 << LBITS];
  uint32_t mark_[(1 << LBITS) / 32] = {0};
};

using MkInflate = MkInflateT<uint8_t>;

// gzip member framing around MkInflate over a whole file in memory: header, deflate stream, CRC-32 and
// ISIZE trailer, any number of members, zero padding between/after them skipped (as gzip.py does).
class MkGzReader {
 public:
  enum Status { MORE = 0, END = 1, BAD_DATA = -1, TRUNCATED = -2, BAD_HEADER = -3 };
  MkGzReader(const uint8_t* data, size_t n) : p_(data), end_(data + n) {}
  // Length of the gzip member header at p: 0 ok (*len set), 1 truncated, -1 not a gzip member.
  static int header_length(const uint8_t* p, const uint8_t* end, size_t* len) {
    if (end - p < 10) return (end - p >= 2 && !(p[0] == 0x1f && p[1] == 0x8b)) ? -1 : 1;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return -1;
    const unsigned flg = p[3];
    const uint8_t* q = p + 10;
    if (flg & 4) {
      if (end - q < 2) return 1;
      const size_t xlen = (size_t)q[0] | ((size_t)q[1] << 8);
      q += 2;
      if ((size_t)(end - q) < xlen) return 1;
      q += xlen;
    }
    for (unsigned bit = 8; bit <= 16; bit <<= 1) {  // FNAME, FCOMMENT: zero-terminated
      if (flg & bit) {
        const uint8_t* z = (const uint8_t*)memchr(q, 0, (size_t)(end - q));
        if (!z) return 1;
        q = z + 1;
      }
    }
    if (flg & 2) {
      if (end - q < 2) return 1;
      q += 2;
    }
    *len = (size_t)(q - p);
    return 0;
  }
  int members() const { return members_; }
  // The trailer of the member that ended inside the last fill() (valid while member_ended()):
  bool member_ended() const { return ended_; }
  uint32_t member_crc() const { return crc_; }
  uint32_t member_isize() const { return isize_; }

  // Fill [out, out_end) with the next bytes of the text; stops early at the end of a member
  // (member_ended() is then set: the caller checks CRC/length of the bytes since the previous one).
  Status fill(uint8_t* out, uint8_t* out_end, const uint8_t* window, size_t* produced) {
    *produced = 0;
    ended_ = false;
    if (!in_member_) {
      while (p_ < end_ && *p_ == 0) ++p_;  // padding
      if (p_ == end_) return END;
      const int h = header();
      if (h) return h == 1 ? TRUNCATED : BAD_HEADER;
      inf_.reset(p_, end_);
      in_member_ = true;
    }
    const MkInflate::Status s = inf_.run(out, out_end, window, produced);
    if (s == MkInflate::OUT_FULL) return MORE;
    if (s == MkInflate::BAD_DATA) return BAD_DATA;
    if (s == MkInflate::TRUNCATED) return TRUNCATED;
    p_ = inf_.input_pos();
    if (end_ - p_ < 8) return TRUNCATED;
    crc_ = (uint32_t)p_[0] | ((uint32_t)p_[1] << 8) | ((uint32_t)p_[2] << 16) | ((uint32_t)p_[3] << 24);
    isize_ = (uint32_t)p_[4] | ((uint32_t)p_[5] << 8) | ((uint32_t)p_[6] << 16) | ((uint32_t)p_[7] << 24);
    p_ += 8;
    in_member_ = false;
    ended_ = true;
    ++members_;
    return MORE;
  }

 private:
  int header() {  // 0 ok, 1 truncated, -1 not gzip
    size_t len = 0;
    const int rc = header_length(p_, end_, &len);
    if (rc == 0) p_ += len;
    return rc;
  }
  const uint8_t* p_;
  const uint8_t* end_;
  MkInflate inf_;
  bool in_member_ = false, ended_ = false;
  uint32_t crc_ = 0, isize_ = 0;
  int members_ = 0;
};

#endif
