// mk_ingest.hip -- native ingest: one FASTA file (plain or gzip) -> chunks -> contexts.
//
// Replaces, for one sample file, the reference's chunk_files + one countKmers task per chunk +
// the dict sum (bin/mercat2.py:86-106, 112-127) and the file reading of find_kmers
// (lib/mercat2_kmers.py:47-50) without writing chunk files: reader threads fill a ring of pinned
// host blocks (pread for plain files, zlib inflate for '.gz'), the calling thread runs the
// streaming Chunker rule over the blocks (mk_cutscan.h) and copies each byte range straight into
// the raw buffer of the context that owns the current chunk (hipMemcpyAsync on that context's
// stream), and one worker thread per context runs mk_chunk_end while the next chunk is being read.
// Host code only; all GPU work is behind the per-context calls of mk_api.hip.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "mk_common.h"
#include "mk_cutscan.h"
#include "mk_inflate.h"
#include "mk_crc32.h"
#include "mk_pgunzip.h"
#include <sys/mman.h>

namespace {

using Clock = std::chrono::steady_clock;
static double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

// ---------------------------------------------------------------------------------- the ring
constexpr size_t MK_GZ_WINDOW = 32768;  // how far back a DEFLATE match may reach
struct Ring {
  size_t block = 0;
  int slots = 0;
  uint8_t* mem = nullptr;  // pinned, slots * block bytes
  std::vector<size_t> len;
  std::vector<char> has_cr;
  std::vector<uint64_t> holds;  // slot -> 1 + index of the block it holds (0 = none)
  std::mutex mu;
  std::condition_variable cv;
  uint64_t released = 0;            // blocks [0, released) may be overwritten
  uint64_t total = UINT64_MAX;      // number of blocks of the text, once known
  std::atomic<uint64_t> next{0};    // next block index a reader takes
  bool abort = false;
  int rc = MK_OK;
  std::string err;

  uint8_t* at(uint64_t i) const { return mem + (size_t)(i % (uint64_t)slots) * block; }
  void fail(int code, const std::string& what) {
    std::lock_guard<std::mutex> g(mu);
    if (rc == MK_OK) { rc = code; err = what; }
    abort = true;
    cv.notify_all();
  }
  // reader side: wait until block i may be written; false = give up
  bool wait_writable(uint64_t i) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return abort || i < released + (uint64_t)slots; });
    return !abort;
  }
  void publish(uint64_t i, size_t n, bool cr) {
    std::lock_guard<std::mutex> g(mu);
    const int s = (int)(i % (uint64_t)slots);
    len[s] = n;
    has_cr[s] = cr;
    holds[s] = i + 1;
    cv.notify_all();
  }
  void set_total(uint64_t t) {
    std::lock_guard<std::mutex> g(mu);
    total = t;
    cv.notify_all();
  }
  // consumer side: 1 = block i is ready, 0 = the text ended before block i, <0 = error
  int wait_ready(uint64_t i) {
    std::unique_lock<std::mutex> g(mu);
    cv.wait(g, [&] { return abort || holds[i % (uint64_t)slots] == i + 1 || i >= total; });
    if (abort) return rc ? rc : MK_ERR_STATE;
    return holds[i % (uint64_t)slots] == i + 1 ? 1 : 0;
  }
  void release_upto(uint64_t r) {
    std::lock_guard<std::mutex> g(mu);
    if (r > released) released = r;
    cv.notify_all();
  }
};

static bool read_fully(int fd, uint8_t* dst, size_t want, off_t off, size_t* got) {
  size_t done = 0;
  while (done < want) {
    const ssize_t r = pread(fd, dst + done, want - done, off + (off_t)done);
    if (r < 0) return false;
    if (r == 0) break;
    done += (size_t)r;
  }
  *got = done;
  return true;
}

// plain file: block i is bytes [i*block, (i+1)*block) of the file; any thread may read any block
static void plain_reader(Ring* R, int fd, uint64_t nblocks, const std::string* path) {
  for (;;) {
    const uint64_t i = R->next.fetch_add(1);
    if (i >= nblocks) return;
    if (!R->wait_writable(i)) return;
    size_t got = 0;
    if (!read_fully(fd, R->at(i), R->block, (off_t)(i * R->block), &got)) {
      R->fail(MK_ERR_IO, "read " + *path + ": " + strerror(errno));
      return;
    }
    R->publish(i, got, got && memchr(R->at(i), '\r', got) != nullptr);
  }
}

// gzip: one thread decodes member after member (as Python's gzip module does) into successive blocks,
// with the reader's own decoder (mk_inflate.h, about twice zlib's speed on FASTA) over the mmap-ed file.
// Matches reach back up to 32 KiB: the blocks follow each other in memory, and when the ring wraps its
// last 32 KiB are copied in front of its first block.  Every member's CRC-32 and length are checked, as
// gzip.py does (carry-less-multiply CRC, mk_crc32.h, over the bytes just written while they are in cache).
static void gz_reader(Ring* R, const uint8_t* file, size_t file_len, const std::string* path, int* members_out) {
  MkGzReader rd(file, file_len);
  const uint8_t* const window = R->mem - MK_GZ_WINDOW;
  const size_t ring_bytes = (size_t)R->slots * R->block;
  uint64_t i = 0;
  bool done = false;
  uint32_t crc = mk_crc32(0, nullptr, 0);
  uint64_t member_len = 0;
  while (!done) {
    if (!R->wait_writable(i)) break;
    const int s = (int)(i % (uint64_t)R->slots);
    if (s == 0 && i > 0) memcpy(R->mem - MK_GZ_WINDOW, R->mem + ring_bytes - MK_GZ_WINDOW, MK_GZ_WINDOW);
    uint8_t* const base = R->at(i);
    size_t fill = 0;
    while (fill < R->block) {
      size_t got = 0;
      const MkGzReader::Status st = rd.fill(base + fill, base + R->block, window, &got);
      crc = mk_crc32(crc, base + fill, got);
      member_len += got;
      fill += got;
      if (st == MkGzReader::END) { done = true; break; }
      if (st != MkGzReader::MORE) {
        R->fail(MK_ERR_IO, *path + (st == MkGzReader::TRUNCATED ? ": gzip stream is truncated"
                                    : st == MkGzReader::BAD_HEADER ? ": not a gzip file (or data after the last member)"
                                                                   : ": corrupt gzip data"));
        done = true;
        fill = 0;
        break;
      }
      if (rd.member_ended()) {
        if (crc != rd.member_crc() || (uint32_t)member_len != rd.member_isize()) {
          R->fail(MK_ERR_IO, *path + ": gzip CRC check failed");
          done = true;
          fill = 0;
          break;
        }
        crc = mk_crc32(0, nullptr, 0);
        member_len = 0;
      }
    }
    if (fill) {
      R->publish(i, fill, memchr(base, '\r', fill) != nullptr);
      ++i;
    }
  }
  *members_out = rd.members();
  R->set_total(i);
}

// The same file through several threads (mk_pgunzip.h: block starts found by search, unknown history
// carried as place holders, every piece verified against the one before it).  This thread runs the
// rounds and copies their text into the ring.
static void gz_parallel_reader(Ring* R, const uint8_t* file, size_t file_len, const std::string* path, int* members_out,
                               int threads) {
  MkParallelGunzip rd(file, file_len, threads, (size_t)4 << 20);
  MkRawBuf<uint8_t> buf[2];  // the text of one round is copied into the ring while the next round is decoded
  std::thread copier;
  std::atomic<bool> stop{false};
  uint64_t i = 0;            // next ring block (touched by the copier only, one copier at a time)
  double s_copy = 0;
  for (unsigned r = 0;; ++r) {
    const uint8_t* text = nullptr;
    size_t n = 0;
    const MkParallelGunzip::Status st = rd.next(buf[r & 1], &text, &n);
    if (copier.joinable()) copier.join();  // (it read the other buffer, which the next round will overwrite)
    if (st == MkParallelGunzip::END || stop.load()) break;
    if (st != MkParallelGunzip::MORE) {
      R->fail(MK_ERR_IO, *path + (st == MkParallelGunzip::TRUNCATED ? ": gzip stream is truncated"
                                  : st == MkParallelGunzip::BAD_HEADER ? ": not a gzip file (or data after the last member)"
                                  : st == MkParallelGunzip::BAD_CRC  ? ": gzip CRC check failed"
                                                                     : ": corrupt gzip data"));
      break;
    }
    copier = std::thread([R, text, n, &i, &stop, &s_copy] {
      const auto t0 = Clock::now();
      for (size_t off = 0; off < n; off += R->block) {
        const size_t m = n - off < R->block ? n - off : R->block;
        if (!R->wait_writable(i)) { stop.store(true); return; }
        memcpy(R->at(i), text + off, m);
        R->publish(i, m, memchr(text + off, '\r', m) != nullptr);
        ++i;
      }
      s_copy += seconds_since(t0);
    });
  }
  if (copier.joinable()) copier.join();
  if (getenv("MK_VERBOSE"))
    fprintf(stderr, "[mk] parallel gunzip: find %.3f s, decode %.3f s, stitch %.3f s, copy+wait %.3f s (overlapped); pieces %zu started, %zu kept\n",
            rd.engine().s_find, rd.engine().s_decode, rd.engine().s_stitch, s_copy, rd.engine().pieces_started, rd.engine().pieces_kept);
  *members_out = rd.members();
  R->set_total(i);
}

// BGZF (bgzip): a gzip file of independent members of at most 64 KiB, each announcing its own size in
// an extra header field ("BC", BSIZE = member bytes - 1), so the member boundaries are known without
// decoding: consecutive members are grouped into jobs of at most one ring block of text, and any reader
// thread decodes any job (no history crosses a member).  plan is empty when the file is not BGZF all
// the way through -- the sequential reader takes it then.
struct BgzfJob {
  size_t in_off, in_len;  // members [in_off, in_off + in_len) of the file
  size_t out_len;         // sum of their ISIZE fields
};
static std::vector<BgzfJob> bgzf_plan(const uint8_t* f, size_t n, size_t block) {
  std::vector<BgzfJob> jobs;
  size_t p = 0;
  BgzfJob cur{0, 0, 0};
  while (p < n) {
    if (n - p < 18 + 8 || f[p] != 0x1f || f[p + 1] != 0x8b || f[p + 2] != 8 || !(f[p + 3] & 4)) return {};
    const size_t xlen = (size_t)f[p + 10] | ((size_t)f[p + 11] << 8);
    if (n - p < 12 + xlen) return {};
    size_t bsize = 0;
    for (size_t q = p + 12; q + 4 <= p + 12 + xlen;) {  // sub-fields: SI1 SI2 LEN(2) data
      const size_t len = (size_t)f[q + 2] | ((size_t)f[q + 3] << 8);
      if (f[q] == 'B' && f[q + 1] == 'C' && len == 2 && q + 6 <= p + 12 + xlen) bsize = ((size_t)f[q + 4] | ((size_t)f[q + 5] << 8)) + 1;
      q += 4 + len;
    }
    if (bsize < 12 + xlen + 8 + 2 || p + bsize > n) return {};
    const uint8_t* t = f + p + bsize - 4;
    const size_t isize = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    if (isize > 65536 || isize > block) return {};
    if (cur.in_len && cur.out_len + isize > block) {
      jobs.push_back(cur);
      cur = BgzfJob{p, 0, 0};
    }
    if (!cur.in_len) cur.in_off = p;
    cur.in_len += bsize;
    cur.out_len += isize;
    p += bsize;
  }
  if (cur.in_len) jobs.push_back(cur);
  // (jobs without text -- e.g. the empty end-of-file member on its own -- are kept: they cost nothing)
  return jobs;
}

static void bgzf_reader(Ring* R, const uint8_t* file, const std::vector<BgzfJob>* jobs, const std::string* path,
                        std::atomic<int>* members_out) {
  for (;;) {
    const uint64_t i = R->next.fetch_add(1);
    if (i >= jobs->size()) return;
    if (!R->wait_writable(i)) return;
    const BgzfJob& job = (*jobs)[i];
    uint8_t* const base = R->at(i);
    MkGzReader rd(file + job.in_off, job.in_len);
    size_t fill = 0, member_start = 0;
    bool ok = true;
    for (;;) {
      size_t got = 0;
      const MkGzReader::Status st = rd.fill(base + fill, base + job.out_len, base, &got);
      fill += got;
      if (st == MkGzReader::END) break;
      if (st != MkGzReader::MORE) { ok = false; break; }
      if (rd.member_ended()) {
        if (mk_crc32(0, base + member_start, fill - member_start) != rd.member_crc() ||
            (uint32_t)(fill - member_start) != rd.member_isize()) { ok = false; break; }
        member_start = fill;
      } else if (got == 0) { ok = false; break; }  // more text than the trailers announced
    }
    if (!ok || fill != job.out_len) {
      R->fail(MK_ERR_IO, *path + ": corrupt gzip data (BGZF block)");
      return;
    }
    members_out->fetch_add(rd.members());
    R->publish(i, fill, fill && memchr(base, '\r', fill) != nullptr);
  }
}

// ------------------------------------------------------------------------- context workers
struct Lane {  // one context, its worker thread and the hand-over between dispatcher and worker
  mk_ctx* c = nullptr;
  std::mutex mu;
  std::condition_variable cv;
  enum { IDLE, FEEDING, ENDING, QUIT } state = IDLE;
  uint64_t min_count = 0;
  int rc = MK_OK;
  std::thread th;
};

static void lane_worker(Lane* L) {
  for (;;) {
    std::unique_lock<std::mutex> g(L->mu);
    L->cv.wait(g, [&] { return L->state == Lane::ENDING || L->state == Lane::QUIT; });
    if (L->state == Lane::QUIT) return;
    g.unlock();
    const int rc = mk_chunk_end(L->c, L->min_count);
    g.lock();
    if (rc && !L->rc) L->rc = rc;
    L->state = Lane::IDLE;
    L->cv.notify_all();
  }
}

struct Dispatcher : MkCutSink {
  std::vector<Lane>* lanes = nullptr;
  Ring* ring = nullptr;
  int cur = -1;          // lane that owns the open chunk (-1: none open)
  int next_lane = 0;
  uint64_t chunks = 0;
  uint64_t min_count = 0;
  size_t reserve = 0;    // raw-buffer bytes to set aside when a chunk is opened
  double s_wait_gpu = 0;
  double s_feed = 0;     // seconds inside the copy calls (hipMemcpyAsync out of the pinned ring, raw-buffer growth)
  std::vector<int> touched;  // lanes that received bytes of the block being processed
  int rc_lane = -1;

  int open_chunk() {
    Lane& L = (*lanes)[next_lane];
    {
      const auto t0 = Clock::now();
      std::unique_lock<std::mutex> g(L.mu);
      L.cv.wait(g, [&] { return L.state == Lane::IDLE; });
      s_wait_gpu += seconds_since(t0);
      if (L.rc) { rc_lane = next_lane; return L.rc; }
      L.state = Lane::FEEDING;
    }
    cur = next_lane;
    next_lane = (next_lane + 1) % (int)lanes->size();
    int rc = mk_chunk_begin(L.c);
    if (!rc && reserve) rc = mk_reserve_raw(L.c, reserve);
    if (rc) rc_lane = cur;
    return rc;
  }
  int close_chunk() {
    if (cur < 0) return MK_OK;
    Lane& L = (*lanes)[cur];
    {
      std::lock_guard<std::mutex> g(L.mu);
      L.min_count = min_count;
      L.state = Lane::ENDING;
      L.cv.notify_all();
    }
    cur = -1;
    ++chunks;
    return MK_OK;
  }
  int feed(const uint8_t* p, size_t n) override {
    if (n == 0) return MK_OK;
    int rc;
    if (cur < 0 && (rc = open_chunk())) return rc;
    mk_ctx* c = (*lanes)[cur].c;
    const bool pinned = p >= ring->mem && p < ring->mem + (size_t)ring->slots * ring->block;
    const auto t0 = Clock::now();
    rc = mk_feed_host_async(c, p, n, /*wait=*/!pinned);
    s_feed += seconds_since(t0);
    if (rc) { rc_lane = cur; return rc; }
    if (pinned && (touched.empty() || touched.back() != cur)) touched.push_back(cur);
    return MK_OK;
  }
  int cut(uint64_t) override { return close_chunk(); }
};

}  // namespace

extern "C" int mk_count_file(mk_ctx* const* ctxs, int nctx, const char* path, uint64_t chunk_bytes, uint64_t min_count,
                             int threads, mk_file_stats_t* st) {
  if (!ctxs || nctx < 1 || !ctxs[0]) return MK_ERR_ARG;
  mk_ctx* c0 = ctxs[0];
  if (!path) { c0->err = "mk_count_file: path is NULL"; return MK_ERR_ARG; }
  for (int j = 0; j < nctx; ++j) {
    if (!ctxs[j]) { c0->err = "mk_count_file: a context is NULL"; return MK_ERR_ARG; }
    if (ctxs[j]->in_chunk) { c0->err = "mk_count_file: a chunk is open"; return MK_ERR_STATE; }
    if (ctxs[j]->alphabet != c0->alphabet || ctxs[j]->k != c0->k || ctxs[j]->canonical != c0->canonical) {
      c0->err = "mk_count_file: contexts differ in alphabet, k or canonical mode";
      return MK_ERR_ARG;
    }
    for (int i = 0; i < j; ++i)
      if (ctxs[i] == ctxs[j]) { c0->err = "mk_count_file: the same context twice"; return MK_ERR_ARG; }
  }
  const auto t_begin = Clock::now();
  const std::string spath(path);
  const int fd = open(path, O_RDONLY | O_CLOEXEC);
  if (fd < 0) { c0->err = "open " + spath + ": " + strerror(errno); return MK_ERR_IO; }
  struct stat sb;
  if (fstat(fd, &sb) != 0) { c0->err = "stat " + spath + ": " + strerror(errno); close(fd); return MK_ERR_IO; }
  const uint64_t disk = (uint64_t)sb.st_size;
  // '.gz' iff the last suffix says so: the reference's test (lib/mercat2_kmers.py:47, lib/mercat2_Chunker.py:42)
  const bool gz = spath.size() >= 3 && spath.compare(spath.size() - 3, 3, ".gz") == 0;
  // chunk iff the ON-DISK size reaches the chunk size (bin/mercat2.py:101)
  const bool chunked = chunk_bytes > 0 && disk >= chunk_bytes;
  // the GPUs the contexts are on, in order of first appearance
  // (MK_DEVICE_PER_CONTEXT=1, for tests on a one-GPU box: every context counts as a GPU of its own, so the several-GPU
  // code path -- per-GPU leaders, mk_merge_devices, the split of a single filter unit -- runs with device list [0, 0, ..])
  const bool each_own = getenv("MK_DEVICE_PER_CONTEXT") != nullptr;
  auto group_of = [&](int j) { return each_own ? -1 - j : ctxs[j]->device; };
  std::vector<int> devs;
  for (int j = 0; j < nctx; ++j)
    if (std::find(devs.begin(), devs.end(), group_of(j)) == devs.end()) devs.push_back(group_of(j));
  // One filter unit (not chunked) on several GPUs: counted in pieces cut at record starts, unfiltered, filtered after
  // the sum -- when it is large enough to be worth the larger merge.  MK_SPLIT_MIN = bytes of text.
  uint64_t text_guess = disk;
  if (gz && disk >= 18) {
    uint8_t tail[4];
    text_guess = disk * 4;
    if (pread(fd, tail, 4, (off_t)disk - 4) == 4) {  // ISIZE of the last member: exact for a one-member file below 4 GiB
      const uint64_t isize = (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) | ((uint64_t)tail[3] << 24);
      text_guess = std::max<uint64_t>(isize, disk);
    }
  }
  const uint64_t split_min = getenv("MK_SPLIT_MIN") ? (uint64_t)atoll(getenv("MK_SPLIT_MIN")) : ((uint64_t)64 << 20);
  const bool split = !chunked && devs.size() > 1 && text_guess >= split_min && c0->mode != MK_MODE_BYREF;
  const uint64_t piece_bytes = split ? std::max<uint64_t>((text_guess + (uint64_t)nctx - 1) / (uint64_t)nctx, std::min<uint64_t>(split_min, (uint64_t)8 << 20)) : 0;
  const int lanes_n = (chunked || split) ? nctx : 1;

  const bool auto_threads = threads <= 0;
  if (auto_threads) threads = (int)std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
  if (hipSetDevice(c0->device) != hipSuccess) { c0->err = "hipSetDevice failed"; close(fd); return MK_ERR_HIP; }

  Ring R;
  {
    const char* e = getenv("MK_INGEST_BLOCK");  // (tests shrink the blocks to put every boundary case in reach)
    R.block = e && atoll(e) > 0 ? (size_t)atoll(e) : ((size_t)4 << 20);
    if (!(e && atoll(e) > 0)) {
      // small samples: pinning the ring costs ~0.4 ms per MiB, more than reading the file -- size it to the text
      const uint64_t text_guess = gz ? disk * 6 : disk;
      while (R.block > ((size_t)64 << 10) && (uint64_t)R.block * 2 > text_guess + R.block / 2) R.block >>= 1;
      if (auto_threads && text_guess <= ((uint64_t)8 << 20)) threads = 1;
    }
  }
  void* gz_map = nullptr;
  bool gz_parallel = false;
  std::vector<BgzfJob> bgzf;
  if (gz) {
    if (disk) {
      gz_map = mmap(nullptr, (size_t)disk, PROT_READ, MAP_PRIVATE, fd, 0);
      if (gz_map == MAP_FAILED) { c0->err = "mmap " + spath + ": " + strerror(errno); close(fd); return MK_ERR_IO; }
      bgzf = bgzf_plan((const uint8_t*)gz_map, (size_t)disk, R.block);
      (void)madvise(gz_map, (size_t)disk, bgzf.empty() ? MADV_SEQUENTIAL : MADV_WILLNEED);
    }
    // one DEFLATE stream: several threads pay off from a few pieces on (MK_GZ_SERIAL=1: always front to back)
    // (decoding scales further than reading: measured 1.1 s with 8 threads, 0.75 s with 16 on the 1.6 GB S2 file)
    if (auto_threads) threads = (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (bgzf.empty()) {
      gz_parallel = threads > 1 && disk >= ((size_t)16 << 20) && !getenv("MK_GZ_SERIAL");
      if (!gz_parallel) threads = 1;
    }
  }
  {
    R.slots = 2 * threads + 4;
    // gzip: the decoder reads up to 32 KiB back across the blocks, which must therefore still be in the ring
    if (gz && (size_t)R.slots * R.block < 2 * MK_GZ_WINDOW + 2 * R.block)
      R.slots = (int)((2 * MK_GZ_WINDOW + R.block - 1) / R.block) + 2;
    const size_t bytes = (size_t)R.slots * R.block + MK_GZ_WINDOW;  // (+ room in front for the window at the wrap)
    if (c0->ingest_ring_bytes < bytes) {
      if (c0->ingest_ring) (void)hipHostFree(c0->ingest_ring);
      c0->ingest_ring = nullptr;
      c0->ingest_ring_bytes = 0;
      // (portable: every GPU the contexts are on copies out of it)
      const hipError_t he = hipHostMalloc(&c0->ingest_ring, bytes, hipHostMallocPortable);
      if (he != hipSuccess) {
        c0->err = "hipHostMalloc(" + std::to_string(bytes) + "): " + hipGetErrorString(he);
        if (gz_map) (void)munmap(gz_map, (size_t)disk);
        close(fd);
        return MK_ERR_NOMEM;
      }
      c0->ingest_ring_bytes = bytes;
    }
    R.mem = (uint8_t*)c0->ingest_ring + MK_GZ_WINDOW;
    R.len.assign(R.slots, 0);
    R.has_cr.assign(R.slots, 0);
    R.holds.assign(R.slots, 0);
  }

  // raw-buffer size to set aside per chunk, so that feeding never has to grow it mid-chunk
  size_t reserve = 0;
  if (chunked) {
    reserve = (size_t)chunk_bytes + ((size_t)8 << 20);
  } else if (split) {
    reserve = (size_t)piece_bytes + ((size_t)8 << 20);
  } else if (!gz) {
    reserve = (size_t)disk + 64;
  } else if (disk >= 18) {
    uint8_t tail[4];
    if (pread(fd, tail, 4, (off_t)disk - 4) == 4) {  // ISIZE of the last member: a hint, nothing more
      const uint64_t isize = (uint64_t)tail[0] | ((uint64_t)tail[1] << 8) | ((uint64_t)tail[2] << 16) | ((uint64_t)tail[3] << 24);
      reserve = (size_t)std::max<uint64_t>(isize, disk) + 64;
    }
  }

  std::vector<Lane> lanes(lanes_n);
  for (int j = 0; j < lanes_n; ++j) {
    lanes[j].c = ctxs[j];
    lanes[j].th = std::thread(lane_worker, &lanes[j]);
  }
  std::vector<hipEvent_t> events((size_t)R.slots * lanes_n, nullptr);
  std::vector<char> ev_set((size_t)R.slots * lanes_n, 0);
  int rc = MK_OK;
  for (size_t e = 0; e < events.size(); ++e)  // (an event belongs to the device that is current when it is made)
    if (hipSetDevice(lanes[e % (size_t)lanes_n].c->device) != hipSuccess ||
        hipEventCreateWithFlags(&events[e], hipEventDisableTiming) != hipSuccess) rc = MK_ERR_HIP;

  int members = 0;
  std::atomic<int> bgzf_members{0};
  std::vector<std::thread> readers;
  uint64_t nblocks = 0;
  if (rc == MK_OK) {
    if (gz && !bgzf.empty()) {
      R.total = bgzf.size();
      for (int t = 0; t < threads && (size_t)t < bgzf.size(); ++t)
        readers.emplace_back(bgzf_reader, &R, (const uint8_t*)gz_map, &bgzf, &spath, &bgzf_members);
    } else if (gz && gz_parallel) {
      readers.emplace_back(gz_parallel_reader, &R, (const uint8_t*)gz_map, (size_t)disk, &spath, &members, threads);
    } else if (gz) {
      readers.emplace_back(gz_reader, &R, (const uint8_t*)gz_map, (size_t)disk, &spath, &members);
    } else {
      nblocks = (disk + R.block - 1) / R.block;
      R.total = nblocks;
      for (int t = 0; t < threads && (uint64_t)t < std::max<uint64_t>(nblocks, 1); ++t)
        readers.emplace_back(plain_reader, &R, fd, nblocks, &spath);
    }
  }

  Dispatcher D;
  D.lanes = &lanes;
  D.ring = &R;
  D.min_count = split ? 0 : min_count;  // (pieces of one filter unit: the filter comes after the sum)
  D.reserve = reserve;
  MkCutScanner scan(chunked ? chunk_bytes : (split ? piece_bytes : UINT64_MAX), &D, /*record_starts_only=*/split);
  uint64_t text_bytes = 0;
  double s_wait_io = 0, s_block = 0, s_retire = 0;
  const double s_setup = seconds_since(t_begin);
  const uint64_t lag = (uint64_t)R.slots / 2;
  auto retire = [&](uint64_t i) {  // wait for the copies out of block i, then let the readers have its slot
    const auto t0 = Clock::now();
    struct Acc { double& a; Clock::time_point t; ~Acc() { a += seconds_since(t); } } acc{s_retire, t0};
    for (int j = 0; j < lanes_n; ++j) {
      const size_t e = (size_t)(i % (uint64_t)R.slots) * lanes_n + j;
      if (ev_set[e]) { (void)hipEventSynchronize(events[e]); ev_set[e] = 0; }
    }
    R.release_upto(i + 1);
  };
  uint64_t i = 0;
  for (; rc == MK_OK; ++i) {
    const auto t0 = Clock::now();
    const int ready = R.wait_ready(i);
    s_wait_io += seconds_since(t0);
    if (ready < 0) { rc = ready; break; }
    if (ready == 0) break;
    const int s = (int)(i % (uint64_t)R.slots);
    const size_t n = R.len[s];
    D.touched.clear();
    const auto tb = Clock::now();
    rc = scan.block(R.at(i), n, R.has_cr[s] != 0);
    s_block += seconds_since(tb);
    if (rc != MK_OK) break;
    text_bytes += n;
    for (int j : D.touched) {
      const size_t e = (size_t)s * lanes_n + j;
      if (hipSetDevice(lanes[j].c->device) != hipSuccess || hipEventRecord(events[e], lanes[j].c->stream) != hipSuccess) { rc = MK_ERR_HIP; break; }
      ev_set[e] = 1;
    }
    if (i >= lag) retire(i - lag);
  }
  const auto t_drain = Clock::now();
  if (rc == MK_OK) rc = scan.finish();
  if (rc == MK_OK && D.cur < 0 && D.chunks == 0) rc = D.open_chunk();  // an empty file is one empty chunk
  if (rc == MK_OK) rc = D.close_chunk();
  if (rc != MK_OK) R.fail(rc, "");
  // drain: copies first (the readers may be waiting for slots), then the readers, then the workers
  for (uint64_t r = (i > lag ? i - lag : 0); r < i; ++r) retire(r);
  for (auto& t : readers) t.join();
  if (rc == MK_OK && R.rc != MK_OK) rc = R.rc;
  for (int j = 0; j < lanes_n; ++j) {
    Lane& L = lanes[j];
    {
      std::unique_lock<std::mutex> g(L.mu);
      if (L.state == Lane::FEEDING) {  // a chunk left open by an error: drop it
        L.c->in_chunk = false;
        L.c->raw_len = 0;
        L.state = Lane::IDLE;
      }
      L.cv.wait(g, [&] { return L.state == Lane::IDLE; });
      L.state = Lane::QUIT;
      L.cv.notify_all();
    }
    L.th.join();
    if (rc == MK_OK && L.rc != MK_OK) { rc = L.rc; D.rc_lane = j; }
  }
  for (auto& e : events)
    if (e) (void)hipEventDestroy(e);
  if (gz_map) (void)munmap(gz_map, (size_t)disk);
  close(fd);
  if (rc != MK_OK) {
    if (!R.err.empty()) c0->err = R.err;
    else if (D.rc_lane > 0) c0->err = ctxs[D.rc_lane]->err;
    return rc;
  }
  const double s_drain = seconds_since(t_drain);
  // the sample's table ends up in ctxs[0]: the contexts of one GPU are summed on that GPU into the first of them,
  // then the GPUs' tables are summed into ctxs[0] (peer copies, mk_multi.hip)
  const auto t_merge = Clock::now();
  std::vector<mk_ctx*> leaders;
  std::vector<int> leader_group;
  for (int j = 0; j < lanes_n; ++j) {
    mk_ctx* lead = nullptr;
    for (size_t l = 0; l < leaders.size(); ++l)
      if (leader_group[l] == group_of(j)) lead = leaders[l];
    if (!lead) { leaders.push_back(ctxs[j]); leader_group.push_back(group_of(j)); continue; }
    if ((rc = mk_merge_from(lead, ctxs[j])) != MK_OK) { if (lead != c0) c0->err = lead->err; return rc; }
    if ((rc = mk_reset(ctxs[j])) != MK_OK) { c0->err = ctxs[j]->err; return rc; }
  }
  if (leaders.size() > 1 && (rc = mk_merge_devices(leaders.data(), (int)leaders.size(), MK_MERGE_GATHER, nullptr)) != MK_OK) return rc;
  if (split && (rc = mk_filter_min(c0, min_count)) != MK_OK) return rc;
  const double s_merge = seconds_since(t_merge);
  if (st) {
    memset(st, 0, sizeof *st);
    st->disk_bytes = disk;
    st->text_bytes = text_bytes;
    st->chunks = split ? 1 : D.chunks;
    st->gz = gz ? 1 : 0;
    st->chunked = chunked ? 1 : 0;
    st->members = members + bgzf_members.load();
    st->threads = threads;
    st->contexts = lanes_n;
    st->devices = (int)leaders.size();
    st->split_pieces = split ? (int)D.chunks : 0;
    st->s_merge = s_merge;
    st->s_wait_io = s_wait_io;
    st->s_wait_gpu = D.s_wait_gpu;
    st->s_total = seconds_since(t_begin);
    st->s_setup = s_setup;
    st->s_feed = D.s_feed;
    st->s_scan = s_block - D.s_feed - D.s_wait_gpu;  // (the dispatcher's own work on the text: the Chunker rule)
    st->s_retire = s_retire;
    st->s_drain = s_drain;
  }
  return MK_OK;
}
