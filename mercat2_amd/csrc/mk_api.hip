// mk_api.hip -- the C ABI (include/mercat_hip.h): context, chunk pipeline, export.
//
// Chunk pipeline == one reference find_kmers call (lib/mercat2_kmers.py:32-78):
//   raw bytes -> parse -> [pack] -> count (dense | hash64 | by-reference) -> keep count >= min_count
//   -> add into the running table (the dict sum of run_mercat2, bin/mercat2.py:121-127).
// Export == sorted(kmers.items()) + the TSV print loop (bin/mercat2.py:128-137).
#include "mk_common.h"
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

typedef unsigned long long u64;

static thread_local std::string g_err;  // errors that have no context to live in

static size_t pow2_at_least(size_t v) {
  size_t p = 1024;
  while (p < v) p <<= 1;
  return p;
}

static void drain_table_users(mk_ctx* t);  // (mk_share_table, below)

// ------------------------------------------------------------------------------ buffers
int mk_buf_reserve(mk_ctx* c, MkDevBuf& b, size_t bytes, bool keep) {
  if (bytes <= b.cap) return MK_OK;
  size_t want = bytes;
  if (keep && b.cap) want = std::max(bytes, b.cap + b.cap / 2);
  want = (want + 255) & ~(size_t)255;
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    c->err = "hipMalloc(" + std::to_string(want) + " bytes): " + hipGetErrorString(e);
    return MK_ERR_NOMEM;
  }
  if (b.p) {
    if (keep) {
      e = hipMemcpyAsync(p, b.p, b.cap, hipMemcpyDeviceToDevice, c->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
      if (e != hipSuccess) {
        (void)hipFree(p);
        c->err = std::string("hipMemcpy (grow): ") + hipGetErrorString(e);
        return MK_ERR_HIP;
      }
    } else {
      (void)hipStreamSynchronize(c->stream);
    }
    (void)hipFree(b.p);
  }
  b.p = p;
  b.cap = want;
  return MK_OK;
}

static void buf_free(MkDevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

// ---------------------------------------------------------------------------- profiling
static hipEvent_t get_event(mk_ctx* c) {
  if (!c->event_pool.empty()) {
    hipEvent_t e = c->event_pool.back();
    c->event_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void mk_prof_begin(mk_ctx* c, int id) {
  if (!c->profile) return;
  MkEventPair p{get_event(c), get_event(c), id};
  (void)hipEventRecord(p.a, c->stream);
  c->events.push_back(p);
}

void mk_prof_end(mk_ctx* c) {
  if (!c->profile || c->events.empty()) return;
  (void)hipEventRecord(c->events.back().b, c->stream);
}

static void prof_collect(mk_ctx* c) {
  if (c->events.empty()) return;
  (void)hipStreamSynchronize(c->stream);
  double* ms[MK_K_NUM] = {&c->st.ms_parse, &c->st.ms_pack, &c->st.ms_count, &c->st.ms_exotic, &c->st.ms_filter, &c->st.ms_export, &c->st.ms_part};
  uint64_t* nn[MK_K_NUM] = {&c->st.n_parse, &c->st.n_pack, &c->st.n_count, &c->st.n_exotic, &c->st.n_filter, &c->st.n_export, &c->st.n_part};
  for (auto& p : c->events) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p.a, p.b) == hipSuccess) {
      *ms[p.id] += t;
      *nn[p.id] += 1;
    }
    c->event_pool.push_back(p.a);
    c->event_pool.push_back(p.b);
  }
  c->events.clear();
}

// ----------------------------------------------------------------------------- lifetime
// "mercat_hip <abi>.<minor> (gfx950)": the ABI number changes whenever a struct or a signature of include/mercat_hip.h does
// (native.py checks it against its own MK_ABI before it trusts the struct layouts)
extern "C" const char* mk_version(void) { return "mercat_hip 4.0 (gfx950)"; }

extern "C" int mk_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n < 0 ? 0 : n;
}

extern "C" const char* mk_last_error(const mk_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }

extern "C" int mk_words_per_key(const mk_ctx* c) { return c ? (c->mode == MK_MODE_HASH128 ? 2 : 1) : 0; }

// Row totals of the last merge that were read back without waiting (process_chunk_fast): add them up. Only call
// when the stream is known to have passed that copy.
static void fold_pending(mk_ctx* c) {
  if (!c->pending_rows) return;
  const MkChunkInfo* p = c->h_info + 1;
  if (c->mode == MK_MODE_HASH128) c->run128_rows += (size_t)p->new_rows;
  else c->run_rows += (size_t)p->new_rows;
  c->run_ref_rows += (size_t)p->new_rows_ref;
  c->pending_rows = false;
}

static int pull_info(mk_ctx* c) {
  if (c->clean_mode && c->clean_meta.p) MK_HIP(hipMemcpyAsync(c->h_clean, c->clean_meta.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipMemcpyAsync(c->h_info, c->info.p, sizeof(MkChunkInfo), hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  fold_pending(c);  // (the stream is idle: whatever was in flight has landed)
  return MK_OK;
}

// Before anything reads run_rows & co.
static int settle(mk_ctx* c) {
  if (!c->pending_rows) return MK_OK;
  MK_HIP(hipSetDevice(c->device));
  MK_HIP(hipStreamSynchronize(c->stream));
  fold_pending(c);
  return MK_OK;
}

int mk_settle(mk_ctx* c) { return settle(c); }
int mk_pull_info(mk_ctx* c) { return pull_info(c); }

extern "C" int mk_create(int device, int alphabet, int k, mk_ctx** out) {
  if (!out) { g_err = "mk_create: out is NULL"; return MK_ERR_ARG; }
  *out = nullptr;
  if (k < 1 || k > (1 << 20)) { g_err = "mk_create: k must be in [1, 2^20]"; return MK_ERR_ARG; }
  if (alphabet != MK_ALPHABET_NT2 && alphabet != MK_ALPHABET_AA5 && alphabet != MK_ALPHABET_RAW) {
    g_err = "mk_create: unknown alphabet";
    return MK_ERR_ARG;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_err = std::string("mk_create: no HIP device (") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
            "); this engine has no CPU fallback";
    return MK_ERR_HIP;
  }
  if (device < 0 || device >= ndev) { g_err = "mk_create: device index out of range"; return MK_ERR_ARG; }
  mk_ctx* c = new (std::nothrow) mk_ctx();
  if (!c) { g_err = "mk_create: out of host memory"; return MK_ERR_NOMEM; }
  c->device = device;
  c->alphabet = alphabet;
  c->k = k;
  c->bits = alphabet == MK_ALPHABET_NT2 ? 2 : (alphabet == MK_ALPHABET_AA5 ? 5 : 0);
  c->syms_per_word = alphabet == MK_ALPHABET_NT2 ? 32 : 12;
  const long kb = (long)k * c->bits;
  c->mode = (c->bits == 0) ? MK_MODE_BYREF : (kb <= 15 ? MK_MODE_DENSE : (kb <= 64 ? MK_MODE_HASH64 : MK_MODE_BYREF));
  if (alphabet == MK_ALPHABET_NT2 && k >= 33 && k <= 64) c->mode = MK_MODE_HASH128;  // packed by-reference
  // amino acids, 13 <= k <= 25: 5 k <= 125 bits, the same two-word tables (mk_count_ref128aa_k); MK_NO_AA128 = by bytes, as before round 3
  if (alphabet == MK_ALPHABET_AA5 && k >= 13 && k <= 25 && !getenv("MK_NO_AA128")) c->mode = MK_MODE_HASH128;
  c->st.mode = c->mode;
  c->use_partition = getenv("MK_NO_PARTITION") ? 0 : 1;
  c->use_fast_parse = getenv("MK_NO_FAST_PARSE") ? 0 : 1;
  c->use_superkmer = getenv("MK_NO_SUPERKMER") ? 0 : 1;
  c->use_superkmer2 = getenv("MK_NO_SUPERKMER2") ? 0 : 1;
  c->run_bucket_major = (getenv("MK_BUCKET_MAJOR") && alphabet == MK_ALPHABET_NT2 && c->mode == MK_MODE_HASH64 && k >= 12 && k <= 32) ? 1 : 0;
  if (const char* e = getenv("MK_SK_MIN_K")) { const int v = atoi(e); if (v >= 12 && v <= 33) c->sk_min_k = v; }
  int rc = MK_OK;
  auto fail = [&](int code, const std::string& msg) {
    g_err = msg;
    mk_destroy(c);
    return code;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return fail(MK_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
    return fail(MK_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  c->use_speculation = getenv("MK_NO_SPECULATION") ? 0 : 1;
  c->use_reuse = getenv("MK_NO_REUSE") ? 0 : 1;
  c->use_fused = getenv("MK_NO_FUSE") ? 0 : 1;
  // What a new context assumes about its first chunk (later chunks go by the chunk before): two windows per distinct key,
  // five per record.  The count kernels plan their sub-range passes from it; a bucket that does not fit is split anyway.
  // (1 and 8 -- every window a new key, full records -- made the first chunk of a read set take eight passes per bucket,
  // 1.35 ms instead of 0.3.)
  c->dup_hint = 2.0;
  c->nk_hint = 5.0;
  if ((e = hipHostMalloc((void**)&c->h_info, 2 * sizeof(MkChunkInfo) + 8 * sizeof(u64), hipHostMallocDefault)) != hipSuccess)
    return fail(MK_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e));
  c->h_clean = (u64*)(c->h_info + 2);
  if ((rc = mk_buf_reserve(c, c->info, sizeof(MkChunkInfo) + 64)) != MK_OK) return fail(rc, c->err);
  if (c->mode == MK_MODE_DENSE) {
    const size_t bytes = ((size_t)1 << kb) * sizeof(u64);
    if ((rc = mk_buf_reserve(c, c->ctab, bytes)) != MK_OK) return fail(rc, c->err);
    if ((rc = mk_buf_reserve(c, c->run, bytes)) != MK_OK) return fail(rc, c->err);
    (void)hipMemsetAsync(c->ctab.p, 0, bytes, c->stream);
    (void)hipMemsetAsync(c->run.p, 0, bytes, c->stream);
    c->ctab_slots = c->run_slots = (size_t)1 << kb;
  }
  if ((e = hipStreamSynchronize(c->stream)) != hipSuccess)
    return fail(MK_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
  *out = c;
  return MK_OK;
}

extern "C" void mk_destroy(mk_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->share_owner) (void)mk_share_table(c, nullptr);
  for (mk_ctx* s : std::vector<mk_ctx*>(c->sharers)) (void)mk_share_table(s, nullptr);  // (their rows in this table go with it)
  for (auto& p : c->events) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  for (auto& e : c->event_pool) (void)hipEventDestroy(e);
  MkDevBuf* all[] = {&c->raw, &c->seq, &c->codes, &c->bad, &c->tile_maps, &c->info, &c->ctab, &c->rtab_chunk, &c->run,
                     &c->run_ref, &c->arena, &c->run128, &c->ex128, &c->ex128_out, &c->ex_keys, &c->ex_cnts, &c->ex_keys2, &c->ex_cnts2, &c->ex_tmp, &c->part, &c->part_meta, &c->surv_keys, &c->surv_cnts, &c->surv_keys2,
                     &c->xfer_out, &c->xfer_in, &c->xfer_meta, &c->clean_meta, &c->clean_runs};
  for (auto* b : all) buf_free(*b);
  if (c->h_info) (void)hipHostFree(c->h_info);
  if (c->ingest_ring) (void)hipHostFree(c->ingest_ring);
  if (c->tsv_pin) { (void)hipHostUnregister(c->tsv_pin); free(c->tsv_pin); }
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// expect_rows != 0: the packed table is also SIZED for about that many distinct keys when that is less than it has now
// (never more: growing is what the imports do) -- after a merge across GPUs an owner keeps 1/N of the rows, and a table
// that fits the caches takes the imports several times faster than the sample-sized one it had
static int reset_impl(mk_ctx* c, size_t expect_rows) {
  if (!c) return MK_ERR_ARG;
  MK_HIP(hipSetDevice(c->device));
  if (c->pending_rows) { MK_HIP(hipStreamSynchronize(c->stream)); c->pending_rows = false; }
  // (a table other contexts launch into: nobody launches while it is cleared, and what was launched has finished.  The
  // ORDER of a sharer's chunks against this reset is the caller's: reset the owner before the sharers count the next sample)
  std::unique_lock<std::shared_mutex> table_lock(c->table_mu, std::defer_lock);
  if (!c->sharers.empty()) { table_lock.lock(); drain_table_users(c); }
  if (expect_rows && c->mode != MK_MODE_DENSE) {
    const size_t want = pow2_at_least(4 * expect_rows);
    if (c->run_slots > want) c->run_slots = want;
    if (c->run128_slots > want) c->run128_slots = want;
  }
  if (c->mode == MK_MODE_DENSE) {
    MK_HIP(hipMemsetAsync(c->run.p, 0, c->run_slots * sizeof(u64), c->stream));
  } else if (c->run_slots) {
    int rc = mk_launch_clear_slots(c, (MkSlot*)c->run.p, c->run_slots);
    if (rc) return rc;
  }
  if (c->run_ref_slots) {
    int rc = mk_launch_clear_slots(c, (MkSlot*)c->run_ref.p, c->run_ref_slots);
    if (rc) return rc;
  }
  if (c->run128_slots) MK_HIP(hipMemsetAsync(c->run128.p, 0, c->run128_slots * sizeof(MkSlot128), c->stream));
  c->run_rows = 0;
  c->run_ref_rows = 0;
  c->run128_rows = 0;
  c->run_side = 0;
  c->in_chunk = false;
  c->raw_len = 0;
  c->part_reuse_ok = false;  // (a new sample sizes its own bucket regions: nothing is inherited across samples)
  c->dup_known = false;
  // (surv_hint stays, like dup_hint and nk_hint: a context that is reset counts the next sample, and the survivors of the
  // last sample's last full chunk are the best guess there is for its first chunk -- the fused launch spills what a wrong
  // guess leaves no room for.  Only a new context has no guess and hands its first chunk's survivors over through regions.)
  c->fuse_cap = 0;
  c->clean_n_runs = c->clean_n_bytes = c->clean_gc = c->clean_symbols = c->clean_raw = c->clean_headers = c->clean_last_runs = 0;
  MK_HIP(hipStreamSynchronize(c->stream));
  return MK_OK;
}
extern "C" int mk_reset(mk_ctx* c) { return reset_impl(c, 0); }
extern "C" int mk_reset_for(mk_ctx* c, uint64_t expect_rows) { return reset_impl(c, (size_t)(expect_rows ? expect_rows : 1)); }

extern "C" int mk_set_canonical(mk_ctx* c, int on) {
  if (!c) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (on && c->alphabet != MK_ALPHABET_NT2) { c->err = "mk_set_canonical: only the nucleotide alphabet has a reverse complement"; return MK_ERR_ARG; }
  if (on && c->mode != MK_MODE_DENSE && c->mode != MK_MODE_HASH64 && !(c->mode == MK_MODE_HASH128 && c->use_superkmer2)) {
    c->err = "mk_set_canonical: canonical counting is implemented for nucleotide k <= 64 (two-word keys: on the partitioned path only)";
    return MK_ERR_ARG;
  }
  c->part_reuse_ok = false;  // (bucket regions sized in the other mode are not inherited)
  if (c->in_chunk || c->run_rows || c->run_ref_rows || c->run128_rows || c->run_side || c->st.chunks) {
    if ((on != 0) != (c->canonical != 0) && (c->run_rows || c->run_ref_rows || c->run128_rows || c->run_side || c->in_chunk)) {
      c->err = "mk_set_canonical: the running table already holds rows counted in the other mode (mk_reset first)";
      return MK_ERR_STATE;
    }
  }
  c->canonical = on ? 1 : 0;
  return MK_OK;
}

extern "C" int mk_set_clean(mk_ctx* c, int on, int toupper) {
  if (!c) return MK_ERR_ARG;
  if (on && c->alphabet != MK_ALPHABET_NT2) { c->err = "mk_set_clean: removeN applies to nucleotide FASTA (bin/mercat2.py:276)"; return MK_ERR_ARG; }
  if (c->in_chunk) { c->err = "mk_set_clean: a chunk is open"; return MK_ERR_STATE; }
  c->clean_mode = on ? 1 : 0;
  c->clean_upper = (on && toupper) ? 1 : 0;
  return MK_OK;
}

extern "C" int mk_clean_stats(mk_ctx* c, mk_clean_gpu_t* out) {
  if (!c || !out) return MK_ERR_ARG;
  out->raw_bytes = c->clean_raw;
  out->symbols = c->clean_symbols;
  out->gc_count = c->clean_gc;
  out->n_bytes = c->clean_n_bytes;
  out->n_runs = c->clean_n_runs;
  out->header_lines = c->clean_headers;
  out->last_runs = c->clean_last_runs;
  return MK_OK;
}

extern "C" int mk_clean_runs(mk_ctx* c, uint64_t* starts, uint64_t* ends, size_t cap, size_t* n) {
  if (!c || !n || (cap && (!starts || !ends))) return MK_ERR_ARG;
  // (the kernel lists run starts and run ends apart, each up to the lists' capacity: past it the two lists would not
  // hold the same runs, and pairing them by rank would invent intervals -- refuse instead of returning a wrong list)
  if ((size_t)c->clean_last_runs > ((size_t)1 << 16)) {
    *n = 0;
    c->err = "mk_clean_runs: the last chunk holds " + std::to_string(c->clean_last_runs) + " runs of N, more than the 65536 the list keeps";
    return MK_ERR_RANGE;
  }
  const size_t kept = (size_t)c->clean_last_runs;
  *n = kept;
  if (!kept || !cap) return MK_OK;
  MK_HIP(hipSetDevice(c->device));
  std::vector<u64> a(kept), b(kept);
  MK_HIP(hipMemcpyAsync(a.data(), c->clean_runs.p, kept * 8, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipMemcpyAsync(b.data(), (const u64*)c->clean_runs.p + ((size_t)1 << 16), kept * 8, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  std::sort(a.begin(), a.end());  // (written in the order the waves got to them)
  std::sort(b.begin(), b.end());
  for (size_t i = 0; i < kept && i < cap; ++i) { starts[i] = a[i]; ends[i] = b[i]; }
  return MK_OK;
}

// ------------------------------------------------------------- one running table for several contexts of a GPU
// The chunks of a sample are dealt to several contexts of one GPU (HIP streams: one's kernels fill the other's gaps), and
// up to round 3 every context summed its survivors into a table of its own, the tables were summed table to table at
// the end (canonical S2: ~1 ms of the step for the second context's 9.9 M rows).  With fused launches (mk_skcount.hip)
// every access to a running table from a count kernel is an atomic, so the kernels of several streams can upsert into
// ONE table.  mk_share_table(ctx, owner): from now on the fused launches of ctx put their survivors into owner's table;
// whatever ctx still merges on its own (a new context's first chunk, spills, rows kept as text) stays in ctx's table and
// is summed at the end as before (mk_merge_from), only it is small now.  Locking: a launch that uses a shared table
// reads its pointer and size under the table's lock (shared); growing or clearing the table takes the lock exclusively,
// waits for the streams of all contexts that launch into it, and only then replaces it.
static int grow_run64(mk_ctx* c, size_t need_rows);
static int grow_run64_body(mk_ctx* c, size_t need_rows);
static mk_ctx* table_of_ctx(mk_ctx* c) { return c->share_owner ? c->share_owner : c; }

static void drain_table_users(mk_ctx* t) {  // (t->table_mu is held exclusively: nobody can launch into the table meanwhile)
  (void)hipStreamSynchronize(t->stream);
  for (mk_ctx* s : t->sharers) (void)hipStreamSynchronize(s->stream);
}

extern "C" int mk_share_table(mk_ctx* c, mk_ctx* owner) {
  if (!c || c == owner) return MK_ERR_ARG;
  if (c->in_chunk) { c->err = "mk_share_table: a chunk is open"; return MK_ERR_STATE; }
  if (c->share_owner == owner) return MK_OK;
  if (c->share_owner) {  // leave the table it launched into
    mk_ctx* old = c->share_owner;
    std::unique_lock<std::shared_mutex> wr(old->table_mu);
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    old->sharers.erase(std::remove(old->sharers.begin(), old->sharers.end(), c), old->sharers.end());
    c->share_owner = nullptr;
    c->fuse_target = nullptr;
  }
  if (!owner) return MK_OK;
  if (owner->share_owner || !c->sharers.empty()) { c->err = "mk_share_table: tables are shared one level deep (the owner must own its table, a sharer cannot be an owner)"; return MK_ERR_ARG; }
  if (owner->device != c->device || owner->alphabet != c->alphabet || owner->k != c->k || owner->canonical != c->canonical || owner->mode != c->mode) {
    c->err = "mk_share_table: contexts differ in device, alphabet, k or canonical mode";
    return MK_ERR_ARG;
  }
  if (c->mode != MK_MODE_HASH64) { c->err = "mk_share_table: only one-word hashed tables are shared"; return MK_ERR_ARG; }
  std::unique_lock<std::shared_mutex> wr(owner->table_mu);
  owner->sharers.push_back(c);
  c->share_owner = owner;
  return MK_OK;
}

// ----------------------------------------------------------------------------- chunk feed
extern "C" int mk_chunk_begin(mk_ctx* c) {
  if (!c) return MK_ERR_ARG;
  if (c->in_chunk) { c->err = "mk_chunk_begin: a chunk is already open"; return MK_ERR_STATE; }
  c->in_chunk = true;
  c->raw_len = 0;
  return MK_OK;
}

static int feed(mk_ctx* c, const uint8_t* p, size_t n, hipMemcpyKind kind) {
  if (!c) return MK_ERR_ARG;
  if (!c->in_chunk) { c->err = "mk_chunk_feed: no open chunk (call mk_chunk_begin)"; return MK_ERR_STATE; }
  if (n == 0) return MK_OK;
  if (!p) { c->err = "mk_chunk_feed: text is NULL"; return MK_ERR_ARG; }
  MK_HIP(hipSetDevice(c->device));
  int rc = mk_buf_reserve(c, c->raw, c->raw_len + n + 64, true);
  if (rc) return rc;
  MK_HIP(hipMemcpyAsync((uint8_t*)c->raw.p + c->raw_len, p, n, kind, c->stream));
  if (kind == hipMemcpyHostToDevice) MK_HIP(hipStreamSynchronize(c->stream));  // caller may reuse its buffer
  c->raw_len += n;
  return MK_OK;
}

int mk_feed_host_async(mk_ctx* c, const uint8_t* p, size_t n, bool wait) {
  if (!c || !c->in_chunk) return MK_ERR_STATE;
  if (n == 0) return MK_OK;
  MK_HIP(hipSetDevice(c->device));
  int rc = mk_buf_reserve(c, c->raw, c->raw_len + n + 64, true);
  if (rc) return rc;
  MK_HIP(hipMemcpyAsync((uint8_t*)c->raw.p + c->raw_len, p, n, hipMemcpyHostToDevice, c->stream));
  if (wait) MK_HIP(hipStreamSynchronize(c->stream));
  c->raw_len += n;
  return MK_OK;
}

int mk_reserve_raw(mk_ctx* c, size_t bytes) {
  MK_HIP(hipSetDevice(c->device));
  return mk_buf_reserve(c, c->raw, bytes, true);
}

extern "C" int mk_chunk_feed(mk_ctx* c, const uint8_t* text, size_t n) { return feed(c, text, n, hipMemcpyHostToDevice); }
extern "C" int mk_chunk_feed_device(mk_ctx* c, const uint8_t* d_text, size_t n) {
  return feed(c, d_text, n, hipMemcpyDeviceToDevice);
}

// ------------------------------------------------------------------------ running tables
// Slots of a running table that has to take need_rows rows: the next power of two above 2.5 x (load 20-40 % after a
// growth, 50 % before the next; MK_GROW_X4=1: above 4 x, as in rounds 1-2).  Compaction, table-to-table sums and
// clears scan the slots, so a table twice as sparse costs every sample ~0.15 ms.
static size_t run_slots_for(size_t need_rows) {
  static const bool x4 = getenv("MK_GROW_X4") != nullptr;
  return pow2_at_least(x4 ? 4 * need_rows : need_rows * 5 / 2);
}

static int grow_run64_body(mk_ctx* c, size_t need_rows);
// (a table other contexts launch into -- mk_share_table -- is only replaced under its lock, with their streams drained)
static int grow_run64(mk_ctx* c, size_t need_rows) {
  if (c->sharers.empty()) return grow_run64_body(c, need_rows);
  std::unique_lock<std::shared_mutex> wr(c->table_mu);
  if (2 * need_rows <= c->run_slots) return MK_OK;
  drain_table_users(c);
  return grow_run64_body(c, need_rows);
}
static int grow_run64_body(mk_ctx* c, size_t need_rows) {
  if (2 * need_rows <= c->run_slots) return MK_OK;
  const size_t slots = run_slots_for(need_rows);
  MkDevBuf nb;
  int rc = mk_buf_reserve(c, nb, slots * sizeof(MkSlot));
  if (rc) return rc;
  if ((rc = mk_launch_clear_slots(c, (MkSlot*)nb.p, slots)) != MK_OK) return rc;
  if (c->run_slots && (rc = mk_launch_rehash64(c, (const MkSlot*)c->run.p, c->run_slots, (MkSlot*)nb.p, slots)) != MK_OK) return rc;
  MK_HIP(hipStreamSynchronize(c->stream));
  buf_free(c->run);
  c->run = nb;
  c->run_slots = slots;
  return MK_OK;
}

static int grow_run128(mk_ctx* c, size_t need_rows) {
  if (2 * need_rows <= c->run128_slots) return MK_OK;
  const size_t slots = pow2_at_least(4 * need_rows);
  MkDevBuf nb;
  int rc = mk_buf_reserve(c, nb, slots * sizeof(MkSlot128));
  if (rc) return rc;
  MK_HIP(hipMemsetAsync(nb.p, 0, slots * sizeof(MkSlot128), c->stream));
  if (c->run128_slots && (rc = mk_launch_rehash128(c, (const MkSlot128*)c->run128.p, c->run128_slots, (MkSlot128*)nb.p, slots)) != MK_OK) return rc;
  MK_HIP(hipStreamSynchronize(c->stream));
  buf_free(c->run128);
  c->run128 = nb;
  c->run128_slots = slots;
  return MK_OK;
}

static int grow_run_ref(mk_ctx* c, size_t need_rows) {
  int rc;
  if (need_rows > c->arena_rows_cap) {
    const size_t rows = std::max(need_rows, c->arena_rows_cap * 2);
    if ((rc = mk_buf_reserve(c, c->arena, rows * (size_t)c->k + 64, true)) != MK_OK) return rc;
    c->arena_rows_cap = rows;
  }
  if (2 * need_rows <= c->run_ref_slots) return MK_OK;
  const size_t slots = pow2_at_least(4 * need_rows);
  MkDevBuf nb;
  if ((rc = mk_buf_reserve(c, nb, slots * sizeof(MkSlot))) != MK_OK) return rc;
  if ((rc = mk_launch_clear_slots(c, (MkSlot*)nb.p, slots)) != MK_OK) return rc;
  if (c->run_ref_slots &&
      (rc = mk_launch_rehash_ref(c, (const MkSlot*)c->run_ref.p, c->run_ref_slots, (MkSlot*)nb.p, slots)) != MK_OK)
    return rc;
  MK_HIP(hipStreamSynchronize(c->stream));
  buf_free(c->run_ref);
  c->run_ref = nb;
  c->run_ref_slots = slots;
  return MK_OK;
}

int mk_grow_run(mk_ctx* c, size_t more_rows) {
  if (c->mode == MK_MODE_HASH64) return grow_run64(c, c->run_rows + more_rows);
  if (c->mode == MK_MODE_HASH128) return grow_run128(c, c->run128_rows + more_rows);
  return MK_OK;  // dense bins are allocated once; by-reference rows have their own table
}

// mk_bin.hip
bool mk_binned_takes(const mk_ctx* c);
int mk_launch_count_binned(mk_ctx* c, size_t seq_len, uint64_t min_count);

// --------------------------------------------------------------------------- the pipeline
#define MK_RETRY_GENERAL 1  // (internal) the speculative lane met input it does not handle: take the general path

// The partitioned nucleotide paths (one-word keys 18 <= k <= 32, two-word keys 33 <= k <= 64) with ONE host
// read-back per chunk instead of three.  Everything up to the count kernel is launched on the assumption that the
// fast parser will do (no blank inside a sequence line) and with buffers and grids sized from the raw length (the
// kernels read the true seq_len on the device); the one read-back after the count kernel tells whether that held
// (otherwise MK_RETRY_GENERAL), whether symbols outside the alphabet need the by-reference kernel (then it runs
// now: one more read-back, rare), and how many rows survive; the merge is launched and its row totals are copied
// back without waiting -- they are added up when the next read-back (or settle()) has passed them.
static int process_chunk_fast(mk_ctx* c, const uint8_t* d_raw, size_t n, u64 min_count) {
  int rc;
  const size_t begin = (size_t)((uintptr_t)d_raw & 15);
  const uint8_t* d_al = d_raw - begin;
  MK_HIP(hipSetDevice(c->device));
  MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  if ((rc = mk_buf_reserve(c, c->seq, n + 256)) != MK_OK) return rc;
  const size_t bad_words = n / 64 + 4;
  if ((rc = mk_buf_reserve(c, c->bad, (bad_words + 2) * 8)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->codes, (2 * bad_words + 8) * 8)) != MK_OK) return rc;
  // (the parsed stream itself is not written: only the by-reference kernel reads it, and that runs only when the chunk
  // holds characters outside the alphabet -- the chunk is then parsed once more with the stream, below)
  static const bool always_seq = getenv("MK_ALWAYS_SEQ") != nullptr;
  if ((rc = mk_launch_fparse(c, d_al, begin, n, /*fuse_pack_nt=*/true, /*write_seq=*/always_seq)) != MK_OK) return rc;
  const bool two = c->mode == MK_MODE_HASH128;
  c->rtab_chunk_slots = 0;
  c->surv_regions = 0;
  c->ctab_slots = 0;
  // Fused upsert (mk_skcount.hip): from a sample's second chunk on the count kernel puts the survivors into the running
  // table itself -- no import kernel, no waiting for their number.  The table is sized HERE for what the chunk before
  // kept, twice over; the kernel spills what a table that fills up all the same cannot take, and that is imported below.
  c->fuse_cap = 0;
  if (!two && c->use_fused && min_count >= 2 && c->surv_hint_ok && !c->run_bucket_major) {
    static const bool cores = getenv("MK_CORES") != nullptr;  // (half-sized buckets: half the survivors per bucket, half the list)
    const unsigned long long per_bucket = (c->surv_hint >> 13) * (cores ? 2 : 1);  // (8192 buckets on chunks of this size; smaller chunks: fewer of both)
    const int cap = per_bucket <= 110 ? 512 : (per_bucket <= 360 ? 1024 : 0);
    if (cap) {
      if ((rc = settle(c)) != MK_OK) return rc;  // (run_rows must be what the table holds)
      // Which table: the owner's when this context shares one (mk_share_table) AND that table has room for what this
      // chunk is expected to add -- only the owner ever replaces its table (it sizes it for its sharers as well), a sharer
      // that finds it too small upserts into its own for this chunk; the sum at the end is the same.
      mk_ctx* t = table_of_ctx(c);
      // (what this chunk is expected to add: the last full chunk's survivors -- late in a sample most of them are keys
      // the table already holds; the spill list takes what a bad guess leaves no room for)
      static const bool roomy = getenv("MK_FUSE_ROOMY") != nullptr;  // (A/B: twice the survivors, as first built)
      const size_t expect = (roomy ? 2 * (size_t)c->surv_hint : (size_t)c->surv_hint) + 4096;
      if (t != c) {
        std::shared_lock<std::shared_mutex> rd(t->table_mu);
        size_t rows_now;
        { std::lock_guard<std::mutex> g(t->rows_mu); rows_now = t->run_rows; }
        if (t->run_slots < 1024 || 2 * (rows_now + expect) > t->run_slots) t = c;
      }
      if (t == c) {
        size_t rows_now;
        { std::lock_guard<std::mutex> g(c->rows_mu); rows_now = c->run_rows; }
        if ((rc = grow_run64(c, rows_now + expect * (1 + c->sharers.size()))) != MK_OK) return rc;
      }
      c->fuse_target = t;
      c->fuse_cap = cap;
    }
  }
  {
    // (a launch into ANOTHER context's table reads its pointer and size under that table's lock: see mk_share_table)
    mk_ctx* t = c->fuse_cap ? c->fuse_target : c;
    std::shared_lock<std::shared_mutex> rd(t->table_mu, std::defer_lock);
    if (t != c) rd.lock();
    rc = two ? mk_launch_count_superkmer2(c, n, min_count) : mk_launch_count_superkmer(c, n, min_count);  // (seq_len <= n)
  }
  c->fuse_cap = 0;
  if (rc) return rc;
  const bool fused = !two && c->fused_last;
  if ((rc = pull_info(c)) != MK_OK) return rc;  // the one read-back
  MkChunkInfo* h = c->h_info;
  if (h->parse_fallback) { c->st.parse_retries += 1; return MK_RETRY_GENERAL; }
  if (h->non_ascii) {
    c->err = "input holds " + std::to_string(h->non_ascii) +
             " sequence byte(s) >= 0x80 (non-ASCII sequence text is not supported; the chunk was not counted)";
    return MK_ERR_NON_ASCII;
  }
  const size_t seq_len = (size_t)h->seq_len;
  if (h->bad_symbols) {  // windows holding a symbol outside the alphabet: by reference, now
    if (!always_seq) {
      // the by-reference kernel reads the parsed stream, which the first parse did not write: parse again (the raw text
      // is still there), this time for the stream only -- the packed words, the bitmap and the chunk's counters stand
      // (the second parse adds to the chunk's counters again -- kept bytes >= 0x80 -- so they are set aside and put back)
      if ((rc = mk_buf_reserve(c, c->ex_tmp, sizeof(MkChunkInfo) + 64)) != MK_OK) return rc;
      MK_HIP(hipMemcpyAsync(c->ex_tmp.p, c->info.p, sizeof(MkChunkInfo), hipMemcpyDeviceToDevice, c->stream));
      if ((rc = mk_launch_fparse(c, d_al, begin, n, /*fuse_pack_nt=*/false, /*write_seq=*/true)) != MK_OK) return rc;
      MK_HIP(hipMemcpyAsync(c->info.p, c->ex_tmp.p, sizeof(MkChunkInfo), hipMemcpyDeviceToDevice, c->stream));
    }
    const u64 bound = std::min<u64>((u64)seq_len, h->bad_symbols * (u64)c->k);
    c->rtab_chunk_slots = pow2_at_least(2 * (size_t)bound);
    if ((rc = mk_buf_reserve(c, c->rtab_chunk, c->rtab_chunk_slots * sizeof(MkSlot))) != MK_OK) return rc;
    if ((rc = mk_launch_clear_slots(c, (MkSlot*)c->rtab_chunk.p, c->rtab_chunk_slots)) != MK_OK) return rc;
    if ((rc = mk_launch_count_byref(c, seq_len, true)) != MK_OK) return rc;
    if ((rc = mk_launch_count_survivors(c, min_count)) != MK_OK) return rc;
    if ((rc = pull_info(c)) != MK_OK) return rc;
  }
  if (h->part_overflow) {  // (see process_chunk: partition again from the exact histogram)
    if (!c->part_sampled) { c->err = "partition overflow without sampling (internal error)"; return MK_ERR_STATE; }
    if (getenv("MK_VERBOSE")) fprintf(stderr, "[mk] sampled partition too small (where=%llu): exact pass\n", h->part_overflow);
    h->windows = h->records = h->distinct = h->survivors = h->side = h->errors = h->part_overflow = 0;
    MK_HIP(hipMemcpyAsync(c->info.p, h, sizeof(MkChunkInfo), hipMemcpyHostToDevice, c->stream));
    c->st.part_retries += 1;
    // (a fused count kernel that met the flag stopped before its first bucket: the running table is as it was; the exact
    // pass hands its survivors over through their regions)
    rc = two ? mk_launch_count_superkmer2(c, seq_len, min_count, /*exact=*/true) : mk_launch_count_superkmer(c, seq_len, min_count, /*exact=*/true);
    if (rc) return rc;
    if ((rc = pull_info(c)) != MK_OK) return rc;
    if (h->part_overflow) { c->err = "partition overflow after the exact pass (internal error: nothing was counted)"; return MK_ERR_STATE; }
  }
  if (h->errors) {
    c->err = "counting kernel reported " + std::to_string(h->errors) + " unrecoverable condition(s) (bucket too large to split)";
    return MK_ERR_RANGE;
  }
  c->part_dirty = false;  // the count kernel ran to its end: every cursor is back at its region's start
  const bool fused_done = !two && c->fused_last;  // (of the launch that counted: the exact pass is never fused)
  if (fused_done) {
    {
      mk_ctx* t = c->fuse_target ? c->fuse_target : c;  // (counted by the kernel, in the same read-back; the table may be another context's)
      std::lock_guard<std::mutex> g(t->rows_mu);
      t->run_rows += (size_t)h->new_rows;
    }
    h->new_rows = 0;
    c->st.fused_chunks += 1;
    c->st.fuse_spilled += h->spilled;
    // (what is launched below adds to the device's copy again, and that copy is read back later: start it from zero.
    // Nearly always nothing is: no spill, no rows kept as text -- then neither this fill nor that read-back is issued:
    // two of the five tiny device operations a chunk cost besides its kernels)
    if (h->spilled || h->survivors_ref)
      MK_HIP(hipMemsetAsync(&((MkChunkInfo*)c->info.p)->new_rows, 0, sizeof(unsigned long long), c->stream));
    if (h->spilled) {  // the table was filling up: what the kernel set aside goes in now, into a table with room
      if ((rc = grow_run64(c, c->run_rows + (size_t)h->spilled)) != MK_OK) return rc;
      if ((rc = mk_launch_import_pairs(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_cnts.p, (size_t)h->spilled)) != MK_OK) return rc;
    }
  }
  (void)fused;
  if (!two && !fused_done && h->survivors && (rc = grow_run64(c, c->run_rows + (size_t)h->survivors)) != MK_OK) return rc;
  if (h->survivors_ref && (rc = grow_run_ref(c, c->run_ref_rows + (size_t)h->survivors_ref)) != MK_OK) return rc;
  if (two && (h->survivors || h->survivors_ref) &&
      (rc = grow_run128(c, c->run128_rows + (size_t)h->survivors + (size_t)h->survivors_ref)) != MK_OK) return rc;
  if (h->survivors && seq_len && !fused_done) {
    const size_t p1 = (size_t)1 << c->p1_log2;
    const uint64_t* meta = (const uint64_t*)c->part_meta.p;  // hist|start|cursor|khist|kstart|kcursor|nsurv
    mk_prof_begin(c, MK_K_FILTER);
    rc = two ? mk_launch_import128_regions(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_keys2.p,
                                           (const uint64_t*)c->surv_cnts.p, meta + 4 * p1 + 1, meta + 6 * p1 + 2, p1)
             : mk_launch_import_regions(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_cnts.p,
                                        meta + 4 * p1 + 1, meta + 6 * p1 + 2, p1, (size_t)h->survivors);
    mk_prof_end(c);
    if (rc) return rc;
  }
  if ((rc = mk_launch_accumulate(c, min_count)) != MK_OK) return rc;  // (survivors of the by-reference chunk table, if any)
  // the merge's row totals: copied back, not waited for (a fused launch that set nothing aside has reported them already)
  if (!fused_done || h->spilled || h->survivors_ref) {
    MK_HIP(hipMemcpyAsync(c->h_info + 1, c->info.p, sizeof(MkChunkInfo), hipMemcpyDeviceToHost, c->stream));
    c->pending_rows = true;
  }
  if (h->side && h->side >= min_count) c->run_side += h->side;
  // (hints for the next chunk come from FULL chunks: a sample's short last chunk -- a third of the coverage, half the
  // windows per distinct key, a fraction of the survivors -- made the first chunk of the next sample plan two sub-range
  // passes per bucket, 480 instead of 305 us, and would size the fused launch's table for nothing)
  const bool full_chunk = !c->dup_known || seq_len * 4 >= c->part_prev_len * 3;
  if (!two && (full_chunk || !c->surv_hint_ok)) { c->surv_hint = h->survivors; c->surv_hint_ok = true; }
  if (h->distinct && full_chunk) { c->dup_hint = (double)h->windows / (double)h->distinct; c->dup_known = true; }
  if (h->records) { c->nk_hint = (double)(h->windows + h->exotic) / (double)h->records; c->items_hint = (double)h->records * 32.0 / (double)(seq_len ? seq_len : 1); }
  if (getenv("MK_VERBOSE"))
    fprintf(stderr, "[mk] chunk (one read-back): raw=%zu seq=%zu windows=%llu records=%llu distinct=%llu survivors=%llu p1=2^%d dup=%.2f nk=%.2f fused=%d spilled=%llu rows=%zu slots=%zu\n",
            n, seq_len, (unsigned long long)h->windows, (unsigned long long)h->records, (unsigned long long)h->distinct,
            (unsigned long long)h->survivors, c->p1_log2, c->dup_hint, c->nk_hint, fused_done ? 1 : 0, (unsigned long long)h->spilled,
            c->run_rows, c->run_slots);
  c->st.table_slots = c->rtab_chunk_slots;
  c->st.raw_bytes += n;
  c->st.symbols += h->symbols;
  c->st.windows += h->windows + h->exotic;
  c->st.exotic_windows += h->exotic;
  c->st.chunks += 1;
  c->st.records += h->records;
  c->st.distinct += h->distinct;
  c->st.survivors += h->survivors + h->survivors_ref + ((h->side && h->side >= min_count) ? 1 : 0);
  return MK_OK;
}

// d_raw may be unaligned: the fast parser reads from the 16-byte boundary below it and ignores the
// bytes in front; only the (rare) general-parser fallback needs an aligned copy.
static int process_chunk(mk_ctx* c, const uint8_t* d_raw, size_t n, u64 min_count) {
  int rc;
  bool known_blank = false;
  if (c->clean_mode) {  // (one read-back more than the speculative lane: the chunk must be known to be reproducible BEFORE it is merged)
    if (!c->use_fast_parse) { c->err = "clean mode needs the fast parser (MK_NO_FAST_PARSE is set)"; return MK_ERR_UNSUPPORTED; }
    if (d_raw != (const uint8_t*)c->raw.p) { c->err = "clean mode rewrites the text in place: feed it (mk_chunk_feed), do not pass caller memory"; return MK_ERR_STATE; }
    MK_HIP(hipSetDevice(c->device));
    if ((rc = mk_launch_clean_pre(c, (uint8_t*)c->raw.p, n)) != MK_OK) return rc;
  }
  if (!c->clean_mode && c->use_speculation && c->use_fast_parse && c->alphabet == MK_ALPHABET_NT2 && n && n < 0xFE000000ull &&
      ((c->mode == MK_MODE_HASH64 && c->use_partition && c->use_superkmer && c->k >= c->sk_min_k && c->k <= 32) ||
       (c->mode == MK_MODE_HASH128 && c->use_superkmer2))) {
    rc = process_chunk_fast(c, d_raw, n, min_count);
    if (rc != MK_RETRY_GENERAL) return rc;
    known_blank = true;  // (the fast parser has just said so: straight to the general one)
  }
  if ((rc = settle(c)) != MK_OK) return rc;
  const size_t begin = (size_t)((uintptr_t)d_raw & 15);
  const uint8_t* d_al = d_raw - begin;
  MK_HIP(hipSetDevice(c->device));
  MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  if ((rc = mk_buf_reserve(c, c->seq, n + 256)) != MK_OK) return rc;
  const bool packed = c->mode != MK_MODE_BYREF;
  const size_t bad_words = n / 64 + 4;
  const size_t code_words = c->alphabet == MK_ALPHABET_NT2 ? 2 * bad_words : (bad_words * 64 + 11) / 12;
  if (packed) {
    if ((rc = mk_buf_reserve(c, c->bad, (bad_words + 2) * 8)) != MK_OK) return rc;
    if ((rc = mk_buf_reserve(c, c->codes, (code_words + 8) * 8)) != MK_OK) return rc;
  }
  for (int attempt = 0; attempt < 2; ++attempt) {
    const bool fast = c->use_fast_parse && attempt == 0 && !known_blank;
    const bool fused = fast && packed && c->alphabet == MK_ALPHABET_NT2;  // the nt pack rides on the parser's LDS image
    if (!fast && begin) {  // aligned copy for the general transducer
      if ((rc = mk_buf_reserve(c, c->raw, n + 64)) != MK_OK) return rc;
      MK_HIP(hipMemcpyAsync(c->raw.p, d_raw, n, hipMemcpyDeviceToDevice, c->stream));
      d_raw = (const uint8_t*)c->raw.p;
    }
    if ((rc = fast ? mk_launch_fparse(c, d_al, begin, n, fused) : mk_launch_parse(c, d_raw, n)) != MK_OK) return rc;
    if (packed && !fused && (rc = mk_launch_pack(c, n)) != MK_OK) return rc;
    if (c->clean_mode && (rc = mk_launch_clean_post(c, n)) != MK_OK) return rc;
    if ((rc = pull_info(c)) != MK_OK) return rc;
    if (c->clean_mode) {
      const u64* m = c->h_clean;  // first header | '>' bytes | marker bytes in the input | N bytes | runs | G+C | starts | ends
      const u64 headers = c->h_info->seq_len - c->h_info->symbols;
      const char* why = c->h_info->parse_fallback ? "a blank inside a sequence line"
                        : m[2]                    ? "a 0x7F byte in the text, or blanks in front of the first header"
                        : m[1] != headers         ? "a '>' that does not start a header line"
                                                  : nullptr;
      if (why) {
        c->err = std::string("clean mode: ") + why + " (removeN's rewrite of such text is not reproduced on the GPU; nothing was counted)";
        return MK_ERR_UNSUPPORTED;
      }
      c->clean_raw += n;
      c->clean_headers += headers;
      c->clean_n_bytes += m[3];
      c->clean_n_runs += m[4];
      c->clean_gc += m[5];
      c->clean_symbols += c->h_info->symbols - m[3];
      c->clean_last_runs = m[4];
      c->h_info->symbols -= m[3];  // (the N bytes are separators now)
      break;
    }
    if (!fast || !c->h_info->parse_fallback) break;
    c->st.parse_retries += 1;
    // a blank inside a sequence line: the general transducer handles strip() exactly
    MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  }
  if (c->h_info->non_ascii) {
    c->err = "input holds " + std::to_string(c->h_info->non_ascii) +
             " sequence byte(s) >= 0x80 (non-ASCII sequence text is not supported; the chunk was not counted)";
    return MK_ERR_NON_ASCII;
  }
  const size_t seq_len = (size_t)c->h_info->seq_len;
  const u64 bad_symbols = c->h_info->bad_symbols;

  // chunk tables
  c->rtab_chunk_slots = 0;
  const bool partitioned = c->mode == MK_MODE_HASH64 && c->use_partition;
  c->surv_regions = 0;
  c->ctab_slots = c->mode == MK_MODE_DENSE ? c->ctab_slots : 0;
  if (c->mode == MK_MODE_HASH64 && !partitioned) {
    c->ctab_slots = pow2_at_least(2 * seq_len);
    if ((rc = mk_buf_reserve(c, c->ctab, c->ctab_slots * sizeof(MkSlot))) != MK_OK) return rc;
    if ((rc = mk_launch_clear_slots(c, (MkSlot*)c->ctab.p, c->ctab_slots)) != MK_OK) return rc;
  }
  // partitioned path: no global chunk table (32-bit record indices in the scatter's LDS: chunks below 4 G symbols)
  const bool sk2 = c->mode == MK_MODE_HASH128 && c->alphabet == MK_ALPHABET_NT2 && c->use_superkmer2 && seq_len < 0xFFFFFF00ull;
  if (c->mode == MK_MODE_HASH128 && c->canonical && !sk2) {
    c->err = "canonical counting of 33..64-mers needs the partitioned path (chunk of 4 G symbols or more)";
    return MK_ERR_RANGE;
  }
  if (c->mode == MK_MODE_BYREF || (c->mode == MK_MODE_HASH128 && !sk2)) {
    c->rtab_chunk_slots = pow2_at_least(2 * seq_len);
  } else if (bad_symbols) {
    const u64 bound = std::min<u64>((u64)seq_len, bad_symbols * (u64)c->k);
    c->rtab_chunk_slots = pow2_at_least(2 * (size_t)bound);
  }
  if (c->rtab_chunk_slots) {
    if ((rc = mk_buf_reserve(c, c->rtab_chunk, c->rtab_chunk_slots * sizeof(MkSlot))) != MK_OK) return rc;
    if ((rc = mk_launch_clear_slots(c, (MkSlot*)c->rtab_chunk.p, c->rtab_chunk_slots)) != MK_OK) return rc;
  }
  c->st.table_slots = (c->mode == MK_MODE_BYREF || c->mode == MK_MODE_HASH128) ? c->rtab_chunk_slots : c->ctab_slots;

  // count
  if (c->mode == MK_MODE_DENSE) rc = mk_launch_count_dense(c, seq_len);
  else if (partitioned) {
    // (the super-k-mer scatter keeps 32-bit record indices in LDS)
    const bool sk = c->use_superkmer && c->alphabet == MK_ALPHABET_NT2 && c->k >= c->sk_min_k && c->k <= 32 && seq_len < 0xFE000000ull;
    // (keys of 16..26 bits -- nucleotide 8 <= k <= 11, protein k = 4, 5 -- are counted by direct index: mk_bin.hip)
    const bool binned = !sk && mk_binned_takes(c) && seq_len < 0xFFFFFF00ull;
    rc = sk ? mk_launch_count_superkmer(c, seq_len, min_count)
            : (binned ? mk_launch_count_binned(c, seq_len, min_count) : mk_launch_count_partitioned(c, seq_len, min_count));
  }
  else if (c->mode == MK_MODE_HASH64) rc = mk_launch_count_hash64(c, seq_len);
  else if (c->mode == MK_MODE_HASH128) rc = sk2 ? mk_launch_count_superkmer2(c, seq_len, min_count) : mk_launch_count_ref128(c, seq_len);
  if (rc) return rc;
  // by reference, byte-wise: every window (raw mode) or only those holding a symbol outside the alphabet
  if (c->rtab_chunk_slots && (c->mode != MK_MODE_HASH128 || bad_symbols) &&
      (rc = mk_launch_count_byref(c, seq_len, packed)) != MK_OK) return rc;

  // filter (per chunk!) + merge
  if ((rc = mk_launch_count_survivors(c, min_count)) != MK_OK) return rc;
  if ((rc = pull_info(c)) != MK_OK) return rc;
  if (c->h_info->part_overflow) {
    // a bucket (or survivor) region sized from the sampled histogram was too small: the kernels stopped
    // short of writing past it; partition and count again from the exact histogram
    if (!c->part_sampled) { c->err = "partition overflow without sampling (internal error)"; return MK_ERR_STATE; }
    MkChunkInfo* h = c->h_info;
    if (getenv("MK_VERBOSE")) fprintf(stderr, "[mk] sampled partition too small (where=%llu: 1 records total, 2 survivors total, 4 a bucket, 8 a survivor region): exact pass\n", h->part_overflow);
    // (the fields the partitioned kernels own; what the by-reference kernel added for odd windows stays)
    h->windows = h->records = h->distinct = h->survivors = h->side = h->errors = h->part_overflow = 0;
    MK_HIP(hipMemcpyAsync(c->info.p, h, sizeof(MkChunkInfo), hipMemcpyHostToDevice, c->stream));
    c->st.part_retries += 1;
    rc = sk2 ? mk_launch_count_superkmer2(c, seq_len, min_count, /*exact=*/true)
             : mk_launch_count_superkmer(c, seq_len, min_count, /*exact=*/true);
    if (rc) return rc;
    if ((rc = pull_info(c)) != MK_OK) return rc;
    if (c->h_info->part_overflow) { c->err = "partition overflow after the exact pass (internal error: nothing was counted)"; return MK_ERR_STATE; }
  }
  if (c->h_info->errors) {
    c->err = "counting kernel reported " + std::to_string(c->h_info->errors) + " unrecoverable condition(s) (bucket too large to split)";
    return MK_ERR_RANGE;
  }
  c->part_dirty = false;
  if (c->mode == MK_MODE_HASH64 && c->h_info->survivors)
    if ((rc = grow_run64(c, c->run_rows + (size_t)c->h_info->survivors)) != MK_OK) return rc;
  if (c->h_info->survivors_ref)
    if ((rc = grow_run_ref(c, c->run_ref_rows + (size_t)c->h_info->survivors_ref)) != MK_OK) return rc;
  // two-word keys: survivors of the partitioned kernel, or (unpartitioned path) of the by-reference chunk table,
  // whose clean rows are packed on their way into the running table
  if (c->mode == MK_MODE_HASH128 && (c->h_info->survivors || c->h_info->survivors_ref))
    if ((rc = grow_run128(c, c->run128_rows + (size_t)c->h_info->survivors + (size_t)c->h_info->survivors_ref)) != MK_OK) return rc;
  if (sk2 && c->surv_regions == 2 && seq_len && c->h_info->survivors) {
    const size_t p1 = (size_t)1 << c->p1_log2;
    const uint64_t* meta = (const uint64_t*)c->part_meta.p;  // hist|start|cursor|khist|kstart|kcursor|nsurv
    mk_prof_begin(c, MK_K_FILTER);
    rc = mk_launch_import128_regions(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_keys2.p,
                                     (const uint64_t*)c->surv_cnts.p, meta + 4 * p1 + 1, meta + 6 * p1 + 2, p1);
    mk_prof_end(c);
    if (rc) return rc;
  }
  if (partitioned && c->h_info->survivors) {
    mk_prof_begin(c, MK_K_FILTER);
    if (c->surv_regions) {
      const size_t p1 = (size_t)1 << c->p1_log2;
      const uint64_t* meta = (const uint64_t*)c->part_meta.p;  // hist|start|cursor|khist|kstart|kcursor|nsurv
      rc = mk_launch_import_regions(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_cnts.p,
                                    meta + 4 * p1 + 1, meta + 6 * p1 + 2, p1, (size_t)c->h_info->survivors);
    } else {
      // (a chunk's survivors from the direct-index / 8-byte-key paths: each key once)
      rc = mk_launch_import_pairs(c, (const uint64_t*)c->surv_keys.p, (const uint64_t*)c->surv_cnts.p, (size_t)c->h_info->survivors, true);
    }
    mk_prof_end(c);
    if (rc) return rc;
  }
  if ((rc = mk_launch_accumulate(c, min_count)) != MK_OK) return rc;
  if ((rc = pull_info(c)) != MK_OK) return rc;
  if (c->mode == MK_MODE_HASH128) c->run128_rows += (size_t)c->h_info->new_rows;
  else c->run_rows += (size_t)c->h_info->new_rows;
  c->run_ref_rows += (size_t)c->h_info->new_rows_ref;
  if (c->h_info->side && c->h_info->side >= min_count) c->run_side += c->h_info->side;
  if (partitioned && c->h_info->distinct) { c->dup_hint = (double)c->h_info->windows / (double)c->h_info->distinct; c->dup_known = true; }
  if (sk2 && c->h_info->distinct) { c->dup_hint = (double)c->h_info->windows / (double)c->h_info->distinct; c->dup_known = true; }
  if ((partitioned || sk2) && c->h_info->records) {
    c->nk_hint = (double)(c->h_info->windows + c->h_info->exotic) / (double)c->h_info->records;
    c->items_hint = (double)c->h_info->records * 32.0 / (double)(seq_len ? seq_len : 1);
  }

  if (getenv("MK_VERBOSE"))
    fprintf(stderr, "[mk] chunk: raw=%zu seq=%zu windows=%llu records=%llu distinct=%llu survivors=%llu new_rows=%llu p1=2^%d dup=%.2f nk=%.2f\n", n, seq_len,
            (unsigned long long)c->h_info->windows, (unsigned long long)c->h_info->records, (unsigned long long)c->h_info->distinct,
            (unsigned long long)c->h_info->survivors, (unsigned long long)c->h_info->new_rows, c->p1_log2, c->dup_hint, c->nk_hint);
  c->st.raw_bytes += n;
  c->st.symbols += c->h_info->symbols;
  c->st.windows += c->h_info->windows + c->h_info->exotic;
  c->st.exotic_windows += c->h_info->exotic;
  c->st.chunks += 1;
  c->st.records += c->h_info->records;
  c->st.distinct += c->h_info->distinct;
  c->st.survivors += c->h_info->survivors + c->h_info->survivors_ref +
                     ((c->h_info->side && c->h_info->side >= min_count) ? 1 : 0);
  return MK_OK;
}

extern "C" int mk_chunk_end(mk_ctx* c, uint64_t min_count) {
  if (!c) return MK_ERR_ARG;
  if (!c->in_chunk) { c->err = "mk_chunk_end: no open chunk"; return MK_ERR_STATE; }
  c->in_chunk = false;
  int rc = process_chunk(c, (const uint8_t*)c->raw.p, c->raw_len, min_count);
  c->raw_len = 0;
  return rc;
}

extern "C" int mk_count_device(mk_ctx* c, const uint8_t* d_text, size_t n, uint64_t min_count) {
  if (!c) return MK_ERR_ARG;
  if (c->in_chunk) { c->err = "mk_count_device: a chunk is open"; return MK_ERR_STATE; }
  if (n && !d_text) { c->err = "mk_count_device: d_text is NULL"; return MK_ERR_ARG; }
  if (!c->clean_mode && (((uintptr_t)d_text & 15) == 0 || c->use_fast_parse)) return process_chunk(c, d_text, n, min_count);
  int rc = mk_chunk_begin(c);
  if (!rc) rc = mk_chunk_feed_device(c, d_text, n);
  if (rc) { c->in_chunk = false; return rc; }
  return mk_chunk_end(c, min_count);
}

// --------------------------------------------------------------------------------- export
struct ExportView {
  std::vector<u64> pkeys, pcnts;     // packed rows, sorted by key (two-word keys: pkeys holds {hi, lo} pairs)
  int words = 1;                     // 64-bit words per packed key
  size_t packed_rows() const { return pcnts.size(); }
  std::vector<uint8_t> rstr;         // by-reference rows: k bytes each, arena order
  std::vector<u64> rcnt;             // counts in arena order
  std::vector<u64> rorder;           // arena rows sorted by string
};

static int gather_packed(mk_ctx* c, ExportView& v, u64* d_keys_out, u64* d_cnts_out, size_t cap, size_t* rows_out,
                         bool to_host) {
  int rc;
  size_t rows = 0;
  const auto t_gather = std::chrono::steady_clock::now();
  if (c->mode == MK_MODE_DENSE) {
    const size_t nbins = c->run_slots;
    std::vector<u64> bins(nbins);
    MK_HIP(hipMemcpyAsync(bins.data(), c->run.p, nbins * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < nbins; ++i)
      if (bins[i]) { v.pkeys.push_back(i); v.pcnts.push_back(bins[i]); }
    rows = v.pkeys.size();
    if (!to_host) {
      if (rows > cap) { c->err = "export: device buffers too small"; return MK_ERR_RANGE; }
      if (rows) {
        MK_HIP(hipMemcpyAsync(d_keys_out, v.pkeys.data(), rows * 8, hipMemcpyHostToDevice, c->stream));
        MK_HIP(hipMemcpyAsync(d_cnts_out, v.pcnts.data(), rows * 8, hipMemcpyHostToDevice, c->stream));
        MK_HIP(hipStreamSynchronize(c->stream));
      }
    }
  } else if (c->mode == MK_MODE_HASH64) {
    rows = c->run_rows;
    const size_t side = c->run_side ? 1 : 0;
    if (!to_host && rows + side > cap) { c->err = "export: device buffers too small"; return MK_ERR_RANGE; }
    if (rows) {
      mk_prof_begin(c, MK_K_EXPORT);
      if ((rc = mk_buf_reserve(c, c->ex_keys, rows * 8 + 64)) != MK_OK) return rc;
      if ((rc = mk_buf_reserve(c, c->ex_cnts, rows * 8 + 64)) != MK_OK) return rc;
      u64* d_cursor = (u64*)((char*)c->info.p + sizeof(MkChunkInfo));
      MK_HIP(hipMemsetAsync(d_cursor, 0, 8, c->stream));
      if ((rc = mk_launch_compact(c, (const MkSlot*)c->run.p, c->run_slots, (uint64_t*)c->ex_keys.p,
                                  (uint64_t*)c->ex_cnts.p, rows, (uint64_t*)d_cursor)) != MK_OK) return rc;
      u64* ok = d_keys_out;
      u64* oc = d_cnts_out;
      if (to_host) {
        if ((rc = mk_buf_reserve(c, c->ex_keys2, rows * 8 + 64)) != MK_OK) return rc;
        if ((rc = mk_buf_reserve(c, c->ex_cnts2, rows * 8 + 64)) != MK_OK) return rc;
        ok = (u64*)c->ex_keys2.p;
        oc = (u64*)c->ex_cnts2.p;
      }
      if ((rc = mk_sort_pairs(c, (const uint64_t*)c->ex_keys.p, (const uint64_t*)c->ex_cnts.p, (uint64_t*)ok,
                              (uint64_t*)oc, rows, c->bits * c->k)) != MK_OK) return rc;
      mk_prof_end(c);
      if (to_host) {
        MK_HIP(hipStreamSynchronize(c->stream));  // (so that sort and copy are timed apart: ~10 us)
        c->ex_st.s_sort += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gather).count();
        v.pkeys.resize(rows);
        v.pcnts.resize(rows);
        MK_HIP(hipMemcpyAsync(v.pkeys.data(), ok, rows * 8, hipMemcpyDeviceToHost, c->stream));
        MK_HIP(hipMemcpyAsync(v.pcnts.data(), oc, rows * 8, hipMemcpyDeviceToHost, c->stream));
      }
      u64 got = 0;
      MK_HIP(hipMemcpyAsync(&got, d_cursor, 8, hipMemcpyDeviceToHost, c->stream));
      MK_HIP(hipStreamSynchronize(c->stream));
      if (got != rows) {
        c->err = "export: table holds " + std::to_string(got) + " rows, expected " + std::to_string(rows);
        return MK_ERR_STATE;
      }
    }
    if (side) {  // the all-ones key (32 x 'T'): the largest key, so it goes last
      if (to_host) { v.pkeys.push_back(MK_EMPTY); v.pcnts.push_back(c->run_side); }
      else {
        u64 kk = MK_EMPTY, cc = c->run_side;
        MK_HIP(hipMemcpyAsync(d_keys_out + rows, &kk, 8, hipMemcpyHostToDevice, c->stream));
        MK_HIP(hipMemcpyAsync(d_cnts_out + rows, &cc, 8, hipMemcpyHostToDevice, c->stream));
        MK_HIP(hipStreamSynchronize(c->stream));
      }
      rows += 1;
    }
  }
  else if (c->mode == MK_MODE_HASH128) {
    v.words = 2;
    rows = c->run128_rows;
    if (!to_host && rows > cap) { c->err = "export: device buffers too small"; return MK_ERR_RANGE; }
    if (rows) {
      mk_prof_begin(c, MK_K_EXPORT);
      // compacted {hi | lo | count} + 4 n words of sort scratch
      if ((rc = mk_buf_reserve(c, c->ex128, 7 * rows * 8 + 64)) != MK_OK) return rc;
      u64* hi = (u64*)c->ex128.p;
      u64* lo = hi + rows;
      u64* cn = lo + rows;
      u64* scratch = cn + rows;
      u64* d_cursor = (u64*)((char*)c->info.p + sizeof(MkChunkInfo));
      MK_HIP(hipMemsetAsync(d_cursor, 0, 8, c->stream));
      if ((rc = mk_launch_compact128(c, (const MkSlot128*)c->run128.p, c->run128_slots, (uint64_t*)hi, (uint64_t*)lo,
                                     (uint64_t*)cn, rows, (uint64_t*)d_cursor)) != MK_OK) return rc;
      u64* ok = d_keys_out;
      u64* oc = d_cnts_out;
      if (to_host) {
        if ((rc = mk_buf_reserve(c, c->ex128_out, 3 * rows * 8 + 64)) != MK_OK) return rc;
        ok = (u64*)c->ex128_out.p;
        oc = ok + 2 * rows;
      }
      if ((rc = mk_sort_pairs128(c, (const uint64_t*)hi, (const uint64_t*)lo, (const uint64_t*)cn, rows, 2 * (c->k - 32),
                                 (uint64_t*)scratch, (uint64_t*)ok, (uint64_t*)oc)) != MK_OK) return rc;
      mk_prof_end(c);
      if (to_host) {
        MK_HIP(hipStreamSynchronize(c->stream));
        c->ex_st.s_sort += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_gather).count();
        v.pkeys.resize(2 * rows);
        v.pcnts.resize(rows);
        MK_HIP(hipMemcpyAsync(v.pkeys.data(), ok, 2 * rows * 8, hipMemcpyDeviceToHost, c->stream));
        MK_HIP(hipMemcpyAsync(v.pcnts.data(), oc, rows * 8, hipMemcpyDeviceToHost, c->stream));
      }
      u64 got = 0;
      MK_HIP(hipMemcpyAsync(&got, d_cursor, 8, hipMemcpyDeviceToHost, c->stream));
      MK_HIP(hipStreamSynchronize(c->stream));
      if (got != rows) {
        c->err = "export: two-word table holds " + std::to_string(got) + " rows, expected " + std::to_string(rows);
        return MK_ERR_STATE;
      }
    }
  }
  if (rows_out) *rows_out = rows;
  return MK_OK;
}

static int gather_ref(mk_ctx* c, ExportView& v, bool sorted) {
  const size_t rows = c->run_ref_rows;
  if (!rows) return MK_OK;
  int rc;
  const size_t k = (size_t)c->k;
  if ((rc = mk_buf_reserve(c, c->ex_keys, rows * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_cnts, rows * 8 + 64)) != MK_OK) return rc;
  u64* d_cursor = (u64*)((char*)c->info.p + sizeof(MkChunkInfo));
  MK_HIP(hipMemsetAsync(d_cursor, 0, 16, c->stream));  // [0] rows compacted, [1] rows with a bad index
  if ((rc = mk_launch_compact(c, (const MkSlot*)c->run_ref.p, c->run_ref_slots, (uint64_t*)c->ex_keys.p,
                              (uint64_t*)c->ex_cnts.p, rows, (uint64_t*)d_cursor)) != MK_OK) return rc;
  // counts by arena row (the slots know their row), on the device
  if ((rc = mk_buf_reserve(c, c->surv_cnts, rows * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_launch_rows_by_slot(c, (const uint64_t*)c->ex_keys.p, (const uint64_t*)c->ex_cnts.p, rows,
                                   (uint64_t*)c->surv_cnts.p, (uint64_t*)d_cursor + 1)) != MK_OK) return rc;
  u64 got[2] = {0, 0};
  v.rstr.resize(rows * k);
  v.rcnt.resize(rows);
  v.rorder.resize(rows);
  for (size_t i = 0; i < rows; ++i) v.rorder[i] = i;
  if (sorted) {
    // rows in byte order: radix sort of the row indices, then the rows and counts gathered in that order
    // on the device, so that the host walks them front to back
    uint64_t* d_order = nullptr;
    if ((rc = mk_sort_rows(c, (const uint8_t*)c->arena.p, rows, c->k, &d_order)) != MK_OK) return rc;
    if ((rc = mk_buf_reserve(c, c->surv_keys, rows * k + 64)) != MK_OK) return rc;
    if ((rc = mk_launch_rows_gather(c, (const uint8_t*)c->arena.p, d_order, (const uint64_t*)c->surv_cnts.p, rows, c->k,
                                    (uint8_t*)c->surv_keys.p, (uint64_t*)c->ex_keys.p)) != MK_OK) return rc;
    MK_HIP(hipMemcpyAsync(v.rstr.data(), c->surv_keys.p, rows * k, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipMemcpyAsync(v.rcnt.data(), c->ex_keys.p, rows * 8, hipMemcpyDeviceToHost, c->stream));
  } else {
    MK_HIP(hipMemcpyAsync(v.rstr.data(), c->arena.p, rows * k, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipMemcpyAsync(v.rcnt.data(), c->surv_cnts.p, rows * 8, hipMemcpyDeviceToHost, c->stream));
  }
  MK_HIP(hipMemcpyAsync(got, d_cursor, 16, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  if (got[0] != rows || got[1] != 0) {
    c->err = "export: by-reference table holds " + std::to_string(got[0]) + " rows (" + std::to_string(got[1]) +
             " with a corrupt row index), expected " + std::to_string(rows);
    return MK_ERR_STATE;
  }
  return MK_OK;
}

// packed row i of a view -> its k characters
static inline void decode_row(const mk_ctx* c, const ExportView& v, size_t i, uint8_t* out);
static inline void decode_key(const mk_ctx* c, u64 key, uint8_t* out) {
  const int k = c->k;
  if (c->alphabet == MK_ALPHABET_NT2) {
    for (int j = k - 1; j >= 0; --j) { out[j] = "ACGT"[key & 3]; key >>= 2; }
  } else {
    for (int j = k - 1; j >= 0; --j) { out[j] = (uint8_t)('A' + (key & 31)); key >>= 5; }
  }
}

static inline void decode_row(const mk_ctx* c, const ExportView& v, size_t i, uint8_t* out) {
  if (v.words == 1) { decode_key(c, v.pkeys[i], out); return; }
  const u64 hi = v.pkeys[2 * i], lo = v.pkeys[2 * i + 1];  // left-aligned: base j < 32 in hi, the rest in lo
  const int k = c->k;
  if (c->alphabet == MK_ALPHABET_AA5) {  // amino acids: the number sum(code_j * 32^(k-1-j)) in (hi, lo)
    unsigned __int128 x = ((unsigned __int128)hi << 64) | lo;
    for (int j = k - 1; j >= 0; --j) { out[j] = (uint8_t)('A' + (unsigned)(x & 31)); x >>= 5; }
    return;
  }
  for (int j = 0; j < 32; ++j) out[j] = "ACGT"[(hi >> (62 - 2 * j)) & 3];
  for (int j = 32; j < k; ++j) out[j] = "ACGT"[(lo >> (62 - 2 * (j - 32))) & 3];
}

// Visit every row in sorted(str) order: a 2-way merge of the packed rows (decoded on the fly)
// and the by-reference rows.
template <class F>
static void merged_rows(const mk_ctx* c, const ExportView& v, F&& f) {
  const size_t k = (size_t)c->k, np = v.packed_rows(), nr = v.rorder.size();
  std::vector<uint8_t> buf(k ? k : 1);
  size_t i = 0, j = 0;
  bool have = false;
  while (i < np || j < nr) {
    if (i < np && !have) { decode_row(c, v, i, buf.data()); have = true; }
    bool take_packed;
    if (i >= np) take_packed = false;
    else if (j >= nr) take_packed = true;
    else take_packed = memcmp(buf.data(), v.rstr.data() + v.rorder[j] * k, k) < 0;
    if (take_packed) { f(buf.data(), v.pcnts[i]); ++i; have = false; }
    else { f(v.rstr.data() + v.rorder[j] * k, v.rcnt[v.rorder[j]]); ++j; }
  }
}

static int write_view_tsv(mk_ctx* c, const ExportView& v, const char* path, const char* basename, size_t* rows_out);
static int build_view(mk_ctx* c, ExportView& v) {
  MK_HIP(hipSetDevice(c->device));
  { int rc_ = settle(c); if (rc_) return rc_; }
  const auto t0 = std::chrono::steady_clock::now();
  c->ex_st = mk_export_stats_t{};
  int rc = gather_packed(c, v, nullptr, nullptr, 0, nullptr, true);
  if (rc) return rc;
  rc = gather_ref(c, v, true);
  // (s_sort was added up inside; the rest of the gathering is the copies to the host)
  c->ex_st.s_d2h = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() - c->ex_st.s_sort;
  c->ex_st.s_total = c->ex_st.s_sort + c->ex_st.s_d2h;
  c->ex_st.rows = v.packed_rows() + v.rorder.size();
  return rc;
}

extern "C" int mk_export_stats(mk_ctx* c, mk_export_stats_t* out) {
  if (!c || !out) return MK_ERR_ARG;
  *out = c->ex_st;
  return MK_OK;
}

extern "C" int mk_export_size(mk_ctx* c, size_t* rows) {
  if (!c || !rows) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (c->mode == MK_MODE_DENSE) {
    ExportView v;
    MK_HIP(hipSetDevice(c->device));
    int rc = gather_packed(c, v, nullptr, nullptr, 0, nullptr, true);
    if (rc) return rc;
    *rows = v.packed_rows() + c->run_ref_rows;
  } else {
    *rows = c->run_rows + (c->run_side ? 1 : 0) + c->run_ref_rows + c->run128_rows;
  }
  c->st.rows = *rows;
  return MK_OK;
}

extern "C" int mk_export(mk_ctx* c, uint8_t* kmers, uint64_t* counts, size_t rows_cap) {
  if (!c) return MK_ERR_ARG;
  ExportView v;
  int rc = build_view(c, v);
  if (rc) return rc;
  const size_t rows = v.packed_rows() + v.rorder.size();
  if (rows > rows_cap) { c->err = "mk_export: rows_cap too small"; return MK_ERR_RANGE; }
  if (rows && (!kmers || !counts)) { c->err = "mk_export: NULL output"; return MK_ERR_ARG; }
  const size_t k = (size_t)c->k;
  size_t at = 0;
  const auto t_f = std::chrono::steady_clock::now();
  merged_rows(c, v, [&](const uint8_t* s, u64 n) {
    memcpy(kmers + at * k, s, k);
    counts[at] = n;
    ++at;
  });
  c->ex_st.s_format = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f).count();
  c->ex_st.s_total += c->ex_st.s_format;
  return MK_OK;
}

// mk_tsv.hip: sorted device rows -> TSV text in c->seq
int mk_launch_tsv_format(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_cnts, size_t rows, int words, uint64_t* d_len,
                         uint64_t* d_off, size_t* text_bytes);

// The table written from the device: compaction and sort as for any export, the rows formatted by a kernel (mk_tsv.hip),
// the text copied out through two pinned blocks while the one before is written to the file.  For tables whose rows are
// all packed keys (no rows kept as text: those are merged in on the host, write_view_tsv).
static int write_tsv_from_device(mk_ctx* c, const char* path, const char* basename, size_t* rows_out) {
  using Clk = std::chrono::steady_clock;
  auto since = [](Clk::time_point t) { return std::chrono::duration<double>(Clk::now() - t).count(); };
  MK_HIP(hipSetDevice(c->device));
  int rc;
  if ((rc = settle(c)) != MK_OK) return rc;
  const auto t0 = Clk::now();
  c->ex_st = mk_export_stats_t{};
  const int words = c->mode == MK_MODE_HASH128 ? 2 : 1;
  size_t cap = 0;
  if (c->mode == MK_MODE_DENSE) cap = c->run_slots;
  else if (c->mode == MK_MODE_HASH64) cap = c->run_rows + 1;
  else cap = c->run128_rows;
  if (rows_out) *rows_out = 0;
  if (!cap) return MK_OK;
  MkDevBuf& kb = c->mode == MK_MODE_HASH128 ? c->ex128_out : c->ex_keys2;
  if ((rc = mk_buf_reserve(c, kb, (cap * (size_t)words + cap) * 8 + 64)) != MK_OK) return rc;  // keys, then counts
  u64* d_keys = (u64*)kb.p;
  u64* d_cnts = d_keys + cap * (size_t)words;
  ExportView v;
  size_t rows = 0;
  if ((rc = gather_packed(c, v, d_keys, d_cnts, cap, &rows, /*to_host=*/false)) != MK_OK) return rc;
  MK_HIP(hipStreamSynchronize(c->stream));
  c->ex_st.s_sort = since(t0);
  c->ex_st.rows = rows;
  if (rows_out) *rows_out = rows;
  if (!rows) return MK_OK;  // bin/mercat2.py:135-137: no file when nothing survives
  const auto t1 = Clk::now();
  // offsets and lengths: two scratch arrays of rows + 1 words (the compaction's buffers are free again)
  if ((rc = mk_buf_reserve(c, c->ex_keys, (rows + 1) * 8 + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_cnts, (rows + 1) * 8 + 64)) != MK_OK) return rc;
  size_t text = 0;
  if ((rc = mk_launch_tsv_format(c, (const uint64_t*)d_keys, (const uint64_t*)d_cnts, rows, words, (uint64_t*)c->ex_keys.p,
                                 (uint64_t*)c->ex_cnts.p, &text)) != MK_OK) return rc;
  // two registered blocks, kept with the context
  const size_t piece = (size_t)8 << 20;
  if (c->tsv_pin_bytes < 2 * piece) {
    void* p = aligned_alloc(4096, 2 * piece);
    if (!p) { c->err = "mk_write_tsv: out of host memory"; return MK_ERR_NOMEM; }
    memset(p, 0, 2 * piece);  // (touched before it is pinned)
    const hipError_t he0 = hipHostRegister(p, 2 * piece, hipHostRegisterDefault);
    if (he0 != hipSuccess) { free(p); c->err = std::string("hipHostRegister: ") + hipGetErrorString(he0); return MK_ERR_HIP; }
    c->tsv_pin = p;
    c->tsv_pin_bytes = 2 * piece;
  }
  char* pin[2] = {(char*)c->tsv_pin, (char*)c->tsv_pin + piece};
  hipEvent_t ev[2] = {nullptr, nullptr};
  for (auto& e : ev) MK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  FILE* f = fopen(path, "wb");
  if (!f) {
    for (auto& e : ev) (void)hipEventDestroy(e);
    c->err = std::string("mk_write_tsv: cannot open ") + path;
    return MK_ERR_IO;
  }
  setvbuf(f, nullptr, _IONBF, 0);  // (whole blocks: no second copy through stdio's buffer)
  const std::string head = std::string("k-mer\t") + basename + "_Count\n";
  double s_write = 0, s_wait = 0;
  auto timed_write = [&](const void* p, size_t n) {
    const auto tw = Clk::now();
    const size_t put = fwrite(p, 1, n, f);
    s_write += since(tw);
    return put == n;
  };
  bool ok = timed_write(head.data(), head.size());
  const size_t npieces = (text + piece - 1) / piece;
  hipError_t he = hipSuccess;
  for (size_t i = 0; i <= npieces && ok && he == hipSuccess; ++i) {
    if (i < npieces) {  // copy piece i out while piece i - 1 is written
      const size_t a = i * piece, n = std::min(piece, text - a);
      he = hipMemcpyAsync(pin[i & 1], (const char*)c->seq.p + a, n, hipMemcpyDeviceToHost, c->stream);
      if (he == hipSuccess) he = hipEventRecord(ev[i & 1], c->stream);
    }
    if (i > 0 && he == hipSuccess) {
      const size_t a = (i - 1) * piece, n = std::min(piece, text - a);
      const auto tw = Clk::now();
      he = hipEventSynchronize(ev[(i - 1) & 1]);
      s_wait += since(tw);
      if (he == hipSuccess) ok = timed_write(pin[(i - 1) & 1], n);
    }
  }
  (void)hipStreamSynchronize(c->stream);
  for (auto& e : ev) (void)hipEventDestroy(e);
  const auto tc = Clk::now();
  const bool bad_close = fclose(f) != 0;
  s_write += since(tc);
  const double all = since(t1);
  c->ex_st.bytes = head.size() + text;
  c->ex_st.s_write = s_write;
  c->ex_st.s_d2h = s_wait;
  c->ex_st.s_format = all - s_write - s_wait;  // (lengths, scan, fill kernel and what the loop itself costs)
  c->ex_st.s_total = since(t0);
  if (he != hipSuccess) { c->err = std::string("mk_write_tsv: copy of the text: ") + hipGetErrorString(he); return MK_ERR_HIP; }
  if (!ok || bad_close) { c->err = std::string("mk_write_tsv: write failed: ") + path; return MK_ERR_IO; }
  return MK_OK;
}

extern "C" int mk_write_tsv(mk_ctx* c, const char* path, const char* basename, size_t* rows_out) {
  if (!c || !path || !basename) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  static const bool host_only = getenv("MK_TSV_HOST") != nullptr;  // (A/B and tests: the host formatter for every table)
  if (!host_only && c->mode != MK_MODE_BYREF && c->run_ref_rows == 0 && c->bits != 0)
    return write_tsv_from_device(c, path, basename, rows_out);
  ExportView v;
  int rc = build_view(c, v);
  if (rc) return rc;
  return write_view_tsv(c, v, path, basename, rows_out);
}

static int write_view_tsv(mk_ctx* c, const ExportView& v, const char* path, const char* basename, size_t* rows_out) {
  const size_t rows = v.packed_rows() + v.rorder.size();
  if (rows_out) *rows_out = rows;
  if (!rows) return MK_OK;  // bin/mercat2.py:135-137: no file when nothing survives
  FILE* f = fopen(path, "wb");
  if (!f) { c->err = std::string("mk_write_tsv: cannot open ") + path; return MK_ERR_IO; }
  std::vector<char> out;
  out.reserve(1 << 22);
  const size_t k = (size_t)c->k;
  const auto t_f = std::chrono::steady_clock::now();
  double s_write = 0;
  uint64_t bytes = 0;
  auto flush = [&]() {
    if (!out.empty()) {
      const auto tw = std::chrono::steady_clock::now();
      fwrite(out.data(), 1, out.size(), f);
      s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      bytes += out.size();
      out.clear();
    }
  };
  const std::string head = std::string("k-mer\t") + basename + "_Count\n";
  out.insert(out.end(), head.begin(), head.end());
  merged_rows(c, v, [&](const uint8_t* s, u64 n) {
    out.insert(out.end(), (const char*)s, (const char*)s + k);
    out.push_back('\t');
    char num[24];
    int len = 0;
    do { num[len++] = (char)('0' + n % 10); n /= 10; } while (n);
    while (len) out.push_back(num[--len]);
    out.push_back('\n');
    if (out.size() > (1u << 22) - 4096 - k) flush();
  });
  flush();
  const bool bad = ferror(f) != 0;
  const auto tc = std::chrono::steady_clock::now();
  const bool bad_close = fclose(f) != 0;
  s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tc).count();
  c->ex_st.bytes = bytes;
  c->ex_st.s_write = s_write;
  c->ex_st.s_format = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_f).count() - s_write;
  c->ex_st.s_total += c->ex_st.s_format + s_write;
  if (bad_close || bad) { c->err = std::string("mk_write_tsv: write failed: ") + path; return MK_ERR_IO; }
  return MK_OK;
}

// ----------------------------------------------------- one table spread over several contexts by key range
// After mk_merge_devices(MK_MERGE_RANGES) context i holds the rows of key range i.  Every context sorts its own
// rows on its own GPU (one host thread each), the host concatenates the packed rows in context order and merges
// the few rows kept as text into them.
static int build_view_multi(mk_ctx* const* ctxs, int n, ExportView& all) {
  if (!ctxs || n < 1 || !ctxs[0]) return MK_ERR_ARG;
  mk_ctx* c0 = ctxs[0];
  for (int j = 0; j < n; ++j) {
    if (!ctxs[j]) { c0->err = "multi export: a context is NULL"; return MK_ERR_ARG; }
    if (ctxs[j]->k != c0->k || ctxs[j]->alphabet != c0->alphabet || ctxs[j]->mode != c0->mode) {
      c0->err = "multi export: contexts differ in alphabet or k";
      return MK_ERR_ARG;
    }
    if (ctxs[j]->in_chunk) { c0->err = "multi export: a chunk is open"; return MK_ERR_STATE; }
  }
  std::vector<ExportView> views((size_t)n);
  std::vector<int> rcs((size_t)n, MK_OK);
  {
    std::vector<std::thread> th;
    for (int j = 1; j < n; ++j) th.emplace_back([&, j] { rcs[j] = build_view(ctxs[j], views[j]); });
    rcs[0] = build_view(ctxs[0], views[0]);
    for (auto& t : th) t.join();
  }
  for (int j = 0; j < n; ++j)
    if (rcs[j]) { if (j) c0->err = ctxs[j]->err; return rcs[j]; }
  all.words = views[0].words;
  const size_t w = (size_t)all.words, k = (size_t)c0->k;
  size_t np = 0, nr = 0;
  for (auto& v : views) { np += v.packed_rows(); nr += v.rorder.size(); }
  all.pkeys.reserve(np * w);
  all.pcnts.reserve(np);
  for (int j = 0; j < n; ++j) {
    const ExportView& v = views[j];
    if (!v.packed_rows()) continue;
    if (!all.pcnts.empty()) {  // ranges ascending and disjoint: last key so far < first key of this context
      const u64* a = all.pkeys.data() + all.pkeys.size() - w;
      const u64* b = v.pkeys.data();
      const bool less = w == 1 ? a[0] < b[0] : (a[0] < b[0] || (a[0] == b[0] && a[1] < b[1]));
      if (!less) {
        c0->err = "multi export: context " + std::to_string(j) + " does not continue the key ranges of the contexts before it "
                  "(call mk_merge_devices with MK_MERGE_RANGES first)";
        return MK_ERR_STATE;
      }
    }
    all.pkeys.insert(all.pkeys.end(), v.pkeys.begin(), v.pkeys.end());
    all.pcnts.insert(all.pcnts.end(), v.pcnts.begin(), v.pcnts.end());
  }
  if (nr) {  // rows kept as text (after a merge they all sit in ctxs[0]; accept them anywhere): one sorted list
    all.rstr.reserve(nr * k);
    for (auto& v : views)
      for (size_t i = 0; i < v.rorder.size(); ++i) {
        all.rstr.insert(all.rstr.end(), v.rstr.begin() + v.rorder[i] * k, v.rstr.begin() + (v.rorder[i] + 1) * k);
        all.rcnt.push_back(v.rcnt[v.rorder[i]]);
      }
    all.rorder.resize(nr);
    for (size_t i = 0; i < nr; ++i) all.rorder[i] = i;
    std::sort(all.rorder.begin(), all.rorder.end(), [&](u64 x, u64 y) { return memcmp(all.rstr.data() + x * k, all.rstr.data() + y * k, k) < 0; });
    for (size_t i = 1; i < nr; ++i)
      if (memcmp(all.rstr.data() + all.rorder[i - 1] * k, all.rstr.data() + all.rorder[i] * k, k) == 0) {
        c0->err = "multi export: the same text row in two contexts (merge the contexts first)";
        return MK_ERR_STATE;
      }
  }
  return MK_OK;
}

extern "C" int mk_export_size_multi(mk_ctx* const* ctxs, int n, size_t* rows) {
  if (!ctxs || n < 1 || !rows) return MK_ERR_ARG;
  size_t total = 0;
  for (int j = 0; j < n; ++j) {
    size_t r = 0;
    int rc = mk_export_size(ctxs[j], &r);
    if (rc) { if (j && ctxs[0] && ctxs[j]) ctxs[0]->err = ctxs[j]->err; return rc; }
    total += r;
  }
  *rows = total;
  return MK_OK;
}

extern "C" int mk_export_multi(mk_ctx* const* ctxs, int n, uint8_t* kmers, uint64_t* counts, size_t rows_cap) {
  ExportView v;
  int rc = build_view_multi(ctxs, n, v);
  if (rc) return rc;
  mk_ctx* c = ctxs[0];
  const size_t rows = v.packed_rows() + v.rorder.size();
  if (rows > rows_cap) { c->err = "mk_export_multi: rows_cap too small"; return MK_ERR_RANGE; }
  if (rows && (!kmers || !counts)) { c->err = "mk_export_multi: NULL output"; return MK_ERR_ARG; }
  const size_t k = (size_t)c->k;
  size_t at = 0;
  merged_rows(c, v, [&](const uint8_t* s, u64 cnt) {
    memcpy(kmers + at * k, s, k);
    counts[at] = cnt;
    ++at;
  });
  return MK_OK;
}

extern "C" int mk_write_tsv_multi(mk_ctx* const* ctxs, int n, const char* path, const char* basename, size_t* rows_out) {
  if (!path || !basename) return MK_ERR_ARG;
  ExportView v;
  int rc = build_view_multi(ctxs, n, v);
  if (rc) return rc;
  return write_view_tsv(ctxs[0], v, path, basename, rows_out);
}

// ------------------------------------------------------------- combined table of several samples
// merge_tsv (lib/mercat2_report.py:98-156) from the tables themselves: a k-way merge of the samples'
// sorted rows (each: device radix sort of the packed keys + by-reference rows, as for mk_export).
namespace {
struct RowIter {  // the rows of one sample in sorted(str) order
  const mk_ctx* c;
  const ExportView* v;
  size_t i = 0, j = 0;
  std::vector<uint8_t> buf;
  const uint8_t* cur = nullptr;
  u64 cnt = 0;
  RowIter(const mk_ctx* c_, const ExportView* v_) : c(c_), v(v_), buf((size_t)c_->k + 1) {}
  bool next() {
    const size_t k = (size_t)c->k, np = v->packed_rows(), nr = v->rorder.size();
    if (i >= np && j >= nr) { cur = nullptr; return false; }
    bool take_packed;
    if (i < np) decode_row(c, *v, i, buf.data());
    if (i >= np) take_packed = false;
    else if (j >= nr) take_packed = true;
    else take_packed = memcmp(buf.data(), v->rstr.data() + v->rorder[j] * k, k) < 0;
    if (take_packed) { cur = buf.data(); cnt = v->pcnts[i]; ++i; }
    else { cur = v->rstr.data() + v->rorder[j] * k; cnt = v->rcnt[v->rorder[j]]; ++j; }
    return true;
  }
};

// f(kmer, counts[n]) for every k-mer present in any sample, in sorted order; absent = 0.
// as_reference: the rows exactly as merge_tsv's streaming loop produces them (lib/mercat2_report.py:128-152).  That
// loop picks the next k-mer only among the samples that ADVANCED in the current step (:131, :149-150) and, for a
// sample whose current key is not greater than the k-mer being written, writes that sample's count whatever its
// key is (:137-140).  So a key held only by samples that did not advance is never written as a row of its own: its
// count lands in a later row.  With as_reference the same rows come out (tables that share nearly all their keys --
// k = 5 on genomes -- are not affected); without it the table is the true union.
template <class F>
int merged_samples(mk_ctx* const* ctxs, int n, F&& f, bool as_reference = false) {
  if (!ctxs || n < 1 || !ctxs[0]) return MK_ERR_ARG;
  mk_ctx* c0 = ctxs[0];
  for (int s = 0; s < n; ++s) {
    if (!ctxs[s]) { c0->err = "merged table: a context is NULL"; return MK_ERR_ARG; }
    if (ctxs[s]->k != c0->k) { c0->err = "merged table: contexts differ in k"; return MK_ERR_ARG; }
    if (ctxs[s]->in_chunk) { c0->err = "merged table: a chunk is open"; return MK_ERR_STATE; }
  }
  std::vector<ExportView> views((size_t)n);
  std::vector<RowIter> it;
  it.reserve((size_t)n);
  for (int s = 0; s < n; ++s) {
    int rc = build_view(ctxs[s], views[s]);
    if (rc) { if (s) c0->err = ctxs[s]->err; return rc; }
    it.emplace_back(ctxs[s], &views[s]);
    it.back().next();
  }
  const size_t k = (size_t)c0->k;
  std::vector<u64> row((size_t)n);
  std::vector<uint8_t> key(k + 1);
  if (as_reference) {
    const uint8_t* best = nullptr;
    for (int s = 0; s < n; ++s)
      if (it[s].cur && (!best || memcmp(it[s].cur, best, k) < 0)) best = it[s].cur;
    if (!best) return MK_OK;
    memcpy(key.data(), best, k);
    std::vector<uint8_t> next(k + 1);
    for (;;) {
      bool have_next = false;
      for (int s = 0; s < n; ++s) {
        if (!it[s].cur || memcmp(it[s].cur, key.data(), k) > 0) { row[s] = 0; continue; }
        row[s] = it[s].cnt;  // (whatever this sample's key is: see above)
        it[s].next();
        if (it[s].cur && (!have_next || memcmp(it[s].cur, next.data(), k) < 0)) { memcpy(next.data(), it[s].cur, k); have_next = true; }
      }
      f(key.data(), row.data());
      if (!have_next) break;
      key.swap(next);
    }
    return MK_OK;
  }
  for (;;) {
    const uint8_t* best = nullptr;
    for (int s = 0; s < n; ++s)
      if (it[s].cur && (!best || memcmp(it[s].cur, best, k) < 0)) best = it[s].cur;
    if (!best) break;
    memcpy(key.data(), best, k);
    for (int s = 0; s < n; ++s) {
      if (it[s].cur && memcmp(it[s].cur, key.data(), k) == 0) { row[s] = it[s].cnt; it[s].next(); }
      else row[s] = 0;
    }
    f(key.data(), row.data());
  }
  return MK_OK;
}
}  // namespace

extern "C" int mk_merged_export(mk_ctx* const* ctxs, int n, uint8_t* kmers, uint64_t* matrix, size_t rows_cap, size_t* rows) {
  if (!rows) return MK_ERR_ARG;
  size_t at = 0;
  bool short_cap = false;
  const size_t k = ctxs && ctxs[0] ? (size_t)ctxs[0]->k : 0;
  int rc = merged_samples(ctxs, n, [&](const uint8_t* s, const u64* counts) {
    if (kmers && matrix) {
      if (at < rows_cap) {
        memcpy(kmers + at * k, s, k);
        memcpy(matrix + at * (size_t)n, counts, (size_t)n * sizeof(u64));
      } else short_cap = true;
    }
    ++at;
  });
  if (rc) return rc;
  *rows = at;
  if (short_cap) { ctxs[0]->err = "mk_merged_export: rows_cap too small"; return MK_ERR_RANGE; }
  return MK_OK;
}

static int write_merged(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column, const char* path,
                        size_t* rows_out, bool as_reference);
extern "C" int mk_write_merged_tsv(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column,
                                   const char* path, size_t* rows_out) {
  return write_merged(ctxs, n, names, first_column, path, rows_out, false);
}
extern "C" int mk_write_merged_tsv_as_reference(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column,
                                                const char* path, size_t* rows_out) {
  return write_merged(ctxs, n, names, first_column, path, rows_out, true);
}
static int write_merged(mk_ctx* const* ctxs, int n, const char* const* names, const char* first_column, const char* path,
                        size_t* rows_out, bool as_reference) {
  if (!ctxs || n < 1 || !ctxs[0] || !names || !first_column || !path) return MK_ERR_ARG;
  mk_ctx* c = ctxs[0];
  FILE* f = fopen(path, "wb");
  if (!f) { c->err = std::string("mk_write_merged_tsv: cannot open ") + path; return MK_ERR_IO; }
  std::vector<char> out;
  out.reserve(1 << 22);
  const auto t_f = std::chrono::steady_clock::now();
  double s_write = 0;
  uint64_t bytes = 0;
  auto flush = [&]() {
    if (!out.empty()) {
      const auto tw = std::chrono::steady_clock::now();
      fwrite(out.data(), 1, out.size(), f);
      s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      bytes += out.size();
      out.clear();
    }
  };
  {
    std::string head = first_column;
    for (int s = 0; s < n; ++s) { head += '\t'; head += names[s] ? names[s] : ""; }
    head += '\n';
    out.insert(out.end(), head.begin(), head.end());
  }
  const size_t k = (size_t)c->k;
  size_t rows = 0;
  int rc = merged_samples(ctxs, n, [&](const uint8_t* s, const u64* counts) {
    out.insert(out.end(), (const char*)s, (const char*)s + k);
    for (int q = 0; q < n; ++q) {
      out.push_back('\t');
      u64 v = counts[q];
      char num[24];
      int len = 0;
      do { num[len++] = (char)('0' + v % 10); v /= 10; } while (v);
      while (len) out.push_back(num[--len]);
    }
    out.push_back('\n');
    ++rows;
    if (out.size() > (1u << 22) - 4096 - k - 24 * (size_t)n) flush();
  }, as_reference);
  flush();
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad) { c->err = std::string("mk_write_merged_tsv: write failed: ") + path; return MK_ERR_IO; }
  if (rc) return rc;
  if (rows_out) *rows_out = rows;
  return MK_OK;
}

// merge_tsv_T (lib/mercat2_report.py:160-194): the same matrix with samples as rows: "sample\t<k-mers>\n", then one
// line per sample.  The reference lists the k-mer columns in the iteration order of a Python set (different in
// every process); here they are sorted.  Consumers address columns by label (bin/mercat2.py:354-355).
extern "C" int mk_write_merged_tsv_t(mk_ctx* const* ctxs, int n, const char* const* names, const char* path, size_t* rows_out) {
  if (!ctxs || n < 1 || !ctxs[0] || !names || !path) return MK_ERR_ARG;
  mk_ctx* c = ctxs[0];
  const size_t k = (size_t)c->k;
  std::vector<uint8_t> kmers;
  std::vector<u64> matrix;  // rows x n
  int rc = merged_samples(ctxs, n, [&](const uint8_t* s, const u64* counts) {
    kmers.insert(kmers.end(), s, s + k);
    matrix.insert(matrix.end(), counts, counts + n);
  });
  if (rc) return rc;
  const size_t rows = k ? kmers.size() / k : 0;
  FILE* f = fopen(path, "wb");
  if (!f) { c->err = std::string("mk_write_merged_tsv_t: cannot open ") + path; return MK_ERR_IO; }
  std::vector<char> out;
  out.reserve(1 << 22);
  const auto t_f = std::chrono::steady_clock::now();
  double s_write = 0;
  uint64_t bytes = 0;
  auto flush = [&]() {
    if (!out.empty()) {
      const auto tw = std::chrono::steady_clock::now();
      fwrite(out.data(), 1, out.size(), f);
      s_write += std::chrono::duration<double>(std::chrono::steady_clock::now() - tw).count();
      bytes += out.size();
      out.clear();
    }
  };
  const char* head = "sample";
  out.insert(out.end(), head, head + 6);
  for (size_t r = 0; r < rows; ++r) {
    out.push_back('\t');
    out.insert(out.end(), (const char*)kmers.data() + r * k, (const char*)kmers.data() + (r + 1) * k);
    if (out.size() > (1u << 22) - 4096 - k) flush();
  }
  out.push_back('\n');
  for (int s = 0; s < n; ++s) {
    const char* nm = names[s] ? names[s] : "";
    out.insert(out.end(), nm, nm + strlen(nm));
    for (size_t r = 0; r < rows; ++r) {
      out.push_back('\t');
      u64 v = matrix[r * (size_t)n + (size_t)s];
      char num[24];
      int len = 0;
      do { num[len++] = (char)('0' + v % 10); v /= 10; } while (v);
      while (len) out.push_back(num[--len]);
      if (out.size() > (1u << 22) - 4096) flush();
    }
    out.push_back('\n');
  }
  flush();
  const bool bad = ferror(f) != 0;
  if (fclose(f) != 0 || bad) { c->err = std::string("mk_write_merged_tsv_t: write failed: ") + path; return MK_ERR_IO; }
  if (rows_out) *rows_out = rows;
  return MK_OK;
}

// ------------------------------------------------------------------------ alpha diversity
extern "C" int mk_alpha_stats(mk_ctx* c, mk_alpha_t* out) {
  if (!c || !out) return MK_ERR_ARG;
  if (c->in_chunk) { c->err = "mk_alpha_stats: a chunk is open"; return MK_ERR_STATE; }
  MK_HIP(hipSetDevice(c->device));
  int rc;
  if ((rc = mk_buf_reserve(c, c->ex_tmp, 16 * sizeof(u64))) != MK_OK) return rc;
  if ((rc = mk_launch_alpha(c, (unsigned long long*)c->ex_tmp.p)) != MK_OK) return rc;
  u64 h[16];
  MK_HIP(hipMemcpyAsync(h, c->ex_tmp.p, sizeof h, hipMemcpyDeviceToHost, c->stream));
  MK_HIP(hipStreamSynchronize(c->stream));
  memset(out, 0, sizeof *out);
  out->observed = h[0];
  out->total = h[1];
  for (int i = 1; i <= 10; ++i) out->freq[i] = h[2 + i];
  memcpy(&out->sum_sq, &h[13], 8);
  memcpy(&out->sum_clnc, &h[14], 8);
  if (c->run_side) {  // the one key that lives beside the packed table (32 x 'T')
    const u64 v = c->run_side;
    out->observed += 1;
    out->total += v;
    if (v <= 10) out->freq[v] += 1;
    out->sum_sq += (double)v * (double)v;
    out->sum_clnc += (double)v * log((double)v);
  }
  return MK_OK;
}

// Give back the per-chunk working memory (raw text, packed words, partition and survivor buffers,
// chunk tables); the running table stays, so the sample can still be exported or merged. The
// buffers come back on the next chunk.
extern "C" int mk_trim(mk_ctx* c) {
  if (!c) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (c->in_chunk) { c->err = "mk_trim: a chunk is open"; return MK_ERR_STATE; }
  MK_HIP(hipSetDevice(c->device));
  MK_HIP(hipStreamSynchronize(c->stream));
  MkDevBuf* scratch[] = {&c->raw, &c->seq, &c->codes, &c->bad, &c->tile_maps, &c->ctab, &c->rtab_chunk, &c->part,
                         &c->surv_keys, &c->surv_cnts, &c->surv_keys2, &c->ex_keys, &c->ex_cnts, &c->ex_keys2, &c->ex_cnts2, &c->ex_tmp, &c->ex128, &c->ex128_out,
                         &c->xfer_out, &c->xfer_in};
  for (auto* b : scratch)
    if (!(b == &c->ctab && c->mode == MK_MODE_DENSE)) buf_free(*b);  // the dense bins are allocated once, at mk_create
  if (c->mode != MK_MODE_DENSE) c->ctab_slots = 0;
  c->rtab_chunk_slots = 0;
  c->part_reuse_ok = false;
  if (c->ingest_ring) { (void)hipHostFree(c->ingest_ring); c->ingest_ring = nullptr; c->ingest_ring_bytes = 0; }
  return MK_OK;
}

// ------------------------------------------------------------------- multi-GPU plumbing
extern "C" int mk_export_pairs_device(mk_ctx* c, uint64_t* d_keys, uint64_t* d_counts, size_t cap, size_t* rows) {
  if (!c || !rows) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (c->mode == MK_MODE_BYREF) { *rows = 0; return MK_OK; }  // rows travel as text (mk_export_exotic)
  MK_HIP(hipSetDevice(c->device));
  ExportView v;
  return gather_packed(c, v, (u64*)d_keys, (u64*)d_counts, cap, rows, false);
}

extern "C" int mk_import_pairs_device(mk_ctx* c, const uint64_t* d_keys, const uint64_t* d_counts, size_t rows) {
  if (!c) return MK_ERR_ARG;
  if (!rows) return MK_OK;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (c->mode == MK_MODE_BYREF) { c->err = "mk_import_pairs_device: context has no packed table"; return MK_ERR_STATE; }
  MK_HIP(hipSetDevice(c->device));
  int rc;
  MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  if (c->mode == MK_MODE_HASH128) {  // d_keys: {hi, lo} per row
    if ((rc = grow_run128(c, c->run128_rows + rows)) != MK_OK) return rc;
    if ((rc = mk_launch_import128_pairs(c, d_keys, d_counts, rows)) != MK_OK) return rc;
    if ((rc = pull_info(c)) != MK_OK) return rc;
    c->run128_rows += (size_t)c->h_info->new_rows;
    return MK_OK;
  }
  // (the all-ones key travels as an ordinary pair, anywhere in the rows: the kernel sets it aside)
  if (c->mode == MK_MODE_HASH64 && (rc = grow_run64(c, c->run_rows + rows)) != MK_OK) return rc;
  if ((rc = mk_launch_import_pairs(c, d_keys, d_counts, rows)) != MK_OK) return rc;
  if ((rc = pull_info(c)) != MK_OK) return rc;
  c->run_rows += (size_t)c->h_info->new_rows;
  if (c->mode == MK_MODE_HASH64) c->run_side += c->h_info->side;
  return MK_OK;
}

extern "C" int mk_export_exotic(mk_ctx* c, uint8_t* kmers, uint64_t* counts, size_t cap, size_t* rows) {
  if (!c || !rows) return MK_ERR_ARG;
  MK_HIP(hipSetDevice(c->device));
  { int rc_ = settle(c); if (rc_) return rc_; }
  ExportView v;
  int rc = gather_ref(c, v, true);
  if (rc) return rc;
  *rows = v.rorder.size();
  if (!kmers && !counts) return MK_OK;  // size query
  if (*rows > cap) { c->err = "mk_export_exotic: cap too small"; return MK_ERR_RANGE; }
  const size_t k = (size_t)c->k;
  for (size_t i = 0; i < v.rorder.size(); ++i) {
    memcpy(kmers + i * k, v.rstr.data() + v.rorder[i] * k, k);
    counts[i] = v.rcnt[v.rorder[i]];
  }
  return MK_OK;
}

extern "C" int mk_import_exotic(mk_ctx* c, const uint8_t* kmers, const uint64_t* counts, size_t rows) {
  if (!c) return MK_ERR_ARG;
  if (!rows) return MK_OK;
  if (!kmers || !counts) return MK_ERR_ARG;
  MK_HIP(hipSetDevice(c->device));
  { int rc_ = settle(c); if (rc_) return rc_; }
  int rc;
  const size_t k = (size_t)c->k;
  if ((rc = mk_buf_reserve(c, c->ex_keys2, rows * k + 64)) != MK_OK) return rc;
  if ((rc = mk_buf_reserve(c, c->ex_cnts2, rows * 8 + 64)) != MK_OK) return rc;
  MK_HIP(hipMemcpyAsync(c->ex_keys2.p, kmers, rows * k, hipMemcpyHostToDevice, c->stream));
  MK_HIP(hipMemcpyAsync(c->ex_cnts2.p, counts, rows * 8, hipMemcpyHostToDevice, c->stream));
  MK_HIP(hipMemsetAsync(c->info.p, 0, sizeof(MkChunkInfo), c->stream));
  if ((rc = grow_run_ref(c, c->run_ref_rows + rows)) != MK_OK) return rc;
  if ((rc = mk_launch_import_ref(c, (const uint8_t*)c->ex_keys2.p, (const uint64_t*)c->ex_cnts2.p, rows)) != MK_OK) return rc;
  if ((rc = pull_info(c)) != MK_OK) return rc;
  c->run_ref_rows += (size_t)c->h_info->new_rows_ref;
  return MK_OK;
}

// Drop every row of the running table whose count is below min_count: the filter of a sample that is ONE
// chunk but was counted in pieces (record ranges on several GPUs, unfiltered) and merged
// (lib/mercat2_kmers.py:73-76 applies it once per file).
extern "C" int mk_filter_min(mk_ctx* c, uint64_t min_count) {
  if (!c) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  if (c->in_chunk) { c->err = "mk_filter_min: a chunk is open"; return MK_ERR_STATE; }
  if (min_count <= 1) return MK_OK;
  MK_HIP(hipSetDevice(c->device));
  int rc;
  u64* d_kept = (u64*)((char*)c->info.p + sizeof(MkChunkInfo));
  u64 kept = 0;
  if (c->mode == MK_MODE_DENSE) {
    if ((rc = mk_launch_refilter_dense(c, (uint64_t*)c->run.p, c->run_slots, min_count)) != MK_OK) return rc;
  } else if (c->mode == MK_MODE_HASH64 && c->run_slots) {
    MkDevBuf nb;
    if ((rc = mk_buf_reserve(c, nb, c->run_slots * sizeof(MkSlot))) != MK_OK) return rc;
    if ((rc = mk_launch_clear_slots(c, (MkSlot*)nb.p, c->run_slots)) != MK_OK) return rc;
    MK_HIP(hipMemsetAsync(d_kept, 0, 8, c->stream));
    if ((rc = mk_launch_refilter64(c, (const MkSlot*)c->run.p, (MkSlot*)nb.p, c->run_slots, min_count, (uint64_t*)d_kept)) != MK_OK) return rc;
    MK_HIP(hipMemcpyAsync(&kept, d_kept, 8, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    buf_free(c->run);
    c->run = nb;
    c->run_rows = (size_t)kept;
    if (c->run_side < min_count) c->run_side = 0;
  } else if (c->mode == MK_MODE_HASH128 && c->run128_slots) {
    MkDevBuf nb;
    if ((rc = mk_buf_reserve(c, nb, c->run128_slots * sizeof(MkSlot128))) != MK_OK) return rc;
    MK_HIP(hipMemsetAsync(nb.p, 0, c->run128_slots * sizeof(MkSlot128), c->stream));
    MK_HIP(hipMemsetAsync(d_kept, 0, 8, c->stream));
    if ((rc = mk_launch_refilter128(c, (const MkSlot128*)c->run128.p, (MkSlot128*)nb.p, c->run128_slots, min_count, (uint64_t*)d_kept)) != MK_OK) return rc;
    MK_HIP(hipMemcpyAsync(&kept, d_kept, 8, hipMemcpyDeviceToHost, c->stream));
    MK_HIP(hipStreamSynchronize(c->stream));
    buf_free(c->run128);
    c->run128 = nb;
    c->run128_rows = (size_t)kept;
  }
  if (c->run_ref_rows) {  // rows kept as text (few): through the host
    size_t n = 0;
    if ((rc = mk_export_exotic(c, nullptr, nullptr, 0, &n)) != MK_OK) return rc;
    std::vector<uint8_t> km(n * (size_t)c->k + 1), km2;
    std::vector<uint64_t> cn(n + 1), cn2;
    if ((rc = mk_export_exotic(c, km.data(), cn.data(), n, &n)) != MK_OK) return rc;
    for (size_t i = 0; i < n; ++i)
      if (cn[i] >= min_count) {
        km2.insert(km2.end(), km.begin() + i * (size_t)c->k, km.begin() + (i + 1) * (size_t)c->k);
        cn2.push_back(cn[i]);
      }
    if ((rc = mk_launch_clear_slots(c, (MkSlot*)c->run_ref.p, c->run_ref_slots)) != MK_OK) return rc;
    c->run_ref_rows = 0;
    if (!cn2.empty() && (rc = mk_import_exotic(c, km2.data(), cn2.data(), cn2.size())) != MK_OK) return rc;
  }
  MK_HIP(hipStreamSynchronize(c->stream));
  return MK_OK;
}

extern "C" int mk_merge_from(mk_ctx* dst, mk_ctx* src) {
  if (!dst || !src || dst == src) return MK_ERR_ARG;
  mk_ctx* c = dst;
  if (dst->device != src->device || dst->alphabet != src->alphabet || dst->k != src->k || dst->canonical != src->canonical) {
    c->err = "mk_merge_from: contexts differ in device, alphabet, k or canonical mode";
    return MK_ERR_ARG;
  }
  if (dst->in_chunk || src->in_chunk) { c->err = "mk_merge_from: a chunk is open"; return MK_ERR_STATE; }
  MK_HIP(hipSetDevice(dst->device));
  int rc;
  // both streams idle before one context's kernels touch the other's buffers (whatever the mode: the export of src
  // writes into dst's survivor buffers, which dst's own merge kernels may still be reading)
  if ((rc = settle(src)) != MK_OK) { dst->err = src->err; return rc; }
  if ((rc = settle(dst)) != MK_OK) return rc;
  MK_HIP(hipStreamSynchronize(src->stream));
  MK_HIP(hipStreamSynchronize(dst->stream));
  if (src->mode == MK_MODE_HASH64) {
    // table to table, on the device: no compaction, no sort (the rows' order does not matter for a sum)
    if (src->run_rows) {
      if ((rc = grow_run64(dst, dst->run_rows + src->run_rows)) != MK_OK) return rc;
      MK_HIP(hipMemsetAsync(dst->info.p, 0, sizeof(MkChunkInfo), dst->stream));
      if ((rc = mk_launch_merge_table64(dst, (const MkSlot*)src->run.p, src->run_slots)) != MK_OK) return rc;
      if ((rc = pull_info(dst)) != MK_OK) return rc;
      dst->run_rows += (size_t)dst->h_info->new_rows;
    }
    dst->run_side += src->run_side;
  } else if (src->mode == MK_MODE_DENSE || src->mode == MK_MODE_HASH128) {
    size_t cap = 0;
    if ((rc = mk_export_size(src, &cap)) != MK_OK) { dst->err = src->err; return rc; }
    cap += 1;
    if ((rc = mk_buf_reserve(dst, dst->surv_keys, cap * 8 * (size_t)mk_words_per_key(src) + 64)) != MK_OK) return rc;
    if ((rc = mk_buf_reserve(dst, dst->surv_cnts, cap * 8 + 64)) != MK_OK) return rc;
    size_t rows = 0;
    if ((rc = mk_export_pairs_device(src, (uint64_t*)dst->surv_keys.p, (uint64_t*)dst->surv_cnts.p, cap, &rows)) != MK_OK) {
      dst->err = src->err;
      return rc;
    }
    if (rows && (rc = mk_import_pairs_device(dst, (const uint64_t*)dst->surv_keys.p, (const uint64_t*)dst->surv_cnts.p, rows)) != MK_OK)
      return rc;
  }
  if (src->run_ref_rows) {
    size_t n = 0;
    if ((rc = mk_export_exotic(src, nullptr, nullptr, 0, &n)) != MK_OK) { dst->err = src->err; return rc; }
    std::vector<uint8_t> km(n * (size_t)src->k + 1);
    std::vector<uint64_t> cn(n + 1);
    if ((rc = mk_export_exotic(src, km.data(), cn.data(), n, &n)) != MK_OK) { dst->err = src->err; return rc; }
    if ((rc = mk_import_exotic(dst, km.data(), cn.data(), n)) != MK_OK) return rc;
  }
  return MK_OK;
}

// ----------------------------------------------------------------------------------- stats
extern "C" int mk_set_profiling(mk_ctx* c, int on) {
  if (!c) return MK_ERR_ARG;
  prof_collect(c);
  c->profile = on != 0;
  c->st.profiled = c->profile ? 1 : 0;
  return MK_OK;
}

extern "C" int mk_get_stats(mk_ctx* c, mk_stats_t* out) {
  if (!c || !out) return MK_ERR_ARG;
  { int rc_ = settle(c); if (rc_) return rc_; }
  (void)hipSetDevice(c->device);
  prof_collect(c);
  c->st.mode = c->mode;
  if (c->mode != MK_MODE_DENSE) c->st.rows = c->run_rows + (c->run_side ? 1 : 0) + c->run_ref_rows + c->run128_rows;
  *out = c->st;
  return MK_OK;
}

extern "C" int mk_reset_stats(mk_ctx* c) {
  if (!c) return MK_ERR_ARG;
  prof_collect(c);
  const int mode = c->st.mode, prof = c->st.profiled;
  c->st = mk_stats_t{};
  c->st.mode = mode;
  c->st.profiled = prof;
  return MK_OK;
}
