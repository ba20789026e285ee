/* kmer_oracle.c -- CPU oracle of the MerCat2 counting path in plain C.  TEST INFRASTRUCTURE ONLY.
 *
 * A second, independent restatement of the reference algorithm (the first is oracle/cpu_ref.py),
 * fast enough to check the GPU path at the benchmark's full sizes.  It is pinned against the
 * same golden vectors (tests/test_oracle_c.py) and is never linked, loaded or called by the
 * product package mercat2_amd.
 *
 * Semantics restated (reference checkout paths):
 *   lib/mercat2_kmers.py:49-63  text-mode lines (universal newlines: "\n", "\r\n", lone "\r"),
 *                               line.strip(), header iff the stripped line starts with '>',
 *                               otherwise seq += line.replace("*", "")
 *   lib/mercat2_kmers.py:52-61, 65-69  every window seq[i:i+k] of a record counts once, keys are
 *                               the substrings themselves (no canonical form, any character)
 *   lib/mercat2_kmers.py:73-76  keep keys with count >= min_count
 *   bin/mercat2.py:132          rows in sorted(str) order == byte-wise order for ASCII
 * Input must be ASCII (bytes >= 0x80 -> ORACLE_ERR_NON_ASCII), like the product.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_ERR_NOMEM (-3)
#define ORACLE_ERR_NON_ASCII (-5)

typedef struct {
  const uint8_t* key; /* k bytes in the arena, NULL = free slot */
  uint64_t count;
  uint64_t hash;
} slot_t;

typedef struct {
  slot_t* slots;
  size_t cap, used;
  uint8_t* arena;
  size_t arena_len, arena_cap;
  int k;
} table_t;

static int py_isspace(unsigned c) { /* str.strip() set, ASCII part */
  return c == ' ' || (c >= 9 && c <= 13) || (c >= 28 && c <= 31);
}

static int table_grow(table_t* t) {
  size_t ncap = t->cap ? t->cap * 2 : (1u << 16);
  slot_t* ns = (slot_t*)calloc(ncap, sizeof(slot_t));
  if (!ns) return ORACLE_ERR_NOMEM;
  for (size_t i = 0; i < t->cap; ++i) {
    if (!t->slots[i].key) continue;
    size_t j = t->slots[i].hash & (ncap - 1);
    while (ns[j].key) j = (j + 1) & (ncap - 1);
    ns[j] = t->slots[i];
  }
  free(t->slots);
  t->slots = ns;
  t->cap = ncap;
  return ORACLE_OK;
}

static uint64_t mix(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

/* The arena stores keys as OFFSETS would move on realloc, so slots keep offsets+1 in `key` cast
 * through uintptr_t until the end; simpler: arena grows by chunks that never move. */
typedef struct chunk { struct chunk* next; size_t len, cap; uint8_t data[]; } chunk_t;
static chunk_t* g_dummy;

typedef struct {
  table_t t;
  chunk_t* chunks;
} state_t;

static const uint8_t* arena_put(state_t* s, const uint8_t* p, int k) {
  chunk_t* c = s->chunks;
  if (!c || c->len + (size_t)k > c->cap) {
    size_t cap = (size_t)1 << 24;
    if (cap < (size_t)k) cap = (size_t)k;
    chunk_t* n = (chunk_t*)malloc(sizeof(chunk_t) + cap);
    if (!n) return NULL;
    n->next = c; n->len = 0; n->cap = cap;
    s->chunks = c = n;
  }
  uint8_t* dst = c->data + c->len;
  memcpy(dst, p, (size_t)k);
  c->len += (size_t)k;
  return dst;
}

static int add_window(state_t* s, const uint8_t* p, uint64_t h) {
  table_t* t = &s->t;
  if ((t->used + 1) * 2 > t->cap) { int rc = table_grow(t); if (rc) return rc; }
  size_t j = h & (t->cap - 1);
  for (;;) {
    slot_t* e = &t->slots[j];
    if (!e->key) {
      const uint8_t* kp = arena_put(s, p, t->k);
      if (!kp) return ORACLE_ERR_NOMEM;
      e->key = kp; e->count = 1; e->hash = h; t->used++;
      return ORACLE_OK;
    }
    if (e->hash == h && memcmp(e->key, p, (size_t)t->k) == 0) { e->count++; return ORACLE_OK; }
    j = (j + 1) & (t->cap - 1);
  }
}

/* every window of one de-wrapped record (lib/mercat2_kmers.py:56-60) */
static int count_record(state_t* s, const uint8_t* rec, size_t len) {
  const int k = s->t.k;
  if (len < (size_t)k) return ORACLE_OK;
  const uint64_t B = 0x100000001b3ULL;
  uint64_t bk = 1, h = 0;
  for (int i = 0; i < k; ++i) { bk *= B; h = h * B + rec[i]; }
  for (size_t i = 0;; ++i) {
    int rc = add_window(s, rec + i, mix(h));
    if (rc) return rc;
    if (i + (size_t)k >= len) break;
    h = h * B + rec[i + k] - bk * rec[i];
  }
  return ORACLE_OK;
}

static int cmp_k;
static int cmp_slots(const void* a, const void* b) {
  return memcmp(((const slot_t*)a)->key, ((const slot_t*)b)->key, (size_t)cmp_k);
}

int oracle_count(const uint8_t* text, size_t n, int k, uint64_t min_count, uint8_t** out_kmers, uint64_t** out_counts,
                 size_t* rows) {
  (void)g_dummy;
  state_t s;
  memset(&s, 0, sizeof s);
  s.t.k = k;
  *out_kmers = NULL; *out_counts = NULL; *rows = 0;
  for (size_t i = 0; i < n; ++i) if (text[i] >= 0x80) return ORACLE_ERR_NON_ASCII;
  uint8_t* rec = NULL;
  size_t rlen = 0, rcap = 0;
  int rc = ORACLE_OK;
  size_t pos = 0;
  while (pos <= n && rc == ORACLE_OK) {
    if (pos == n) break;
    /* one text-mode line: up to '\n', '\r\n' or lone '\r' */
    size_t e = pos;
    while (e < n && text[e] != '\n' && text[e] != '\r') ++e;
    size_t next = e;
    if (e < n) next = (text[e] == '\r' && e + 1 < n && text[e + 1] == '\n') ? e + 2 : e + 1;
    size_t a = pos, b = e; /* strip() */
    while (a < b && py_isspace(text[a])) ++a;
    while (b > a && py_isspace(text[b - 1])) --b;
    if (a < b && text[a] == '>') { /* header: flush the record (lib/mercat2_kmers.py:52-61) */
      rc = count_record(&s, rec, rlen);
      rlen = 0;
    } else {
      if (rlen + (b - a) > rcap) {
        rcap = (rlen + (b - a)) * 2 + 1024;
        uint8_t* nr = (uint8_t*)realloc(rec, rcap);
        if (!nr) { rc = ORACLE_ERR_NOMEM; break; }
        rec = nr;
      }
      for (size_t i = a; i < b; ++i) if (text[i] != '*') rec[rlen++] = text[i]; /* replace("*","") */
    }
    pos = next;
  }
  if (rc == ORACLE_OK) rc = count_record(&s, rec, rlen); /* last record (lib/mercat2_kmers.py:64-69) */
  free(rec);
  if (rc == ORACLE_OK) {
    size_t m = 0;
    for (size_t i = 0; i < s.t.cap; ++i)
      if (s.t.slots[i].key && s.t.slots[i].count >= min_count) s.t.slots[m++] = s.t.slots[i];
    cmp_k = k;
    qsort(s.t.slots, m, sizeof(slot_t), cmp_slots);
    uint8_t* ok = (uint8_t*)malloc(m * (size_t)k + 1);
    uint64_t* oc = (uint64_t*)malloc(m * sizeof(uint64_t) + 8);
    if (!ok || !oc) { free(ok); free(oc); rc = ORACLE_ERR_NOMEM; }
    else {
      for (size_t i = 0; i < m; ++i) { memcpy(ok + i * (size_t)k, s.t.slots[i].key, (size_t)k); oc[i] = s.t.slots[i].count; }
      *out_kmers = ok; *out_counts = oc; *rows = m;
    }
  }
  free(s.t.slots);
  for (chunk_t* c = s.chunks; c;) { chunk_t* nx = c->next; free(c); c = nx; }
  return rc;
}

void oracle_free(void* p) { free(p); }
