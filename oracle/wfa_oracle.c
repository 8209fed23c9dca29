/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see wfa_oracle.h).
 *
 * Plain-C restatement of WFA2-lib v2.3's gap-affine, end-to-end, exact
 * (heuristic none) wavefront alignment, i.e. what quim0/WFA-GPU reaches
 * through utils/wfa_cpu.c:166-189 (compute_alignment_cpu) and what its -c
 * path treats as the truth (lib/align.cu:300-317).
 *
 * Each function cites the reference lines it restates; paths are relative
 * to /root/reference/external/WFA unless they start with lib/ or utils/.
 * Nothing here is copied: the data structures are a flat per-score table of
 * (lo, hi, offsets) triples instead of WFA2's slab/components machinery.
 */
#include "wfa_oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* wavefront/wavefront_offset.h:44 */
#define ONULL (INT32_MIN / 2)

#define OMAX(a, b) ((a) > (b) ? (a) : (b))
#define OMIN(a, b) ((a) < (b) ? (a) : (b))

/* One component (M, I or D) of one score.  `null` rows read as ONULL
 * everywhere and carry the canonical limits lo=1, hi=-1
 * (wavefront/wavefront.c:110-117). */
typedef struct {
  int lo, hi;      /* trimmed limits (inclusive)                        */
  int base;        /* diagonal stored at off[0]                          */
  int null;
  int32_t* off;
} owf_t;

struct oslab { struct oslab* next; size_t cap, used; int32_t data[]; };

struct oracle_aligner {
  int x, o, e;
  int ring;            /* max(x, o+e) + 1                                 */
  /* per-score rows: index = score (keep_all) or score % ring            */
  owf_t *M, *I, *D;
  size_t rows_cap;
  /* offset storage (keep_all mode): linked slabs, newest first */
  struct oslab* slabs;
  /* ring-mode storage: 3*ring rows of row_cap ints                      */
  int32_t* ring_buf;
  size_t ring_row_cap, ring_buf_cap;
  /* backtrace scratch */
  char* ops;
  size_t ops_cap;
  /* last run */
  int keep_all;
  int plen, tlen;
  const char *pattern, *text;
};

static const owf_t OWF_NULL = {1, -1, 0, 1, NULL};

oracle_aligner_t* oracle_aligner_new(int x, int o, int e) {
  /* wavefront/wavefront_penalties.c:96-105: X>0, O>=0, E>0 */
  if (x <= 0 || o < 0 || e <= 0) return NULL;
  oracle_aligner_t* al = (oracle_aligner_t*)calloc(1, sizeof(*al));
  if (!al) return NULL;
  al->x = x; al->o = o; al->e = e;
  al->ring = OMAX(x, o + e) + 1;
  return al;
}

void oracle_aligner_delete(oracle_aligner_t* al) {
  if (!al) return;
  free(al->M); free(al->I); free(al->D);
  while (al->slabs) { struct oslab* n = al->slabs->next; free(al->slabs); al->slabs = n; }
  free(al->ring_buf); free(al->ops);
  free(al);
}

static void rows_reserve(oracle_aligner_t* al, size_t n) {
  if (n <= al->rows_cap) return;
  size_t cap = al->rows_cap ? al->rows_cap : 256;
  while (cap < n) cap *= 2;
  al->M = (owf_t*)realloc(al->M, cap * sizeof(owf_t));
  al->I = (owf_t*)realloc(al->I, cap * sizeof(owf_t));
  al->D = (owf_t*)realloc(al->D, cap * sizeof(owf_t));
  if (!al->M || !al->I || !al->D) { fprintf(stderr, "[oracle] OOM\n"); exit(1); }
  al->rows_cap = cap;
}

/* keep_all storage: rows are carved out of malloc'd slabs so that row
 * pointers stay valid while the table of rows grows. */
static int32_t* arena_alloc(oracle_aligner_t* al, size_t n) {
  struct oslab* head = al->slabs;
  if (!head || head->used + n > head->cap) {
    size_t cap = (size_t)1 << 20;
    if (cap < n) cap = n;
    struct oslab* s = (struct oslab*)malloc(sizeof(struct oslab) + cap * sizeof(int32_t));
    if (!s) { fprintf(stderr, "[oracle] OOM\n"); exit(1); }
    s->next = head; s->cap = cap; s->used = 0;
    al->slabs = s;
    head = s;
  }
  int32_t* p = head->data + head->used;
  head->used += n;
  return p;
}

static void arena_reset(oracle_aligner_t* al) {
  /* keep the newest slab, free the rest */
  struct oslab* head = al->slabs;
  if (!head) return;
  struct oslab* s = head->next;
  while (s) { struct oslab* n = s->next; free(s); s = n; }
  head->next = NULL;
  head->used = 0;
}

static inline int row_index(const oracle_aligner_t* al, int s) {
  return al->keep_all ? s : s % al->ring;
}

/* wavefront/wavefront_compute.c:255-296 (get_*wavefront): negative score
 * or null row -> the null wavefront */
static inline const owf_t* fetch(const oracle_aligner_t* al, const owf_t* rows, int s) {
  if (s < 0) return &OWF_NULL;
  const owf_t* w = &rows[row_index(al, s)];
  return w->null ? &OWF_NULL : w;
}

/* reads outside [lo,hi] are NULL (wavefront_compute.c:480-520 init_ends) */
static inline int32_t rd(const owf_t* w, int k) {
  if (k < w->lo || k > w->hi) return ONULL;
  return w->off[k - w->base];
}

static void row_alloc(oracle_aligner_t* al, owf_t* w, int s, int comp, int lo, int hi) {
  w->lo = lo; w->hi = hi; w->base = lo; w->null = 0;
  size_t n = (size_t)(hi - lo + 1);
  if (al->keep_all) {
    w->off = arena_alloc(al, n);
  } else {
    if (n > al->ring_row_cap) { fprintf(stderr, "[oracle] ring row overflow\n"); exit(1); }
    w->off = al->ring_buf + ((size_t)(comp * al->ring + s % al->ring)) * al->ring_row_cap;
  }
}

/* wavefront/wavefront_compute.c:570-603 (trim_ends) */
static void trim(const oracle_aligner_t* al, owf_t* w) {
  const uint32_t plen = (uint32_t)al->plen, tlen = (uint32_t)al->tlen;
  int k;
  for (k = w->hi; k >= w->lo; --k) {
    const int32_t off = w->off[k - w->base];
    const uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
    if (h <= tlen && v <= plen) break;
  }
  w->hi = k;
  for (k = w->lo; k <= w->hi; ++k) {
    const int32_t off = w->off[k - w->base];
    const uint32_t h = (uint32_t)off, v = (uint32_t)(off - k);
    if (h <= tlen && v <= plen) break;
  }
  w->lo = k;
  w->null = (w->lo > w->hi);
}

/* EXPERIMENT, COMPILE-TIME ONLY (-DORACLE_EXPERIMENT_NULL_INVALID_GAPS; the checker the tests load is never built with it:
 * a process-wide run-time switch could leave the ground truth in non-WFA2 mode for whoever calls next): null every I and D
 * value that ran past a sequence end the moment it is computed -- what a kernel that keeps one diagonal per lane and no
 * per-row limits does (wfa-gpu_amd/csrc/short_kernel.hip).  WFA2 keeps such values and only trims them at the ends of a row
 * (wavefront_compute.c:570-603).  scratch/short_cigar_semantics.py builds a second library with the flag and looks for pairs
 * whose CIGAR changes. */
#ifdef ORACLE_EXPERIMENT_NULL_INVALID_GAPS
#define GAP_FIX(val, kk) do { if ((uint32_t)(val) > tlen || (uint32_t)((val) - (kk)) > plen) (val) = ONULL; } while (0)
#else
#define GAP_FIX(val, kk) do { } while (0)
#endif

/* wavefront/wavefront_compute_affine.c:228-259 + :45-87 (kernel) +
 * wavefront_compute.c:41-71 (limits) + :401-437 (which outputs exist) */
static void compute_step(oracle_aligner_t* al, int s, oracle_stats_t* st) {
  const int x = al->x, oe = al->o + al->e, e = al->e;
  const owf_t* mx  = fetch(al, al->M, s - x);
  const owf_t* moe = fetch(al, al->M, s - oe);
  const owf_t* ie  = fetch(al, al->I, s - e);
  const owf_t* de  = fetch(al, al->D, s - e);
  owf_t* om = &al->M[row_index(al, s)];
  owf_t* oi = &al->I[row_index(al, s)];
  owf_t* od = &al->D[row_index(al, s)];
  if (mx->null && moe->null && ie->null && de->null) {
    *om = OWF_NULL; *oi = OWF_NULL; *od = OWF_NULL;
    return;
  }
  int lo = mx->lo, hi = mx->hi;
  if (lo > moe->lo - 1) lo = moe->lo - 1;
  if (hi < moe->hi + 1) hi = moe->hi + 1;
  if (lo > ie->lo + 1) lo = ie->lo + 1;
  if (hi < ie->hi + 1) hi = ie->hi + 1;
  if (lo > de->lo - 1) lo = de->lo - 1;
  if (hi < de->hi - 1) hi = de->hi - 1;
  /* ring-mode aliasing: inputs live in other ring slots than s%ring because
   * x, o+e, e are all in [1, ring-1]; safe to overwrite slot s%ring. */
  const int have_i = !moe->null || !ie->null;
  const int have_d = !moe->null || !de->null;
  /* copy the input descriptors: in ring mode om/oi/od never alias them, but
   * keep_all reallocs cannot happen here either (rows reserved by caller) */
  const owf_t MX = *mx, MOE = *moe, IE = *ie, DE = *de;
  row_alloc(al, om, s, 0, lo, hi);
  if (have_i) row_alloc(al, oi, s, 1, lo, hi); else *oi = OWF_NULL;
  if (have_d) row_alloc(al, od, s, 2, lo, hi); else *od = OWF_NULL;
  const uint32_t plen = (uint32_t)al->plen, tlen = (uint32_t)al->tlen;
  /* Interior diagonals, where all five reads fall inside their rows, run without the range tests of rd()
   * (same arithmetic; lets the compiler vectorise the loop that dominates long alignments). */
  int in_lo = hi + 1, in_hi = hi;
  if (!MX.null && !MOE.null && !IE.null && !DE.null && have_i && have_d) {
    in_lo = MX.lo; in_hi = MX.hi;
    if (in_lo < MOE.lo + 1) in_lo = MOE.lo + 1;
    if (in_hi > MOE.hi - 1) in_hi = MOE.hi - 1;
    if (in_lo < IE.lo + 1) in_lo = IE.lo + 1;
    if (in_hi > IE.hi + 1) in_hi = IE.hi + 1;
    if (in_lo < DE.lo - 1) in_lo = DE.lo - 1;
    if (in_hi > DE.hi - 1) in_hi = DE.hi - 1;
    if (in_lo < lo) in_lo = lo;
    if (in_hi > hi) in_hi = hi;
    if (in_lo > in_hi) { in_lo = hi + 1; in_hi = hi; }
  }
  for (int k = lo; k <= hi; ++k) {
    if (k == in_lo) {
      const int32_t* restrict pmo = MOE.off - MOE.base;
      const int32_t* restrict pie = IE.off - IE.base;
      const int32_t* restrict pde = DE.off - DE.base;
      const int32_t* restrict pmx = MX.off - MX.base;
      int32_t* restrict qi = oi->off - lo; int32_t* restrict qd = od->off - lo; int32_t* restrict qm = om->off - lo;
      for (int kk = in_lo; kk <= in_hi; ++kk) {
        int32_t ins = OMAX(pmo[kk - 1], pie[kk - 1]) + 1;
        int32_t del = OMAX(pmo[kk + 1], pde[kk + 1]);
        GAP_FIX(ins, kk); GAP_FIX(del, kk);
        const int32_t mis = pmx[kk] + 1;
        int32_t mx3 = OMAX(del, OMAX(mis, ins));
        const uint32_t h = (uint32_t)mx3, v = (uint32_t)(mx3 - kk);
        if (h > tlen) mx3 = ONULL;
        if (v > plen) mx3 = ONULL;
        qi[kk] = ins; qd[kk] = del; qm[kk] = mx3;
      }
      k = in_hi;
      continue;
    }
    int32_t ins = OMAX(rd(&MOE, k - 1), rd(&IE, k - 1)) + 1;
    int32_t del = OMAX(rd(&MOE, k + 1), rd(&DE, k + 1));
    GAP_FIX(ins, k); GAP_FIX(del, k);
    const int32_t mis = rd(&MX, k) + 1;
    int32_t mx3 = OMAX(del, OMAX(mis, ins));
    const uint32_t h = (uint32_t)mx3, v = (uint32_t)(mx3 - k);
    if (h > tlen) mx3 = ONULL;
    if (v > plen) mx3 = ONULL;
    if (have_i) oi->off[k - lo] = ins;
    if (have_d) od->off[k - lo] = del;
    om->off[k - lo] = mx3;
  }
  if (st) {
    st->cells += hi - lo + 1;
    st->steps += 1;
    if (-lo > st->max_abs_k) st->max_abs_k = -lo;
    if (hi > st->max_abs_k) st->max_abs_k = hi;
  }
  trim(al, om);
  if (have_i) trim(al, oi);
  if (have_d) trim(al, od);
}

/* wavefront/wavefront_extend.c:174-215: longest common prefix from (v,h).
 * WFA2 compares raw bytes in 8-byte blocks against sentinel-padded copies
 * (utils/string_padded.c:100-137); explicit bounds are equivalent. */
static void extend_row(const oracle_aligner_t* al, owf_t* m) {
  if (m->null) return;
  const char* p = al->pattern; const char* t = al->text;
  const int plen = al->plen, tlen = al->tlen;
  for (int k = m->lo; k <= m->hi; ++k) {
    int32_t off = m->off[k - m->base];
    if (off == ONULL) continue;
    int h = off, v = off - k;
    while (v < plen && h < tlen && p[v] == t[h]) { ++v; ++h; }
    m->off[k - m->base] = h;
  }
}

/* wavefront/wavefront_extend.c:47-67 */
static int reached_end(const oracle_aligner_t* al, const owf_t* m) {
  const int kend = al->tlen - al->plen;
  if (m->null || m->lo > kend || kend > m->hi) return 0;
  return m->off[kend - m->base] >= al->tlen;
}

/* wavefront/wavefront_unialign.c:413-449 (main loop) and :126-151 (init) */
static int run(oracle_aligner_t* al, const char* pattern, int plen,
               const char* text, int tlen, int keep_all, int max_score,
               oracle_stats_t* st) {
  al->keep_all = keep_all;
  al->pattern = pattern; al->text = text; al->plen = plen; al->tlen = tlen;
  if (st) memset(st, 0, sizeof(*st));
  if (keep_all) {
    arena_reset(al);
    rows_reserve(al, 1024);
  } else {
    rows_reserve(al, (size_t)al->ring);
    const size_t row_cap = (size_t)plen + (size_t)tlen + 8;
    const size_t need = row_cap * 3 * (size_t)al->ring;
    if (need > al->ring_buf_cap) {
      free(al->ring_buf);
      al->ring_buf = (int32_t*)malloc(need * sizeof(int32_t));
      if (!al->ring_buf) { fprintf(stderr, "[oracle] OOM\n"); exit(1); }
      al->ring_buf_cap = need;
    }
    al->ring_row_cap = row_cap;
  }
  /* score 0: M = {k=0: offset 0}; I and D do not exist */
  row_alloc(al, &al->M[0], 0, 0, 0, 0);
  al->M[0].off[0] = 0;
  al->I[0] = OWF_NULL; al->D[0] = OWF_NULL;
  if (st) { st->cells = 1; st->steps = 1; }
  int s = 0;
  for (;;) {
    owf_t* m = &al->M[row_index(al, s)];
    extend_row(al, m);
    if (reached_end(al, m)) return s;
    ++s;
    if (max_score > 0 && s > max_score) return -2;
    if (keep_all) rows_reserve(al, (size_t)s + 1);
    compute_step(al, s, st);
  }
}

int oracle_score(oracle_aligner_t* al, const char* pattern, int plen,
                 const char* text, int tlen, int max_score,
                 oracle_stats_t* stats) {
  if (!al || !pattern || !text || plen < 0 || tlen < 0) return -1;
  return run(al, pattern, plen, text, tlen, 0, max_score, stats);
}

/* ---- backtrace: wavefront/wavefront_backtrace.c:36-59, 318-526 ---- */

enum { BT_I_OPEN = 1, BT_I_EXT = 2, BT_D_OPEN = 5, BT_D_EXT = 6, BT_M = 9 };

static inline int64_t piggy(int64_t off, int type) { return (off * 16) | type; }
/* NB: WFA2 writes ((int64)offset << 4) | type; for negative offsets the
 * arithmetic shift equals multiplication by 16, which is what we spell. */

static int64_t bt_get(const oracle_aligner_t* al, const owf_t* rows, int s,
                      int k, int add, int type) {
  if (s < 0) return ONULL;
  const owf_t* w = &rows[s];
  if (w->null || k < w->lo || k > w->hi) return ONULL;
  return piggy((int64_t)w->off[k - w->base] + add, type);
}

static int backtrace(oracle_aligner_t* al, int score, oracle_stats_t* st,
                     char** ops_begin, char** ops_end) {
  const int plen = al->plen, tlen = al->tlen;
  const int x = al->x, oe = al->o + al->e, e = al->e;
  const size_t cap = (size_t)plen + (size_t)tlen + 2;
  if (cap > al->ops_cap) {
    free(al->ops);
    al->ops = (char*)malloc(cap);
    if (!al->ops) { fprintf(stderr, "[oracle] OOM\n"); exit(1); }
    al->ops_cap = cap;
  }
  char* ops = al->ops;
  size_t pos = cap - 1;           /* next free slot, filled backwards */
  int nops = 0;
  enum { ST_M, ST_I, ST_D } state = ST_M;
  int s = score, k = tlen - plen, offset = tlen;
  int h = offset, v = offset - k;
  while (v > 0 && h > 0 && s > 0) {
    const int s_x = s - x, s_oe = s - oe, s_e = s - e;
    int64_t best;
    if (state == ST_M) {
      const int64_t mis = bt_get(al, al->M, s_x, k, 1, BT_M);
      const int64_t io = bt_get(al, al->M, s_oe, k - 1, 1, BT_I_OPEN);
      const int64_t ie = bt_get(al, al->I, s_e, k - 1, 1, BT_I_EXT);
      const int64_t d_o = bt_get(al, al->M, s_oe, k + 1, 0, BT_D_OPEN);
      const int64_t d_e = bt_get(al, al->D, s_e, k + 1, 0, BT_D_EXT);
      best = OMAX(mis, OMAX(OMAX(io, ie), OMAX(d_o, d_e)));
    } else if (state == ST_I) {
      const int64_t io = bt_get(al, al->M, s_oe, k - 1, 1, BT_I_OPEN);
      const int64_t ie = bt_get(al, al->I, s_e, k - 1, 1, BT_I_EXT);
      best = OMAX(io, ie);
    } else {
      const int64_t d_o = bt_get(al, al->M, s_oe, k + 1, 0, BT_D_OPEN);
      const int64_t d_e = bt_get(al, al->D, s_e, k + 1, 0, BT_D_EXT);
      best = OMAX(d_o, d_e);
    }
    if (best < 0) break;
    if (state == ST_M) {
      const int src = (int)(best >> 4);
      for (int n = offset - src; n > 0; --n) ops[pos--] = 'M';
      offset = src;
      v = offset - k; h = offset;
      if (v <= 0 || h <= 0) break;
    }
    const int type = (int)(best & 15);
    switch (type) {
      case BT_M:      s = s_x;  state = ST_M; ops[pos--] = 'X'; --offset; break;
      case BT_I_OPEN: s = s_oe; state = ST_M; ops[pos--] = 'I'; --k; --offset; break;
      case BT_I_EXT:  s = s_e;  state = ST_I; ops[pos--] = 'I'; --k; --offset; break;
      case BT_D_OPEN: s = s_oe; state = ST_M; ops[pos--] = 'D'; ++k; break;
      case BT_D_EXT:  s = s_e;  state = ST_D; ops[pos--] = 'D'; ++k; break;
      default: return -1;
    }
    ++nops;
    v = offset - k; h = offset;
  }
  if (state == ST_M) {
    if (v > 0 && h > 0) {
      const int n = OMIN(v, h);
      for (int i = 0; i < n; ++i) ops[pos--] = 'M';
      v -= n; h -= n;
    }
    while (v > 0) { ops[pos--] = 'D'; --v; ++nops; }
    while (h > 0) { ops[pos--] = 'I'; --h; ++nops; }
  } else if (v != 0 || h != 0 || s != 0) {
    return -1;  /* wavefront_backtrace.c:512-520: "Beginning backtrace error" */
  }
  if (st) st->num_ops = nops;
  *ops_begin = ops + pos + 1;
  *ops_end = ops + cap;
  return 0;
}

/* alignment/cigar.c:394-426 (cigar_sprint, print_matches=true) */
static long rle_print(const char* b, const char* e, char* out, size_t cap) {
  size_t pos = 0;
  while (b < e) {
    const char op = *b; int n = 0;
    while (b < e && *b == op) { ++b; ++n; }
    char tmp[16];
    const int w = snprintf(tmp, sizeof tmp, "%d%c", n, op);
    if (pos + (size_t)w + 1 > cap) return -1;
    memcpy(out + pos, tmp, (size_t)w);
    pos += (size_t)w;
  }
  if (pos + 1 > cap) return -1;
  out[pos] = '\0';
  return (long)pos;
}

int oracle_align(oracle_aligner_t* al, const char* pattern, int plen,
                 const char* text, int tlen, char* cigar_out, size_t cigar_cap,
                 oracle_stats_t* stats) {
  if (!al || !pattern || !text || plen < 0 || tlen < 0 || !cigar_out) return -1;
  const int score = run(al, pattern, plen, text, tlen, 1, 0, stats);
  char *b, *e;
  if (backtrace(al, score, stats, &b, &e) != 0) {
    fprintf(stderr, "[oracle] backtrace failed\n");
    return -1;
  }
  if (rle_print(b, e, cigar_out, cigar_cap) < 0) return -3;
  return score;
}

int64_t oracle_batch(const char* seqbuf, const int64_t* offsets, int64_t n,
                     int x, int o, int e, int32_t* scores, char* cigar_buf,
                     size_t cigar_stride, int64_t* cells, int nthreads) {
  int64_t done = 0, total_cells = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads) reduction(+ : done, total_cells)
#endif
  {
    oracle_aligner_t* al = oracle_aligner_new(x, o, e);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < n; ++i) {
      if (!al) continue;
      const char* p = seqbuf + offsets[4 * i + 0];
      const int plen = (int)offsets[4 * i + 1];
      const char* t = seqbuf + offsets[4 * i + 2];
      const int tlen = (int)offsets[4 * i + 3];
      oracle_stats_t st;
      int sc;
      if (cigar_buf) sc = oracle_align(al, p, plen, t, tlen, cigar_buf + (size_t)i * cigar_stride, cigar_stride, &st);
      else sc = oracle_score(al, p, plen, t, tlen, 0, &st);
      if (scores) scores[i] = sc;
      total_cells += st.cells;
      ++done;
    }
    oracle_aligner_delete(al);
  }
  (void)nthreads;
  if (cells) *cells = total_cells;
  return done;
}

/* lib/kernels/sequence_packing_kernel.cu:69-90: code = (c & 6) >> 1
 * (A=0 C=1 T=2 G=3).  Word layout is this build's (little-endian, see .h). */
int oracle_pack2(const char* seq, int len, uint32_t* words_out) {
  int bad = 0;
  const int nwords = (len + 15) / 16;
  for (int w = 0; w < nwords; ++w) words_out[w] = 0;
  for (int i = 0; i < len; ++i) {
    const unsigned char c = (unsigned char)seq[i];
    if (c != 'A' && c != 'C' && c != 'G' && c != 'T') bad = 1;
    words_out[i >> 4] |= (uint32_t)((c & 6u) >> 1) << (2 * (i & 15));
  }
  return bad;
}

/* utils/verification.c:27-89 (check_cigar_edit) and :91-146
 * (check_affine_distance: a new gap is charged when I follows D or D
 * follows I as well) */
int oracle_check_cigar(const char* pattern, int plen, const char* text,
                       int tlen, const char* cigar, int x, int o, int e,
                       int* cost_out) {
  int v = 0, h = 0, cost = 0;
  char prev = 0;
  const char* c = cigar;
  while (*c) {
    int n = 0;
    if (*c < '0' || *c > '9') return 0;
    while (*c >= '0' && *c <= '9') { n = n * 10 + (*c - '0'); ++c; }
    const char op = *c++;
    if (n <= 0) return 0;
    switch (op) {
      case 'M':
        for (int i = 0; i < n; ++i, ++v, ++h)
          if (v >= plen || h >= tlen || pattern[v] != text[h]) return 0;
        break;
      case 'X':
        for (int i = 0; i < n; ++i, ++v, ++h)
          if (v >= plen || h >= tlen || pattern[v] == text[h]) return 0;
        cost += n * x;
        break;
      case 'I':
        h += n; if (h > tlen) return 0;
        cost += (prev == 'I' ? 0 : o) + n * e;
        break;
      case 'D':
        v += n; if (v > plen) return 0;
        cost += (prev == 'D' ? 0 : o) + n * e;
        break;
      default:
        return 0;
    }
    prev = op;
  }
  if (cost_out) *cost_out = cost;
  return v == plen && h == tlen;
}
