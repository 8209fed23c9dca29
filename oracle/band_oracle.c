/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see wfa_oracle.h).
 *
 * Deterministic CPU restatement of the reference's ADAPTIVE-BAND distance kernel
 * (lib/kernels/sequence_distance_kernel_aband.cu; the CIGAR kernel
 * lib/kernels/sequence_alignment_kernel_aband.cu:147-205 applies the same window rule), so that the heuristic this build
 * ships (band_cut in wfa-gpu_amd/csrc/align_kernel.hip) can be compared with the reference's on the same pairs: which pairs
 * finish inside the band, with which score.  Citations are lines of sequence_distance_kernel_aband.cu unless a file is named.
 *
 * What is restated, in the order the kernel does it:
 *   ring of A = max(o + e, x) + 1 wavefronts per component; a wavefront = {lo, hi, exist, beta offsets stored relative to lo}
 *   (:36-41 of the .cuh); reads outside [lo, hi] give OFFSET_NULL = -32000 (get_offset :28-33), stores outside are dropped
 *   (set_offset :35-43); all slots start as lo = hi = 0, exist = false, offsets NULL (:262-281)
 *   score 0: M[0][0] = extend(0) (:285-295); then per distance d (:325-402):
 *     GAP_exist = d >= o+e and (M[d-o-e].exist or I[d-e].exist); M_exist = GAP_exist or (d >= x and M[d-x].exist)   (:331-345)
 *     neither: the slot is marked non-existent -- its limits and offsets stay what its previous occupant left (:347-351)
 *     M only: next_M (:45-76)    gaps: next_MDI (:78-174), steps++
 *     then, if |k*| <= d: M[d][k*] == tlen -> finished; > tlen -> unfinished (:380-387; never true: extend nulls h > tlen)
 *   the loop runs while steps < max_steps - 1 (:325); steps counts next_MDI scores only (:373)
 *   next_MDI window (:91-130): hi = max(Mx.hi, max(Mo.hi, I.hi, D.hi) + 1), lo = min(Mx.lo, min(Mo.lo, I.lo, D.lo) - 1);
 *     while hi - lo > beta - 1: hi--, and if still too wide lo++ (:100-104);
 *     if the MISMATCH-source wavefront is full width (Mx.hi - Mx.lo >= beta - 1) and d % lambda == 0: the diagonal i in
 *     [Mx.lo, Mx.hi) -- Mx.hi itself is not looked at (:118) -- whose offset is closest to the end, max(plen - v, tlen - h),
 *     first minimum wins, becomes the centre: lo = i - beta / 2, hi = lo + beta - 1, unconditionally (:114-130)
 *   cells (:150-173): I = max(Mo[k-1], I[k-1]) + 1, D = max(Mo[k+1], D[k+1]), M = extend(max(Mx[k] + 1, D, I)) if >= 0;
 *     I and D are stored as computed (no range nulling), extend returns NULL for h > tlen or v > plen
 *     (common_alignment_kernels.cuh:38)
 * Determinism: every thread of the reference's block computes the window redundantly from the previous wavefronts and thread 0
 * publishes it before a barrier (:132-148), so next_MDI has no race.  next_M stores through the window its slot still holds
 * from the previous occupant and thread 0 replaces that window afterwards without a barrier in between (:59-76 vs :35-43): the
 * restatement takes "every store sees the old window" (thread 0 last).  M-only scores occur before the first gap wavefront
 * exists, when every window is still [0, 0], so the other order gives the same values.
 * Offsets are kept in 32 bits: the reference's int16 I offsets may wrap once tlen + steps exceeds 32767 -- not reachable with the
 * sequences (<= 10 kbp) and step limits (<= 14 000) this is used with.
 */
#include "wfa_oracle.h"

#include <limits.h>
#include <stdlib.h>
#include <string.h>

#define BNULL (-32000)
#define BMAX(a, b) ((a) > (b) ? (a) : (b))
#define BMIN(a, b) ((a) < (b) ? (a) : (b))

typedef struct { int lo, hi, exist; int* off; } bwf_t;

static int b_get(const bwf_t* w, int k) { return (k > w->hi || k < w->lo) ? BNULL : w->off[k - w->lo]; }
static void b_set(bwf_t* w, int k, int v, int beta) {
  if (k > w->hi || k < w->lo) return;
  if (k - w->lo < beta) w->off[k - w->lo] = v;      /* (a stale window wider than beta cannot occur: widths never exceed beta) */
}
/* common_alignment_kernels.cuh:29-111 on the ASCII sequences (same result as on the packed words for ACGT input) */
static int b_extend(const char* text, const char* pattern, int tlen, int plen, int k, int off) {
  int v = off - k, h = off;
  if (off < 0 || v > plen || h > tlen) return BNULL;
  while (v < plen && h < tlen && pattern[v] == text[h]) { ++v; ++h; }
  return h;
}

int oracle_band_ref(const char* pattern, int plen, const char* text, int tlen, int x, int o, int e,
                    int beta, int lambda, int max_steps, int* finished_out, int* steps_out) {
  if (!pattern || !text || plen < 0 || tlen < 0 || x <= 0 || o < 0 || e <= 0 || beta < 1 || lambda < 1) return -1;
  const int A = BMAX(o + e, x) + 1;
  bwf_t* W = (bwf_t*)calloc((size_t)3 * A, sizeof(bwf_t));
  int* store = (int*)malloc(sizeof(int) * (size_t)3 * A * beta);
  if (!W || !store) { free(W); free(store); return -1; }
  for (int i = 0; i < 3 * A * beta; ++i) store[i] = BNULL;
  for (int i = 0; i < 3 * A; ++i) { W[i].lo = W[i].hi = 0; W[i].exist = 0; W[i].off = store + (size_t)i * beta; }
  bwf_t *M = W, *I = W + A, *D = W + 2 * A;
  int curr = 0;
  b_set(&M[curr], 0, b_extend(text, pattern, tlen, plen, 0, 0), beta);
  M[curr].exist = 1;
  const int tk = tlen - plen, tk_abs = tk >= 0 ? tk : -tk;
  int finished = 0, distance = 0, steps = 0;
  if (!(tk_abs <= distance && M[curr].exist && b_get(&M[curr], tk) == tlen)) {
    curr = (curr - 1 + A) % A;
    distance++; steps++;
    while (steps < max_steps - 1) {
      const int od = (curr + o + e) % A, ed = (curr + e) % A, xd = (curr + x) % A;
      int gap_exist = 0, m_exist = 0;
      if (distance - o - e >= 0) gap_exist = M[od].exist || I[ed].exist;
      if (gap_exist) m_exist = 1;
      else if (distance - x >= 0) m_exist = M[xd].exist;
      if (!gap_exist && !m_exist) {
        M[curr].exist = D[curr].exist = I[curr].exist = 0;
        distance++;
      } else {
        if (m_exist && !gap_exist) {
          /* next_M (:45-76) */
          const bwf_t* p = &M[xd];
          const int hi = p->hi, lo = p->lo;
          for (int k = lo; k <= hi; ++k) {
            int c = b_get(p, k) + 1;
            if (c >= 0) c = b_extend(text, pattern, tlen, plen, k, c);
            b_set(&M[curr], k, c, beta);      /* through the window the slot still holds */
          }
          M[curr].hi = hi; M[curr].lo = lo; M[curr].exist = 1;
          D[curr].exist = 0; I[curr].exist = 0;
        } else {
          /* next_MDI (:78-174) */
          const bwf_t *px = &M[xd], *po = &M[od], *pi = &I[ed], *pd = &D[ed];
          int hi = BMAX(px->hi, BMAX(po->hi, BMAX(pi->hi, pd->hi)) + 1);
          int lo = BMIN(px->lo, BMIN(po->lo, BMIN(pi->lo, pd->lo)) - 1);
          while (hi - lo > beta - 1) {
            hi--;
            if (hi - lo <= beta - 1) break;
            lo++;
          }
          const int plo = px->lo, phi = px->hi;
          if (phi - plo >= beta - 1 && distance % lambda == 0) {
            int bmind = 2 * (tlen + plen), centre = plo;
            for (int i = plo; i < phi; ++i) {
              const int bo = b_get(px, i);
              const int dist = bo >= 0 ? BMAX(plen - (bo - i), tlen - bo) : INT_MAX;
              if (dist < bmind) { bmind = dist; centre = i; }
            }
            lo = centre - beta / 2;
            hi = lo + beta - 1;
          }
          /* the new rows are computed from the four input rows before anything of the slot is overwritten (the inputs are
           * other slots of the ring: x, o + e, e >= 1) */
          M[curr].lo = I[curr].lo = D[curr].lo = lo;
          M[curr].hi = I[curr].hi = D[curr].hi = hi;
          M[curr].exist = I[curr].exist = D[curr].exist = 1;
          for (int k = lo; k <= hi; ++k) {
            const int ins = BMAX(b_get(po, k - 1) + 1, b_get(pi, k - 1) + 1);
            const int del = BMAX(b_get(po, k + 1), b_get(pd, k + 1));
            const int mis = b_get(px, k) + 1;
            int m = BMAX(BMAX(mis, del), ins);
            if (m >= 0) m = b_extend(text, pattern, tlen, plen, k, m);
            b_set(&I[curr], k, ins, beta);
            b_set(&D[curr], k, del, beta);
            b_set(&M[curr], k, m, beta);
          }
          steps++;
        }
        if (tk_abs <= distance) {
          const int t = b_get(&M[curr], tk);
          if (t == tlen) { finished = 1; break; }
          if (t > tlen) { finished = 0; break; }
        }
        distance++;
      }
      curr = (curr - 1 + A) % A;
    }
  } else {
    finished = 1;
  }
  free(W); free(store);
  if (finished_out) *finished_out = finished;
  if (steps_out) *steps_out = steps;
  return distance;
}

/* Batch driver over the WFA-GPU buffer layout (see oracle_batch): scores[i] = the banded distance, or -1 when the pair did
 * not finish inside the band within max_steps (the reference then hands it to its CPU fallback). */
int64_t oracle_band_ref_batch(const char* seqbuf, const int64_t* offsets, int64_t n, int x, int o, int e, int beta, int lambda,
                              int max_steps, int32_t* scores, int nthreads) {
  int64_t i;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (i = 0; i < n; ++i) {
    int fin = 0;
    const int d = oracle_band_ref(seqbuf + offsets[4 * i], (int)offsets[4 * i + 1], seqbuf + offsets[4 * i + 2], (int)offsets[4 * i + 3],
                                  x, o, e, beta, lambda, max_steps, &fin, NULL);
    scores[i] = fin ? d : -1;
  }
  (void)nthreads;
  return n;
}
