/* See verification.h.  Used only when check_correctness is set. */
#include "verification.h"

#include <stdlib.h>
#include <string.h>

static const char* next_item(const char* c, long* n, char* op) {
    long v = 0;
    if (*c < '0' || *c > '9') return NULL;
    /* (a run longer than any sequence is not a CIGAR of this pair: refused before the accumulator can overflow --
     * found by the UBSan harness, tests/host_api_asan.c) */
    while (*c >= '0' && *c <= '9') { v = v * 10 + (*c - '0'); ++c; if (v > 0x7FFFFFFFL) return NULL; }
    if (!*c || v <= 0) return NULL;
    *n = v; *op = *c;
    return c + 1;
}

bool check_cigar_edit(const char* text, const char* pattern, size_t tlen, size_t plen,
                      const char* cigar) {
    size_t v = 0, h = 0;
    const char* c = cigar;
    while (*c) {
        long n; char op;
        c = next_item(c, &n, &op);
        if (!c) return false;
        for (long i = 0; i < n; ++i) {
            switch (op) {
                case 'M': if (v >= plen || h >= tlen || pattern[v] != text[h]) return false; ++v; ++h; break;
                case 'X': if (v >= plen || h >= tlen || pattern[v] == text[h]) return false; ++v; ++h; break;
                case 'I': if (h >= tlen) return false; ++h; break;
                case 'D': if (v >= plen) return false; ++v; break;
                default: return false;
            }
        }
    }
    return v == plen && h == tlen;
}

bool check_affine_distance(const char* text, const char* pattern, size_t tlen, size_t plen,
                           int distance, int x, int o, int e, const char* cigar) {
    (void)text; (void)pattern; (void)tlen; (void)plen;
    long cost = 0;
    char prev = 0;
    const char* c = cigar;
    while (*c) {
        long n; char op;
        c = next_item(c, &n, &op);
        if (!c) return false;
        if (op == 'X') cost += n * x;
        else if (op == 'I' || op == 'D') cost += (prev == op ? 0 : o) + n * e;
        else if (op != 'M') return false;
        prev = op;
    }
    return cost == distance;
}

/* Furthest-reaching points per (score, diagonal): the plain textbook formulation (the span grows by one diagonal per
 * side and score, nothing is trimmed, three score-indexed tables) so that it shares nothing with the kernels it
 * checks.  Storage is a ring of depth max(x, o+e)+1 rows per table over columns k + plen + 1, kept in a scratch the
 * caller owns (one per worker) that is grown, never freed per pair, and never bulk-filled: spans only grow, so a ring slot is always
 * reused by a WIDER row, and the only cells a read can reach that the slot's current row has not written are columns
 * that entered the span after the slot's previous (narrower) row was written -- those are set to NONE in every slot
 * of the ring at the moment they enter (two columns per score). */
void verification_scratch_free(verification_scratch_t* sc) {
    if (!sc) return;
    free(sc->buf);
    sc->buf = NULL; sc->cap = 0;
}

int verification_cpu_score(const char* pattern, const char* text, size_t plen, size_t tlen, int x, int o, int e) {
    verification_scratch_t sc = {NULL, 0};
    const int s = verification_cpu_score_scratch(pattern, text, plen, tlen, x, o, e, &sc);
    verification_scratch_free(&sc);
    return s;
}

int verification_cpu_score_scratch(const char* pattern, const char* text, size_t plen_, size_t tlen_,
                                   int x, int o, int e, verification_scratch_t* sc) {
    const int plen = (int)plen_, tlen = (int)tlen_;
    const int W = plen + tlen + 5, K0 = plen + 2;
    const int NONE = -(1 << 28);
    const int oe = o + e;
    const int depth = (x > oe ? x : oe) + 1;
    const size_t need = (size_t)W * 3 * (size_t)depth;
    if (!sc) return -1;
    if (need > sc->cap) {
        free(sc->buf);
        sc->cap = need + need / 4;
        sc->buf = (int*)malloc(sizeof(int) * sc->cap);
        if (!sc->buf) { sc->cap = 0; return -1; }
    }
    int* M = sc->buf; int* I = M + (size_t)W * depth; int* D = I + (size_t)W * depth;
    const int kend = tlen - plen;
    /* columns -1, 0, 1 of every slot: the span of score 0 and its guards */
    for (int r = 0; r < depth; ++r)
        for (int k = -1; k <= 1; ++k) { M[(size_t)r * W + K0 + k] = NONE; I[(size_t)r * W + K0 + k] = NONE; D[(size_t)r * W + K0 + k] = NONE; }
    {
        int h = 0;
        while (h < plen && h < tlen && pattern[h] == text[h]) ++h;
        M[K0] = h;
        if (kend == 0 && h >= tlen) return 0;
    }
    int lo = 0, hi = 0;     /* diagonal span that can be non-empty at the current score */
    int s;
    for (s = 1;; ++s) {
        int* Ms = M + (size_t)(s % depth) * W; int* Is = I + (size_t)(s % depth) * W; int* Ds = D + (size_t)(s % depth) * W;
        const int* Mx = s >= x ? M + (size_t)((s - x) % depth) * W : NULL;
        const int* Mo = s >= oe ? M + (size_t)((s - oe) % depth) * W : NULL;
        const int* Ie = s >= e ? I + (size_t)((s - e) % depth) * W : NULL;
        const int* De = s >= e ? D + (size_t)((s - e) % depth) * W : NULL;
        if (lo > -plen) {
            --lo;
            for (int r = 0; r < depth; ++r) { M[(size_t)r * W + K0 + lo - 1] = NONE; I[(size_t)r * W + K0 + lo - 1] = NONE; D[(size_t)r * W + K0 + lo - 1] = NONE; }
        }
        if (hi < tlen) {
            ++hi;
            for (int r = 0; r < depth; ++r) { M[(size_t)r * W + K0 + hi + 1] = NONE; I[(size_t)r * W + K0 + hi + 1] = NONE; D[(size_t)r * W + K0 + hi + 1] = NONE; }
        }
        for (int k = lo; k <= hi; ++k) {
            int ins = NONE, del = NONE, mis = NONE;
            if (Mo && Mo[K0 + k - 1] > ins) ins = Mo[K0 + k - 1];
            if (Ie && Ie[K0 + k - 1] > ins) ins = Ie[K0 + k - 1];
            if (ins >= 0) ++ins;
            if (Mo && Mo[K0 + k + 1] > del) del = Mo[K0 + k + 1];
            if (De && De[K0 + k + 1] > del) del = De[K0 + k + 1];
            if (Mx && Mx[K0 + k] >= 0) mis = Mx[K0 + k] + 1;
            if (ins >= 0 && (ins > tlen || ins - k > plen)) ins = NONE;
            if (del >= 0 && (del > tlen || del - k > plen)) del = NONE;
            if (mis >= 0 && (mis > tlen || mis - k > plen)) mis = NONE;
            Is[K0 + k] = ins; Ds[K0 + k] = del;
            int m = mis > ins ? mis : ins; if (del > m) m = del;
            if (m >= 0) {
                int h = m, v = m - k;
                while (v < plen && h < tlen && pattern[v] == text[h]) { ++v; ++h; }
                m = h;
            }
            Ms[K0 + k] = m;
        }
        if (kend >= lo && kend <= hi && Ms[K0 + kend] >= tlen) break;
    }
    return s;
}
