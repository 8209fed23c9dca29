/* Result array allocation (reference: lib/alignment_results.c:24-55). */
#include <pthread.h>

#include "../../include/wfa_gpu_abi.h"

/* How many records an array handed out by initialize_wfa_results holds.  The aligner object has no field for it (its layout
 * is the reference's ABI), and it needs one: sequences may be added after the parameters -- and with them the results -- were
 * initialised (the reference then frees / indexes the array by the NEW count: found by the AddressSanitizer harness,
 * tests/host_api_asan.c).  A small registry beside the arrays, keyed by their address.
 * CONTRACT (include/wfa_gpu_abi.h): an array made by initialize_wfa_results is released through destroy_wfa_results --
 * the one place an entry leaves the registry.  (A hidden header in front of the array would make the count travel with the
 * array, but the reference's arrays are plain calloc blocks and its callers may treat them as such.)  An array that fails to
 * register (out of memory for the registry itself) is handed back as a failure, not as an array of unknown capacity. */
static pthread_mutex_t g_reg_mu = PTHREAD_MUTEX_INITIALIZER;
static struct { const void* p; size_t n; }* g_reg = NULL;
static size_t g_reg_len = 0, g_reg_cap = 0;

static bool reg_put(const void* p, size_t n) {
    pthread_mutex_lock(&g_reg_mu);
    /* (an entry for this address can only be a stale one -- its array was freed behind the library's back and the allocator
     * has handed the address out again: the new array's count replaces it) */
    for (size_t i = 0; i < g_reg_len; ++i)
        if (g_reg[i].p == p) { g_reg[i].n = n; pthread_mutex_unlock(&g_reg_mu); return true; }
    if (g_reg_len == g_reg_cap) {
        const size_t cap = g_reg_cap ? 2 * g_reg_cap : 16;
        void* grown = realloc(g_reg, cap * sizeof(*g_reg));
        if (grown == NULL) { pthread_mutex_unlock(&g_reg_mu); return false; }
        g_reg = grown; g_reg_cap = cap;
    }
    g_reg[g_reg_len].p = p; g_reg[g_reg_len].n = n; ++g_reg_len;
    pthread_mutex_unlock(&g_reg_mu);
    return true;
}

static size_t reg_take(const void* p, int remove) {
    size_t n = (size_t)-1;
    pthread_mutex_lock(&g_reg_mu);
    for (size_t i = 0; i < g_reg_len; ++i)
        if (g_reg[i].p == p) {
            n = g_reg[i].n;
            if (remove) g_reg[i] = g_reg[--g_reg_len];
            break;
        }
    pthread_mutex_unlock(&g_reg_mu);
    return n;
}

/* Records of an array made by initialize_wfa_results; (size_t)-1 for any other pointer. */
size_t wfagpu_amd_results_capacity(const wfa_alignment_result_t* results) { return results ? reg_take(results, 0) : (size_t)-1; }

bool initialize_wfa_results(wfa_alignment_result_t** results, const size_t num_alignments,
                            const size_t cigar_length) {
    if (results == NULL) return false;
    wfa_alignment_result_t* r = (wfa_alignment_result_t*)calloc(num_alignments ? num_alignments : 1, sizeof(*r));
    if (r == NULL) return false;
    if (!reg_put(r, num_alignments)) { free(r); return false; }
    *results = r;
    const size_t bytes = cigar_length ? cigar_length : 1;
    for (size_t i = 0; i < num_alignments; ++i) {
        r[i].cigar.buffer = (char*)calloc(bytes, 1);
        if (r[i].cigar.buffer == NULL) return false;
        r[i].cigar.buffer_size = bytes;
    }
    return true;
}

bool destroy_wfa_results(wfa_alignment_result_t* results, const size_t num_alignments) {
    if (results == NULL) return false;
    /* (an array of this library is freed by the count it was made with, whatever the caller believes it holds) */
    const size_t made = reg_take(results, 1);
    const size_t n = made != (size_t)-1 ? made : num_alignments;
    for (size_t i = 0; i < n; ++i) free(results[i].cigar.buffer);
    free(results);
    return true;
}
