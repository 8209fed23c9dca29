/* Wall-clock helper (reference: utils/wf_clock.h:29-36 macros). */
#ifndef WFAGPU_WF_CLOCK_H
#define WFAGPU_WF_CLOCK_H
#include <time.h>
static inline double wf_now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
#endif
