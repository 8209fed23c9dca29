/* stderr logging macros with the reference's spellings (utils/logger.h:27-55
 * of quim0/WFA-GPU): LOG_DEBUG compiles away unless -DDEBUG. */
#ifndef WFAGPU_LOGGER_H
#define WFAGPU_LOGGER_H

#include <stdio.h>

#define WFAGPU_LOG_(tag, ...)                                              \
    do {                                                                   \
        char wfagpu_log_buf_[1024];                                        \
        snprintf(wfagpu_log_buf_, sizeof wfagpu_log_buf_, __VA_ARGS__);    \
        fprintf(stderr, tag "%s (%s:%d)\n", wfagpu_log_buf_, __FILE__, __LINE__); \
    } while (0);

#ifdef DEBUG
#define LOG_DEBUG(...) WFAGPU_LOG_("DEBUG: ", __VA_ARGS__)
#else
#define LOG_DEBUG(...)
#endif
#define LOG_ERROR(...) WFAGPU_LOG_("[!] ERROR: ", __VA_ARGS__)
#define LOG_INFO(...) WFAGPU_LOG_("INFO: ", __VA_ARGS__)
#define LOG_WARN(...) WFAGPU_LOG_("[!] WARNING: ", __VA_ARGS__)

#endif
