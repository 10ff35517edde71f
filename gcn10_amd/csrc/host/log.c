/* log.c -- per-worker log files in the reference's format.
 *
 * "<log_dir>/rank_<r>.log", opened in append mode, one line per message:
 *   [2026-10-04T17:08:12] [INFO] [rank 0] processing block 2234
 * with the start / stop lines mirrored to stderr
 * (/root/reference/src/log.c:67-86, 104-117, 149-166, 255-272).  A "rank" is a
 * GPU worker thread here; the MPI progress plumbing of the reference
 * (src/log.c:124-248) has no counterpart -- completions are logged directly.
 */
#include "gcn10_host.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>

struct gcn10_log {
    FILE *fp;
    int rank;
    pthread_mutex_t mu;
};

static void timestamp(char *buf, size_t n)
{
    time_t t = time(NULL);
    struct tm tmv;

    localtime_r(&t, &tmv);
    strftime(buf, n, "%Y-%m-%dT%H:%M:%S", &tmv);        /* src/log.c:88-97 */
}

gcn10_log *gcn10_log_open(const char *log_dir, int rank)
{
    gcn10_log *lg = calloc(1, sizeof *lg);
    char path[4096], ts[64];
    struct stat st;

    if (!lg)
        return NULL;
    lg->rank = rank;
    pthread_mutex_init(&lg->mu, NULL);

    if (log_dir && *log_dir) {                          /* src/log.c:45-64 */
        if (stat(log_dir, &st) != 0) {
            if (mkdir(log_dir, 0775) != 0 && errno != EEXIST)
                fprintf(stderr, "log: failed to create directory '%s': %s\n", log_dir,
                        strerror(errno));
        }
        else if (!S_ISDIR(st.st_mode)) {
            fprintf(stderr, "log: path '%s' exists and is not a directory\n", log_dir);
        }
    }
    snprintf(path, sizeof path, "%s/rank_%d.log", (log_dir && *log_dir) ? log_dir : ".", rank);
    lg->fp = fopen(path, "a");
    if (!lg->fp)
        fprintf(stderr, "log: failed to open %s: %s (fallback to stderr only)\n", path,
                strerror(errno));

    timestamp(ts, sizeof ts);
    if (lg->fp) {
        fprintf(lg->fp, "[%s] [rank %d] logging started\n", ts, rank);  /* src/log.c:111-114 */
        fflush(lg->fp);
    }
    fprintf(stderr, "[%s] [rank %d] logging started\n", ts, rank);
    return lg;
}

void gcn10_log_message(gcn10_log *lg, const char *level, const char *msg, bool also_console)
{
    char ts[64];
    int rank = lg ? lg->rank : 0;

    timestamp(ts, sizeof ts);
    if (!level)
        level = "INFO";
    if (!msg)
        msg = "";
    if (lg) {
        pthread_mutex_lock(&lg->mu);
        if (lg->fp) {
            fprintf(lg->fp, "[%s] [%s] [rank %d] %s\n", ts, level, rank, msg);
            fflush(lg->fp);
        }
        pthread_mutex_unlock(&lg->mu);
    }
    if (also_console || !lg || !lg->fp)
        fprintf(stderr, "[%s] [%s] [rank %d] %s\n", ts, level, rank, msg);
}

void gcn10_log_close(gcn10_log *lg)
{
    char ts[64];

    if (!lg)
        return;
    timestamp(ts, sizeof ts);
    if (lg->fp) {
        fprintf(lg->fp, "[%s] [rank %d] logging finished\n", ts, lg->rank);    /* src/log.c:262-268 */
        fclose(lg->fp);
    }
    fprintf(stderr, "[%s] [rank %d] logging finished\n", ts, lg->rank);
    pthread_mutex_destroy(&lg->mu);
    free(lg);
}
