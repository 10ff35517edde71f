/* lookup.c -- lookup-CSV loader of the MI355X curve-number generator.
 *
 * Keeps the observable behaviour of load_lookup_table() in the reference
 * (/root/reference/src/cn.c:13-85): same file naming, same 255 default, same
 * row grammar ("<class>_<letter>,<cn>" after one header line), same tolerance
 * of BOM / CRLF / stray commas, same rows rejected with the same log text.
 * The tokenizer is written without strtok so it is reentrant: the nine tables
 * of a run are parsed once, up front, not 18 times per block.
 */
#include "gcn10_host.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

const char *const gcn10_conds[2] = { "drained", "undrained" };
const char *const gcn10_hcs[3] = { "p", "f", "g" };
const char *const gcn10_arcs[3] = { "i", "ii", "iii" };

enum { LOOKUP_LINE_MAX = 128 };     /* char line[128], src/cn.c:17 */

static void report(gcn10_row_error_fn fn, void *user, const char *msg)
{
    if (fn)
        fn(user, msg);
}

/* one comma-delimited field starting at *cursor; NULL when none is left.
 * Empty fields are skipped, as strtok(.., ",") does (src/cn.c:52, 67). */
static const char *next_field(const char **cursor, size_t *len)
{
    const char *p = *cursor;
    const char *start;

    while (*p == ',')
        p++;
    if (*p == '\0') {
        *cursor = p;
        return NULL;
    }
    start = p;
    while (*p != '\0' && *p != ',')
        p++;
    *len = (size_t)(p - start);
    *cursor = (*p == ',') ? p + 1 : p;
    return start;
}

int gcn10_load_lookup_file(const char *path, int table[256][5],
                           gcn10_row_error_fn on_error, void *user)
{
    char line[LOOKUP_LINE_MAX];
    char msg[PATH_MAX + 256];
    FILE *f = fopen(path, "r");

    if (!f)
        return -1;

    for (int lc = 0; lc < 256; lc++)
        for (int sg = 0; sg < 5; sg++)
            table[lc][sg] = 255;            /* nodata, src/cn.c:36-40 */

    if (!fgets(line, sizeof line, f)) {     /* header, src/cn.c:43 */
        fclose(f);
        return -2;
    }

    while (fgets(line, sizeof line, f)) {
        const char *cur = line;
        size_t code_len = 0, cn_len = 0;
        const char *code = next_field(&cur, &code_len);
        const char *us, *cn_field;
        int lc, sg, cn;

        if (!code)
            continue;                       /* src/cn.c:53-55 */
        us = memchr(code, '_', code_len);
        if (!us) {                          /* src/cn.c:58-63 */
            snprintf(msg, sizeof msg, "invalid grid_code in %s: %.*s", path,
                     (int)code_len, code);
            report(on_error, user, msg);
            continue;
        }
        lc = atoi(code);                    /* digits before '_', src/cn.c:65 */
        /* letter after '_' (or the field's end): A, B, C; anything else is D,
         * src/cn.c:66 */
        {
            char letter = (us + 1 < code + code_len) ? us[1] : '\0';

            sg = letter == 'A' ? 1 : letter == 'B' ? 2 : letter == 'C' ? 3 : 4;
        }
        cn_field = next_field(&cur, &cn_len);
        if (!cn_field) {                    /* src/cn.c:68-73 */
            snprintf(msg, sizeof msg, "invalid row in %s: missing cn", path);
            report(on_error, user, msg);
            continue;
        }
        cn = atoi(cn_field);                /* src/cn.c:74 */
        if (lc >= 0 && lc < 256) {          /* src/cn.c:75-77 */
            table[lc][sg] = cn;
        }
        else {                              /* src/cn.c:78-82 */
            snprintf(msg, sizeof msg, "invalid values in %s: lc=%d, sg=%d", path,
                     lc, sg);
            report(on_error, user, msg);
        }
    }
    fclose(f);
    return 0;
}

int gcn10_load_lookup_table(const char *dir, const char *hc, const char *arc,
                            int table[256][5], gcn10_row_error_fn on_error,
                            void *user)
{
    char path[PATH_MAX];

    if (snprintf(path, sizeof path, "%s/default_lookup_%s_%s.csv", dir, hc, arc)
        >= (int)sizeof path)
        return -3;                          /* src/cn.c:21-26 */
    return gcn10_load_lookup_file(path, table, on_error, user);
}

int gcn10_load_all_lookup_tables(const char *dir, int tables[9][256][5],
                                 int *failed_k, gcn10_row_error_fn on_error,
                                 void *user)
{
    for (int hi = 0; hi < 3; hi++) {
        for (int ai = 0; ai < 3; ai++) {
            int k = hi * 3 + ai;
            int rc = gcn10_load_lookup_table(dir, gcn10_hcs[hi], gcn10_arcs[ai],
                                             tables[k], on_error, user);

            if (rc != 0) {
                if (failed_k)
                    *failed_k = k;
                return rc;
            }
        }
    }
    return 0;
}
