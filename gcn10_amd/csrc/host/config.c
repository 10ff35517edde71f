/* config.c -- key=value run configuration.
 *
 * Same file grammar and the same five required keys as the reference's
 * parse_config() (/root/reference/src/config.c:44-114): '#' comment lines,
 * lines without '=' ignored, unknown keys ignored, surrounding whitespace
 * trimmed, 511-byte lines.  Existing gcn10 config files work unchanged; a few
 * optional keys steer the GPU pipeline.
 */
#include "gcn10_host.h"

#include <ctype.h>
#include <stdbool.h>
#include <errno.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char *strip(char *s)
{
    char *e;

    while (*s && isspace((unsigned char)*s))
        s++;
    e = s + strlen(s);
    while (e > s && isspace((unsigned char)e[-1]))
        e--;
    *e = '\0';
    return s;
}

static int set_str(char **slot, const char *val)
{
    char *copy = strdup(val);

    if (!copy)
        return -1;
    free(*slot);        /* a repeated key: the last one wins, as in the reference */
    *slot = copy;
    return 0;
}

int gcn10_parse_lookups(const char *text, unsigned *mask)
{
    char buf[256];
    unsigned m = 0;

    if (!text || strlen(text) >= sizeof buf)
        return -1;
    strcpy(buf, text);
    for (char *tok = strtok(buf, ", \t"); tok; tok = strtok(NULL, ", \t")) {
        bool found = false;

        if (!strcmp(tok, "all")) {
            m |= 0x1ffu;
            continue;
        }
        for (int hi = 0; hi < 3 && !found; hi++)
            for (int ai = 0; ai < 3 && !found; ai++) {
                char name[16];

                snprintf(name, sizeof name, "%s_%s", gcn10_hcs[hi], gcn10_arcs[ai]);
                if (!strcmp(tok, name)) {
                    m |= 1u << (hi * 3 + ai);
                    found = true;
                }
            }
        if (!found)
            return -1;
    }
    if (!m)
        return -1;
    *mask = m;
    return 0;
}

int gcn10_parse_conditions(const char *text, unsigned *mask)
{
    char buf[64];
    unsigned m = 0;

    if (!text || strlen(text) >= sizeof buf)
        return -1;
    strcpy(buf, text);
    for (char *tok = strtok(buf, ", \t"); tok; tok = strtok(NULL, ", \t")) {
        if (!strcmp(tok, "both") || !strcmp(tok, "all"))
            m |= 3u;
        else if (!strcmp(tok, gcn10_conds[0]))
            m |= 1u;
        else if (!strcmp(tok, gcn10_conds[1]))
            m |= 2u;
        else
            return -1;
    }
    if (!m)
        return -1;
    *mask = m;
    return 0;
}

int gcn10_config_parse(const char *path, gcn10_config *cfg, char *err, size_t errcap)
{
    char line[512];                                     /* src/config.c:47 */
    FILE *f;

    memset(cfg, 0, sizeof *cfg);
    cfg->gpu_deflate = 2;
    cfg->gpu_inflate = 1;
    cfg->prefetch_blocks = 1;
    cfg->table_mask = 0x1ffu;
    cfg->cond_mask = 3u;
    f = fopen(path, "r");
    if (!f) {
        snprintf(err, errcap, "cannot open config '%s'", path);     /* src/config.c:52 */
        return -1;
    }
    while (fgets(line, sizeof line, f)) {
        char *p = strip(line);
        char *eq, *key, *val;
        int rc = 0;

        if (*p == '\0' || *p == '#')
            continue;
        eq = strchr(p, '=');
        if (!eq)
            continue;
        *eq = '\0';
        key = strip(p);
        val = strip(eq + 1);

        if (!strcmp(key, "hysogs_data_path"))
            rc = set_str(&cfg->hysogs_data_path, val);
        else if (!strcmp(key, "esa_data_path"))
            rc = set_str(&cfg->esa_data_path, val);
        else if (!strcmp(key, "blocks_shp_path"))
            rc = set_str(&cfg->blocks_shp_path, val);
        else if (!strcmp(key, "lookup_table_path"))
            rc = set_str(&cfg->lookup_table_path, val);
        else if (!strcmp(key, "log_dir"))
            rc = set_str(&cfg->log_dir, val);
        else if (!strcmp(key, "esa_tile_dir"))
            rc = set_str(&cfg->esa_tile_dir, val);
        else if (!strcmp(key, "gpus"))
            cfg->gpus = atoi(val);
        else if (!strcmp(key, "workers_per_gpu"))
            cfg->workers_per_gpu = atoi(val);
        else if (!strcmp(key, "strip_rows"))
            cfg->strip_rows = atoi(val);
        else if (!strcmp(key, "gpu_inflate"))
            cfg->gpu_inflate = atoi(val) != 0;
        else if (!strcmp(key, "io_threads"))
            cfg->io_threads = atoi(val);
        else if (!strcmp(key, "direct_io"))
            cfg->direct_io = atoi(val) != 0;
        else if (!strcmp(key, "prefetch_blocks"))
            cfg->prefetch_blocks = atoi(val) != 0;
        else if (!strcmp(key, "deflate_level"))
            cfg->deflate_level = atoi(val);
        else if (!strcmp(key, "gpu_deflate"))
            cfg->gpu_deflate = atoi(val) < 0 ? 0 : (atoi(val) > 2 ? 2 : atoi(val));
        else if (!strcmp(key, "lookups") || !strcmp(key, "conditions")) {
            const bool lk = key[0] == 'l';

            if ((lk ? gcn10_parse_lookups(val, &cfg->table_mask) : gcn10_parse_conditions(val, &cfg->cond_mask)) != 0) {
                fclose(f);
                snprintf(err, errcap, "bad value for %s: '%s' (%s)", key, val,
                         lk ? "names like g_ii, p_i,f_iii or all" : "drained, undrained or both");
                gcn10_config_free(cfg);
                return -3;
            }
        }
        if (rc != 0) {
            fclose(f);
            snprintf(err, errcap, "malloc failed for %s", key);     /* src/config.c:71 */
            gcn10_config_free(cfg);
            return -1;
        }
    }
    fclose(f);

    if (!cfg->hysogs_data_path || !cfg->esa_data_path || !cfg->blocks_shp_path ||
        !cfg->lookup_table_path || !cfg->log_dir) {
        snprintf(err, errcap,                                       /* src/config.c:109-111 */
                 "missing one of: hysogs_data_path, esa_data_path,\n"
                 "blocks_shp_path, lookup_table_path, log_dir");
        gcn10_config_free(cfg);
        return -2;
    }
    return 0;
}

void gcn10_config_free(gcn10_config *cfg)
{
    free(cfg->hysogs_data_path);
    free(cfg->esa_data_path);
    free(cfg->blocks_shp_path);
    free(cfg->lookup_table_path);
    free(cfg->log_dir);
    free(cfg->esa_tile_dir);
    memset(cfg, 0, sizeof *cfg);
}
