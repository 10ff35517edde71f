/* blocks.c -- block list file and the block-extent shapefile, without OGR.
 *
 * The reference asks OGR for the feature whose "ID" equals the block id and
 * uses the envelope of its geometry (/root/reference/src/cn.c:155-184), and
 * lists every feature's "ID" when no list file is given (src/raster.c:68-103).
 * Both need only two things from the shapefile: each record's bounding box
 * (stored in the .shp record header of Polygon / PolygonZ / PolygonM shapes)
 * and the integer "ID" column of the .dbf.  ESRI Shapefile Technical
 * Description (1998) and the dBASE III header layout are the format sources.
 */
#include "gcn10_host.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int *gcn10_read_block_list(const char *path, int *n_blocks)
{
    FILE *f = fopen(path, "r");
    int cap = 128, n = 0;
    int *ids;

    *n_blocks = 0;
    if (!f)
        return NULL;                    /* "cannot open block list file", src/raster.c:31-35 */
    ids = malloc((size_t)cap * sizeof *ids);
    if (!ids) {
        fclose(f);
        return NULL;
    }
    while (fscanf(f, "%d", &ids[n]) == 1) {     /* src/raster.c:46 */
        if (++n == cap) {
            int *grown = realloc(ids, (size_t)cap * 2 * sizeof *ids);

            if (!grown) {
                free(ids);
                fclose(f);
                return NULL;
            }
            ids = grown;
            cap *= 2;
        }
    }
    fclose(f);
    *n_blocks = n;
    return ids;
}

static uint32_t be32(const unsigned char *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

static uint32_t le32(const unsigned char *p)
{
    return ((uint32_t)p[3] << 24) | ((uint32_t)p[2] << 16) | ((uint32_t)p[1] << 8) | p[0];
}

static double le_f64(const unsigned char *p)
{
    uint64_t v = 0;
    double d;

    for (int i = 7; i >= 0; i--)
        v = (v << 8) | p[i];
    memcpy(&d, &v, sizeof d);
    return d;
}

static unsigned char *slurp(const char *path, size_t *len)
{
    FILE *f = fopen(path, "rb");
    unsigned char *buf;
    long n;

    if (!f)
        return NULL;
    if (fseek(f, 0, SEEK_END) != 0 || (n = ftell(f)) < 0 || fseek(f, 0, SEEK_SET) != 0) {
        fclose(f);
        return NULL;
    }
    buf = malloc((size_t)n + 1);
    if (!buf || fread(buf, 1, (size_t)n, f) != (size_t)n) {
        free(buf);
        fclose(f);
        return NULL;
    }
    fclose(f);
    *len = (size_t)n;
    return buf;
}

/* "<base>.shp" -> "<base>.dbf" (keeps the case of the extension) */
static char *sibling(const char *shp_path, const char *lower, const char *upper)
{
    size_t n = strlen(shp_path);
    char *out = malloc(n + 5);

    if (!out)
        return NULL;
    memcpy(out, shp_path, n + 1);
    if (n >= 4 && out[n - 4] == '.') {
        const char *ext = isupper((unsigned char)out[n - 1]) ? upper : lower;

        memcpy(out + n - 3, ext, 3);
    }
    else {
        memcpy(out + n, ".dbf", 5);
    }
    return out;
}

/* reads the integer column "ID" (matched case-insensitively, as OGR matches
 * field names) of every record, in record order: .dbf record i belongs to .shp
 * record i. */
static int read_dbf_ids(const char *dbf_path, int n_expected, int *ids, char *err, size_t errcap)
{
    size_t len = 0;
    unsigned char *d = slurp(dbf_path, &len);
    uint32_t nrec;
    unsigned hdr, rec;
    int fld_off = 1, fld_len = -1;

    if (!d || len < 32) {
        snprintf(err, errcap, "cannot read %s", dbf_path);
        free(d);
        return -1;
    }
    nrec = le32(d + 4);
    hdr = d[8] | (d[9] << 8);
    rec = d[10] | (d[11] << 8);
    for (unsigned p = 32; p + 32 <= hdr && p + 32 <= len && d[p] != 0x0D; p += 32) {
        char name[12];

        memcpy(name, d + p, 11);
        name[11] = '\0';
        if (strcasecmp(name, "ID") == 0) {
            fld_len = d[p + 16];
            break;
        }
        fld_off += d[p + 16];
    }
    if (fld_len < 0) {
        snprintf(err, errcap, "%s has no \"ID\" field", dbf_path);
        free(d);
        return -1;
    }
    if ((int)nrec < n_expected)
        n_expected = (int)nrec;
    for (int i = 0; i < n_expected; i++) {
        size_t at = (size_t)hdr + (size_t)i * rec + (size_t)fld_off;
        char tmp[64];
        int n = fld_len < 63 ? fld_len : 63;

        if (at + (size_t)fld_len > len) {
            n_expected = i;
            break;
        }
        memcpy(tmp, d + at, (size_t)n);
        tmp[n] = '\0';
        ids[i] = atoi(tmp);         /* OGR_F_GetFieldAsInteger, src/raster.c:98 */
    }
    free(d);
    return n_expected;
}

int gcn10_blocks_open(const char *shp_path, gcn10_blocks *out, char *err, size_t errcap)
{
    size_t len = 0;
    unsigned char *s = slurp(shp_path, &len);
    size_t pos = 100;
    int cap = 1024, n = 0;
    char *dbf;

    memset(out, 0, sizeof *out);
    if (!s || len < 100 || be32(s) != 9994) {
        snprintf(err, errcap, "ogr open failed: %s", shp_path);     /* src/cn.c:157 */
        free(s);
        return -1;
    }
    out->id = malloc((size_t)cap * sizeof *out->id);
    out->bbox = malloc((size_t)cap * sizeof *out->bbox);
    if (!out->id || !out->bbox)
        goto oom;

    while (pos + 8 <= len) {
        size_t content = (size_t)be32(s + pos + 4) * 2;     /* 16-bit words */
        const unsigned char *rec = s + pos + 8;
        uint32_t type;

        if (pos + 8 + content > len)
            break;
        if (n == cap) {
            int *gi = realloc(out->id, (size_t)cap * 2 * sizeof *out->id);
            double (*gb)[4];

            if (!gi)
                goto oom;
            out->id = gi;
            gb = realloc(out->bbox, (size_t)cap * 2 * sizeof *out->bbox);
            if (!gb)
                goto oom;
            out->bbox = gb;
            cap *= 2;
        }
        type = content >= 4 ? le32(rec) : 0;
        out->id[n] = 0;
        if ((type == 5 || type == 15 || type == 25 || type == 3 || type == 13 || type == 23 ||
             type == 8 || type == 18 || type == 28) && content >= 36) {
            /* box = Xmin, Ymin, Xmax, Ymax right after the shape type */
            for (int k = 0; k < 4; k++)
                out->bbox[n][k] = le_f64(rec + 4 + 8 * k);
        }
        else if ((type == 1 || type == 11 || type == 21) && content >= 20) {
            out->bbox[n][0] = out->bbox[n][2] = le_f64(rec + 4);
            out->bbox[n][1] = out->bbox[n][3] = le_f64(rec + 12);
        }
        else {
            /* null shape: OGR gives an empty envelope (all zero) */
            memset(out->bbox[n], 0, sizeof out->bbox[n]);
        }
        n++;
        pos += 8 + content;
    }
    free(s);
    s = NULL;

    dbf = sibling(shp_path, "dbf", "DBF");
    if (!dbf)
        goto oom;
    n = read_dbf_ids(dbf, n, out->id, err, errcap);
    free(dbf);
    if (n < 0) {
        gcn10_blocks_free(out);
        return -1;
    }
    out->n = n;
    return 0;

oom:
    snprintf(err, errcap, "malloc failed for shapefile ids");       /* src/raster.c:89 */
    free(s);
    gcn10_blocks_free(out);
    return -1;
}

void gcn10_blocks_free(gcn10_blocks *b)
{
    free(b->id);
    free(b->bbox);
    memset(b, 0, sizeof *b);
}

int gcn10_blocks_find(const gcn10_blocks *b, int block_id)
{
    for (int i = 0; i < b->n; i++)
        if (b->id[i] == block_id)
            return i;
    return -1;
}
