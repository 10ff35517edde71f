/* main.c -- the gcn10 command line.
 *
 * Same flags as the reference program (/root/reference/src/main.c:85-98):
 *   gcn10 -c|--config <file> [-l|--blocks <file>] [-o|--overwrite] [-h|--help] [-v|--version]
 * plus -b as a synonym of -l (the reference's usage text advertises -b,
 * src/main.c:28, while its parser only takes -l, src/main.c:90), --gpus N, and --lookups /
 * --conditions to produce a subset of the 18 rasters (BASELINE config 3: "single lookup").
 * No mpirun: one process drives every GPU of the node.
 */
#include "gcn10_host.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void usage(FILE *fp)
{
    fprintf(fp,
            "gcn10 - high-resolution curve number generator, MI355X edition\n"
            "usage:\n"
            "  gcn10 --config <config.txt> [--blocks <blocks.txt>] [--overwrite] [--gpus <n>]\n"
            "        [--lookups <names>] [--conditions drained|undrained|both]\n"
            "  gcn10 --help | -h | --version | -v\n"
            "\n"
            "options:\n"
            "  --config, -c <file>\tpath to config file (required)\n"
            "  --blocks, -l, -b <file>\toptional list of block ids to process\n"
            "  --overwrite, -o\toverwrite existing outputs if present (optional)\n"
            "  --gpus <n>\t\tnumber of GPUs to use (default: all visible)\n"
            "  --lookups <names>\tonly these lookups, e.g. g_ii or p_i,f_iii (default: all nine)\n"
            "  --conditions <c>\tdrained, undrained or both (default: both)\n"
            "  --help, -h\t\tshow this help and exit\n"
            "  --version, -v\tprint version and exit\n"
            "\n"
            "notes:\n"
            "  one process drives all GPUs of the node; 'mpirun -n <ranks>' is not needed.\n"
            "  outputs go to ./cn_rasters_drained and ./cn_rasters_undrained.\n");
}

int main(int argc, char **argv)
{
    gcn10_run_options opt;

    /* --help / --version before anything else (src/main.c:39-56, 66-69) */
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-h")) {
            usage(stdout);
            return 0;
        }
        if (!strcmp(argv[i], "--version") || !strcmp(argv[i], "-v")) {
            printf("gcn10 %s\n", GCN10_VERSION);
            return 0;
        }
    }
    memset(&opt, 0, sizeof opt);
    for (int i = 1; i < argc; i++) {            /* src/main.c:85-98 */
        if ((!strcmp(argv[i], "-c") || !strcmp(argv[i], "--config")) && i + 1 < argc)
            opt.config_path = argv[++i];
        else if ((!strcmp(argv[i], "-l") || !strcmp(argv[i], "-b") || !strcmp(argv[i], "--blocks")) &&
                 i + 1 < argc)
            opt.blocks_file = argv[++i];
        else if (!strcmp(argv[i], "-o") || !strcmp(argv[i], "--overwrite"))
            opt.overwrite = true;
        else if (!strcmp(argv[i], "--gpus") && i + 1 < argc)
            opt.gpus = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--lookups") && i + 1 < argc)
            opt.lookups = argv[++i];
        else if (!strcmp(argv[i], "--conditions") && i + 1 < argc)
            opt.conditions = argv[++i];
    }
    return gcn10_run(&opt);
}
