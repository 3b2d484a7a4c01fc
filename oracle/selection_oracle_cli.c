/*
 * selection_oracle_cli.c -- CPU ORACLE driver (test infrastructure, NOT product code).
 *
 * Restates main() of the reference's src/selection.cpp:70-304 on top of selection_oracle.c so that
 * its stdout can be compared byte for byte with the reference binary built in oracle/_ref/.
 * Options as in src/selection.cpp:86 ("xl:t:a:h:c:"), plus -n (no CB: the "smh_a" timing mode of
 * experiments/src/time_smh.cpp:229-257), -S (print "evaluated survivors selected" to stderr) and
 * -F 0|1 (estimator flavour: 1 = FMA build of the reference (default), 0 = -ffp-contract=off build).
 *
 * Sort: the reference uses std::sort (unstable) on the double cardinality (selection.cpp:251-256); the order among exactly
 * equal cardinalities is whatever GNU libstdc++'s introsort leaves.  orc_std_sort_perm (selection_oracle.c) restates that
 * algorithm, so ties land as in the reference binary (pinned by tests/golden/expected/ties_*).
 */
#include "selection_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { char *name; double card; int64_t idx; } entry_t;

static char *trim(char *s)
{   /* selection.cpp:56-57 */
    size_t b = strspn(s, " \t\r\n");
    s += b;
    size_t e = strlen(s);
    while (e && strchr(" \t\r\n", s[e - 1])) --e;
    s[e] = 0;
    return s;
}

int main(int argc, char **argv)
{
    const char *list_file = "";
    int threads = 8;
    unsigned aux_bytes = 256;
    float threshold = 0.9f;
    const char *criterion = "";
    int use_cb = 1, print_stats = 0;
    int c;
    while ((c = getopt(argc, argv, "xl:t:a:h:c:nSF:")) != -1) {
        switch (c) {
            case 'x': printf("Usage: -l -t -a -h -c\n"); return 0;
            case 'l': list_file = optarg; break;
            case 't': threads = atoi(optarg); break;
            case 'a': aux_bytes = (unsigned)atoi(optarg); break;
            case 'h': threshold = strtof(optarg, NULL); break;     /* std::stof */
            case 'c': criterion = optarg; break;
            case 'n': use_cb = 0; break;
            case 'S': print_stats = 1; break;
            case 'F': orc_set_fma(atoi(optarg)); break;   /* estimator flavour, see selection_oracle.c */
            default: break;
        }
    }
    if (!*list_file) { fprintf(stderr, "No input file provided\n"); exit(-1); }
    FILE *lf = fopen(list_file, "r");
    if (!lf) { fprintf(stderr, "No valid input file provided\n"); exit(-1); }

    int crit;
    if (!strcmp(criterion, "smh_a")) crit = 0;
    else if (!strcmp(criterion, "hll_a")) crit = 1;
    else if (!strcmp(criterion, "hll_an")) crit = 2;
    else if (!strcmp(criterion, "hll_a+smh_a")) crit = 3;          /* oracle extension: two-stage */
    else { printf("Option -c invalid. The accepted criteria are hll_a, hll_an and smh_a.\n"); return 0; }

    int64_t N = 0, capN = 0;
    entry_t *ent = NULL;
    char *line = NULL; size_t lcap = 0;
    while (getline(&line, &lcap, lf) != -1) {
        char *t = trim(line);
        if (!*t) continue;
        if (N == capN) { capN = capN ? capN * 2 : 64; ent = (entry_t *)realloc(ent, (size_t)capN * sizeof(entry_t)); }
        ent[N].name = strdup(t); ent[N].card = 0; ent[N].idx = N; ++N;
    }
    fclose(lf);
    free(line);

    const unsigned p = 14;
    unsigned m = aux_bytes / 8;                                    /* selection.cpp:231 */
    unsigned p_aux = (unsigned)__builtin_ctz(aux_bytes ? aux_bytes : 1);   /* :125 */
    if (crit == 3) { /* two-stage: -a gives the smh bytes; aux hll fixed at p=8 as in BASELINE config 5 */
        p_aux = 8;
    }
    const size_t hb = (size_t)1 << p, ab = (size_t)1 << p_aux;
    uint8_t *hll = (uint8_t *)calloc((size_t)(N ? N : 1), hb);
    uint8_t *auxh = (crit != 0) ? (uint8_t *)calloc((size_t)(N ? N : 1), ab) : NULL;
    uint64_t *auxs = (crit == 0 || crit == 3) ? (uint64_t *)calloc((size_t)(N ? N : 1) * (m ? m : 1), 8) : NULL;
    uint8_t *hll_s = (uint8_t *)calloc((size_t)(N ? N : 1), hb);
    uint8_t *auxh_s = auxh ? (uint8_t *)calloc((size_t)(N ? N : 1), ab) : NULL;
    uint64_t *auxs_s = auxs ? (uint64_t *)calloc((size_t)(N ? N : 1) * (m ? m : 1), 8) : NULL;
    double *cards = (double *)calloc((size_t)(N ? N : 1), sizeof(double));

    char path[4096];
    for (int64_t i = 0; i < N; ++i) {
        uint32_t np, hdr[4]; double val;
        snprintf(path, sizeof path, "%s.hll", ent[i].name);
        if (orc_read_hll(path, hll + (size_t)i * hb, hb, &np, hdr, &val) || np != p) {
            fprintf(stderr, "Could not read '%s'\n", path); return 2;
        }
        if (hdr[1] != 2 || val >= 0.) { fprintf(stderr, "unsupported estimator/value in '%s'\n", path); return 2; }
        ent[i].card = orc_hll_report(hll + (size_t)i * hb, p);
        if (auxh) {
            snprintf(path, sizeof path, "%s.hll_%u", ent[i].name, p_aux);
            if (orc_read_hll(path, auxh + (size_t)i * ab, ab, &np, hdr, &val) || np != p_aux) {
                fprintf(stderr, "Could not read '%s'\n", path); return 2;
            }
        }
        if (auxs) {
            snprintf(path, sizeof path, "%s.smh%u", ent[i].name, m);
            int64_t n = orc_read_smh(path, auxs + (size_t)i * m, m);
            if (n < 0) { fprintf(stderr, "Could not read '%s'\n", path); return 2; }
            if ((uint64_t)n != m) { fprintf(stderr, "ERROR: Number of bands and rows doesnt match the MinHash sketch size.\n"); }
        }
    }

    {
        entry_t *tmp = (entry_t *)malloc((size_t)(N ? N : 1) * sizeof(entry_t));
        int64_t *perm = (int64_t *)malloc((size_t)(N ? N : 1) * sizeof(int64_t));
        double *cv = (double *)malloc((size_t)(N ? N : 1) * sizeof(double));
        for (int64_t i = 0; i < N; ++i) cv[i] = ent[i].card;
        orc_std_sort_perm(cv, N, perm);
        for (int64_t r = 0; r < N; ++r) tmp[r] = ent[perm[r]];
        memcpy(ent, tmp, (size_t)N * sizeof(entry_t));
        free(tmp); free(perm); free(cv);
    }
    for (int64_t r = 0; r < N; ++r) {
        int64_t s = ent[r].idx;
        memcpy(hll_s + (size_t)r * hb, hll + (size_t)s * hb, hb);
        if (auxh) memcpy(auxh_s + (size_t)r * ab, auxh + (size_t)s * ab, ab);
        if (auxs) memcpy(auxs_s + (size_t)r * m, auxs + (size_t)s * m, (size_t)m * 8);
        cards[r] = ent[r].card;
    }

    int n_rows = 1, n_bands = 1;
    if (auxs) orc_banding(m, threshold, &n_rows, &n_bands);

    int64_t cap = N * (N - 1) / 2 + 1;
    if (cap > (int64_t)1 << 26) cap = (int64_t)1 << 26;
    orc_pair_t *out = (orc_pair_t *)malloc((size_t)cap * sizeof(orc_pair_t));
    int64_t stats[2];
    int64_t n = orc_select(hll_s, p, auxs_s, m, auxh_s, p_aux, cards, N, threshold, n_rows, n_bands,
                           use_cb, crit, out, cap, stats, threads);
    if (n > cap) { fprintf(stderr, "oracle output capacity exceeded\n"); return 3; }
    char jb[64];
    for (int64_t j = 0; j < n; ++j) {
        orc_format_jacc(out[j].jacc, jb, sizeof jb);
        printf("%s %s %s\n", ent[out[j].i].name, ent[out[j].k].name, jb);
    }
    if (print_stats)
        fprintf(stderr, "evaluated=%lld survivors=%lld selected=%lld rows=%d bands=%d\n",
                (long long)stats[0], (long long)stats[1], (long long)n, n_rows, n_bands);
    return 0;
}
