/*
 * csrc/smatcher_main.c -- the `smatcher` command: the reference driver's role for the AC / WM path.
 *
 * Mirrors main.c of the reference around the hot path, on top of include/smatcher.h only:
 *   argument parsing          main.c:345-368   (<algorithm> -m -p_size -n -alphabet [-c])
 *   select_data_file          main.c:32-123    (paths derived from n / m / alphabet under a data dir)
 *   table setup               main.c:410-449
 *   load_files                main.c:453       (helper is absent upstream: formats defined below)
 *   shard ranges              main.c:464-477   (MPI ranks -> `-ranks R`, run one after the other, each
 *                                               on device rank % device_count; counts summed as
 *                                               MPI_Reduce does, main.c:656)
 *   multiac / multiwm2        main.c:125-157 / 268-298   ("search_ac matches \t%i\t time \t%f\n")
 *   multish                   main.c:158-196             (preBmBc, preproc_sh, search_sh)
 *   multisbom                 main.c:197-231             (preproc_sbom, search_sbom)
 *   cuda_ac1..5 / cuda_wm1..5 main.c:582-648, cuda_sh1..5 / cuda_sbom1..5 main.c:595-618
 *   report                    main.c:662-670   ("Total results: %d." ...)
 *
 * Data formats (upstream's load_files / create_multiple_pattern_with_hits live in the missing
 * ../helper.o, so they are pinned here):
 *   text file     `-coding raw`     one byte per symbol, value < alphabet (what `-c` writes)
 *                 `-coding dna`     FASTA / plain nucleotides: header lines ('>' to end of line) and
 *                                   whitespace skipped, A C G T (either case) -> 0 1 2 3, anything
 *                                   else (N, IUPAC codes) skipped
 *                 `-coding protein` FASTA / plain residues: the 20 standard amino acids
 *                                   ACDEFGHIKLMNPQRSTVWY -> 0..19, anything else skipped
 *                 `-coding ascii`   bytes as they are (alphabet 128 or 256); bytes >= alphabet fail
 *   pattern file  p_size * m symbol bytes, pattern-major (always raw symbols)
 *   `-c`          create what is missing: a synthetic text (the corpus generator of the tests and
 *                 the bench) when the text file does not exist, and a pattern file whose even
 *                 patterns are cut from the text at splitmix64-chosen offsets and whose odd patterns
 *                 are uniform random (create_multiple_pattern_with_hits' role: guaranteed hits)
 */
#include "smatcher.h"
#include "smatcher_hip.h"

#include <errno.h>
#include <sys/stat.h>
#include <time.h>

static double now_seconds(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static void usage(void)
{
    printf("smatcher - multiple pattern matching (Aho-Corasick, Wu-Manber) on MI355X\n");
    printf("Usage: smatcher <ac|sh|sbom|wm|all> -m <m> -p_size <p_size> -n <n> -alphabet <alphabet> [options]\n");
    printf("-h,--help\t\t print this help message\n");
    printf("-c\t\t\t create the data files that are missing\n");
    printf("-data <dir>\t\t data directory (default ./data-cuda-multi)\n");
    printf("-text <file>\t\t text file (default <dir>/text/text<alphabet>_<n>)\n");
    printf("-pattern <file>\t\t pattern file (default <dir>/pattern/<n>/<m>/<alphabet>/pattern)\n");
    printf("-coding <raw|dna|protein|ascii>\t how the text file encodes symbols (default raw)\n");
    printf("-ranks <R>\t\t split the text into R byte ranges as the reference's MPI ranks do\n");
    printf("-multi\t\t\t also run the ranks side by side, one device each, counts added by RCCL (default for R > 1)\n");
    printf("-dry\t\t\t load / create the data and build the tables, then stop (no GPU needed)\n");
    exit(0);
}

static void mkdirs_for(const char *path)
{
    char buf[1024];
    size_t len = strlen(path);
    if (len >= sizeof buf) fail("path too long\n");
    memcpy(buf, path, len + 1);
    for (size_t i = 1; i < len; ++i)
        if (buf[i] == '/') {
            buf[i] = 0;
            if (mkdir(buf, 0777) != 0 && errno != EEXIST) {
                fprintf(stderr, "mkdir %s: %s\n", buf, strerror(errno));
                exit(1);
            }
            buf[i] = '/';
        }
}

static int file_exists(const char *path)
{
    struct stat st;
    return stat(path, &st) == 0;
}

/* symbol code of one file byte, -1 = skip, -2 = error */
static int decode_byte(int coding, int alphabet, int ch)
{
    static const char amino[] = "ACDEFGHIKLMNPQRSTVWY";
    switch (coding) {
    case 0: return ch < alphabet ? ch : -2;
    case 1:
        switch (ch) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return -1;
        }
    case 2: {
        if (ch >= 'a' && ch <= 'z') ch -= 32;
        const char *p = ch ? strchr(amino, ch) : NULL;
        return p ? (int)(p - amino) : -1;
    }
    default: return ch < alphabet ? ch : -2;
    }
}

/* load_files' text half: up to n symbols; returns the number read */
static long load_text(const char *path, int coding, int alphabet, unsigned char *text, long n)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "cannot open text file %s: %s\n", path, strerror(errno));
        exit(1);
    }
    long got = 0;
    int in_header = 0, at_line_start = 1;
    static unsigned char buf[1 << 16];
    size_t r;
    while (got < n && (r = fread(buf, 1, sizeof buf, f)) > 0) {
        for (size_t i = 0; i < r && got < n; ++i) {
            int ch = buf[i];
            if (coding == 1 || coding == 2) {
                if (at_line_start && ch == '>') in_header = 1;
                at_line_start = ch == '\n';
                if (in_header) {
                    if (ch == '\n') in_header = 0;
                    continue;
                }
            }
            int s = decode_byte(coding, alphabet, ch);
            if (s == -2) {
                fprintf(stderr, "%s: byte %d at symbol %ld is outside the alphabet (%d)\n", path, ch, got, alphabet);
                exit(1);
            }
            if (s >= 0) text[got++] = (unsigned char)s;
        }
    }
    fclose(f);
    return got;
}

static void write_file(const char *path, const unsigned char *data, size_t bytes)
{
    mkdirs_for(path);
    FILE *f = fopen(path, "wb");
    if (!f || fwrite(data, 1, bytes, f) != bytes) {
        fprintf(stderr, "cannot write %s: %s\n", path, strerror(errno));
        exit(1);
    }
    fclose(f);
}

/* create_multiple_pattern_with_hits' role: even patterns are cut from the text, odd ones random */
static void create_patterns(unsigned char *pattern2, int m, int p_size, int alphabet, const unsigned char *text, long n)
{
    for (int j = 0; j < p_size; ++j) {
        unsigned char *dst = pattern2 + (size_t)j * m;
        if ((j & 1) == 0 && n >= m) {
            uint64_t off = smh_splitmix64_at(7, (uint64_t)j) % (uint64_t)(n - m + 1);
            memcpy(dst, text + off, (size_t)m);
        } else {
            for (int i = 0; i < m; ++i)
                dst[i] = (unsigned char)(smh_splitmix64_at(7 + 0x5bd1e995u, (uint64_t)j * (uint64_t)m + (uint64_t)i) % (uint64_t)alphabet);
        }
    }
}

int main(int argc, char **argv)
{
    int m = 0, p_size = 0, nFull = 0, alphabet = 0, B = 3, create_data = 0, ranks = 1, dry = 0, coding = 0, force_multi = 0;
    const char *data_dir = "./data-cuda-multi", *text_arg = NULL, *pattern_arg = NULL;
    int i, j;

    /* main.c:345-362 */
    for (i = 1; i < argc; i++) {
        if (strcmp(argv[i], "--help") == 0 || strcmp(argv[i], "-h") == 0) usage();
        if (strcmp(argv[i], "-c") == 0) create_data = 1;
        if (strcmp(argv[i], "-dry") == 0) dry = 1;
        if (strcmp(argv[i], "-multi") == 0) force_multi = 1;
        if (i + 1 >= argc) continue;
        if (strcmp(argv[i], "-m") == 0) m = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-n") == 0) nFull = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-p_size") == 0) p_size = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-alphabet") == 0) alphabet = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-ranks") == 0) ranks = atoi(argv[i + 1]);
        if (strcmp(argv[i], "-data") == 0) data_dir = argv[i + 1];
        if (strcmp(argv[i], "-text") == 0) text_arg = argv[i + 1];
        if (strcmp(argv[i], "-pattern") == 0) pattern_arg = argv[i + 1];
        if (strcmp(argv[i], "-coding") == 0) {
            const char *c = argv[i + 1];
            coding = strcmp(c, "raw") == 0 ? 0 : strcmp(c, "dna") == 0 ? 1 : strcmp(c, "protein") == 0 ? 2
                     : strcmp(c, "ascii") == 0 ? 3 : -1;
            if (coding < 0) fail("-coding must be raw, dna, protein or ascii\n");
        }
    }
    const char *algo = argc > 1 ? argv[1] : "";
    const int run_ac = strcmp(algo, "ac") == 0 || strcmp(algo, "all") == 0;
    const int run_wm = strcmp(algo, "wm") == 0 || strcmp(algo, "all") == 0;
    const int run_sh = strcmp(algo, "sh") == 0 || strcmp(algo, "all") == 0;
    const int run_sbom = strcmp(algo, "sbom") == 0 || strcmp(algo, "all") == 0;
    if (m == 0 || nFull == 0 || p_size == 0 || alphabet == 0 || (!run_ac && !run_wm && !run_sh && !run_sbom)) usage();
    if (p_size > 100000) fail("Only up to 100.000 patterns are supported\n"); /* main.c:370-371 */
    if (m < 3) fail("The pattern length must be at least 3 (Wu-Manber block size)\n");
    if (ranks < 1 || ranks > 4096) fail("-ranks must be between 1 and 4096\n");
    if (coding == 1 && alphabet != 4) fail("For DNA sequences, you must use an alphabet size of 4\n");       /* main.c:68 */
    if (coding == 2 && alphabet != 20) fail("For protein sequences, you must use an alphabet size of 20\n"); /* main.c:81 */

    /* select_data_file, main.c:32-123 */
    char text_filename[1024], pattern_filename[1024];
    if (text_arg) snprintf(text_filename, sizeof text_filename, "%s", text_arg);
    else snprintf(text_filename, sizeof text_filename, "%s/text/text%i_%i", data_dir, alphabet, nFull);
    if (pattern_arg) snprintf(pattern_filename, sizeof pattern_filename, "%s", pattern_arg);
    else snprintf(pattern_filename, sizeof pattern_filename, "%s/pattern/%i/%i/%i/pattern", data_dir, nFull, m, alphabet);

    unsigned char *textFull = (unsigned char *)malloc((size_t)nFull + 64);
    unsigned char **pattern = (unsigned char **)malloc((size_t)p_size * sizeof(unsigned char *));
    unsigned char *pattern2 = (unsigned char *)malloc((size_t)m * p_size);
    if (!textFull || !pattern || !pattern2) fail("Failed to allocate array\n");

    /* load_files (+ -c), main.c:451-461 */
    double timeReadFile = now_seconds();
    if (!file_exists(text_filename)) {
        if (!create_data) {
            fprintf(stderr, "text file %s does not exist (use -c to create a synthetic one)\n", text_filename);
            exit(1);
        }
        if (coding != 0) fail("-c writes raw symbol files: use -coding raw\n");
        smh_corpus_text_host(textFull, (uint64_t)nFull, 0, 42, alphabet);
        write_file(text_filename, textFull, (size_t)nFull);
        printf("created text \t%s\t symbols \t%i\n", text_filename, nFull);
    }
    long got = load_text(text_filename, coding, alphabet, textFull, nFull);
    if (got < nFull) {
        fprintf(stderr, "%s holds %ld symbols, fewer than -n %i\n", text_filename, got, nFull);
        exit(1);
    }
    if (!file_exists(pattern_filename)) {
        if (!create_data) {
            fprintf(stderr, "pattern file %s does not exist (use -c to create it)\n", pattern_filename);
            exit(1);
        }
        create_patterns(pattern2, m, p_size, alphabet, textFull, nFull);
        write_file(pattern_filename, pattern2, (size_t)m * p_size);
        printf("created patterns \t%s\t count \t%i\n", pattern_filename, p_size);
    }
    {
        FILE *f = fopen(pattern_filename, "rb");
        if (!f || fread(pattern2, 1, (size_t)m * p_size, f) != (size_t)m * p_size) {
            fprintf(stderr, "%s does not hold %i patterns of length %i\n", pattern_filename, p_size, m);
            exit(1);
        }
        fclose(f);
        for (size_t k = 0; k < (size_t)m * p_size; ++k)
            if (pattern2[k] >= alphabet) fail("pattern symbol outside the alphabet\n");
    }
    for (j = 0; j < p_size; j++) {
        pattern[j] = (unsigned char *)calloc((size_t)m + 1, 1); /* m+1 zeroed: ac/ac.c:136-143 reads pattern[j][m] */
        if (!pattern[j]) fail("Failed to allocate array!\n");
        memcpy(pattern[j], pattern2 + (size_t)j * m, (size_t)m);
    }
    timeReadFile = now_seconds() - timeReadFile;
    {
        /* what was loaded, so that a run can be tied to its input: FNV-1a 64 of the symbol stream */
        uint64_t h = 0xcbf29ce484222325ull;
        for (long k = 0; k < nFull; ++k) h = (h ^ textFull[k]) * 0x100000001b3ull;
        printf("text symbols \t%i\t fnv1a64 \t%016llx\n", nFull, (unsigned long long)h);
    }

    /* main.c:410-449 */
    int *state_transition = NULL;
    unsigned int *state_supply = NULL, *state_final = NULL;
    if (run_ac) {
        size_t rows = (size_t)m * p_size + 1;
        state_transition = (int *)malloc(rows * alphabet * sizeof(int));
        state_supply = (unsigned int *)calloc(rows, sizeof(unsigned int));
        state_final = (unsigned int *)calloc(rows, sizeof(unsigned int));
        if (!state_transition || !state_supply || !state_final) fail("Failed to allocate array\n");
        memset(state_transition, -1, rows * alphabet * sizeof(int));
    }
    int *sh_transition = NULL, *bmBc = NULL;
    unsigned int *sh_final = NULL;
    if (run_sh) { /* main.c:410-427: the Set-Horspool trie uses the same table shapes */
        size_t rows = (size_t)m * p_size + 1;
        sh_transition = (int *)malloc(rows * alphabet * sizeof(int));
        sh_final = (unsigned int *)calloc(rows, sizeof(unsigned int));
        bmBc = (int *)malloc(alphabet * sizeof(int));
        if (!sh_transition || !sh_final || !bmBc) fail("Failed to allocate array\n");
        memset(sh_transition, -1, rows * alphabet * sizeof(int));
    }
    int *sbom_transition = NULL;
    unsigned int *state_final_multi = NULL;
    if (run_sbom) { /* main.c:410-412,422-425 */
        size_t rows = (size_t)m * p_size + 1;
        sbom_transition = (int *)malloc(rows * alphabet * sizeof(int));
        state_final_multi = (unsigned int *)calloc(rows * 200, sizeof(unsigned int));
        if (!sbom_transition || !state_final_multi) fail("Failed to allocate array\n");
        memset(sbom_transition, -1, rows * alphabet * sizeof(int));
    }
    int *SHIFT = NULL, *PREFIX_value = NULL, *PREFIX_index = NULL, *PREFIX_size = NULL;
    if (run_wm) {
        wu_determine_shiftsize(alphabet);
        m_nBitsInShift = 2;
        SHIFT = (int *)malloc(shiftsize * sizeof(int));
        PREFIX_value = (int *)malloc((size_t)shiftsize * p_size * sizeof(int));
        PREFIX_index = (int *)malloc((size_t)shiftsize * p_size * sizeof(int));
        PREFIX_size = (int *)malloc(shiftsize * sizeof(int));
        if (!SHIFT || !PREFIX_value || !PREFIX_index || !PREFIX_size) fail("Failed to allocate array\n");
        for (i = 0; i < (int)shiftsize; i++) {
            SHIFT[i] = m - B + 1;
            PREFIX_size[i] = 0;
        }
    }

    /* preprocessing happens once; every rank of the reference repeats it on identical input */
    double t0 = now_seconds();
    struct ac_table *table = NULL;
    if (run_ac) {
        table = preproc_ac(pattern, m, p_size, alphabet, state_transition, state_supply, state_final);
        printf("preproc_ac states \t%u\t patterns \t%u\t time \t%f\n", table->idcounter, table->patterncounter, now_seconds() - t0);
    }
    t0 = now_seconds();
    struct ac_table *sh_table = NULL;
    if (run_sh) {
        preBmBc(pattern, m, p_size, alphabet, bmBc);
        sh_table = preproc_sh(pattern, m, p_size, alphabet, sh_transition, sh_final);
        printf("preproc_sh states \t%u\t patterns \t%u\t time \t%f\n", sh_table->idcounter, sh_table->patterncounter, now_seconds() - t0);
    }
    t0 = now_seconds();
    struct sbom_table *sbom_table = NULL;
    if (run_sbom) {
        sbom_table = preproc_sbom(pattern, m, p_size, alphabet, sbom_transition, state_final_multi);
        printf("preproc_sbom states \t%u\t patterns \t%u\t time \t%f\n", sbom_table->idcounter, sbom_table->patterncounter, now_seconds() - t0);
    }
    t0 = now_seconds();
    if (run_wm) {
        preproc_wu2(pattern2, m, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
        int zero = 0;
        for (i = 0; i < (int)shiftsize; i++) zero += SHIFT[i] == 0;
        printf("preproc_wm2 zero-shift blocks \t%i\t of \t%u\t time \t%f\n", zero, shiftsize, now_seconds() - t0);
    }
    fflush(stdout);
    if (dry) {
        if (table) free_ac(table, alphabet);
        if (sh_table) free_sh(sh_table, alphabet);
        if (sbom_table) free_sbom(sbom_table, m);
        printf("dry run: no search\n");
        fflush(stdout);
        return 0;
    }

    /* main.c:464-489: rank r scans [r*c, min((r+1)*c + m-1, nFull)), c = ceil(nFull / R) */
    const int devices = smh_device_count();
    long long ac_sum = 0, sh_sum = 0, sbom_sum = 0, wm_sum = 0, wm_gpu_sum[5] = {0};
    double timeExecuteCPU = 0, gpuTime_sum[5] = {0};
    for (int r = 0; r < ranks; ++r) {
        uint64_t begin, end;
        smh_shard_range((uint64_t)nFull, ranks, r, m, &begin, &end);
        unsigned char *text = textFull + begin;
        int n = (int)(end - begin);
        if (devices > 0 && smh_set_device(r % devices) != SMH_OK) fail("cannot select the device\n");
        if (run_ac) {
            /* multiac, main.c:125-157 */
            double t2 = now_seconds();
            int matches = (int)search_ac(text, n, table);
            double t3 = now_seconds();
            timeExecuteCPU += t3 - t2;
            printf("search_ac matches \t%i\t time \t%f\n", matches, t3 - t2);
            fflush(stdout);
            ac_sum += matches;
            cuda_ac1(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
            cuda_ac2(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
            cuda_ac3(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
            cuda_ac4(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
            cuda_ac5(m, text, n, p_size, alphabet, state_transition, state_supply, state_final);
        }
        if (run_sh) {
            /* multish, main.c:158-196 */
            double t2 = now_seconds();
            int matches = (int)search_sh(m, text, n, sh_table, bmBc);
            double t3 = now_seconds();
            timeExecuteCPU += t3 - t2;
            printf("search_sh matches \t%i\t time \t%f\n", matches, t3 - t2);
            fflush(stdout);
            sh_sum += matches;
            cuda_sh1(m, text, n, p_size, alphabet, sh_transition, sh_final, bmBc);
            cuda_sh2(m, text, n, p_size, alphabet, sh_transition, sh_final, bmBc);
            cuda_sh3(m, text, n, p_size, alphabet, sh_transition, sh_final, bmBc);
            cuda_sh4(m, text, n, p_size, alphabet, sh_transition, sh_final, bmBc);
            cuda_sh5(m, text, n, p_size, alphabet, sh_transition, sh_final, bmBc);
        }
        if (run_sbom) {
            /* multisbom, main.c:197-231 */
            double t2 = now_seconds();
            int matches = (int)search_sbom(pattern, m, text, n, sbom_table);
            double t3 = now_seconds();
            timeExecuteCPU += t3 - t2;
            printf("search_sbom matches \t%i\t time \t%f\n", matches, t3 - t2);
            fflush(stdout);
            sbom_sum += matches;
            cuda_sbom1(pattern2, m, text, n, p_size, alphabet, sbom_transition, state_final_multi);
            cuda_sbom2(pattern2, m, text, n, p_size, alphabet, sbom_transition, state_final_multi);
            cuda_sbom3(pattern2, m, text, n, p_size, alphabet, sbom_transition, state_final_multi);
            cuda_sbom4(pattern2, m, text, n, p_size, alphabet, sbom_transition, state_final_multi);
            cuda_sbom5(pattern2, m, text, n, p_size, alphabet, sbom_transition, state_final_multi);
        }
        if (run_wm) {
            /* multiwm2, main.c:268-298 */
            double t2 = now_seconds();
            int matches = (int)search_wu2(pattern2, m, p_size, text, n, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size);
            double t3 = now_seconds();
            timeExecuteCPU += t3 - t2;
            printf("search_wm2 matches \t%i\t time \t%f\n", matches, t3 - t2);
            fflush(stdout);
            wm_sum += matches;
            /* main.c:623-648 */
            int (*const gpu[5])(unsigned char *, int, unsigned char *, int, int, int, int, int *, int *, int *, int *, double *) =
                {cuda_wm1, cuda_wm2, cuda_wm3, cuda_wm4, cuda_wm5};
            for (i = 0; i < 5; i++) {
                double secs = 0;
                wm_gpu_sum[i] += gpu[i](pattern2, m, text, n, p_size, alphabet, B, SHIFT, PREFIX_value, PREFIX_index, PREFIX_size, &secs);
                gpuTime_sum[i] += secs;
            }
        }
    }

    /* The same R byte ranges once more, this time the way the reference's MPI job runs them: all at once, one
     * device per rank, text shards resident, ONE all-reduce of the 64-bit counts (main.c:464-489, 654-657 with
     * RCCL for MPI).  Only for R > 1 (or `-multi`, which also runs it with one rank) and only when R devices are
     * visible.  The per-rank results above are complete without it: when the multi-device layer cannot be set up
     * (no RCCL and no host-sum, out of memory) that is a warning; a count that DIFFERS is an error. */
    if ((ranks > 1 || force_multi) && devices >= ranks && (run_ac || run_wm)) {
        smh_multi *mg = NULL;
        if (smh_multi_create(&mg, NULL, ranks, SMH_MULTI_HOST_SUM) != SMH_OK ||
            smh_multi_load_text(mg, textFull, (uint64_t)nFull, m - 1) != SMH_OK) {
            fprintf(stderr, "multi-device run skipped: %s\n", smh_last_error());
            if (mg) smh_multi_free(mg);
            mg = NULL;
        }
        uint64_t total = 0, per[SMH_MULTI_MAX_DEVICES];
        double secs = 0;
        if (mg && run_ac) {
            smh_ac *h = smh_ac_compile_patterns(pattern2, m, p_size, alphabet);
            if (!h || smh_multi_ac_count(mg, h, &total, per, &secs) != SMH_OK) {
                fprintf(stderr, "multi-device ac skipped: %s\n", smh_last_error());
            } else {
                printf("multi-device ac (%d devices, %s) matches \t%llu\t time \t%f\n", ranks,
                       smh_multi_uses_rccl(mg) ? "RCCL all-reduce" : "host sum", (unsigned long long)total, secs);
                if ((long long)total != ac_sum) { fprintf(stderr, "multi-device ac counted %llu, the ranks one by one %lld\n", (unsigned long long)total, ac_sum); exit(1); }
            }
            if (h) smh_ac_free(h);
        }
        if (mg && run_wm) {
            smh_wm *h = smh_wm_compile(pattern2, m, p_size, alphabet);
            if (!h || smh_multi_wm_count(mg, h, &total, per, &secs) != SMH_OK) {
                fprintf(stderr, "multi-device wm skipped: %s\n", smh_last_error());
            } else {
                printf("multi-device wm (%d devices, %s) matches \t%llu\t time \t%f\n", ranks,
                       smh_multi_uses_rccl(mg) ? "RCCL all-reduce" : "host sum", (unsigned long long)total, secs);
                if ((long long)total != wm_sum) { fprintf(stderr, "multi-device wm counted %llu, the ranks one by one %lld\n", (unsigned long long)total, wm_sum); exit(1); }
            }
            if (h) smh_wm_free(h);
        }
        if (mg) smh_multi_free(mg);
        fflush(stdout);
    }

    /* main.c:662-670 */
    if (run_ac) printf("Total results (ac): %lld.\n", ac_sum);
    if (run_sh) printf("Total results (sh): %lld.\n", sh_sum);
    if (run_sbom) printf("Total results (sbom): %lld.\n", sbom_sum);
    if (run_sbom && ((run_ac && sbom_sum != ac_sum) || (run_sh && sbom_sum != sh_sum) || (run_wm && sbom_sum != wm_sum))) {
        fprintf(stderr, "SBOM counted %lld; the other algorithms disagree\n", sbom_sum);
        exit(1);
    }
    if (run_sh && run_ac && sh_sum != ac_sum) {
        fprintf(stderr, "Set-Horspool counted %lld, Aho-Corasick %lld\n", sh_sum, ac_sum);
        exit(1);
    }
    if (run_sh && run_wm && sh_sum != wm_sum) {
        fprintf(stderr, "Set-Horspool counted %lld, Wu-Manber %lld\n", sh_sum, wm_sum);
        exit(1);
    }
    if (run_wm) {
        printf("Total results: %lld.\n", wm_sum);
        for (i = 0; i < 5; i++)
            if (wm_gpu_sum[i] != wm_sum) {
                fprintf(stderr, "cuda_wm%d counted %lld, search_wu2 %lld\n", i + 1, wm_gpu_sum[i], wm_sum);
                exit(1);
            }
    }
    if (run_ac && run_wm && ac_sum != wm_sum) {
        fprintf(stderr, "Aho-Corasick counted %lld, Wu-Manber %lld\n", ac_sum, wm_sum);
        exit(1);
    }
    printf("timeReadFile: %f.\n", timeReadFile);
    printf("timeExecuteCPU: %f.\n", timeExecuteCPU);
    if (run_wm)
        for (i = 0; i < 5; i++) printf("gpuTime[%d]: %f.\n", (i + 1), gpuTime_sum[i] / ranks);

    if (table) free_ac(table, alphabet);
    if (sh_table) free_sh(sh_table, alphabet);
    free(sh_transition); free(sh_final); free(bmBc);
    if (sbom_table) free_sbom(sbom_table, m);
    free(sbom_transition); free(state_final_multi);
    for (j = 0; j < p_size; j++) free(pattern[j]);
    free(pattern); free(pattern2); free(textFull);
    free(state_transition); free(state_supply); free(state_final);
    free(SHIFT); free(PREFIX_value); free(PREFIX_index); free(PREFIX_size);
    fflush(stdout); /* also reached as smatcher_main() inside a host process that goes on living */
    return 0;
}
