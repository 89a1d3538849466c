/* ref_driver.c — driver that exposes the REAL reference implementation (compiled in place from
 * /root/reference by oracle/Makefile into oracle/_ref/libpbwtref.so) through a flat C interface
 * so Python tests can validate the restatement in pbwt_oracle.c against it and generate the
 * golden fixtures under tests/golden/.  TEST INFRASTRUCTURE ONLY; never shipped or called by the
 * product.  This file is our own code: it plays the role pbwtMain.c plays in the reference
 * (owner of the `logFile` global, caller of the library functions) and contains no reference
 * source.  It is only compiled where /root/reference exists. */
#include "pbwt.h"          /* the reference's own header, found via -I/root/reference */
#include <stdint.h>
#include <unistd.h>
#include <time.h>
#include <sys/types.h>
#include <sys/wait.h>

FILE *logFile;             /* normally defined by the reference's main program (pbwtMain.c:179) */

typedef struct { int32_t ai, bi, start, end; } ref_match;
static ref_match *g_rec; static size_t g_n, g_cap;

static void capture(int ai, int bi, int start, int end)
{
    if (g_n == g_cap) { g_cap = g_cap ? 2 * g_cap : 4096; g_rec = realloc(g_rec, g_cap * sizeof(ref_match)); }
    g_rec[g_n].ai = ai; g_rec[g_n].bi = bi; g_rec[g_n].start = start; g_rec[g_n].end = end; ++g_n;
}

void ref_init(void)
{
    static int done;
    if (done) return;
    logFile = fopen("/dev/null", "w");
    pbwtInit();
    done = 1;
}

static PBWT *make_panel(int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart)
{
    PBWT *p = pbwtCreate(M, N);
    if (aFstart) memcpy(p->aFstart, aFstart, sizeof(int) * M);
    p->yz = arrayCreate(nz + 1, uchar);
    if (nz) memcpy(arrp(p->yz, 0, uchar), yz, nz);
    arrayMax(p->yz) = nz;
    return p;
}

/* the pbwtReadMacs build loop (pbwtIO.c:477-483) driven from bit columns, calling the reference's
 * cursor functions; dumps every site's state: a_all (N+1)*M, d_all (N+1)*(M+1) (with_d only) */
long ref_build_bitcols(int M, int N, const uint32_t *bits, int wpc, int with_d,
                       uint8_t *yz_out, long yzcap, int32_t *aFend, int32_t *a_all, int32_t *d_all)
{
    ref_init();
    PBWT *p = pbwtCreate(M, 0);
    PbwtCursor *u = pbwtCursorCreate(p, TRUE, TRUE);
    for (int k = 0; k <= N; ++k) {
        if (a_all) memcpy(a_all + (size_t)k * M, u->a, sizeof(int) * M);
        if (d_all && with_d) memcpy(d_all + (size_t)k * (M + 1), u->d, sizeof(int) * (M + 1));
        if (k == N) break;
        const uint32_t *col = bits + (size_t)k * wpc;
        for (int j = 0; j < M; ++j) { int h = u->a[j]; u->y[j] = (col[h >> 5] >> (h & 31)) & 1; }
        if (with_d) pbwtCursorWriteForwardsAD(u, k); else pbwtCursorWriteForwards(u);
        p->N++;
    }
    pbwtCursorToAFend(u, p);
    long nz = arrayMax(p->yz);
    if (nz > yzcap) nz = -1;
    else { memcpy(yz_out, arrp(p->yz, 0, uchar), nz); memcpy(aFend, p->aFend, sizeof(int) * M); }
    pbwtCursorDestroy(u); pbwtDestroy(p);
    return nz;
}

/* cursor create + ForwardsReadAD loop, as matchMaximalWithin drives it; dumps k=0..N */
void ref_sweep_dump(int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart,
                    int32_t *a_all, int32_t *d_all, uint8_t *y_all, int32_t *c_all)
{
    ref_init();
    PBWT *p = make_panel(M, N, yz, nz, aFstart);
    PbwtCursor *u = pbwtCursorCreate(p, TRUE, TRUE);
    for (int k = 0; k <= N; ++k) {
        memcpy(a_all + (size_t)k * M, u->a, sizeof(int) * M);
        memcpy(d_all + (size_t)k * (M + 1), u->d, sizeof(int) * (M + 1));
        memcpy(y_all + (size_t)k * M, u->y, M);
        c_all[k] = u->c;
        pbwtCursorForwardsReadAD(u, k);
    }
    pbwtCursorDestroy(u); pbwtDestroy(p);
}

/* matchMaximalWithin with a capturing callback: returns record count, *out = malloc'ed records */
long ref_max_within(int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart, ref_match **out)
{
    ref_init();
    PBWT *p = make_panel(M, N, yz, nz, aFstart);
    g_rec = NULL; g_n = g_cap = 0;
    matchMaximalWithin(p, capture);
    pbwtDestroy(p);
    *out = g_rec;
    return (long)g_n;
}

/* -stats -maxWithin: pbwtLongMatches prints the histogram to stdout (pbwtMatch.c:166-175).
 * Run in a forked child with stdout redirected into `path`: the reference keeps its
 * matchLengthHist static non-NULL afterwards (pbwtMatch.c:28,158-159), which would silently switch
 * every later matchMaximalWithin call in this process to histogram mode. */
int ref_max_within_hist_to_file(int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart,
                                const char *path, int with_check)
{
    ref_init();
    fflush(stdout); fflush(stderr);
    pid_t pid = fork();
    if (pid < 0) return -1;
    if (pid == 0) {
        FILE *f = fopen(path, "w");
        if (!f) _exit(2);
        dup2(fileno(f), 1);
        PBWT *p = make_panel(M, N, yz, nz, aFstart);
        isStats = TRUE; isCheck = with_check ? TRUE : FALSE;
        pbwtLongMatches(p, 0);
        fflush(stdout);
        _exit(0);
    }
    int status = 0;
    if (waitpid(pid, &status, 0) < 0) return -1;
    return (WIFEXITED(status) && WEXITSTATUS(status) == 0) ? 0 : -2;
}

/* -maxWithin exactly as the CLI prints it (reportMatch, pbwtMatch.c:46-58), optional -check */
int ref_max_within_text_to_file(int M, int N, const uint8_t *yz, long nz, const int32_t *aFstart,
                                const char *path, int with_check)
{
    ref_init();
    PBWT *p = make_panel(M, N, yz, nz, aFstart);
    fflush(stdout);
    int saved = dup(1);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    dup2(fileno(f), 1);
    isCheck = with_check ? TRUE : FALSE;
    pbwtLongMatches(p, 0);
    isCheck = FALSE;
    fflush(stdout);
    dup2(saved, 1); close(saved); fclose(f);
    pbwtDestroy(p);
    return 0;
}

/* -longWithin L exactly as the CLI prints it (pbwtLongMatches -> matchLongWithin2 -> reportMatch) */
int ref_long_within_text_to_file(int M, int N, int L, const uint8_t *yz, long nz, const int32_t *aFstart, const char *path, int with_check)
{
    ref_init();
    PBWT *p = make_panel(M, N, yz, nz, aFstart);
    fflush(stdout);
    int saved = dup(1);
    FILE *f = fopen(path, "w");
    if (!f) return -1;
    dup2(fileno(f), 1);
    isCheck = with_check ? TRUE : FALSE;
    pbwtLongMatches(p, L);
    isCheck = FALSE;
    fflush(stdout);
    dup2(saved, 1); close(saved); fclose(f);
    pbwtDestroy(p);
    return 0;
}

long ref_match_sweep(int Mp, int N, const uint8_t *pz, long pnz, const int32_t *pStart,
                     int Mq, const uint8_t *qz, long qnz, const int32_t *qStart, ref_match **out)
{
    ref_init();
    PBWT *p = make_panel(Mp, N, pz, pnz, pStart);
    PBWT *q = make_panel(Mq, N, qz, qnz, qStart);
    g_rec = NULL; g_n = g_cap = 0;
    matchSequencesSweep(p, q, capture);
    pbwtDestroy(p); pbwtDestroy(q);
    *out = g_rec;
    return (long)g_n;
}


/* matchSequencesSweep with the reference's log lines (the "no match to query" events of pbwtMatch.c:405-410 and the
 * averages line :438-439) written to `path` */
int ref_match_sweep_log_to_file(int Mp, int N, const uint8_t *pz, long pnz, const int32_t *pStart,
                                int Mq, const uint8_t *qz, long qnz, const int32_t *qStart, const char *path)
{
    ref_init();
    PBWT *p = make_panel(Mp, N, pz, pnz, pStart);
    PBWT *q = make_panel(Mq, N, qz, qnz, qStart);
    FILE *saved = logFile;
    logFile = fopen(path, "w");
    if (!logFile) { logFile = saved; return -1; }
    g_rec = NULL; g_n = g_cap = 0;
    matchSequencesSweep(p, q, capture);
    fclose(logFile); logFile = saved;
    free(g_rec); g_rec = NULL;
    pbwtDestroy(p); pbwtDestroy(q);
    return 0;
}

/* matchSequencesSweepSparse (pbwtMatch.c:501-602) through the reference's own code, 5-field records */
typedef struct { int ai, bi, start, end, sparse; } ref_match5;
static ref_match5 *g_rec5; static size_t g_n5, g_cap5;
static void capture5(int ai, int bi, int start, int end, BOOL isSparse)
{
    if (g_n5 == g_cap5) { g_cap5 = g_cap5 ? 2 * g_cap5 : 1024; g_rec5 = realloc(g_rec5, g_cap5 * sizeof(ref_match5)); }
    ref_match5 m = { ai, bi, start, end, isSparse ? 1 : 0 };
    g_rec5[g_n5++] = m;
}

long ref_match_sweep_sparse(int Mp, int N, const uint8_t *pz, long pnz, const int32_t *pStart,
                            int Mq, const uint8_t *qz, long qnz, const int32_t *qStart, int nSparse, ref_match5 **out)
{
    ref_init();
    PBWT *p = make_panel(Mp, N, pz, pnz, pStart);
    PBWT *q = make_panel(Mq, N, qz, qnz, qStart);
    g_rec5 = NULL; g_n5 = g_cap5 = 0;
    matchSequencesSweepSparse(p, q, nSparse, capture5);
    pbwtDestroy(p); pbwtDestroy(q);
    *out = g_rec5;
    return (long)g_n5;
}

/* file-level entry points of the reference: used to make .pbwt / -haps goldens */
int ref_macs_to_pbwt(const char *macs, const char *pbwt_out, const char *sites_out)
{
    ref_init();
    FILE *fp = fopen(macs, "r"); if (!fp) return -1;
    PBWT *p = pbwtReadMacs(fp); fclose(fp);
    FILE *fo = fopen(pbwt_out, "w"); if (!fo) return -2;
    pbwtWrite(p, fo); fclose(fo);
    if (sites_out) { FILE *fs = fopen(sites_out, "w"); if (!fs) return -3; pbwtWriteSites(p, fs); fclose(fs); }
    pbwtDestroy(p);
    return 0;
}

/* -checkpoint n -readMacs f (pbwtIO.c:27,158-168,481): the reference writes check_A / check_B .pbwt and .sites into the
 * current directory every n sites; run inside `dir` */
int ref_macs_checkpoint(const char *macs, int n, const char *dir)
{
    ref_init();
    char cwd[4096];
    if (!getcwd(cwd, sizeof cwd)) return -1;
    FILE *fp = fopen(macs, "r"); if (!fp) return -2;
    if (chdir(dir)) { fclose(fp); return -3; }
    nCheckPoint = n;
    PBWT *p = pbwtReadMacs(fp); fclose(fp);
    nCheckPoint = 0;
    pbwtDestroy(p);
    if (chdir(cwd)) return -4;
    return 0;
}

int ref_vcfq_to_pbwt(const char *vcfq, const char *pbwt_out, const char *sites_out)
{
    ref_init();
    FILE *fp = fopen(vcfq, "r"); if (!fp) return -1;
    PBWT *p = pbwtReadVcfq(fp); fclose(fp);
    FILE *fo = fopen(pbwt_out, "w"); if (!fo) return -2;
    pbwtWrite(p, fo); fclose(fo);
    if (sites_out) { FILE *fs = fopen(sites_out, "w"); if (!fs) return -3; pbwtWriteSites(p, fs); fclose(fs); }
    pbwtDestroy(p);
    return 0;
}

int ref_pbwt_to_haps(const char *pbwt_in, const char *haps_out)
{
    ref_init();
    FILE *fp = fopen(pbwt_in, "r"); if (!fp) return -1;
    PBWT *p = pbwtRead(fp); fclose(fp);
    FILE *fo = fopen(haps_out, "w"); if (!fo) return -2;
    pbwtWriteHaplotypes(fo, p); fclose(fo);
    pbwtDestroy(p);
    return 0;
}

/* timing helper for bench.py's cpu_baseline (kind "reference"): the reference's own build loop
 * (pbwtCursorWriteForwardsAD over bit columns, as ref_build_bitcols) followed by its -stats
 * -maxWithin sweep, no dumps; returns the number of histogram entries as a checksum */
long ref_time_build_and_within(int M, int N, const uint32_t *bits, int wpc, double *t_build, double *t_within)
{
    ref_init();
    struct timespec t0, t1, t2;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    PBWT *p = pbwtCreate(M, 0);
    PbwtCursor *u = pbwtCursorCreate(p, TRUE, TRUE);
    for (int k = 0; k < N; ++k) {
        const uint32_t *col = bits + (size_t)k * wpc;
        for (int j = 0; j < M; ++j) { int h = u->a[j]; u->y[j] = (col[h >> 5] >> (h & 31)) & 1; }
        pbwtCursorWriteForwardsAD(u, k);
        p->N++;
    }
    pbwtCursorToAFend(u, p);
    pbwtCursorDestroy(u);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    /* histogram mode of matchMaximalWithin needs the file-static matchLengthHist, reachable only through
     * pbwtLongMatches with isStats: run it with stdout sent to /dev/null */
    fflush(stdout);
    int saved = dup(1);
    FILE *f = fopen("/dev/null", "w");
    dup2(fileno(f), 1);
    isStats = TRUE;
    pbwtLongMatches(p, 0);
    isStats = FALSE;
    fflush(stdout);
    dup2(saved, 1); close(saved); fclose(f);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    *t_build = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
    *t_within = (t2.tv_sec - t1.tv_sec) + 1e-9 * (t2.tv_nsec - t1.tv_nsec);
    long nz = arrayMax(p->yz);
    pbwtDestroy(p);
    return nz;
}

/* -read x.pbwt -buildReverse -writeReverse y.pbwt (pbwtCore.c:151-191, pbwtIO.c:121-132) */
int ref_build_reverse(const char *pbwt_in, const char *rev_out)
{
    ref_init();
    FILE *fp = fopen(pbwt_in, "r"); if (!fp) return -1;
    PBWT *p = pbwtRead(fp); fclose(fp);
    pbwtBuildReverse(p);
    FILE *fo = fopen(rev_out, "w"); if (!fo) return -2;
    pbwtWriteReverse(p, fo); fclose(fo);
    pbwtDestroy(p);
    return 0;
}

/* panel transforms of the reference (pbwtSubRange pbwtCore.c:111, pbwtSelectSites :682, pbwtRemoveSites :686, pbwtSubSample
 * pbwtSample.c:59) on a .pbwt + .sites pair, written back as files: op 0 subrange [i0, i1), 1 selectSites(list file),
 * 2 removeSites(list file), 3 subsample with the haplotype list select[0..nsel) */
int ref_transform(const char *pbwt_in, const char *sites_in, int op, int i0, int i1, const char *list_file,
                  const int32_t *select, int nsel, const char *pbwt_out, const char *sites_out)
{
    ref_init();
    FILE *fp = fopen(pbwt_in, "r"); if (!fp) return -1;
    PBWT *p = pbwtRead(fp); fclose(fp);
    if (sites_in) { fp = fopen(sites_in, "r"); if (!fp) return -2; pbwtReadSites(p, fp); fclose(fp); }
    if (op == 0) p = pbwtSubRange(p, i0, i1);
    else if (op == 1 || op == 2) {
        fp = fopen(list_file, "r"); if (!fp) return -3;
        char *chr = 0; Array sites = pbwtReadSitesFile(fp, &chr); fclose(fp);
        p = (op == 1) ? pbwtSelectSites(p, sites, FALSE) : pbwtRemoveSites(p, sites, FALSE);
    } else {
        Array sel = arrayCreate(nsel, int);
        for (int i = 0; i < nsel; ++i) array(sel, i, int) = select[i];
        p = pbwtSubSample(p, sel);
    }
    FILE *fo = fopen(pbwt_out, "w"); if (!fo) return -4;
    pbwtWrite(p, fo); fclose(fo);
    if (sites_out && p->sites) { FILE *fs = fopen(sites_out, "w"); if (!fs) return -5; pbwtWriteSites(p, fs); fclose(fs); }
    return 0;
}

/* the log lines of `-read P -readSites S -selectSites L` (pbwtReadSitesFile's "read %ld sites on chromosome %s from file",
 * pbwtIO.c:263, once per sites file) written to `log_out` */
int ref_read_sites_log(const char *pbwt_in, const char *sites_in, const char *list_file, const char *log_out)
{
    ref_init();
    FILE *fp = fopen(pbwt_in, "r"); if (!fp) return -1;
    PBWT *p = pbwtRead(fp); fclose(fp);
    FILE *saved = logFile;
    logFile = fopen(log_out, "w");
    if (!logFile) { logFile = saved; return -2; }
    fp = fopen(sites_in, "r"); if (!fp) return -3;
    pbwtReadSites(p, fp); fclose(fp);
    fp = fopen(list_file, "r"); if (!fp) return -4;
    char *chr = 0; Array sites = pbwtReadSitesFile(fp, &chr); fclose(fp);
    arrayDestroy(sites);
    fclose(logFile); logFile = saved;
    pbwtDestroy(p);
    return 0;
}

size_t ref_pack3(uint8_t *y_with_sentinel, int M, uint8_t *out) { ref_init(); return pack3(y_with_sentinel, M, out); }
size_t ref_unpack3(uint8_t *z, int M, uint8_t *y, int *n0) { ref_init(); return unpack3(z, M, y, n0); }
void ref_free(void *p) { free(p); }
