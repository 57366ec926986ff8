/* readdata.c -- the numbers of the input file: fixed-point conversion (src/fixed.c) and read_matrix / read_vector
 * (src/linear.c:27-102).  Pure host code (no GPU, no sockets): also part of libhosttest.so for the CPU tests. */
#define _GNU_SOURCE
#include <limits.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <unistd.h>

#include "protocol.h"
#include "protocol_int.h"

/* (fixed_t)(d * (1ll << p)) with the phase-2 type (src/fixed.c:3-5, src/linear.c:51) */
/* Out of range the C cast is undefined; the reference runs on x86-64, where cvttsd2si yields the
 * "integer indefinite" value (INT_MIN of the type) -- the rule the oracle states (orc_double_to_fixed). */
int64_t double_to_fixed(double d, int p, int w) {
    double t = d * (double)(1ll << p);
    if (w == 32) {
        if (!(t > -2147483649.0 && t < 2147483648.0)) return (int64_t)INT32_MIN;
        return (int64_t)(int32_t)t;
    }
    if (!(t >= -9223372036854775808.0 && t < 9223372036854775808.0)) return INT64_MIN;
    return (int64_t)t;
}
double fixed_to_double(int64_t f, int p) { return ((double)f) / (double)(1ll << p); }

/* read_matrix / read_vector (src/linear.c:27-102), values divided by the normalizer */
int read_values(FILE *f, size_t count, int precision, double normalizer, int w2, int64_t *out) {
    for (size_t i = 0; i < count; i++) {
        double val;
        if (fscanf(f, "%lf", &val) != 1) return 1;
        val /= normalizer;
        out[i] = double_to_fixed(val, precision, w2);
    }
    return 0;
}

/* read_matrix / read_vector for a data provider.  The rest of the input file -- n d, n x d numbers, n, n numbers,
 * separated by white space -- is scanned token by token; only the columns this party owns (and the target, if it owns it)
 * are converted -- with strtod, i.e. the same correctly rounded value "%lf" gives -- and quantised, every other entry stays 0
 * (it is never used: a provider only ever touches its own columns).  Token counts and syntax are still checked for the
 * whole file.
 * Round 4: the scan is split over threads.  Config 4's file is 25 million numbers in 500 MB, and one thread took 0.6 s over
 * it -- a quarter of that run's wall clock, with every provider doing the same beside the others.  The file is cut into
 * byte ranges that begin at token starts; pass 1 counts the tokens of each range (and clears a slice of X), a prefix sum
 * gives every range the index of its first token, pass 2 converts: token i < n d is X[i / d][i % d], token n d is the
 * count of the target vector, the rest is y.  Same values, same checks (a converted token must be a number to its last
 * character). */
static int is_space(int ch) { return ch == ' ' || ch == '\n' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; }
static int is_num_char(int ch) { return (ch >= '0' && ch <= '9') || ch == '.' || ch == '-' || ch == '+' || ch == 'e' || ch == 'E' || ch == 'i' || ch == 'n' || ch == 'f' || ch == 'a' || ch == 'I' || ch == 'N' || ch == 'F' || ch == 'A' || ch == 'x' || ch == 'X'; }

/* Every thread streams its byte range of the FILE through a buffer of its own with pread: no 500 MB copy, and no 125 000
 * page faults of a mapping either -- those contend for the process's address-space lock with the HIP runtime that is coming up
 * on another thread at that moment (a mapped file cost the runtime 0.15 s of its start in config 4). */
enum { kScanBlock = 1 << 20, kMaxToken = 4096 };
typedef struct {
    int fd;
    size_t lo, hi;                 /* byte range of the file; lo is a token start or white space */
    size_t first;                  /* index of the first token of the range (after pass 1) */
    size_t ntok;
    size_t n, d, c0, c1;
    int own_y, precision, w2;
    double normalizer;
    int64_t *Xq, *yq;
    size_t z0, z1;                 /* the slice of Xq this job clears in pass 1 */
    int bad;
} scan_job;

/* the next block of the range: whole tokens only, NUL-terminated; returns its length (0 at the end of the range, -1 on error) */
static long next_block(scan_job *j, size_t *pos, char *tmp) {
    if (*pos >= j->hi) return 0;
    size_t want = j->hi - *pos < (size_t)kScanBlock ? j->hi - *pos : (size_t)kScanBlock, got = 0;
    while (got < want) {
        ssize_t r = pread(j->fd, tmp + got, want - got, (off_t)(*pos + got));
        if (r < 0) return -1;
        if (r == 0) break;
        got += (size_t)r;
    }
    if (got == 0) return -1;
    if (*pos + got < j->hi) {                        /* not the last block: stop after the last complete token */
        size_t k = got;
        while (k > 0 && !is_space((unsigned char)tmp[k - 1])) k--;
        if (k == 0 || got - k > (size_t)kMaxToken) return -1;      /* a "token" longer than any number */
        got = k;
    }
    tmp[got] = 0;
    *pos += got;
    return (long)got;
}
static void *count_main(void *arg) {
    scan_job *j = arg;
    char *tmp = malloc((size_t)kScanBlock + 1);
    size_t cnt = 0, pos = j->lo;
    long m;
    if (!tmp) { j->bad = 1; return 0; }
    while ((m = next_block(j, &pos, tmp)) > 0) {
        int in_tok = 0;
        for (long i = 0; i < m; i++) {
            int sp = is_space((unsigned char)tmp[i]);
            if (!sp && !in_tok) cnt++;
            in_tok = !sp;
        }
    }
    if (m < 0) j->bad = 1;
    free(tmp);
    j->ntok = cnt;
    memset(j->Xq + j->z0, 0, (j->z1 - j->z0) * sizeof *j->Xq);
    return 0;
}
static void *parse_main(void *arg) {
    scan_job *j = arg;
    char *tmp = malloc((size_t)kScanBlock + 1);
    const size_t nd = j->n * j->d;
    size_t i = j->first, row = i < nd ? i / j->d : 0, col = i < nd ? i % j->d : 0, pos = j->lo;
    long m;
    if (!tmp) { j->bad = 1; return 0; }
    while ((m = next_block(j, &pos, tmp)) > 0) {
        const char *p = tmp, *end = tmp + m;
        for (;;) {
            while (p < end && is_space((unsigned char)*p)) p++;
            if (p >= end) break;
            if (i < nd) {
                if (col >= j->c0 && col < j->c1) {
                    char *e;
                    double v = strtod(p, &e);
                    if (e == p || (*e && !is_space((unsigned char)*e))) goto bad;
                    p = e;
                    j->Xq[row * j->d + col] = double_to_fixed(v / j->normalizer, j->precision, j->w2);
                } else {
                    /* a column of another provider: not converted, but every character must be one a number is written
                     * with and the token ends where pass 1's tokeniser (is_space) ends it -- a stray control byte or a
                     * letter would otherwise shift every later row / column index silently */
                    if (!is_num_char((unsigned char)*p)) goto bad;
                    while (p < end && !is_space((unsigned char)*p)) { if (!is_num_char((unsigned char)*p)) goto bad; p++; }
                }
                if (++col == j->d) { col = 0; row++; }
            } else if (i == nd) {                       /* read_vector's length (src/linear.c:83-87) */
                char *e;
                unsigned long long n2 = strtoull(p, &e, 10);
                if (e == p || (*e && !is_space((unsigned char)*e)) || n2 != j->n) goto bad;
                p = e;
            } else {
                const size_t k = i - nd - 1;
                if (k >= j->n) goto done;               /* anything after the n-th entry of y is not read (as fscanf would not) */
                char *e;
                double v = strtod(p, &e);
                if (e == p || (*e && !is_space((unsigned char)*e))) goto bad;
                p = e;
                if (j->own_y) j->yq[k] = double_to_fixed(v / j->normalizer, j->precision, j->w2);
            }
            i++;
        }
    }
    if (m < 0) goto bad;
    if (i - j->first != j->ntok) goto bad;             /* pass 2 must have seen exactly the tokens pass 1 counted in this range */
done:
    free(tmp);
    return 0;
bad:
    j->bad = 1;
    free(tmp);
    return 0;
}

int read_own_columns_threads(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
                             int64_t *Xq, int64_t *yq, int threads) {
    long at = ftell(f);
    if (at < 0 || fseek(f, 0, SEEK_END)) return 1;
    long endpos = ftell(f);                            /* (the stream stays at the end: this call consumes the rest of the file) */
    if (endpos < at) return 1;
    const int fd = fileno(f);
    const size_t len = (size_t)endpos;
    int rc = 1;
    scan_job *jobs = 0;
    pthread_t *th = 0;
    size_t body = (size_t)at;
    {   /* read_matrix's header (src/linear.c:30-34) */
        char head[256];
        ssize_t r = pread(fd, head, sizeof head - 1, (off_t)at);
        if (r <= 0) return 1;
        head[r] = 0;
        char *p = head, *e;
        while (is_space((unsigned char)*p)) p++;
        size_t n2 = strtoull(p, &e, 10); if (e == p) return 1; p = e;
        while (is_space((unsigned char)*p)) p++;
        size_t d2 = strtoull(p, &e, 10); if (e == p) return 1; p = e;
        if (!*p && (size_t)r == sizeof head - 1) return 1;           /* a header of 255 digits is not one */
        if (*p && !is_space((unsigned char)*p)) return 1;
        if (n2 != n || d2 != d) return 1;
        body += (size_t)(p - head);
    }
    memset(yq, 0, n * sizeof *yq);                                 /* (Xq: every thread clears its slice in pass 1) */
    if (threads < 1) threads = 1;
    if ((size_t)threads > (len - body) / 64 + 1) threads = (int)((len - body) / 64 + 1);    /* (ranges of at least a few tokens) */
    jobs = calloc((size_t)threads, sizeof *jobs);
    th = calloc((size_t)threads, sizeof *th);
    if (!jobs || !th) goto out;
    for (int t = 0; t < threads; t++) {
        size_t lo = body + (len - body) / (size_t)threads * (size_t)t;
        if (t > 0) {                                               /* a token belongs to the range it starts in */
            char win[kMaxToken + 1];
            for (;;) {
                if (lo >= len) { lo = len; break; }
                ssize_t r = pread(fd, win, sizeof win, (off_t)(lo - 1));
                if (r <= 0) goto out;
                ssize_t k = 0;
                while (k < r && !is_space((unsigned char)win[k])) k++;        /* win[0] is the byte BEFORE lo */
                if (k < r) { lo += (size_t)k; break; }
                if (r == (ssize_t)sizeof win) goto out;            /* no white space in 4 KiB: not a file of numbers */
                lo = len; break;                                   /* the last token runs to the end of the file */
            }
        }
        jobs[t].lo = lo;
    }
    for (int t = 0; t < threads; t++) {
        scan_job *j = &jobs[t];
        j->fd = fd; j->hi = t + 1 < threads ? jobs[t + 1].lo : len;
        if (j->hi < j->lo) j->hi = j->lo;
        j->n = n; j->d = d; j->c0 = c0; j->c1 = c1; j->own_y = own_y; j->precision = precision; j->w2 = w2; j->normalizer = normalizer;
        j->Xq = Xq; j->yq = yq;
        j->z0 = n * d / (size_t)threads * (size_t)t; j->z1 = t + 1 < threads ? n * d / (size_t)threads * (size_t)(t + 1) : n * d;
    }
    for (int pass = 0; pass < 2; pass++) {
        int started = 0;
        for (int t = 1; t < threads; t++) {
            if (pthread_create(&th[t], 0, pass ? parse_main : count_main, &jobs[t])) break;
            started = t;
        }
        if (started != threads - 1) {                          /* could not start a thread: finish what runs, then go serial */
            for (int t = 1; t <= started; t++) pthread_join(th[t], 0);
            for (int t = started + 1; t < threads; t++) (pass ? parse_main : count_main)(&jobs[t]);
            (pass ? parse_main : count_main)(&jobs[0]);
        } else {
            (pass ? parse_main : count_main)(&jobs[0]);
            for (int t = 1; t < threads; t++) pthread_join(th[t], 0);
        }
        for (int t = 0; t < threads; t++) if (jobs[t].bad) goto out;
        if (!pass) {
            size_t at_tok = 0;
            for (int t = 0; t < threads; t++) { jobs[t].first = at_tok; at_tok += jobs[t].ntok; }
            if (at_tok < n * d + 1 + n) goto out;              /* too few numbers (fscanf would have failed at the end of file) */
        }
    }
    rc = 0;
out:
    free(jobs); free(th);
    return rc;
}

/* CPUs this process may really use: the online count, or the cgroup's CPU quota where one is set (cgroup v2 cpu.max, v1
 * cfs_quota_us / cfs_period_us) -- the GPU boxes of this pool show 256 CPUs and grant 16 */
static long usable_cpus(void) {
    long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    if (cpus < 1) cpus = 1;
    long quota = -1, period = -1;
    FILE *g = fopen("/sys/fs/cgroup/cpu.max", "r");
    if (g) {
        char q[64];
        if (fscanf(g, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atol(q);
        fclose(g);
    } else {
        FILE *a = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *b = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (a && b && fscanf(a, "%ld", &quota) == 1 && fscanf(b, "%ld", &period) == 1) { /* both read */ } else quota = -1;
        if (a) fclose(a);
        if (b) fclose(b);
    }
    if (quota > 0 && period > 0) {
        long granted = (quota + period - 1) / period;
        if (granted < cpus) cpus = granted;
    }
    return cpus;
}

/* threads for the scan: LINREG_PARSE_THREADS, else a quarter of the usable CPUs, at most 8 (the other parties of a
 * one-node run parse the same file at the same time, and their HIP runtimes are coming up: config 4 on a 16-CPU grant
 * takes 2.4-2.5 s with one thread per provider, 2.0 with two, 1.9 with four, 1.8-1.95 with eight, 1.9 with sixteen --
 * scripts/exp/parse_threads_ab.sh) */
int read_own_columns(FILE *f, size_t n, size_t d, size_t c0, size_t c1, int own_y, int precision, double normalizer, int w2,
                     int64_t *Xq, int64_t *yq) {
    int threads = 0;
    const char *e = getenv("LINREG_PARSE_THREADS");
    if (e && *e) threads = atoi(e);
    if (threads <= 0) {
        long cpus = usable_cpus();
        threads = cpus >= 8 ? (int)(cpus / 4) : 1;
        if (threads > 8) threads = 8;
    }
    if (n * d < ((size_t)1 << 18)) threads = 1;                  /* small inputs: not worth a thread */
    return read_own_columns_threads(f, n, d, c0, c1, own_y, precision, normalizer, w2, Xq, yq, threads);
}
