/* Where the 0.35 s of `agrep-hip -c word file` on a 1 MiB file go: the C-ABI calls of a count-only
 * scan, timed one by one.  Build + run on the GPU box:
 *   gcc -O2 -Iinclude scripts/c1_timeline.c -o /tmp/c1_timeline -Lagrep_amd -lagrep_hip -Wl,-rpath,$PWD/agrep_amd && /tmp/c1_timeline /tmp/c1.txt */
#include <fcntl.h>
#include <stdio.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include "agrep_hip.h"

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    double t0 = now();
    int nd = agh_device_count();
    double t1 = now();
    agh_query *q = agh_query_literal((const unsigned char *)"haystack", 8, 0, 0, (const unsigned char *)"\n", 1);
    double t2 = now();
    if (!q) { fprintf(stderr, "query: %s\n", agh_last_error()); return 1; }
    agh_result r;
    double ts[4];
    for (int i = 0; i < 3; ++i) {
        int fd = open(argv[1], O_RDONLY);
        double a = now();
        if (agh_scan_fd(q, fd, AGH_COUNT, &r, NULL, 0) != 0) { fprintf(stderr, "scan: %s\n", agh_last_error()); return 1; }
        ts[i] = now() - a;
        close(fd);
    }
    double t3 = now();
    agh_query_free(q);
    double t4 = now();
    printf("devices %d  matched %llu\n", nd, (unsigned long long)r.n_matched);
    printf("agh_device_count (runtime start) %.3f s\nagh_query_literal %.3f s\nagh_scan_fd: first %.3f s, second %.4f s, third %.4f s\nagh_query_free %.3f s\n",
           t1 - t0, t2 - t1, ts[0], ts[1], ts[2], t4 - t3);
    return 0;
}
