/* eagle_scan_demo.c -- the C ABI of include/eagle_hip.h used from plain C, no Python, no R.
 *
 *   eagle_scan_demo <marker text file> <AA> <AB> <BB> <workdir>
 *
 * Converts a whitespace-separated genotype table (what ReadMarker(type="text") takes) into M.ascii / Mt.ascii, builds
 * MM^T and its normalisation (calcMMt.R:13), runs one calculate_a_and_vara scan with S = V = I, a_hat = ones -- for
 * which a_i = sum_j m_ij and vara_i = sum_j m_ij^2, checked here on the host -- and prints the selected marker
 * (find_qtl.R:71-83).  Build:  gcc -O2 -Iinclude examples/eagle_scan_demo.c -Leagleeverything_amd -leaglehip
 *                              -Wl,-rpath,$PWD/eagleeverything_amd -lm -o eagle_scan_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "eagle_hip.h"

static void on_message(const char* text, void* user) { (void)user; fprintf(stderr, "[eagle] %s\n", text); }

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc__ = (call);                                                            \
        if (rc__ != EAGLE_OK) {                                                       \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc__, eagle_last_error(ctx));    \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 6) { fprintf(stderr, "usage: %s <marker text file> <AA> <AB> <BB> <workdir>\n", argv[0]); return 2; }
    eagle_ctx* ctx = eagle_open(0);
    if (!ctx) { fprintf(stderr, "eagle_open: %s\n", eagle_open_error()); return 1; }
    eagle_set_message_callback(ctx, on_message, NULL);
    char fM[4096], fMt[4096];
    snprintf(fM, sizeof fM, "%s/M.ascii", argv[5]);
    snprintf(fMt, sizeof fMt, "%s/Mt.ascii", argv[5]);
    long dims[2];
    CHECK(eagle_get_row_column(ctx, argv[1], dims));                                           /* ReadMarker.R:283 */
    const long n = dims[0], L = dims[1];
    CHECK(eagle_create_M_ascii(ctx, argv[1], fM, "text", argv[2], argv[3], argv[4], 8.0, dims, 1, "NA"));
    CHECK(eagle_create_Mt_ascii(ctx, fM, fMt, "text", 8.0, dims, 1));
    double na = NAN;
    double* MMt = malloc(sizeof(double) * n * n);
    CHECK(eagle_calculateMMt(ctx, fM, 8.0, 4, &na, 1, dims, 1, MMt));                          /* calculateMMt.R:24 */
    double trace = 0, mx = 0;
    for (long i = 0; i < n; i++) trace += MMt[i * n + i];
    CHECK(eagle_last_mmt_normalised(ctx, MMt, &mx));                                           /* calcMMt.R:13 */
    double* S = calloc((size_t)n * n, sizeof(double));
    double* ahat = malloc(sizeof(double) * n);
    for (long i = 0; i < n; i++) { S[i * n + i] = 1.0; ahat[i] = 1.0; }
    double* a = malloc(sizeof(double) * L);
    double* vara = malloc(sizeof(double) * L);
    const long dimsT[2] = {L, n};                                                              /* calculate_a_and_vara.R:21 */
    CHECK(eagle_calculate_a_and_vara(ctx, fMt, &na, 1, S, S, 8.0, dimsT, ahat, 1, a, vara));
    long idx = 0, ties = 0;
    double tsqmax = 0;
    CHECK(eagle_last_scan_argmax(ctx, &idx, &tsqmax, &ties));
    /* host check of the closed form through ReadBlock (a marker row of Mt.ascii as doubles) */
    double* row = malloc(sizeof(double) * n);
    long bad = 0;
    for (long i = 0; i < L; i += (L > 64 ? L / 64 : 1)) {
        CHECK(eagle_read_block(ctx, fMt, i, n, 1, row));
        double s1 = 0, s2 = 0;
        for (long j = 0; j < n; j++) { s1 += row[j]; s2 += row[j] * row[j]; }
        if (fabs(a[i] - s1) > 1e-9 * (1 + fabs(s1)) || fabs(vara[i] - s2) > 1e-7 * (1 + s2)) bad++;
    }
    printf("n=%ld L=%ld trace(MMt)=%.0f max(MMt)=%.0f selected=%ld tsqmax=%.12g near_ties=%ld closed_form_mismatches=%ld\n", n, L, trace, mx, idx,
           tsqmax, ties, bad);
    free(MMt); free(S); free(ahat); free(a); free(vara); free(row);
    eagle_close(ctx);
    return bad ? 1 : 0;
}
