/* C-only caller of libsslrec_hip.so: builds a plan from plain (rowptr, col, val), uploads it and multiplies, with no
 * Python anywhere -- the self-sufficiency check of include/sslrec_hip.h (native plan builder + SpMM entry points).
 * Replaces what a reference user gets from torch.spmm(adj, embeds) (models/general_cf/lightgcn.py:28-29).
 * build: hipcc tests/c_abi_smoke.c -I include -L sslrec_amd/csrc -lsslrec_hip -Wl,-rpath,$PWD/sslrec_amd/csrc -o c_abi_smoke
 * usage: c_abi_smoke [n_rows n_cols nnz_per_row d]   -> prints "OK max_abs_err=..." and exits 0, non-zero on any failure */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "sslrec_hip.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (int)(call);                                                   \
        if (rc_ != 0) { fprintf(stderr, "%s failed: %d\n", #call, rc_); return 2; } \
    } while (0)

static uint32_t lcg(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

static int run(int n_rows, int n_cols, int per_row, int d, int want_kind) {
    const int64_t nnz = (int64_t)n_rows * per_row;
    int64_t *rowptr = (int64_t *)malloc(sizeof(int64_t) * (n_rows + 1));
    int32_t *col = (int32_t *)malloc(sizeof(int32_t) * nnz);
    float *val = (float *)malloc(sizeof(float) * nnz);
    float *x = (float *)malloc(sizeof(float) * (size_t)n_cols * d);
    float *y = (float *)malloc(sizeof(float) * (size_t)n_rows * d);
    double *ref = (double *)calloc((size_t)n_rows * d, sizeof(double));
    uint32_t s = 12345u + (uint32_t)d;
    for (int r = 0; r <= n_rows; ++r) rowptr[r] = (int64_t)r * per_row;
    for (int64_t e = 0; e < nnz; ++e) { col[e] = (int32_t)(lcg(&s) % (uint32_t)n_cols); val[e] = (float)(lcg(&s) % 1000) / 1000.f; }
    for (size_t i = 0; i < (size_t)n_cols * d; ++i) x[i] = (float)(lcg(&s) % 2001) / 1000.f - 1.f;
    for (int r = 0; r < n_rows; ++r)
        for (int64_t e = rowptr[r]; e < rowptr[r + 1]; ++e)
            for (int k = 0; k < d; ++k) ref[(size_t)r * d + k] += (double)val[e] * (double)x[(size_t)col[e] * d + k];

    sslrec_plan_t *plan = NULL;
    CHECK(sslrec_plan_build_csr(rowptr, col, val, n_rows, n_cols, &plan));
    const int kind = sslrec_plan_layout(plan, d, want_kind, 0);
    if (kind != SSLREC_PLAN_SWEPT && kind != SSLREC_PLAN_STREAMED) { fprintf(stderr, "layout: %d\n", kind); return 3; }
    sslrec_plan_info_t info;
    CHECK(sslrec_plan_info(plan, d, kind, &info));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    CHECK(sslrec_plan_upload(plan, d, kind, st));
    float *dx, *dy, *dacc;
    CHECK(hipMalloc((void **)&dx, sizeof(float) * (size_t)n_cols * d));
    CHECK(hipMalloc((void **)&dy, sizeof(float) * (size_t)n_rows * d));
    CHECK(hipMalloc((void **)&dacc, sizeof(float) * (size_t)n_rows * d));
    CHECK(hipMemcpy(dx, x, sizeof(float) * (size_t)n_cols * d, hipMemcpyHostToDevice));
    CHECK(hipMemset(dacc, 0, sizeof(float) * (size_t)n_rows * d));
    CHECK(sslrec_plan_spmm_f32(plan, d, dx, dy, NULL, st));
    sslrec_epilogue_t epi = {0};                       /* the fused layer sum: acc += y, twice */
    epi.acc_in = dacc; epi.acc_out = dacc;
    CHECK(sslrec_plan_spmm_f32(plan, d, dx, NULL, &epi, st));
    CHECK(sslrec_plan_spmm_f32(plan, d, dx, NULL, &epi, st));
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(y, dy, sizeof(float) * (size_t)n_rows * d, hipMemcpyDeviceToHost));
    double err = 0.0, err_acc = 0.0;
    for (size_t i = 0; i < (size_t)n_rows * d; ++i) err = fmax(err, fabs((double)y[i] - ref[i]));
    CHECK(hipMemcpy(y, dacc, sizeof(float) * (size_t)n_rows * d, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)n_rows * d; ++i) err_acc = fmax(err_acc, fabs((double)y[i] - 2.0 * ref[i]));
    printf("kind=%s d=%d rows=%d nnz=%lld n_elem=%d max_abs_err=%.3g acc_err=%.3g\n", kind == SSLREC_PLAN_SWEPT ? "swept" : "streamed", d,
           n_rows, (long long)nnz, info.n_elem, err, err_acc);
    sslrec_plan_free(plan);
    hipFree(dx); hipFree(dy); hipFree(dacc);
    free(rowptr); free(col); free(val); free(x); free(y); free(ref);
    return (err < 1e-4 && err_acc < 2e-4) ? 0 : 4;
}

int main(int argc, char **argv) {
    if (sslrec_abi_version() != SSLREC_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    if (argc == 5) return run(atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), SSLREC_PLAN_AUTO);
    int rc = 0;
    rc |= run(5000, 3000, 12, 64, SSLREC_PLAN_AUTO);          /* fits the LDS: column-swept kernel */
    rc |= run(5000, 3000, 12, 128, SSLREC_PLAN_STREAMED);     /* forced streamed kernel */
    rc |= run(700, 900, 3, 32, SSLREC_PLAN_AUTO);
    if (rc == 0) printf("OK\n");
    return rc;
}
