/* TEST INFRASTRUCTURE (oracle/_ref, MKL variant).  The reference's AO2MOmmm_bra_nr_s2 (pyscf/lib/ao2mo/nr_ao2mo.c:399-419) calls
 * BLAS dsymm_; the only BLAS on this image without a cap on concurrent callers is the MKL linked statically into libtorch_cpu.so,
 * which exports dgemm_ but not dsymm_.  This shim is the dsymm_ of that variant: it completes the referenced triangle of A in
 * place (A is the reference's per-thread scratch image of one tensor row, written by NPdunpack_tril just before the call and not
 * read again afterwards) and hands the product to dgemm_ - the same 2 n^2 m flops through the same library's GEMM kernel. */
#include <stddef.h>
void dgemm_(const char *, const char *, const int *, const int *, const int *, const double *, const double *, const int *,
            const double *, const int *, const double *, double *, const int *);

void pamd_ref_dsymm_(const char *side, const char *uplo, const int *m, const int *n, const double *alpha, double *a, const int *lda,
                     const double *b, const int *ldb, const double *beta, double *c, const int *ldc)
{
    const int left = (*side == 'L' || *side == 'l');
    const int na = left ? *m : *n;
    const size_t ld = (size_t)*lda;
    /* column-major A: element (i, j) at a[i + j * ld]; the caller filled the triangle named by uplo */
    if (*uplo == 'U' || *uplo == 'u') {
        for (int j = 0; j < na; j++)
            for (int i = j + 1; i < na; i++) a[i + j * ld] = a[j + i * ld];
    } else {
        for (int j = 0; j < na; j++)
            for (int i = 0; i < j; i++) a[i + j * ld] = a[j + i * ld];
    }
    const char N = 'N';
    if (left)
        dgemm_(&N, &N, m, n, m, alpha, a, lda, b, ldb, beta, c, ldc);
    else
        dgemm_(&N, &N, m, n, n, alpha, b, ldb, a, lda, beta, c, ldc);
}
