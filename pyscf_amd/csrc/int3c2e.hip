// C ABI of the on-device integral generation and of the cderi = L^-1 (Q|pq) solve.
//   PAMD_int3c2e_class   <- GTOnr3c_drv / GTOnr3c_fill_s2ij + libcint int3c2e_sph
//                           (pyscf/lib/gto/fill_nr_3c.c:127-225), also used for GTOint2c
//                           (pyscf/lib/gto/fill_int2c.c) through a unit "dummy" j shell
//   PAMD_cderi_solve     <- scipy trsm at pyscf/df/incore.py:204-213 (and lib.dot at :216
//                           for the eigen-decomposition fallback)
#include "common.h"
#define RYS_QUAL static
#define RYS_WANT_TABLE
#include "rys_tables.inc"
#undef RYS_QUAL
#include "int3c2e_args.h"

namespace pamd {
int launch_int3c2e_lk0(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk1(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk2(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk3(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk4(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk5(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_lk6(int, int, const Int3c2eArgs &, hipStream_t);
int launch_int3c2e_grad_lk0(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk1(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk2(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk3(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk4(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk5(int, int, const Int3c2eGradArgs &, hipStream_t);
int launch_int3c2e_grad_lk6(int, int, const Int3c2eGradArgs &, hipStream_t);
}

using namespace pamd;

namespace {

constexpr int KB = 16, NT = 128, LDN = NT + 16, LDT = KB + 1;

// C[m][n] = sum_{k < klimit(m tile)} At[k][m] * Bt[n][k]
//   At: [kdim][lda]  (= Linv^T, k = aux function Q, m = local aux row L)
//   Bt: [n][ldb]     (= T, n = pq of the slab, k contiguous)
//   C : cderi + column offset, ldc = nao_pair
// triangular: rows m (global index m_off + m) only need k <= m_off + m.
__global__ __launch_bounds__(256, 2) void cderi_solve_kernel(
    const double *__restrict__ At, int lda, const double *__restrict__ Bt, long ldb,
    double *__restrict__ C, long ldc, int m, long n, int kdim, int m_off, int triangular)
{
    __shared__ double sA[KB * LDN];
    __shared__ double sB[NT * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * NT;
    const long n0 = (long)blockIdx.x * NT;
    int kend = kdim;
    if (triangular) {
        int lim = m_off + m0 + NT;
        if (lim < kend) kend = lim;
    }
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    double pa[8], pb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int e = tid + j * 256;
            int k = e >> 7, c = e & 127;                  // A: row k, col c
            pa[j] = (k0 + k < kend && m0 + c < m) ? At[(long)(k0 + k) * lda + m0 + c] : 0.0;
            int nn = e >> 4, kq = e & 15;                 // B: row nn (pq), 16 consecutive k
            pb[j] = (k0 + kq < kend && n0 + nn < n) ? Bt[(n0 + nn) * ldb + k0 + kq] : 0.0;
        }
    };
    if (kend > 0) fetch(0);
    for (int k0 = 0; k0 < kend; k0 += KB) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int e = tid + j * 256;
            sA[(e >> 7) * LDN + (e & 127)] = pa[j];
            sB[(e >> 4) * LDT + (e & 15)] = pb[j];
        }
        __syncthreads();
        if (k0 + KB < kend) fetch(k0 + KB);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = sA[(kk + fk) * LDN + wr * 64 + a * 16 + fn];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = sB[(wc * 64 + b * 16 + fn) * LDT + kk + fk];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            long col = n0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int row = m0 + wr * 64 + a * 16 + fk + 4 * r;
                if (row < m) C[(long)row * ldc + col] = acc[a][b][r];
            }
        }
}

// r06: the same product with BOTH operands by buffer-resource LDS-DMA, double-buffered, one barrier per
// 16-deep k-tile, the DMA issue spread over the four MFMA groups - the loop of gemm_tn_glds2 / sub_orb_dot2 instead of the r01
// register-staged one above (58 TF/s = 0.74, matrix pipe busy 0.77, two barriers per k-tile, bounds checks in the inner loads).
//   A = At[k][m] (k-major rows: one 1 KiB row DMA per k row of the 128-column tile);
//   B = Bt[n][k] (k contiguous for a fixed n - the "transposed" operand of e2_pk / sub_orb_dot2: one DMA moves eight pq rows x
//       16 k with per-lane source addresses into XOR-swizzled 16-byte chunks, fragment reads conflict-free).
// Out-of-range rows / columns need no branches: both buffer resources carry their true byte counts, loads beyond them return 0
// (k beyond kdim or beyond the diagonal block of a triangular factor multiplies zeros of L^-1 anyway).
__global__ __launch_bounds__(256, 2) void cderi_solve2_kernel(
    const double *__restrict__ At, int lda, const double *__restrict__ Bt, long ldb,
    double *__restrict__ C, long ldc, int m, long n, int kdim, int m_off, int triangular)
{
    __shared__ double sa0[KB * LDN];
    __shared__ double sa1[KB * LDN];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * NT;
    const long n0 = (long)blockIdx.x * NT;
    int kend = kdim;
    if (triangular) {
        const int lim = m_off + m0 + NT;
        if (lim < kend) kend = lim;
    }
    // (A may be handed over with a column offset into a larger [kdim][lda] buffer: only the m columns of the last row are surely there)
    const long a_bytes = ((long)(kdim - 1) * lda + ((m + 1) & ~1) - m0) * 8, b_bytes = (n - n0) * ldb * 8;
    const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc((void *)(At + m0), 0,
                                                                        (int)(a_bytes < 0 ? 0 : (a_bytes < 0x7fffffffL ? a_bytes : 0x7fffffffL)), 0x00020000);
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc((void *)(Bt + n0 * ldb), 0,
                                                                        (int)(b_bytes < 0x7fffffffL ? b_bytes : 0x7fffffffL), 0x00020000);
    const int lda8 = lda * 8;
    const int voff = lane * 16;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn;
    const int pl = fn & 7, bodd = (fn >> 3) & 1;
    const int offb_tr = (wc * 8 + (fn >> 3)) * 128 + (((pl ^ bodd) * 8) + ((fk >> 1) ^ (pl & 1))) * 2 + (fk & 1);
    int atr[4];
#pragma unroll
    for (int g = 0; g < 4; g++) atr[g] = offb_tr + (((2 * g) ^ (pl & 6))) * 2;
    int voff_tr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int blk = wave * 4 + j;
        const int pl_s = (lane >> 3) ^ (blk & 1);
        const int kp = (lane & 7) ^ pl_s;
        voff_tr[j] = (int)((((long)(blk * 8 + pl_s)) * ldb + 2 * kp) * 8);
    }
    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *da, double *db, int j) {
        const int k = wave * 4 + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (__attribute__((address_space(3))) void *)(da + k * LDN), 16, voff, (k0 + k) * lda8, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_b, (__attribute__((address_space(3))) void *)(db + k * 128), 16, voff_tr[j], k0 * 8, 0, 0);
    };
    auto step = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < kend) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = ca[offa + kk * LDN + a * 16];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cb[atr[kk >> 2] + b * 256];
            stage_row(kn, na, nb, kk >> 2);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    if (kend > 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) stage_row(0, sa0, sq0, j);
    }
    for (int k0 = 0; k0 < kend; k0 += 2 * KB) {
        step(sa0, sq0, sa1, sq1, k0);
        if (k0 + KB < kend) step(sa1, sq1, sa0, sq0, k0 + KB);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const long col = n0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + wr * 64 + a * 16 + fk + 4 * r;
                if (row < m) C[(long)row * ldc + col] = acc[a][b][r];
            }
        }
}

// ---- integral-direct J (pyscf/df/df_jk.py:415-506 get_j): contract a freshly generated slab
// T[row][Q] = (pq|Q) in place, without ever forming cderi.
constexpr int DJ_ROWS = 512;
// part[chunk][Q] = sum_{r in chunk} T[r][Q] * d[r]
__global__ __launch_bounds__(256) void vj_direct_pass1_kernel(const double *__restrict__ T, long ldT, long nrows,
                                                              int naux, const double *__restrict__ d,
                                                              double *__restrict__ part)
{
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= naux) return;
    const long r0 = (long)blockIdx.y * DJ_ROWS;
    const long r1 = (r0 + DJ_ROWS < nrows) ? r0 + DJ_ROWS : nrows;
    double acc = 0;
    for (long r = r0; r < r1; r++) acc += T[r * ldT + q] * d[r];
    part[(long)blockIdx.y * naux + q] = acc;
}
// out[r] = sum_Q T[r][Q] * rho[Q]   (one wave per row)
__global__ __launch_bounds__(256) void vj_direct_pass2_kernel(const double *__restrict__ T, long ldT, long nrows,
                                                              int naux, const double *__restrict__ rho,
                                                              double *__restrict__ out)
{
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= nrows) return;
    const int lane = threadIdx.x & 63;
    double acc = 0;
    for (int q = lane; q < naux; q += 64) acc += T[r * ldT + q] * rho[q];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) out[r] = acc;
}

}  // namespace

extern "C" {

long PAMD_vj_direct_pass1_worksize(long nrows, int naux) { return (long)ceil_div(nrows, DJ_ROWS) * naux; }

// d_part[nchunk][naux] partial sums (nchunk = ceil(nrows/512)); the caller adds them up (fixed order)
int PAMD_vj_direct_pass1(const double *d_T, long ldT, long nrows, int naux, const double *d_dmtril_rows,
                         double *d_part, void *stream)
{
    if (nrows == 0 || naux == 0) return 0;
    dim3 grid(ceil_div(naux, 256), ceil_div(nrows, DJ_ROWS));
    vj_direct_pass1_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_T, ldT, nrows, naux, d_dmtril_rows, d_part);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_vj_direct_pass2(const double *d_T, long ldT, long nrows, int naux, const double *d_rho, double *d_vj_rows,
                         void *stream)
{
    if (nrows == 0 || naux == 0) return 0;
    vj_direct_pass2_kernel<<<ceil_div(nrows, 4), 256, 0, (hipStream_t)stream>>>(d_T, ldT, nrows, naux, d_rho,
                                                                               d_vj_rows);
    PAMD_CHECK_LAUNCH();
    return 0;
}


long PAMD_rys_table_len(void) { return RYS_TABLE_LEN; }

// Host copy of the table and its layout (for the CPU-side accuracy test of the tables).
int PAMD_rys_table_host(double *h_dst, int *offsets /*[RYS_NMAX+1]*/, int *nint /*[RYS_NMAX+1]*/,
                        double *herm_u /*[(NMAX+1)*NMAX]*/, double *herm_w)
{
    memcpy(h_dst, RYS_TABLE, sizeof(double) * RYS_TABLE_LEN);
    for (int i = 0; i <= RYS_NMAX; i++) { offsets[i] = RYS_OFFSET[i]; nint[i] = RYS_NINT[i]; }
    memcpy(herm_u, RYS_HERM_U, sizeof(RYS_HERM_U));
    memcpy(herm_w, RYS_HERM_W, sizeof(RYS_HERM_W));
    return RYS_NMAX;
}

int PAMD_rys_table_upload(double *d_dst, void *stream)
{
    PAMD_CHECK_HIP(hipMemcpyAsync(d_dst, RYS_TABLE, sizeof(double) * RYS_TABLE_LEN, hipMemcpyHostToDevice,
                                  (hipStream_t)stream));
    PAMD_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int PAMD_int3c2e_class(int li, int lj, int lk, const PAMD_int3c2e_args *args, void *stream)
{
    PAMD_REQUIRE(li >= lj, "int3c2e class needs l_i >= l_j");
    const Int3c2eArgs &a = *reinterpret_cast<const Int3c2eArgs *>(args);
    hipStream_t st = (hipStream_t)stream;
    switch (lk) {
    case 0: return launch_int3c2e_lk0(li, lj, a, st);
    case 1: return launch_int3c2e_lk1(li, lj, a, st);
    case 2: return launch_int3c2e_lk2(li, lj, a, st);
    case 3: return launch_int3c2e_lk3(li, lj, a, st);
    case 4: return launch_int3c2e_lk4(li, lj, a, st);
    case 5: return launch_int3c2e_lk5(li, lj, a, st);
    case 6: return launch_int3c2e_lk6(li, lj, a, st);
    default: return set_error(-2, "int3c2e: aux angular momentum > 6 unsupported", __FILE__, __LINE__);
    }
}

// grad[rep][atom][3] += sum_{pq,Q} Z[pq][Q] d(pq|Q)/dR_atom for one angular class: the contraction that
// pyscf/df/grad/rhf.py:117-199 performs with int3c2e_ip1 / int3c2e_ip2 / int2c2e_ip1 blocks, and (with
// point-charge aux shells) the int1e_ipnuc / int1e_iprinv part of pyscf/grad/rhf.py:91-146 (hcore_generator).
int PAMD_int3c2e_grad_class(int li, int lj, int lk, const PAMD_int3c2e_grad_args *args, void *stream)
{
    PAMD_REQUIRE(li >= lj, "int3c2e_grad class needs l_i >= l_j");
    PAMD_REQUIRE(args->nrep >= 1 && args->natm >= 1, "int3c2e_grad: nrep, natm >= 1");
    const Int3c2eGradArgs &a = *reinterpret_cast<const Int3c2eGradArgs *>(args);
    hipStream_t st = (hipStream_t)stream;
    switch (lk) {
    case 0: return launch_int3c2e_grad_lk0(li, lj, a, st);
    case 1: return launch_int3c2e_grad_lk1(li, lj, a, st);
    case 2: return launch_int3c2e_grad_lk2(li, lj, a, st);
    case 3: return launch_int3c2e_grad_lk3(li, lj, a, st);
    case 4: return launch_int3c2e_grad_lk4(li, lj, a, st);
    case 5: return launch_int3c2e_grad_lk5(li, lj, a, st);
    case 6: return launch_int3c2e_grad_lk6(li, lj, a, st);
    default: return set_error(-2, "int3c2e_grad: aux angular momentum > 6 unsupported", __FILE__, __LINE__);
    }
}

int PAMD_cderi_solve(const double *d_linvT, int lda, const double *d_T, long ldT, double *d_cderi,
                     long ldc, int nL, long npq, int naux, int l_off, int triangular, void *stream)
{
    if (nL == 0 || npq == 0) return 0;
    dim3 grid(ceil_div(npq, NT), ceil_div(nL, NT));
    // r06: all-DMA kernel when the operands allow 16-byte loads (even leading dimensions, aligned bases) and the 32-bit DMA offsets
    // cover a k range of A / 128 rows of T; PAMD_SOLVE_V2=0 in the environment keeps the r01 kernel (A/B runs)
    static const int solve_v2 = [] { const char *e = getenv("PAMD_SOLVE_V2"); return (e && e[0] == '0') ? 0 : 1; }();
    if (solve_v2 && lda % 2 == 0 && ldT % 2 == 0 && ((uintptr_t)d_linvT % 16 == 0) && ((uintptr_t)d_T % 16 == 0) &&
        (long)naux * lda * 8 < (1L << 31) && 128L * ldT * 8 < (1L << 31)) {
        cderi_solve2_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_linvT, lda, d_T, ldT, d_cderi, ldc, nL, npq, naux, l_off, triangular);
        PAMD_CHECK_LAUNCH();
        return 0;
    }
    cderi_solve_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_linvT, lda, d_T, ldT, d_cderi, ldc, nL, npq,
                                                               naux, l_off, triangular);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
