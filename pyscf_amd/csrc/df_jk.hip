// Density-fitted J/K contraction kernels for gfx950 (MI355X).
//
// Reference algorithm: pyscf/df/df_jk.py:280-413 (get_jk) and the C half-transform it
// calls, pyscf/lib/ao2mo/nr_ao2mo.c:399-419 (AO2MOmmm_bra_nr_s2 = dsymm),
// :1016-1031 (AO2MOtranse2_nr_s2 = unpack_tril one aux row), :1240-1266 (AO2MOnr_e2_drv).
//
// Data layout in HBM (identical to the reference's `_cderi`, pyscf/df/df.py:59-72):
//   cderi[L][pq]   L in [0,naux_local), pq = p(p+1)/2+q (p>=q), row-major, f64.
//
// Kernels
//   vj_pass1   rho[s][L]   = sum_pq cderi[L][pq] * dmtril[s][pq]      (HBM-bound row dots)
//   vj_pass2   vj[s][pq]  += sum_L  rho[s][L]    * cderi[L][pq]       (HBM-bound column axpy)
//   e2_symm    X[L][i][p]  = sum_q  Bsym_L[p][q] * orb[q][i]          (FP64 MFMA, reads the
//              packed row directly: the symmetric unpack is fused into the LDS fill)
//   gemm_tn    C[m][n]    += sum_k  A[k][m] * B[k][n]                 (FP64 MFMA; lower-tri
//              tiles only for the K = X^T X  SYRK)
#include <algorithm>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>
#include "common.h"
#include "mfma_e2.h"

using namespace pamd;

namespace {

// ------------------------------------------------------------------------------------ J
constexpr int J1_THREADS = 256;
constexpr int MAX_NSET = 4;

// v2: R aux rows per workgroup share one register copy of the dmtril chunk, so the L2 traffic
// for dmtril drops R-fold and the kernel streams cderi at the HBM rate.
constexpr int J1R = 16;            // aux rows per workgroup
constexpr int J1E = 8;             // elements per thread per row
constexpr int J1_CHUNK2 = J1_THREADS * J1E;

template <int NSET>
__global__ __launch_bounds__(J1_THREADS) void vj_pass1_rows_kernel(
    const double *__restrict__ cderi, long npair, int naux, const double *__restrict__ dmtril,
    double *__restrict__ partial, int nchunk)
{
    const int chunk = blockIdx.x;
    const int L0 = blockIdx.y * J1R;
    const long base = (long)chunk * J1_CHUNK2 + threadIdx.x;
    double d[NSET][J1E];
#pragma unroll
    for (int s = 0; s < NSET; s++)
#pragma unroll
        for (int e = 0; e < J1E; e++) {
            long i = base + e * J1_THREADS;
            d[s][e] = (i < npair) ? dmtril[(long)s * npair + i] : 0.0;
        }
    __shared__ double red[J1R][NSET][J1_THREADS / 64];
    const int nrow = (naux - L0 < J1R) ? naux - L0 : J1R;
    for (int r = 0; r < nrow; r++) {
        const double *row = cderi + (long)(L0 + r) * npair;
        double b[J1E];
#pragma unroll
        for (int e = 0; e < J1E; e++) {
            long i = base + e * J1_THREADS;
            b[e] = (i < npair) ? __builtin_nontemporal_load(row + i) : 0.0;
        }
#pragma unroll
        for (int s = 0; s < NSET; s++) {
            double v = 0;
#pragma unroll
            for (int e = 0; e < J1E; e++) v += b[e] * d[s][e];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if ((threadIdx.x & 63) == 0) red[r][s][threadIdx.x >> 6] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < nrow * NSET) {
        const int r = threadIdx.x / NSET, s = threadIdx.x - r * NSET;
        double v = 0;
        for (int w = 0; w < J1_THREADS / 64; w++) v += red[r][s][w];
        partial[((long)s * naux + L0 + r) * nchunk + chunk] = v;
    }
}

template <bool ACCUMULATE>
__global__ void vj_pass1_reduce_kernel(const double *__restrict__ partial, double *__restrict__ rho,
                                       int n, int nchunk)
{
    // one wave per output element (fixed summation order -> deterministic)
    int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);   // over nset*naux
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    double v = 0;
    for (int c = lane; c < nchunk; c += 64) v += partial[(long)i * nchunk + c];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) rho[i] = ACCUMULATE ? rho[i] + v : v;
}

template <int NSET>
__global__ __launch_bounds__(256) void vj_pass2_kernel(
    const double *__restrict__ cderi, long npair, int naux, const double *__restrict__ rho,
    double *__restrict__ vj)
{
    // grid-stride over the packed columns: with a capped grid (tuning key "j2wg") the pass becomes a thin background
    // stream beside the MFMA-bound SYRK instead of a burst that takes its issue slots and memory queue
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npair; i += (long)gridDim.x * 256) {
        double acc[NSET];
#pragma unroll
        for (int s = 0; s < NSET; s++) acc[s] = 0;
        const double *col = cderi + i;
        int L = 0;
        for (; L + 8 <= naux; L += 8) {
            double b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) b[u] = __builtin_nontemporal_load(col + (long)(L + u) * npair);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int s = 0; s < NSET; s++) acc[s] += rho[s * naux + L + u] * b[u];
        }
        for (; L < naux; L++) {
            double b = col[(long)L * npair];
#pragma unroll
            for (int s = 0; s < NSET; s++) acc[s] += rho[s * naux + L] * b;
        }
#pragma unroll
        for (int s = 0; s < NSET; s++) vj[(long)s * npair + i] += acc[s];
    }
}

// The same pass with 16-byte loads: every lane owns TWO adjacent packed columns (npair even, 16-byte aligned rows), so a wave
// moves 1 KiB per load instruction and issues half as many loads and half as many scalar rho reads per byte - fewer issue slots
// taken from the MFMA kernel it runs beside (r03; tuning key "j2wide").
template <int NSET>
__global__ __launch_bounds__(256) void vj_pass2_wide_kernel(
    const double *__restrict__ cderi, long npair, int naux, const double *__restrict__ rho,
    double *__restrict__ vj)
{
    const long npair2 = npair >> 1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npair2; i += (long)gridDim.x * 256) {
        double2_t acc[NSET];
#pragma unroll
        for (int s = 0; s < NSET; s++) acc[s] = double2_t{0, 0};
        const double2_t *col = reinterpret_cast<const double2_t *>(cderi) + i;
        int L = 0;
        for (; L + 8 <= naux; L += 8) {
            double2_t b[8];
#pragma unroll
            for (int u = 0; u < 8; u++) b[u] = __builtin_nontemporal_load(col + (long)(L + u) * npair2);
#pragma unroll
            for (int u = 0; u < 8; u++)
#pragma unroll
                for (int s = 0; s < NSET; s++) {
                    const double r = rho[s * naux + L + u];
                    acc[s][0] += r * b[u][0];
                    acc[s][1] += r * b[u][1];
                }
        }
        for (; L < naux; L++) {
            const double2_t b = col[(long)L * npair2];
#pragma unroll
            for (int s = 0; s < NSET; s++) {
                const double r = rho[s * naux + L];
                acc[s][0] += r * b[0];
                acc[s][1] += r * b[1];
            }
        }
#pragma unroll
        for (int s = 0; s < NSET; s++) {
            vj[(long)s * npair + 2 * i] += acc[s][0];
            vj[(long)s * npair + 2 * i + 1] += acc[s][1];
        }
    }
}

// ------------------------------------------------------------------------------------ r06: J on the SQUARE layout
// sq[L][p][q] (rows x ld doubles per aux row, both triangles, pads zero) is the ONLY copy of the tensor when DF.layout = 'square'
// (2x the packed size in HBM instead of packed + image = 3x).  The J passes read the p >= q run of every
// square row - the same 8 nao_pair bytes per aux row as the packed layout, in runs of (p + 1) doubles that start on a 128-byte
// boundary (ld % 16 == 0) - and keep the PACKED density / result vectors (pyscf/df/df_jk.py:329-337,367).
__device__ __forceinline__ void tril_pq(long i, int &p, int &q)
{
    p = (int)((sqrt(8.0 * (double)i + 1.0) - 1.0) * 0.5);
    while ((long)(p + 1) * (p + 2) / 2 <= i) p++;
    while ((long)p * (p + 1) / 2 > i) p--;
    q = (int)(i - (long)p * (p + 1) / 2);
}

template <int NSET>
__global__ __launch_bounds__(J1_THREADS) void vj_pass1_sq_kernel(
    const double *__restrict__ sq, long lstride, int ld, long npair, int naux, const double *__restrict__ dmtril,
    double *__restrict__ partial, int nchunk)
{
    const int chunk = blockIdx.x;
    const int L0 = blockIdx.y * J1R;
    const long base = (long)chunk * J1_CHUNK2 + threadIdx.x;
    double d[NSET][J1E];
    long off[J1E];
#pragma unroll
    for (int e = 0; e < J1E; e++) {
        const long i = base + e * J1_THREADS;
        int p = 0, q = 0;
        if (i < npair) tril_pq(i, p, q);
        off[e] = (i < npair) ? (long)p * ld + q : -1;
#pragma unroll
        for (int s = 0; s < NSET; s++) d[s][e] = (i < npair) ? dmtril[(long)s * npair + i] : 0.0;
    }
    __shared__ double red[J1R][NSET][J1_THREADS / 64];
    const int nrow = (naux - L0 < J1R) ? naux - L0 : J1R;
    for (int r = 0; r < nrow; r++) {
        const double *row = sq + (long)(L0 + r) * lstride;
        double b[J1E];
#pragma unroll
        for (int e = 0; e < J1E; e++) b[e] = (off[e] >= 0) ? __builtin_nontemporal_load(row + off[e]) : 0.0;
#pragma unroll
        for (int s = 0; s < NSET; s++) {
            double v = 0;
#pragma unroll
            for (int e = 0; e < J1E; e++) v += b[e] * d[s][e];
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            if ((threadIdx.x & 63) == 0) red[r][s][threadIdx.x >> 6] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x < nrow * NSET) {
        const int r = threadIdx.x / NSET, s = threadIdx.x - r * NSET;
        double v = 0;
        for (int w = 0; w < J1_THREADS / 64; w++) v += red[r][s][w];
        partial[((long)s * naux + L0 + r) * nchunk + chunk] = v;
    }
}

// Work units of the square second pass: (row p, 256-column chunk c) with c * 256 <= p - every wave then reads 64 consecutive doubles
// that START ON A 512-BYTE BOUNDARY of its row (rows start on 128-byte boundaries: ld % 16 == 0).  The first version walked the flat
// packed index space, so a wave's 512 bytes began anywhere inside the row: 9 sectors of 64 bytes instead of 8, and with the
// non-temporal loads the neighbour fetched the shared sector again - FETCH_SIZE 74.3 GB for 61.3 GB of operand (+21 %,
// profiles/r06/pmc_summary.json) and a pass that cost the co-running SYRK more than the packed pass does (whose 2 KiB chunks are
// sector-aligned by construction).  Rows 256 g .. 256 g + 255 have g + 1 chunks each: unit u -> (g, p, c) in closed form.
__device__ __forceinline__ void sq_unit(long u, int &p, int &c)
{
    int g = (int)((sqrt(1.0 + (double)u / 32.0) - 1.0) * 0.5);          // 128 g (g + 1) <= u
    while (128L * (g + 1) * (g + 2) <= u) g++;
    while (128L * g * (g + 1) > u) g--;
    const int r = (int)(u - 128L * g * (g + 1));
    p = 256 * g + r / (g + 1);
    c = r - (r / (g + 1)) * (g + 1);
}
__host__ __device__ inline long sq_units(int nao)
{
    const long g = nao / 256, rem = nao - 256 * g;                       // full groups of 256 rows, then `rem` rows with g + 1 chunks
    return 128L * g * (g + 1) + rem * (g + 1);
}

template <int NSET>
__global__ __launch_bounds__(256) void vj_pass2_sq_kernel(
    const double *__restrict__ sq, long lstride, int ld, int nao, int naux, const double *__restrict__ rho,
    double *__restrict__ vj)
{
    const long nunit = sq_units(nao);
    const long npair = (long)nao * (nao + 1) / 2;
    for (long u = blockIdx.x; u < nunit; u += gridDim.x) {
        int p, c;
        sq_unit(u, p, c);
        const int q = c * 256 + threadIdx.x;
        if (q > p) continue;                                            // (idle lanes of a row's last chunk: no memory traffic)
        double acc[NSET];
#pragma unroll
        for (int s = 0; s < NSET; s++) acc[s] = 0;
        const double *col = sq + (long)p * ld + q;
        int L = 0;
        for (; L + 8 <= naux; L += 8) {
            double b[8];
#pragma unroll
            for (int k = 0; k < 8; k++) b[k] = __builtin_nontemporal_load(col + (long)(L + k) * lstride);
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int s = 0; s < NSET; s++) acc[s] += rho[s * naux + L + k] * b[k];
        }
        for (; L < naux; L++) {
            const double b = col[(long)L * lstride];
#pragma unroll
            for (int s = 0; s < NSET; s++) acc[s] += rho[s * naux + L] * b;
        }
        const long i = (long)p * (p + 1) / 2 + q;
#pragma unroll
        for (int s = 0; s < NSET; s++) vj[(long)s * npair + i] += acc[s];
    }
}

// sq[L][p][q] = sq[L][q][p] = slab[L][p (p + 1) / 2 + q - r0] for the AO rows p in [p0, p1) of one column slab of the build
// (df/incore.py:189-217 solves the tensor column slab by column slab): 32 x 32 tiles through LDS so that both the direct rows and
// the mirrored column runs are written coalesced.  grid (ceil(p1 / 32) q-tiles, ceil((p1 - p0) / 32) p-tiles, nL)
__global__ __launch_bounds__(256) void unpack_slab_kernel(const double *__restrict__ slab, long ncol, long r0, int p0, int p1,
                                                          double *__restrict__ sq, int ld, long lstride)
{
    __shared__ double t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int pb = p0 + blockIdx.y * 32, qb = blockIdx.x * 32;
    if (qb > pb + 31) return;                                    // the whole tile lies above the diagonal
    const long L = blockIdx.z;
    const double *src = slab + L * ncol;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = pb + ty + 8 * k, q = qb + tx;
        t[ty + 8 * k][tx] = (p < p1 && q <= p) ? src[(long)p * (p + 1) / 2 + q - r0] : 0.0;
    }
    __syncthreads();
    double *dst = sq + L * lstride;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int p = pb + ty + 8 * k, q = qb + tx;
        if (p < p1 && q <= p) dst[(long)p * ld + q] = t[ty + 8 * k][tx];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = qb + ty + 8 * k, p = pb + tx;
        if (p < p1 && q < p) dst[(long)q * ld + p] = t[tx][ty + 8 * k];
    }
}

// tril[L][p (p + 1) / 2 + q] = sq[L][p][q]: the reference's `_cderi` rows (pyscf/df/df.py:59-72) out of the square layout, for
// DF.loop / export / the consumers that want the packed operand
__global__ __launch_bounds__(256) void pack_rows_kernel(const double *__restrict__ sq, long lstride, int ld, int nao,
                                                        double *__restrict__ tril, long npair)
{
    const int p = blockIdx.y;
    const long L = blockIdx.z;
    for (int q = blockIdx.x * 256 + threadIdx.x; q <= p; q += gridDim.x * 256)
        tril[L * npair + (long)p * (p + 1) / 2 + q] = sq[L * lstride + (long)p * ld + q];
}

// ------------------------------------------------------------------------------------ K
// X[L][i][p] = sum_q Bsym_L[q][p] * orb[q][i]: grid x = p tile (128 cols), y = L, z = chunk of MT*16 orbitals
// (body and operand conventions: mfma_e2.h)
template <int MT, bool PLAIN>
__global__ __launch_bounds__(256, 2) void e2_symm_kernel(
    const double *__restrict__ cderi, long npair, int nao, const double *__restrict__ orb, int ldo,
    double *__restrict__ X, int nocc_pad, long ldx, long src_stride, long ncols,
    const unsigned char *__restrict__ kmask, double *__restrict__ rho)
{
    const int L = blockIdx.y;
    const double *row = PLAIN ? cderi + (long)L * src_stride : cderi + (long)L * npair;
    const unsigned char *km = kmask ? kmask + ((long)L * gridDim.x + blockIdx.x) * ((nao + KB - 1) / KB) : nullptr;
    double *slot = rho ? rho + ((long)L * gridDim.z * gridDim.x + (long)blockIdx.z * gridDim.x + blockIdx.x) * 4 : nullptr;
    e2_symm_body<MT, PLAIN, false>(row, npair, nao, orb, ldo, X + (long)L * nocc_pad * ldx, nocc_pad, ldx, ncols, km, slot,
                                   nullptr, blockIdx.x * NT, blockIdx.z * (MT * 16));
}

// C[split][m][n] += sum_{k in split range} A[k][m] * B[k][n]
//   128x128 tile per workgroup, 64x64 per wave (4x4 MFMA tiles).
//   grid: x = tile id (all tiles, or lower-triangular tiles when lower_only), y = k split.
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(
    const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
    double *__restrict__ C, int ldc, int m, int n, long kdim, int lower_only, int ntile_n)
{
    __shared__ double sP[KB * LDN];
    __shared__ double sQ[KB * LDN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tm, tn;
    if (lower_only) {
        // tile id -> (tm >= tn)
        int t = blockIdx.x;
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
        while (tm * (tm + 1) / 2 > t) tm--;
        tn = t - tm * (tm + 1) / 2;
    } else {
        tm = blockIdx.x / ntile_n;
        tn = blockIdx.x - tm * ntile_n;
    }
    const int p0 = tm * NT, q0 = tn * NT;
    const int nsplit = gridDim.y;
    const long kchunk = ((kdim + nsplit - 1) / nsplit + KB - 1) / KB * KB;
    const long kbeg = (long)blockIdx.y * kchunk;
    const long kend = (kbeg + kchunk < kdim) ? kbeg + kchunk : kdim;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    // each thread stages 8 doubles of each panel: row = e/128, col = e%128
    double pa[8], pb[8];
    auto fetch = [&](long k0) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int e = tid + j * 256;
            int k = e >> 7, c = e & 127;
            long kk = k0 + k;
            pa[j] = (kk < kend && p0 + c < m) ? A[kk * lda + p0 + c] : 0.0;
            pb[j] = (kk < kend && q0 + c < n) ? B[kk * ldb + q0 + c] : 0.0;
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (long k0 = kbeg; k0 < kend; k0 += KB) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int e = tid + j * 256;
            int k = e >> 7, c = e & 127;
            sP[k * LDN + c] = pa[j];
            sQ[k * LDN + c] = pb[j];
        }
        __syncthreads();
        if (k0 + KB < kend) fetch(k0 + KB);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = sP[(kk + fk) * LDN + wr * 64 + a * 16 + fn];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = sQ[(kk + fk) * LDN + wc * 64 + b * 16 + fn];
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
        __syncthreads();
    }
    double *out = C + (long)blockIdx.y * m * ldc;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int col = q0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int rowi = p0 + wr * 64 + a * 16 + fk + 4 * r;
                if (rowi < m) unsafeAtomicAdd(out + (long)rowi * ldc + col, acc[a][b][r]);   // no-return FP64 atomic: nothing to wait for
            }
        }
}

// LDS-DMA variant of gemm_tn: both panels are streamed HBM/L2 -> LDS with global_load_lds_dwordx4
// (one 1 KiB wave-instruction = one 128-double panel row), double-buffered, one barrier per
// k-tile, no staging VGPRs.  Requirements (checked by the launcher): lda, ldb even, 16-byte
// aligned bases, every k range a multiple of KB, and 128 readable doubles from any row start
// (edge tiles read past m/n inside the allocation; those columns are never stored).
constexpr int KLMAX = 2048;             // k-tiles per compaction segment of the screened LDS-DMA GEMM
// WA = MFMA tiles per wave row: workgroup tile (32 WA) x 128; WA = 4 is the square 128 x 128 tile (the only shape of the
// lower-triangular SYRK mode), WA = 5 gives 160 x 128 and 80 instead of 64 MFMAs per wave between barriers.
template <int WA>
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(
    const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
    double *__restrict__ C, int ldc, int m, int n, long kdim, int lower_only, int ntile_n,
    const unsigned char *__restrict__ maskA, const unsigned char *__restrict__ maskB, int ntile_m, int nsplit)
{
    // grid (tiles, splits), tile index fastest: the workgroups of one k-split start together on all 8 XCDs and walk
    // the same panel rows in lockstep, so each XCD's L2 serves its tiles from one fetch per row.  (Tried in r01: split
    // index fastest with nsplit = 8, one k-range per XCD - no gain for the SYRK, 15 % slower for the vmat GEMM: the
    // tiles of a split then start at different times and the L2 cannot hold a k-range.)
    const int bsplit = blockIdx.y, btile = blockIdx.x;
    // two separate LDS objects (not one [2][..] array): the compiler can then prove that the LDS-DMA writes of
    // the next tile do not alias the ds_reads of the current one and leaves the DMA in flight during the MFMAs
    constexpr int TM = 32 * WA;
    constexpr int LDM = TM + ((TM % 32 == 16) ? 0 : 16);
    constexpr int NFULL = TM / 128, REM = TM % 128;
    constexpr int PA = KB * LDM;                 // doubles of the A panel; the B panel [k][LDN] follows it
    __shared__ double sb0[PA + KB * LDN];
    __shared__ double sb1[PA + KB * LDN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int tm, tn;
    if (lower_only) {
        int t = btile;
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
        while (tm * (tm + 1) / 2 > t) tm--;
        tn = t - tm * (tm + 1) / 2;
    } else {
        tm = btile / ntile_n;
        tn = btile - tm * ntile_n;
    }
    const int p0 = tm * TM, q0 = tn * NT;
    const long kchunk = ((kdim + nsplit - 1) / nsplit + KB - 1) / KB * KB;
    const long kbeg = (long)bsplit * kchunk;
    const long kend = (kbeg + kchunk < kdim) ? kbeg + kchunk : kdim;

    double4_t acc[WA][4];
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;

    // wave w stages rows 4w..4w+3 of each panel: lane -> 2 doubles (16 B) of the row
    auto stage = [&](long k0, double *dst) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = wave * 4 + j;
            const double *ga = A + (k0 + k) * lda + p0 + lane * 2;
            const double *gb = B + (k0 + k) * ldb + q0 + lane * 2;
#pragma unroll
            for (int pc = 0; pc < NFULL; pc++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga + pc * 128),
                                                 (__attribute__((address_space(3))) void *)(dst + k * LDM + pc * 128), 16, 0, 0);
            if (REM > 0 && lane * 2 < REM)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga + NFULL * 128),
                                                 (__attribute__((address_space(3))) void *)(dst + k * LDM + NFULL * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gb,
                                             (__attribute__((address_space(3))) void *)(dst + PA + k * LDN), 16, 0, 0);
        }
    };
    // optional screening: k-tile kt is skipped when either 16 x 128 panel tile is negligible
    // (maskA[kt][tm], maskB[kt][tn]; VXCdot_ao_ao_sparse's pair_mask idea, nr_numint_sparse.c:890-973).
    // The surviving k-tiles of a segment are compacted (in order) into an LDS list first so the
    // pipelined loop never waits on a mask byte.
    // (the 160-row instance is never screened: its LDS image is exactly half a CU's 160 KiB, the list would not fit)
    constexpr bool MASKED = (WA == 4);
    unsigned short *klist = nullptr;
    int *s_cnt = nullptr;
    if constexpr (MASKED) {
        __shared__ unsigned short klist_s[KLMAX];
        __shared__ int s_cnt_s[4];
        klist = klist_s;
        s_cnt = s_cnt_s;
    }
    const long seglen = (MASKED && maskA) ? (long)KLMAX * KB : (kend > kbeg ? kend - kbeg : 1);
    for (long seg = kbeg; seg < kend; seg += seglen) {
        const long segend = (seg + seglen < kend) ? seg + seglen : kend;
        int nact = (int)((segend - seg) / KB);
        if (MASKED && maskA) {
            const int nt = nact;
            const long t0g = seg / KB;
            int total = 0;
            __syncthreads();
            for (int t0 = 0; t0 < nt; t0 += 256) {
                const int t = t0 + tid;
                const bool act = t < nt && maskA[(t0g + t) * ntile_m + tm] && maskB[(t0g + t) * ntile_n + tn];
                const unsigned long long b = __ballot(act);
                if (lane == 0) s_cnt[wave] = __popcll(b);
                __syncthreads();
                int off = total;
                for (int w = 0; w < wave; w++) off += s_cnt[w];
                if (act) klist[off + __popcll(b & ((1ull << lane) - 1))] = (unsigned short)t;
                total += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
                __syncthreads();
            }
            nact = total;
        }
        auto kof = [&](int i) { return seg + (long)((MASKED && maskA) ? klist[i] : i) * KB; };
        auto step = [&](const double *cur, double *nxt, int i) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (i + 1 < nact) stage(kof(i + 1), nxt);
            const double *sP = cur, *sQ = cur + PA;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 4) {
                double af[WA], bf[4];
#pragma unroll
                for (int a = 0; a < WA; a++) af[a] = sP[(kk + fk) * LDM + wr * (WA * 16) + a * 16 + fn];
#pragma unroll
                for (int b = 0; b < 4; b++) bf[b] = sQ[(kk + fk) * LDN + wc * 64 + b * 16 + fn];
#pragma unroll
                for (int a = 0; a < WA; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
            }
        };
        if (nact > 0) stage(kof(0), sb0);
        for (int i = 0; i < nact; i += 2) {
            step(sb0, sb1, i);
            if (i + 1 < nact) step(sb1, sb0, i + 1);
        }
    }
    double *out = C + (long)bsplit * m * ldc;
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int col = q0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int rowi = p0 + wr * (WA * 16) + a * 16 + fk + 4 * r;
                if (rowi < m) unsafeAtomicAdd(out + (long)rowi * ldc + col, acc[a][b][r]);   // no-return FP64 atomic: nothing to wait for
            }
        }
}

// Half transform on the *square* image of the tensor: X[L][i][p] = sum_q orb[q][i] * sq[L][q][p].
// With 288 GB of HBM the K path can afford a second, unpacked copy of cderi (2x the packed size, written once at
// build time by unpack_tril): the symmetric unpack leaves the hot loop and both operands become plain row panels
// that stream HBM/L2 -> LDS by LDS-DMA exactly as in gemm_tn_glds (no staging VGPRs, no ds_write, one barrier per
// k-tile, DMA of tile t+1 in flight under the MFMAs of tile t).  Workgroup tile: M = 32 WA orbitals x 128 AOs,
// waves 2 x 2, each WA x 4 MFMA tiles.  Requirements (checked by the launcher): q rows padded to a multiple of
// KB (zero orbital rows), ld and ldo even, 16-byte aligned bases, 128 readable doubles from any row start.
// RHO = true additionally accumulates rho[L] += sum_{i,p} X[L][i][p] * orb[p][i] in the epilogue.  With the density
// D = orb orb^T of the MO branch this is rho_L = sum_pq B_L[p][q] D[p][q], the first J pass (pyscf/df/df_jk.py:367
// `dmtril.dot(eri1.T)`): the half-transformed tile is in the accumulators anyway, so J needs one HBM pass over the
// packed tensor instead of two.
template <int WA, bool RHO>
__global__ __launch_bounds__(256, 2) void e2_sq_kernel(
    const double *__restrict__ sq, long ld, long lstride, int kdim, const double *__restrict__ orb, int ldo,
    double *__restrict__ X, int nocc_pad, long ldx, double *__restrict__ rho, int nchunk)
{
    constexpr int M = 2 * WA * 16;
    constexpr int LDM = M + ((M % 32 == 16) ? 0 : 16);
    constexpr int NFULL = M / 128, REM = M % 128;
    __shared__ double sa0[KB * LDM];
    __shared__ double sa1[KB * LDM];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grid.x = orbital chunk (fastest) + nchunk * AO tile: the workgroups that read the same tensor panel for
    // different orbital chunks start together, so the panel comes from HBM once and from the Infinity Cache after
    const int p0 = (blockIdx.x / nchunk) * NT;
    const long L = blockIdx.y;
    const int m0 = (blockIdx.x % nchunk) * M;
    const double *src_sq = sq + L * lstride + p0 + lane * 2;
    const double *src_orb = orb + m0 + lane * 2;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;

    double4_t acc[WA][4];
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage = [&](int k0, double *da, double *db) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = wave * 4 + j;
            const double *ga = src_orb + (long)(k0 + k) * ldo;
            const double *gb = src_sq + (long)(k0 + k) * ld;
#pragma unroll
            for (int pc = 0; pc < NFULL; pc++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga + pc * 128),
                                                 (__attribute__((address_space(3))) void *)(da + k * LDM + pc * 128), 16, 0, 0);
            if (REM > 0 && lane * 2 < REM)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ga + NFULL * 128),
                                                 (__attribute__((address_space(3))) void *)(da + k * LDM + NFULL * 128), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gb,
                                             (__attribute__((address_space(3))) void *)(db + k * LDN), 16, 0, 0);
        }
    };
    auto step = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k0 + KB < kdim) stage(k0 + KB, na, nb);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[WA], bf[4];
#pragma unroll
            for (int a = 0; a < WA; a++) af[a] = ca[(kk + fk) * LDM + wr * (WA * 16) + a * 16 + fn];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cb[(kk + fk) * LDN + wc * 64 + b * 16 + fn];
#pragma unroll
            for (int a = 0; a < WA; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    stage(0, sa0, sq0);
    for (int k0 = 0; k0 < kdim; k0 += 2 * KB) {
        step(sa0, sq0, sa1, sq1, k0);
        if (k0 + KB < kdim) step(sa1, sq1, sa0, sq0, k0 + KB);
    }
    double *out = X + L * nocc_pad * ldx;
    double rho_acc = 0;
#pragma unroll
    for (int a = 0; a < WA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const long p = p0 + wc * 64 + b * 16 + fn;
            if (p >= ldx) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = m0 + wr * (WA * 16) + a * 16 + fk + 4 * r;
                if (i < nocc_pad) {
                    out[(long)i * ldx + p] = acc[a][b][r];
                    if (RHO) rho_acc += acc[a][b][r] * orb[p * ldo + i];     // rows p >= nao and columns i >= nocc of orb are zero
                }
            }
        }
    if (RHO) {
        // one partial per wave (no atomics): rho[L][workgroup][wave], reduced in a fixed order afterwards
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rho_acc += __shfl_xor(rho_acc, off, 64);
        if (lane == 0) rho[(L * gridDim.x + blockIdx.x) * 4 + wave] = rho_acc;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// v2 of the two LDS-DMA MFMA kernels (r02).  What the ISA of v1 showed: every global_load_lds carried ~9 VALU/SALU
// instructions of 64-bit address arithmetic (per-lane pointers), all 12 (8) of them issued in one burst right after the
// barrier, so each wave spent ~115 issue slots plus an exposed first ds_read latency before its first MFMA of a k-tile.
// v2: (i) the wave index is made uniform (readfirstlane) and the panels are addressed through buffer resources - row
// offset in an SGPR (soffset), lane offset in one loop-invariant VGPR - so a DMA is `s_mov m0; s_add; buffer_load ... lds`,
// no VALU; (ii) the DMA rows of tile t+1 are issued one per k-group BETWEEN the MFMA groups of tile t (the matrix pipe
// has free issue slots there), not in a burst before them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const double *p)
{
    // raw buffer, stride 0, 4 GiB window from p (every panel of one workgroup lies inside: checked by the launchers)
    return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, 0xffffffff, 0x00020000);
}

// reads through this pointer type stay single ds_read_b64 instructions (volatile: hipcc never merges them into ds_read2_b64)
typedef const volatile __attribute__((address_space(3))) double *lds_b64_ptr;

__device__ __forceinline__ void dma_row(__amdgpu_buffer_rsrc_t r, double *lds_dst, int voff_bytes, int soff_bytes)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void *)lds_dst, 16, voff_bytes, soff_bytes, 0, 0);
}

// Half transform on the square image, 160 orbitals x 128 AOs per workgroup (the WA = 5 shape of e2_sq_kernel).
// LDS image of a k-tile: orbital columns 0..127 as [16][144] rows (one full 1 KiB DMA each), orbital columns 128..159 of
// all 16 rows as one contiguous [16][32] block - each wave fetches the remainders of its 4 rows with ONE DMA whose
// per-lane source addresses pick (row, column pair), odd rows rotated by 16 doubles so that the two 16-lane halves of a
// fragment read hit different banks - and the tensor panel as [16][144].  9 DMAs per wave and k-tile (v1: 12, four of
// them exec-masked), no branch inside the k-tile body, so the compiler keeps prefetching the fragments of k-group
// g+1 under the MFMAs of group g.  Wave (wr, wc) owns orbital tiles {64 wr + 16 a, a < 4} and {128 + 16 wr}.
// A wave whose 64 AO columns all lie beyond `ncol` (the 128-column tile overhangs the matrix) only stages its DMA rows:
// its matrix-pipe time goes to the co-resident workgroup.
// PAIRM: 0 = every workgroup takes one aux row and 128 columns; 1 = every workgroup takes the 64 valid columns of the last
// column tile for TWO aux rows (separate launch, tuning "e2merge" = 0); 2 (default, r03) = one launch for both: the rows
// blockIdx.y >= nL of the grid carry the pair workgroups - no second kernel boundary, and their ~2 rounds of workgroups fill up
// the last round of the main ones (2 x 0.5 ms of a 110 ms step).
// blockIdx.x -> (AO column tile, orbital chunk) of the two v2 half-transform kernels when there are several chunks (r04).
// Workgroup b of a launch runs on XCD b % 8, so the plain  x = ptile * nchunk + chunk  puts chunk c of EVERY tile on the XCDs
// of one parity (nchunk = 2): the two chunks of a tile - same tensor panel - never share an L2, and with chunks of different
// cost (wide last chunk) half of the XCDs carry all the cheap ones.  Mapped: groups of 8 column tiles, inside a group
// x = chunk * 8 + tile, so the chunks of one tile are 8 workgroups apart = the same XCD, a few dispatches apart, and every XCD
// sees every chunk kind.  The last group holds what is left of the P tiles.
__device__ __forceinline__ void e2_chunk_map(int x, int nchunk, int P, int xmap, int &ptile, int &chunk)
{
    if (!xmap || nchunk == 1) { ptile = x / nchunk; chunk = x - ptile * nchunk; return; }
    const int gsz = 8 * nchunk, g = x / gsz, j = x - g * gsz;
    int pg = P - 8 * g;
    pg = pg > 8 ? 8 : pg;
    chunk = j / pg;
    ptile = 8 * g + j - chunk * pg;
}

// WM > 0 (r04, NA = 4 only): the LAST orbital chunk holds WM < 8 MFMA tiles and runs in a 1 x 4 wave arrangement - every wave
// takes all WM orbital tiles x 32 AO columns (WM + 2 fragment reads per 2 WM MFMAs), so no wave of the workgroup multiplies
// padding: nocc = 226 (taxol) costs 8 + 7 tiles instead of 2 x 8.  Same LDS image, same DMA schedule, same partial layout.
template <int NA, bool RHO, int PAIRM, int WM = 0>
__global__ __launch_bounds__(256, 2) void e2_sq2_kernel(
    const double *__restrict__ sq, long ld, long lstride, int kdim, const double *__restrict__ orb, int ldo,
    double *__restrict__ X, int nocc_pad, long ldx, double *__restrict__ rho, int nchunk, int ncol, int ptile0, int nslot,
    int nL, int prio, int xmap)
{
    static_assert(NA == 4 || NA == 5, "128- or 160-orbital tile");
    static_assert(WM == 0 || (NA == 4 && WM >= 1 && WM < 8), "wide last chunk: 128-orbital panels");
    if (prio) __builtin_amdgcn_s_setprio(3);             // ahead of the tail of a co-running second J pass (tuning "e2prio")
    constexpr int M = NA * 32;                           // NA = 4: no remainder block
    __shared__ double sa0[KB * LDN + (NA == 5 ? KB * 32 : 0)];
    __shared__ double sa1[KB * LDN + (NA == 5 ? KB * 32 : 0)];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    constexpr int RB = KB * LDN;                         // start of the remainder block inside sa*
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    bool PAIR = PAIRM == 1;
    int ptile, chunk;
    if constexpr (PAIRM == 1) { ptile = ptile0 + blockIdx.x / nchunk; chunk = blockIdx.x % nchunk; }
    else e2_chunk_map(blockIdx.x, nchunk, gridDim.x / nchunk, xmap, ptile, chunk);
    long L = PAIRM == 1 ? 2 * (long)blockIdx.y : blockIdx.y;
    if constexpr (PAIRM == 2) {
        if ((long)blockIdx.y >= nL) {                  // ptile0 = the pair tile; main tiles are 0 .. gridDim.x / nchunk - 1
            const long idx = ((long)blockIdx.y - nL) * gridDim.x + blockIdx.x;
            if (idx >= (long)nchunk * ((nL + 1) / 2)) return;
            PAIR = true;
            ptile = ptile0;
            chunk = (int)(idx % nchunk);
            L = 2 * (idx / nchunk);
        }
    }
    const int p0 = ptile * NT;
    const int m0 = chunk * M;
    const __amdgpu_buffer_rsrc_t r_sq = make_rsrc(sq + L * lstride + p0);
    const __amdgpu_buffer_rsrc_t r_orb = make_rsrc(orb + m0);
    const int ldb8 = (int)ld * 8, ldo8 = ldo * 8;
    const int voff = lane * 16;
    // tensor-panel lane offset: PAIR -> lanes 32..63 fetch the same 64 columns of the next aux row (the last row of an odd
    // block pairs with itself; its second copy is not stored)
    const int voff_b = PAIR ? (lane & 31) * 16 + ((lane >> 5) && L + 1 < nL ? (int)(lstride * 8) : 0) : voff;
    // remainder DMA: lane -> LDS doubles [2 lane, 2 lane + 1] of the wave's 4 x 32 block = row (lane >> 4), rotated column
    const int rrow = lane >> 4;
    const int voff_rem = rrow * ldo8 + (128 + ((((lane & 15) * 2) - 16 * (rrow & 1)) & 31)) * 8;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn;                                   // + kk * LDN + 16 a
    const int offr = RB + fk * 32 + ((wr * 16 + fn + 16 * (fk & 1)) & 31);        // + kk * 32
    const int offb = fk * LDN + wc * 64 + fn;                                   // + kk * LDN + 16 b

    double4_t acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *da, double *db, int j) {
        const int k = wave * 4 + j;
        dma_row(r_orb, da + k * LDN, voff, (k0 + k) * ldo8);
        dma_row(r_sq, db + k * LDN, voff_b, (k0 + k) * ldb8);
        if (NA == 5 && j == 0) dma_row(r_orb, da + RB + wave * 128, voff_rem, (k0 + wave * 4) * ldo8);
    };
    auto step = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // last k-tile: a harmless reload of the current tile into the idle buffer keeps the body branch-free
        const int kn = (k0 + KB < kdim) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[NA], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = ca[offa + kk * LDN + a * 16];
            if constexpr (NA == 5) af[4] = ca[offr + kk * 32];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cb[offb + kk * LDN + b * 16];
            stage_row(kn, na, nb, kk >> 2);
#pragma unroll
            for (int a = 0; a < NA; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    auto step_idle = [&](double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < kdim) ? k0 + KB : k0;
#pragma unroll
        for (int j = 0; j < 4; j++) stage_row(kn, na, nb, j);
    };
    if constexpr (WM > 0) {
        if (chunk == nchunk - 1) {                       // workgroup-uniform
            double4_t wacc[WM][2];
#pragma unroll
            for (int a = 0; a < WM; a++) wacc[a][0] = wacc[a][1] = double4_t{0, 0, 0, 0};
            const int woffa = fk * LDN + fn;                                    // + kk * LDN + 16 a
            const int woffb = fk * LDN + wave * 32 + fn;                        // + kk * LDN + 16 b
            auto wstep = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int kn = (k0 + KB < kdim) ? k0 + KB : k0;
#pragma unroll
                for (int kk = 0; kk < KB; kk += 4) {
                    double af[WM], bf[2];
#pragma unroll
                    for (int a = 0; a < WM; a++) af[a] = ca[woffa + kk * LDN + a * 16];
                    bf[0] = cb[woffb + kk * LDN];
                    bf[1] = cb[woffb + kk * LDN + 16];
                    stage_row(kn, na, nb, kk >> 2);
#pragma unroll
                    for (int a = 0; a < WM; a++) {
                        wacc[a][0] = mfma_f64_16x16x4(af[a], bf[0], wacc[a][0]);
                        wacc[a][1] = mfma_f64_16x16x4(af[a], bf[1], wacc[a][1]);
                    }
                }
            };
#pragma unroll
            for (int j = 0; j < 4; j++) stage_row(0, sa0, sq0, j);
            if (PAIR || p0 + wave * 32 < ncol) {
                for (int k0 = 0; k0 < kdim; k0 += 2 * KB) {
                    wstep(sa0, sq0, sa1, sq1, k0);
                    if (k0 + KB < kdim) wstep(sa1, sq1, sa0, sq0, k0 + KB);
                }
            } else {
                for (int k0 = 0; k0 < kdim; k0 += 2 * KB) {
                    step_idle(sa1, sq1, k0);
                    if (k0 + KB < kdim) step_idle(sa0, sq0, k0 + KB);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const int wrow = wave >> 1;                                         // PAIR: waves 0, 1 -> row L, waves 2, 3 -> row L + 1
            const long Lw = L + (PAIR ? wrow : 0);
            const bool row_ok = Lw < nL;
            double *out = X + Lw * nocc_pad * ldx;
            double rho_acc = 0;
#pragma unroll
            for (int a = 0; a < WM; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const long p = p0 + (PAIR ? (wave & 1) * 32 : wave * 32) + b * 16 + fn;
                    if (p >= ldx || !row_ok) continue;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = m0 + a * 16 + fk + 4 * r;
                        if (i < nocc_pad) {
                            out[(long)i * ldx + p] = wacc[a][b][r];
                            if (RHO) rho_acc += wacc[a][b][r] * orb[p * ldo + i];
                        }
                    }
                }
            if (RHO) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) rho_acc += __shfl_xor(rho_acc, off, 64);
                if (lane == 0) {
                    const long slot = (long)ptile * nchunk + chunk;
                    if (row_ok) rho[(Lw * nslot + slot) * 4 + wave] = rho_acc;
                    if (PAIR) {
                        const long Lo = L + (1 - wrow);
                        if (Lo < nL) rho[(Lo * nslot + slot) * 4 + wave] = 0.0;
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) stage_row(0, sa0, sq0, j);
    if (PAIR || p0 + wc * 64 < ncol) {
        for (int k0 = 0; k0 < kdim; k0 += 2 * KB) {
            step(sa0, sq0, sa1, sq1, k0);
            if (k0 + KB < kdim) step(sa1, sq1, sa0, sq0, k0 + KB);
        }
    } else {
        for (int k0 = 0; k0 < kdim; k0 += 2 * KB) {
            step_idle(sa1, sq1, k0);
            if (k0 + KB < kdim) step_idle(sa0, sq0, k0 + KB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const long Lw = L + (PAIR ? wc : 0);                       // the aux row this wave column belongs to
    const bool row_ok = Lw < nL;
    double *out = X + Lw * nocc_pad * ldx;
    double rho_acc = 0;
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const long p = p0 + (PAIR ? 0 : wc * 64) + b * 16 + fn;
            if (p >= ldx || !row_ok) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = m0 + (a < 4 ? wr * 64 + a * 16 : 128 + wr * 16) + fk + 4 * r;
                if (i < nocc_pad) {
                    out[(long)i * ldx + p] = acc[a][b][r];
                    if (RHO) rho_acc += acc[a][b][r] * orb[p * ldo + i];     // rows p >= nao and columns i >= nocc of orb are zero
                }
            }
        }
    if (RHO) {
        // one partial per wave (no atomics): rho[L][slot][wave], reduced in a fixed order afterwards
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rho_acc += __shfl_xor(rho_acc, off, 64);
        if (lane == 0) {
            const long slot = (long)ptile * nchunk + chunk;
            if (row_ok) rho[(Lw * nslot + slot) * 4 + wave] = rho_acc;
            if (PAIR) {                                            // this wave has nothing for the other row of the pair
                const long Lo = L + (1 - wc);
                if (Lo < nL) rho[(Lo * nslot + slot) * 4 + wave] = 0.0;
            }
        }
    }
}

// Half transform straight from the PACKED rows (no unpacked image), all operands by LDS-DMA: what e2_sq2 does, for ranks
// that cannot spend 2x the tensor size on the square copy.  X[L][i][p] = sum_q B_L[max(p,q)(max+1)/2 + min(p,q)] C[q][i].
// Workgroup = (aux row L, 128 AO columns p0.., 160 orbitals).  The q loop runs over VIRTUAL 16-deep k-tiles:
//   * q-tiles entirely above the diagonal (q0 + 15 < p0): source B_L[p][q], contiguous in q for a fixed p.  One DMA moves
//     eight p-rows x 16 q (1 KiB) with per-lane source addresses; the tile lands transposed in LDS as [p][q-pair] 16-byte
//     chunks, XOR-swizzled (chunk = (p ^ blk) & 7 rows, (q-pair ^ p) columns) so that fragment reads spread over the banks.
//   * q-tiles entirely below (q0 >= p0 + 128): source B_L[q][p], one row-contiguous DMA per q as in e2_sq2.
//   * the up to eight q-tiles that cross the diagonal are visited twice, once per layout, and every fragment element is
//     kept only where it is valid (row layout: q >= p, transposed layout: q < p) by a per-lane select with 0.
// The keep-test  t = q - p  (row: t >= 0, transposed: t < 0) is evaluated for every virtual tile - it is all-true on the pure
// ones - and the DMA operands / fragment addresses of the two layouts are chosen by selects, so the k-tile body stays one
// basic block.  Cost against the square image: +ncross/(kdim/16) MFMA work (6.9 % at nao = 1856), ~90 VALU per tile in the
// matrix pipe's shadow; gain: the tensor is read once (30.7 GB per build instead of 61.3) and 2 x its size of HBM is free.
// Reads beyond a packed row (pad rows q >= nao, pad columns p >= nao of the last tile) stay inside the buffer resource
// (num_records = bytes to the end of the block, out-of-range -> 0) and only meet zero orbital rows or never-stored columns.
// DIAG = true (r03): the 128 x 128 blocks on the diagonal of every B_L come from a small side image diag[L][P][128][128]
// (full symmetric blocks, zero padded; 14 % of the packed size at nao = 1856, PAMD_e2_diag_blocks) in the ROW layout: the
// crossing k-tiles are then visited once, without keep-masks, by the same body as the tiles below the diagonal - two loop
// phases instead of three, 116 instead of 124 k-tiles at nao = 1856.
// WM > 0 (r04, NA = 4): last orbital chunk of WM < 8 tiles in the 1 x 4 wave arrangement, as in e2_sq2_kernel.
template <int NA, bool RHO, bool DIAG, int WM = 0>
__global__ __launch_bounds__(256, 2) void e2_pk_kernel(
    const double *__restrict__ cderi, long npair, int nL, int kdim, const double *__restrict__ orb, int ldo,
    double *__restrict__ X, int nocc_pad, long ldx, double *__restrict__ rho, int nchunk, int nao,
    const double *__restrict__ diag, int ntile_p, int xmap)
{
    static_assert(NA == 4 || NA == 5, "128- or 160-orbital tile");
    static_assert(WM == 0 || (NA == 4 && WM >= 1 && WM < 8), "wide last chunk: 128-orbital panels");
    constexpr int M = NA * 32;
    constexpr int WMe = WM > 0 ? WM : 1;
    __shared__ double sa0[KB * LDN + (NA == 5 ? KB * 32 : 0)];
    __shared__ double sa1[KB * LDN + (NA == 5 ? KB * 32 : 0)];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    constexpr int RB = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ptile, chunk;
    e2_chunk_map(blockIdx.x, nchunk, ntile_p, xmap, ptile, chunk);
    const int p0 = ptile * NT;
    const long L = blockIdx.y;
    const int m0 = chunk * M;
    const long bytes_left = (long)(nL - L) * npair * 8;
    const __amdgpu_buffer_rsrc_t r_pk = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(cderi + L * npair), 0, bytes_left > 0xffffffffL ? 0xffffffff : (unsigned)bytes_left, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_orb = make_rsrc(orb + m0);
    const int ldo8 = ldo * 8;
    const int voff = lane * 16;
    const int rrow = lane >> 4;
    const int voff_rem = rrow * ldo8 + (128 + ((((lane & 15) * 2) - 16 * (rrow & 1)) & 31)) * 8;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn;
    const int offr = RB + fk * 32 + ((wr * 16 + fn + 16 * (fk & 1)) & 31);
    // fragment addresses (doubles) of the two B layouts for b = 0, kk = 0; the b / kk strides differ per layout
    const int offb_row = fk * LDN + wc * 64 + fn;                                          // + kk * LDN + 16 b
    const int pl = fn & 7, bodd = (fn >> 3) & 1;
    const int offb_tr = (wc * 8 + (fn >> 3)) * 128 + (((pl ^ bodd) * 8) + ((fk >> 1) ^ (pl & 1))) * 2 + (fk & 1);   // + k-group term + 256 b
    int atr[4];                                            // transposed-layout fragment address of k-group g (b = 0)
#pragma unroll
    for (int g = 0; g < 4; g++) atr[g] = offb_tr + (((2 * g) ^ (pl & 6))) * 2;
    // transposed-layout DMA j of this wave fills LDS block blk = 4 wave + j: lane -> chunk ci = lane:
    //   row slot prow = ci >> 3 holds p-row  pl_s = prow ^ (blk & 1), column slot ci & 7 holds q-pair kp = (ci & 7) ^ pl_s
    int voff_tr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int blk = wave * 4 + j;
        const int pl_s = (lane >> 3) ^ (blk & 1);
        const int kp = (lane & 7) ^ pl_s;
        const long pp = p0 + blk * 8 + pl_s;
        voff_tr[j] = (int)((pp * (pp + 1) / 2 + 2 * kp) * 8);
    }
    // keep-test bases: t(kk, b) = (q0 - p0) + tb[b] + kk  with  tb[b] = fk - fn - 64 wc - 16 b
    int tb[4];
#pragma unroll
    for (int b = 0; b < 4; b++) tb[b] = fk - fn - wc * 64 - b * 16;

    // the same quantities for the 1 x 4 arrangement: wave w owns AO columns [32 w, 32 w + 32) = LDS blocks 4 w .. 4 w + 3
    const int woffa = fk * LDN + fn;
    const int woffb_row = fk * LDN + wave * 32 + fn;
    const int woffb_tr = (wave * 4 + (fn >> 3)) * 128 + (((pl ^ bodd) * 8) + ((fk >> 1) ^ (pl & 1))) * 2 + (fk & 1);
    int watr[4], wtb[2];
#pragma unroll
    for (int g = 0; g < 4; g++) watr[g] = woffb_tr + (((2 * g) ^ (pl & 6))) * 2;
    wtb[0] = fk - fn - wave * 32;
    wtb[1] = wtb[0] - 16;

    bool wide = false;
    if constexpr (WM > 0) wide = chunk == nchunk - 1;                    // workgroup-uniform
    double4_t acc[NA][4];
    double4_t wacc[WMe][2];
    if (!wide) {                                                         // (only one of the two sets is live on either path)
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};
    } else {
#pragma unroll
        for (int a = 0; a < WMe; a++) wacc[a][0] = wacc[a][1] = double4_t{0, 0, 0, 0};
    }

    // virtual tile v -> (q0, transposed? as 0 / 1), by integer arithmetic only: any branch here would split the k-tile
    // body into several scheduling regions (and did: hipcc turns the obvious ternaries into s_cbranch)
    const int nA = p0 / KB;
    const int ncross = (kdim - p0) / KB < 8 ? (kdim - p0) / KB : 8;
    const int nvirt = DIAG ? kdim / KB : kdim / KB + ncross;
    auto tile_q0 = [&](int v) {
        if (DIAG) return v * KB;                           // every k-tile once
        int adj = (v - nA + 1) >> 1;                       // tiles visited twice so far
        adj = adj < 0 ? 0 : adj;
        adj = adj > ncross ? ncross : adj;
        return (v - adj) * KB;
    };
    auto tile_tr = [&](int v) {
        const int u = v - nA;
        if (DIAG) return (u >> 31) & 1;                    // transposed layout above the diagonal block only
        return (int)(u < 0) | ((int)(u < 2 * ncross) & (u & 1) & (int)(u >= 0));
    };
    // DIAG: base address and size of the two sources as integers - the buffer resource of a DMA is rebuilt from them with
    // scalar arithmetic (a select between two resource descriptors would be a branch)
    const long pk_base = (long)(cderi + L * npair);
    const long dg_base = DIAG ? (long)(diag + ((long)L * ntile_p + p0 / NT) * (NT * NT)) : 0;
    const unsigned pk_num = bytes_left > 0xffffffffL ? 0xffffffffu : (unsigned)bytes_left;
    int dvo[4];                                            // (transposed - row) difference of the DMA lane offsets
#pragma unroll
    for (int j = 0; j < 4; j++) dvo[j] = voff_tr[j] - voff;

    // NTR: layout of the tile being staged when the caller knows it at compile time (0 row, 1 transposed; 2 = ask tile_tr):
    // inside the DIAG kernel's two phases it does, and the lane offset / LDS stride / source offset selects (4 v_cndmask per
    // k-tile - ~0.45 % of the kernel each - and a third of the SALU work) fold away
    auto stage_row = [&](auto NTRc, int v, double *da, double *db, int j) {
        constexpr int NTR = decltype(NTRc)::value;
        const int q0 = tile_q0(v);
        const int tr = NTR == 2 ? tile_tr(v) : NTR;
        const int k = wave * 4 + j;
        dma_row(r_orb, da + k * LDN, voff, (q0 + k) * ldo8);
        const int q = q0 + k;
        const int soff_row = (q * (q + 1) / 2 + p0) * 8;
        if constexpr (DIAG && NTR == 1) {
            dma_row(r_pk, db + k * 128, voff_tr[j], q0 * 8);                     // above the diagonal block: never from the side image
        } else if constexpr (DIAG) {
            const int u = v - nA;
            const int is_dg = (~(u >> 31)) & ((u - ncross) >> 31) & 1;           // 0 <= u < ncross: the diagonal block
            const int soff = soff_row + tr * (q0 * 8 - soff_row) + is_dg * ((q - p0) * (NT * 8) - soff_row);
            const long base = pk_base + ((dg_base - pk_base) & -(long)is_dg);
            const unsigned num = pk_num + (unsigned)is_dg * ((unsigned)(NT * NT * 8) - pk_num);
            const __amdgpu_buffer_rsrc_t r_src = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, num, 0x00020000);
            dma_row(r_src, db + k * (LDN - tr * (LDN - 128)), voff + tr * dvo[j], soff);
        } else {
            const int soff = soff_row + tr * (q0 * 8 - soff_row);
            dma_row(r_pk, db + k * (LDN - tr * (LDN - 128)), voff + tr * dvo[j], soff);
        }
        if (NA == 5 && j == 0) dma_row(r_orb, da + RB + wave * 128, voff_rem, (q0 + wave * 4) * ldo8);
    };
    // One k-tile body per (layout, masked?) pair, each a single basic block: measured with a layout-generic body, ~100 VALU
    // selects per tile in a layout-generic body cost 12 % - a VALU between two FP64 MFMAs is not free (~6 cycles each) - so
    // the fragment addresses of a body are immediates again and the keep-masks exist only in the (<= 16) crossing tiles.
    auto step = [&](auto WDc, auto TRc, auto MKc, auto NTRc, const double *ca, const double *cb, double *na, double *nb, int v) {
        constexpr bool WD = decltype(WDc)::value, TR = decltype(TRc)::value, MK = decltype(MKc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int vn = (v + 1 < nvirt - 1) ? v + 1 : nvirt - 1;   // last tile: harmless reload into the idle buffer
        const int d = tile_q0(v) - p0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            if constexpr (WD) {
                double af[WMe], bf[2];
#pragma unroll
                for (int a = 0; a < WMe; a++) af[a] = ca[woffa + kk * LDN + a * 16];
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const double val = TR ? ((lds_b64_ptr)cb)[watr[kk >> 2] + b * 256] : cb[woffb_row + kk * LDN + b * 16];
                    if (MK) {
                        const bool below = (d + wtb[b] + kk) >= 0;
                        bf[b] = (below != TR) ? val : 0.0;
                    } else {
                        bf[b] = val;
                    }
                }
                stage_row(NTRc, vn, na, nb, kk >> 2);
#pragma unroll
                for (int a = 0; a < WMe; a++) {
                    wacc[a][0] = mfma_f64_16x16x4(af[a], bf[0], wacc[a][0]);
                    wacc[a][1] = mfma_f64_16x16x4(af[a], bf[1], wacc[a][1]);
                }
                continue;
            }
            double af[NA], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = ca[offa + kk * LDN + a * 16];
            if constexpr (NA == 5) af[4] = ca[offr + kk * 32];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                // transposed tiles: single ds_read_b64 (32-lane groups, 64-dword bank modulus - what the swizzle is made for).
                // Merged into ds_read2_b64 (16-lane groups, 32-dword modulus) the 16 fn-lanes all read the same half of their
                // 16-byte chunks: an inherent 2-way conflict (PMC, r03: SQ_LDS_BANK_CONFLICT 4.8e9 -> 0, LDS cycles 2.71e10 ->
                // 1.93e10, kernel -1.2 %; every OTHER fragment read is faster merged - profiles/r03/kbench_frag_reads.log)
                const double val = TR ? ((lds_b64_ptr)cb)[atr[kk >> 2] + b * 256] : cb[offb_row + kk * LDN + b * 16];
                if (MK) {
                    const bool below = (d + tb[b] + kk) >= 0;      // q >= p
                    bf[b] = (below != TR) ? val : 0.0;
                } else {
                    bf[b] = val;
                }
            }
            stage_row(NTRc, vn, na, nb, kk >> 2);
#pragma unroll
            for (int a = 0; a < NA; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    using N0 = std::integral_constant<int, 0>;
    using N1 = std::integral_constant<int, 1>;
    using N2 = std::integral_constant<int, 2>;
#pragma unroll
    for (int j = 0; j < 4; j++) stage_row(N2{}, 0, sa0, sq0, j);
    // the phases start on even tiles (nA is a multiple of 8, the crossing tiles come in pairs), so the buffer of
    // every call is a compile-time constant: with run-time buffer pointers hipcc needs waterfall loops for m0
    auto run = [&](auto WDc) {
    int v = 0;
    if constexpr (DIAG) {
        for (; v < nA - 2; v += 2) {                                     // above the diagonal: transposed layout, and so is the next tile
            step(WDc, T_{}, F_{}, N1{}, sa0, sq0, sa1, sq1, v);
            step(WDc, T_{}, F_{}, N1{}, sa1, sq1, sa0, sq0, v + 1);
        }
        if (v < nA) {                                                    // its last pair: the tile after it is the diagonal block (row layout)
            step(WDc, T_{}, F_{}, N1{}, sa0, sq0, sa1, sq1, v);
            step(WDc, T_{}, F_{}, N0{}, sa1, sq1, sa0, sq0, v + 1);
            v += 2;
        }
        for (; v < nvirt; v += 2) {                                      // diagonal block (side image) and below: row layout
            step(WDc, F_{}, F_{}, N0{}, sa0, sq0, sa1, sq1, v);
            if (v + 1 < nvirt) step(WDc, F_{}, F_{}, N0{}, sa1, sq1, sa0, sq0, v + 1);
        }
    } else {
        for (; v < nA; v += 2) {                                         // above the diagonal: transposed layout
            step(WDc, T_{}, F_{}, N2{}, sa0, sq0, sa1, sq1, v);
            step(WDc, T_{}, F_{}, N2{}, sa1, sq1, sa0, sq0, v + 1);
        }
        for (; v < nA + 2 * ncross; v += 2) {                            // crossing tiles: row half, then transposed half
            step(WDc, F_{}, T_{}, N2{}, sa0, sq0, sa1, sq1, v);
            step(WDc, T_{}, T_{}, N2{}, sa1, sq1, sa0, sq0, v + 1);
        }
        for (; v < nvirt; v += 2) {                                      // below: row layout
            step(WDc, F_{}, F_{}, N2{}, sa0, sq0, sa1, sq1, v);
            if (v + 1 < nvirt) step(WDc, F_{}, F_{}, N2{}, sa1, sq1, sa0, sq0, v + 1);
        }
    }
    };
    if constexpr (WM > 0) { if (wide) run(T_{}); else run(F_{}); } else run(F_{});
    double *out = X + L * nocc_pad * ldx;
    double rho_acc = 0;
    if constexpr (WM > 0) {
        if (wide) {
#pragma unroll
            for (int a = 0; a < WMe; a++)
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const long p = p0 + wave * 32 + b * 16 + fn;
                    if (p >= ldx) continue;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = m0 + a * 16 + fk + 4 * r;
                        if (i < nocc_pad) {
                            out[(long)i * ldx + p] = wacc[a][b][r];
                            if (RHO && p < nao) rho_acc += wacc[a][b][r] * orb[p * ldo + i];
                        }
                    }
                }
        }
    }
    if (!wide)
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const long p = p0 + wc * 64 + b * 16 + fn;
            if (p >= ldx) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = m0 + (a < 4 ? wr * 64 + a * 16 : 128 + wr * 16) + fk + 4 * r;
                if (i < nocc_pad) {
                    out[(long)i * ldx + p] = acc[a][b][r];
                    if (RHO && p < nao) rho_acc += acc[a][b][r] * orb[p * ldo + i];
                }
            }
        }
    if (RHO) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rho_acc += __shfl_xor(rho_acc, off, 64);
        if (lane == 0) rho[(L * gridDim.x + blockIdx.x) * 4 + wave] = rho_acc;
    }
}

// 128 x 128 tile of C[split] += A^T B (all tiles, or the lower-triangular ones of the SYRK), v2 DMA scheme
__global__ __launch_bounds__(256, 2) void gemm_tn_glds2_kernel(
    const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb,
    double *__restrict__ C, int ldc, int m, int n, long kdim, int lower_only, int ntile_n, long kchunk, int prio,
    const int *__restrict__ order, int nsplit_o)
{
    if (prio) __builtin_amdgcn_s_setprio(3);        // MFMA waves ahead of a co-resident HBM-bound kernel's waves (tuning "mfmaprio")
    // r06: XCD-aware dispatch order (syrk_xcd_order): a 1-D launch whose workgroup b (XCD b % 8) looks its (tile, split) up
    int bsplit = blockIdx.y, btile = blockIdx.x, nsp = (int)gridDim.y;
    if (order) {
        const int o = __builtin_amdgcn_readfirstlane(order[blockIdx.x]);
        if (o < 0) return;
        btile = o & 0xffff; bsplit = o >> 16; nsp = nsplit_o;
    }
    constexpr int PA = KB * LDN;
    __shared__ double sb0[2 * PA];
    __shared__ double sb1[2 * PA];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tm, tn;
    if (lower_only) {
        int t = btile;
        tm = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((tm + 1) * (tm + 2) / 2 <= t) tm++;
        while (tm * (tm + 1) / 2 > t) tm--;
        tn = t - tm * (tm + 1) / 2;
    } else {
        tm = btile / ntile_n;
        tn = btile - tm * ntile_n;
    }
    const int p0 = tm * NT, q0 = tn * NT;
    // k range of this split: `kchunk` rows each, the LAST split takes what is left (uniform splits: about the same; balanced
    // SYRK: a short remainder piece, see dgemm_tn_impl)
    long kbeg = (long)bsplit * kchunk;
    if (kbeg > kdim) kbeg = kdim;
    const long kend = (kbeg + kchunk < kdim && bsplit + 1 < nsp) ? kbeg + kchunk : kdim;
    const int nk = (int)(kend - kbeg);                 // rows of this split: row offsets stay below 4 GiB (launcher)
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(A + kbeg * lda + p0);
    const __amdgpu_buffer_rsrc_t r_b = make_rsrc(B + kbeg * ldb + q0);
    const int voff = lane * 16;
    const int lda8 = lda * 8, ldb8 = ldb * 8;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn, offb = PA + fk * LDN + wc * 64 + fn;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *dst, int j) {
        const int k = wave * 4 + j;
        dma_row(r_a, dst + k * LDN, voff, (k0 + k) * lda8);
        dma_row(r_b, dst + PA + k * LDN, voff, (k0 + k) * ldb8);
    };
    auto step = [&](const double *cur, double *nxt, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < nk) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = cur[offa + kk * LDN + a * 16];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cur[offb + kk * LDN + b * 16];
            stage_row(kn, nxt, kk >> 2);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    auto step_idle = [&](double *nxt, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < nk) ? k0 + KB : k0;
#pragma unroll
        for (int j = 0; j < 4; j++) stage_row(kn, nxt, j);
    };
    if (nk > 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) stage_row(0, sb0, j);
    }
    // a wave whose 64 x 64 block lies entirely beyond the matrix (edge tiles), or entirely above the diagonal of a
    // diagonal SYRK tile, only stages its DMA rows
    const bool idle = p0 + wr * 64 >= m || q0 + wc * 64 >= n || (lower_only && tm == tn && wc > wr);
    if (!idle) {
        for (int k0 = 0; k0 < nk; k0 += 2 * KB) {
            step(sb0, sb1, k0);
            if (k0 + KB < nk) step(sb1, sb0, k0 + KB);
        }
    } else {
        for (int k0 = 0; k0 < nk; k0 += 2 * KB) {
            step_idle(sb1, k0);
            if (k0 + KB < nk) step_idle(sb0, k0 + KB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    double *out = C + (long)bsplit * m * ldc;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            int col = q0 + wc * 64 + b * 16 + fn;
            if (col >= n) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int rowi = p0 + wr * 64 + a * 16 + fk + 4 * r;
                if (rowi < m) unsafeAtomicAdd(out + (long)rowi * ldc + col, acc[a][b][r]);
            }
        }
}

// SYRK C[split] += X^T X on a re-tiled lower triangle (r03).  The 2 x 2-wave tiling of gemm_tn_glds2 leaves dead 64 x 64
// wave blocks wherever a 128 x 128 tile sticks out of the triangle: one per diagonal tile and, when the matrix has an odd
// number nb of 64-column blocks, two per tile of the last tile row (three in the corner) - 45 of 480 at nao = 1856.  Here a
// work item is FOUR 64 x 64 blocks that share at most four 64-column panels of X ("slots"): slots 0/1 arrive with the "A"
// DMA (lanes 0-31 fetch slot 0, lanes 32-63 slot 1, per-lane source offsets in one loop-invariant VGPR), slots 2/3 with the
// "B" DMA; wave w multiplies slot a[w] by slot b[w] into block (rb[w], cb[w]).  Items (host table, syrk_items()):
//   * off-diagonal 128-tiles (I > J):  slots {2I, 2I+1, 2J, 2J+1}, the usual 2 x 2 arrangement
//   * diagonal tile I + one block of the odd last block row r:  slots {2I, 2I+1, r}: (2I,2I), (2I+1,2I), (2I+1,2I+1), (r,2I)
//   * the other blocks of row r three at a time: slots {r, c1, c2, c3}: (r,c1), (r,c2), (r,c3) [+ the corner (r,r)]
// 110 items with 435 live blocks instead of 120 tiles at nao = 1856.  k loop, DMA scheme and epilogue as gemm_tn_glds2.
// r05 - the second J pass INSIDE the SYRK ("one kernel with a fixed issue order" instead of two kernels that
// share the CUs).  The packed tensor rows of the block, B[nb][npair], are cut into row-loads (one aux row x 512 consecutive packed
// columns = 16 bytes per lane of the workgroup); the row-loads are dealt to the workgroups in dispatch order in proportion to the
// k-tiles each one works through (JStream), every k-tile issues its share (<= 4, one per MFMA group) as plain global loads into
// registers and folds the loads of the PREVIOUS k-tile (landed by the s_waitcnt at the top of the tile) into two accumulators per
// lane: vj[col] += rho[L] * B[L][col].  A workgroup walks down the aux rows of a 512-column chunk and adds its partial sum into vj
// (atomic: two workgroups may share a chunk) when the chunk ends.
struct JStream {
    const double *B;        // packed rows of this K block [nb][npair]
    const double *rho;      // [nb] first-pass result of the same rows
    double *vj;             // [npair], accumulated
    long npair;
    int nb;
    long total_steps;       // k-tiles of ALL workgroups of the launch (nitems x sum over the splits)
    long total_rows;        // row-loads: ceil(npair / 512) * nb
};

template <int PROBE, bool JF>         // PROBE = 1: benchmarking probe, skips the "B" DMA of every k-row (half the L2 -> LDS traffic, wrong results)
__global__ __launch_bounds__(256, 2) void syrk_slots_kernel(
    const double *__restrict__ A, int lda, double *__restrict__ C, int ldc, int m, long kdim, const int *__restrict__ items,
    long kchunk, int prio, JStream js, const int *__restrict__ order, int nsplit_o)
{
    if (prio) __builtin_amdgcn_s_setprio(3);
    int bsplit = blockIdx.y, bitem = blockIdx.x, nsp_grid = (int)gridDim.y;
    if (order) {                         // r06: XCD-aware dispatch order, see gemm_tn_glds2_kernel (never together with JF)
        const int o = __builtin_amdgcn_readfirstlane(order[blockIdx.x]);
        if (o < 0) return;
        bitem = o & 0xffff; bsplit = o >> 16; nsp_grid = nsplit_o;
    }
    constexpr int PA = KB * LDN;
    __shared__ double sb0[2 * PA];
    __shared__ double sb1[2 * PA];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int *it = items + (long)bitem * 16;
    const int c0 = it[0], c1 = it[1], c2 = it[2], c3 = it[3];
    const int wdesc = __builtin_amdgcn_readfirstlane(it[4 + wave * 3]);           // a | b << 8 | live << 16
    const int rb = __builtin_amdgcn_readfirstlane(it[5 + wave * 3]), cb = __builtin_amdgcn_readfirstlane(it[6 + wave * 3]);
    const int sa = wdesc & 0xff, sbt = (wdesc >> 8) & 0xff;
    const bool live = (wdesc >> 16) & 1;
    long kbeg = (long)bsplit * kchunk;
    if (kbeg > kdim) kbeg = kdim;
    const long kend = (kbeg + kchunk < kdim && bsplit + 1 < nsp_grid) ? kbeg + kchunk : kdim;
    const int nk = (int)(kend - kbeg);
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(A + kbeg * lda);
    const int lda8 = lda * 8;
    const int voff_a = (lane & 31) * 16 + ((lane >> 5) ? c1 : c0) * 8;
    const int voff_b = (lane & 31) * 16 + ((lane >> 5) ? c3 : c2) * 8;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = (sa >> 1) * PA + (sa & 1) * 64 + fk * LDN + fn;
    const int offb = (sbt >> 1) * PA + (sbt & 1) * 64 + fk * LDN + fn;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *dst, int j) {
        const int k = wave * 4 + j;
        dma_row(r_a, dst + k * LDN, voff_a, (k0 + k) * lda8);
        if (PROBE == 0) dma_row(r_a, dst + PA + k * LDN, voff_b, (k0 + k) * lda8);
    };
    // ---- fused second J pass (JF): this workgroup's run of row-loads, dealt in dispatch order by k-tile count.  All of this
    // state is wave-uniform; it is pinned to SGPRs with readfirstlane so that rho[L] is a scalar load and the row-load guards are
    // scalar branches
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
    int j_left = 0;                               // row-loads of the run not issued yet
    int j_c = 0, j_L = 0;                         // 512-column chunk and aux row of the next row-load
    int j_q = 0, j_rem = 0, j_bres = 0, j_steps = 1;   // row-loads per k-tile: j_q (+ 1 when the Bresenham remainder rolls over)
    double2_t jv[4];                              // loads of the previous k-tile
    double jrho[4] = {0, 0, 0, 0};
    int jch[4] = {0, 0, 0, 0};
    int jpend = 0;
    double2_t jacc = double2_t{0, 0};
    int jacc_c = -1;
    const long jcol0 = (long)tid * 2;
    if (JF) {
        const int nsp = nsp_grid;
        long before = 0, mine = 0;
        for (int y = 0; y < nsp; y++) {
            long kb = (long)y * kchunk;
            if (kb > kdim) kb = kdim;
            const long ke = (kb + kchunk < kdim && y + 1 < nsp) ? kb + kchunk : kdim;
            const long st_y = (ke - kb + KB - 1) / KB;
            if (y < bsplit) before += st_y * (long)gridDim.x;
            if (y == bsplit) mine = st_y;
        }
        const long base = before + (long)blockIdx.x * mine;
        const long r0 = base * js.total_rows / js.total_steps;
        const long r1 = (base + mine) * js.total_rows / js.total_steps;
        j_c = UNI((int)(r0 / js.nb));
        j_L = UNI((int)(r0 - (long)j_c * js.nb));
        j_left = UNI((int)(r1 - r0));
        j_steps = UNI(mine > 0 ? (int)mine : 1);
        j_q = UNI(j_left / j_steps);
        j_rem = UNI(j_left - j_q * j_steps);
#pragma unroll
        for (int u = 0; u < 4; u++) jv[u] = double2_t{0, 0};
    }
    auto j_flush = [&]() {
        if (jacc_c >= 0) {
            const long col = (long)jacc_c * 512 + jcol0;
            if (col < js.npair) {
                unsafeAtomicAdd(js.vj + col, jacc[0]);
                unsafeAtomicAdd(js.vj + col + 1, jacc[1]);
            }
        }
        jacc = double2_t{0, 0};
    };
    auto j_consume = [&]() {                       // the previous k-tile's loads have landed (s_waitcnt vmcnt(0) at the top of the tile)
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (u < jpend) {
                if (jch[u] != jacc_c) { j_flush(); jacc_c = jch[u]; }
                jacc[0] += jrho[u] * jv[u][0];
                jacc[1] += jrho[u] * jv[u][1];
            }
        jpend = 0;
    };
    int j_quota = 0;
    auto j_begin_step = [&]() {
        j_consume();
        int qn = j_q;
        j_bres += j_rem;
        if (j_bres >= j_steps) { j_bres -= j_steps; qn++; }
        if (qn > 4) qn = 4;                        // (the launcher only fuses shapes with <= 4 row-loads per k-tile; the tail loop takes the rest)
        if (qn > j_left) qn = j_left;
        j_quota = UNI(qn);
        j_bres = UNI(j_bres);
    };
    auto j_issue = [&](int u) {                    // one row-load per MFMA group
        if (u < j_quota) {
            const long col = (long)j_c * 512 + jcol0;
            jv[u] = col < js.npair ? __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(js.B + (long)j_L * js.npair + col))
                                   : double2_t{0, 0};
            jrho[u] = js.rho[j_L];
            jch[u] = j_c;
            jpend = u + 1;
            j_left = UNI(j_left - 1);
            const int ln = j_L + 1;
            const bool wrap = ln == js.nb;
            j_L = UNI(wrap ? 0 : ln);
            j_c = UNI(wrap ? j_c + 1 : j_c);
        }
    };
    auto step = [&](const double *cur, double *nxt, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (JF) j_begin_step();
        const int kn = (k0 + KB < nk) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[4], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = cur[offa + kk * LDN + a * 16];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cur[offb + kk * LDN + b * 16];
            stage_row(kn, nxt, kk >> 2);
            if (JF) j_issue(kk >> 2);
#pragma unroll
            for (int a = 0; a < 4; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    auto step_idle = [&](double *nxt, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (JF) j_begin_step();
        const int kn = (k0 + KB < nk) ? k0 + KB : k0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            stage_row(kn, nxt, j);
            if (JF) j_issue(j);
        }
    };
    auto j_finish = [&]() {                        // loads still pending, row-loads the k-tiles did not get to, the last partial sum
        if (!JF) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        j_consume();
        while (j_left > 0) {
            const long col = (long)j_c * 512 + jcol0;
            if (j_c != jacc_c) { j_flush(); jacc_c = j_c; }
            if (col < js.npair) {
                const double2_t b = *reinterpret_cast<const double2_t *>(js.B + (long)j_L * js.npair + col);
                const double r = js.rho[j_L];
                jacc[0] += r * b[0];
                jacc[1] += r * b[1];
            }
            j_left--;
            if (++j_L == js.nb) { j_L = 0; j_c++; }
        }
        j_flush();
    };
#undef UNI
    if (nk > 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) stage_row(0, sb0, j);
    }
    if (live) {
        for (int k0 = 0; k0 < nk; k0 += 2 * KB) {
            step(sb0, sb1, k0);
            if (k0 + KB < nk) step(sb1, sb0, k0 + KB);
        }
        j_finish();
    } else {
        for (int k0 = 0; k0 < nk; k0 += 2 * KB) {
            step_idle(sb1, k0);
            if (k0 + KB < nk) step_idle(sb0, k0 + KB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        j_finish();
        return;
    }
    double *out = C + (long)bsplit * m * ldc;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int col = cb * 64 + b * 16 + fn;
            if (col >= m) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rowi = rb * 64 + a * 16 + fk + 4 * r;
                if (rowi < m) unsafeAtomicAdd(out + (long)rowi * ldc + col, acc[a][b][r]);
            }
        }
}

// r06 experiment ("syrkring" = 1): the re-tiled SYRK with a FOUR-stage LDS ring of 8-deep k-tiles instead of two stages of 16.
// Same LDS (4 x 2 x 8 x 144 doubles = 73.7 KB: still two workgroups per CU), same work items, same epilogue; what changes is how
// far ahead the operands are requested: the DMA rows of tile t + 3 are issued during tile t and the top of tile t waits with
// s_waitcnt vmcnt(8) for tile t only (tiles t + 1, t + 2 stay in flight) - at least two whole tile times of lead (>= 8 k cycles
// of wall time with two workgroups sharing the matrix pipe) where the 2-stage loop gives the last rows of a tile a quarter of
// one.  r05 read the ~6 ms the co-running second J pass costs the SYRK as DMA LATENCY (tiles arriving late at the k-tile barrier
// while 1.5 TB/s of foreign traffic is in flight) and stopped at "a deeper ring does not fit": it does, at the price of a barrier
// every 32 instead of 64 MFMAs per wave.
__global__ __launch_bounds__(256, 2) void syrk_ring4_kernel(
    const double *__restrict__ A, int lda, double *__restrict__ C, int ldc, int m, long kdim, const int *__restrict__ items,
    long kchunk, const int *__restrict__ order, int nsplit_o)
{
    constexpr int K8 = 8, P8 = K8 * LDN, ST = 2 * P8;          // one stage: "A" panel rows [8][144], then the "B" panel
    // four SEPARATE arrays: the compiler's wait-count insertion tracks LDS-DMA targets per object - one array for the whole ring
    // made it put s_waitcnt vmcnt(0) before every fragment read (any DMA in flight might alias), which is the 2-stage loop again
    __shared__ double st0[ST];
    __shared__ double st1[ST];
    __shared__ double st2[ST];
    __shared__ double st3[ST];
    int bsplit = blockIdx.y, bitem = blockIdx.x, nsp_grid = (int)gridDim.y;
    if (order) {
        const int o = __builtin_amdgcn_readfirstlane(order[blockIdx.x]);
        if (o < 0) return;
        bitem = o & 0xffff; bsplit = o >> 16; nsp_grid = nsplit_o;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int *it = items + (long)bitem * 16;
    const int c0 = it[0], c1 = it[1], c2 = it[2], c3 = it[3];
    const int wdesc = __builtin_amdgcn_readfirstlane(it[4 + wave * 3]);
    const int rb = __builtin_amdgcn_readfirstlane(it[5 + wave * 3]), cb = __builtin_amdgcn_readfirstlane(it[6 + wave * 3]);
    const int sa = wdesc & 0xff, sbt = (wdesc >> 8) & 0xff;
    const bool live = (wdesc >> 16) & 1;
    long kbeg = (long)bsplit * kchunk;
    if (kbeg > kdim) kbeg = kdim;
    const long kend = (kbeg + kchunk < kdim && bsplit + 1 < nsp_grid) ? kbeg + kchunk : kdim;
    const int nk = (int)(kend - kbeg);                              // a multiple of 16 (launcher)
    const int ntile = nk / K8;
    const __amdgpu_buffer_rsrc_t r_a = make_rsrc(A + kbeg * lda);
    const int lda8 = lda * 8;
    const int voff_a = (lane & 31) * 16 + ((lane >> 5) ? c1 : c0) * 8;
    const int voff_b = (lane & 31) * 16 + ((lane >> 5) ? c3 : c2) * 8;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = (sa >> 1) * P8 + (sa & 1) * 64 + fk * LDN + fn;
    const int offb = (sbt >> 1) * P8 + (sbt & 1) * 64 + fk * LDN + fn;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    // tile index -> first k row (tiles beyond the range re-load the last one: the DMA count per step stays uniform for vmcnt)
    auto tile_k0 = [&](int t) { return (t < ntile ? t : ntile - 1) * K8; };
    auto stage_row = [&](int t, double *dst, int j) {              // j = 0, 1: this wave's two rows of the 8-row tile
        const int k = wave * 2 + j, k0 = tile_k0(t);
        dma_row(r_a, dst + k * LDN, voff_a, (k0 + k) * lda8);
        dma_row(r_a, dst + P8 + k * LDN, voff_b, (k0 + k) * lda8);
    };
    if (ntile <= 0) return;
#pragma unroll
    for (int j = 0; j < 2; j++) stage_row(0, st0, j);
#pragma unroll
    for (int j = 0; j < 2; j++) stage_row(1, st1, j);
#pragma unroll
    for (int j = 0; j < 2; j++) stage_row(2, st2, j);
    // Schedule of step t (stage s = t % 4), one barrier per step, fragment reads software-pipelined ACROSS it:
    //   group 0: read the fragments of (tile t, k-group 1); DMA row 0 of tile t + 3 -> stage (t + 3) % 4; MFMAs on (tile t, group 0)
    //   group 1: read the fragments of (tile t + 1, group 0); DMA row 1 of tile t + 3;                    MFMAs on (tile t, group 1)
    //   end:     s_waitcnt vmcnt(4) = this wave's rows of tile t + 2 have landed (tile t + 3 may fly); s_barrier = tile t + 2 is
    //            visible to everybody from now on, and nobody reads stage t % 4 any more (tile t + 4 goes there in step t + 1).
    // Tile t + 1 is readable during step t because the barrier at the end of step t - 1 covered it.
    double a0[4], b0[4], a1[4], b1[4];
    auto frag = [&](const double *cur, int kk, double (&af)[4], double (&bf)[4]) {
#pragma unroll
        for (int a = 0; a < 4; a++) af[a] = cur[offa + kk * LDN + a * 16];
#pragma unroll
        for (int b = 0; b < 4; b++) bf[b] = cur[offb + kk * LDN + b * 16];
    };
    auto mma = [&](const double (&af)[4], const double (&bf)[4]) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
    };
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");          // tiles 0 and 1 landed and visible
    if (live) {
        frag(st0, 0, a0, b0);
        auto step = [&](const double *cur, const double *nextt, double *dma, int t) {
            frag(cur, 4, a1, b1);
            stage_row(t + 3, dma, 0);
            mma(a0, b0);
            frag(nextt, 0, a0, b0);
            stage_row(t + 3, dma, 1);
            mma(a1, b1);
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        };
        for (int t = 0; t < ntile; t += 4) {
            step(st0, st1, st3, t);
            if (t + 1 < ntile) step(st1, st2, st0, t + 1);
            if (t + 2 < ntile) step(st2, st3, st1, t + 2);
            if (t + 3 < ntile) step(st3, st0, st2, t + 3);
        }
    } else {
        auto step_idle = [&](double *dma, int t) {                      // a wave without a live block only stages its DMA rows
#pragma unroll
            for (int j = 0; j < 2; j++) stage_row(t + 3, dma, j);
            asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        };
        for (int t = 0; t < ntile; t += 4) {
            step_idle(st3, t);
            if (t + 1 < ntile) step_idle(st0, t + 1);
            if (t + 2 < ntile) step_idle(st1, t + 2);
            if (t + 3 < ntile) step_idle(st2, t + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!live) return;
    double *out = C + (long)bsplit * m * ldc;
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int col = cb * 64 + b * 16 + fn;
            if (col >= m) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rowi = rb * 64 + a * 16 + fk + 4 * r;
                if (rowi < m) unsafeAtomicAdd(out + (long)rowi * ldc + col, acc[a][b][r]);
            }
        }
}

// flag[rt][ct] = max |src[16 rt .. 16 rt + 15][16 ct .. 16 ct + 15]| > thr   (rows >= nrows count as zero)
__global__ __launch_bounds__(256) void tile_mask_kernel(const double *__restrict__ src, long ld, long nrows, double thr,
                                                        unsigned char *__restrict__ out, int nct)
{
    const long rt = blockIdx.x;
    for (int c = threadIdx.x; c < nct * 16; c += 256) {
        double mx = 0;
        if (c < ld)
            for (int r = 0; r < 16; r++) {
                long row = rt * 16 + r;
                if (row < nrows) mx = fmax(mx, fabs(src[row * ld + c]));
            }
        for (int off = 8; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
        if ((c & 15) == 0) out[rt * nct + (c >> 4)] = mx > thr;
    }
}

// out[i][j] = sum_s part[s][i][j]  (i>=j when lower), mirrored to the upper triangle when sym
__global__ void reduce_splits_kernel(const double *__restrict__ part, int nsplit, int m, int ldc,
                                     double *__restrict__ out, int ldo, int sym)
{
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    int i = blockIdx.y;
    if (j >= m || i >= m) return;
    if (sym && j > i) return;
    double v = 0;
    for (int s = 0; s < nsplit; s++) v += part[((long)s * m + i) * ldc + j];
    out[(long)i * ldo + j] = v;
    if (sym) out[(long)j * ldo + i] = v;
}

// full[L][p][q] (L stride = rows*ld, rows >= nao) from tril rows; both triangles filled
__global__ void unpack_tril_kernel(const double *__restrict__ tril, long npair, int nao,
                                   double *__restrict__ full, int ld, int rows)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    int p = blockIdx.y;
    long L = blockIdx.z;
    if (q >= ld) return;
    double v = 0;
    if (q < nao) v = (p >= q) ? tril[L * npair + (long)p * (p + 1) / 2 + q]
                              : tril[L * npair + (long)q * (q + 1) / 2 + p];
    full[(L * rows + p) * ld + q] = v;
}

// tril[s][pq] = (D[s][p][q] + D[s][q][p]) * (p==q ? 0.5 : 1)   (pyscf/df/df_jk.py:329-332)
__global__ void pack_dm_kernel(const double *__restrict__ dm, int nao, double *__restrict__ tril,
                               long npair)
{
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    int p = blockIdx.y;
    long s = blockIdx.z;
    if (q > p) return;
    const double *d = dm + s * nao * nao;
    double v = d[(long)p * nao + q] + d[(long)q * nao + p];
    if (p == q) v *= 0.5;
    tril[s * npair + (long)p * (p + 1) / 2 + q] = v;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
static int g_use_glds = 1;
static int g_gemm_wide = 1;
static int g_e2_mtmax = 10;   // orbital tiles (of 16) per workgroup, upper bound
static int g_e2_prio = 0;     // s_setprio 3 in the half transform (A/B switch)
static int g_mfma_prio = 0;   // s_setprio 3 in the SYRK kernels (A/B switch)
static int g_j2_wide = 0;     // second J pass with 16-byte loads (two packed columns per lane); A/B switch, default set after measurement
static int g_j2_maxwg = 0;    // cap on the workgroups of the second J pass (0: one per 256 columns)
static int g_pair_tail = 1;   // half-empty last column tile of e2_sq2 as one workgroup per pair of aux rows
static int g_sq_shift = 0;    // benchmarking probe only: read the square image from a base shifted by this many doubles
static int g_e2_merge = 1;    // e2_sq2: pair-tail workgroups inside the main launch ("e2merge")
static int g_e2_xmap = 1;     // e2_sq2 / e2_pk with several orbital chunks: the chunks of a column tile on one XCD ("e2xmap", r04)
static int g_e2_wide = 1;     // e2_sq2 / e2_pk: last orbital chunk of < 8 tiles in the 1 x 4 wave arrangement ("e2wide", r04)
static int g_pk_diag = 1;     // e2_pk reads the diagonal 128 x 128 blocks from the side image when the caller passes one ("pkdiag")
static int g_pk_dma = 1;      // packed-operand half transform by LDS-DMA (e2_pk) when the chunk shape allows
static int g_syrk_probe = 0;  // benchmarking probe only: the re-tiled SYRK without its second panel DMA (results meaningless)
static int g_syrk_slots = 1;  // SYRK on the re-tiled triangle (syrk_slots_kernel) when the matrix has an odd number of 64-column blocks
static int g_syrk_frac = 1;   // balanced SYRK: full pieces + one short remainder piece per tile (dgemm_tn_impl)
static int g_num_cu = 256;    // MI355X
static int g_syrk_reserve = 0; // balanced SYRK: workgroup slots (of 2 x 256) left free for a co-running J pass 2 ("syrkreserve")
static int g_dma_v2 = 1;      // buffer-resource LDS-DMA with the issue spread over the MFMA groups (e2_sq2 / gemm_tn_glds2)
static int g_syrk_ring = 0;   // r06 experiment: re-tiled SYRK with a 4-stage ring of 8-deep k-tiles (syrk_ring4_kernel; "syrkring")
static int g_syrk_xmap = 0;   // r06: SYRK work items in an XCD-aware dispatch order (syrk_xcd_order; "syrkxmap": 1 split-major, 2 item-major)

extern "C" {

// Runtime tuning switches (benchmarking aid): key "glds" = 0/1.
int PAMD_set_tuning(const char *key, int value)
{
    if (strcmp(key, "glds") == 0) { g_use_glds = value; return 0; }
    if (strcmp(key, "gemmwide") == 0) { g_gemm_wide = value; return 0; }
    if (strcmp(key, "dmav2") == 0) { g_dma_v2 = value; return 0; }
    if (strcmp(key, "syrkfrac") == 0) { g_syrk_frac = value; return 0; }
    if (strcmp(key, "syrkreserve") == 0 && value >= 0 && value < 256) { g_syrk_reserve = value; return 0; }
    if (strcmp(key, "syrkslots") == 0) { g_syrk_slots = value; return 0; }
    if (strcmp(key, "syrkprobe") == 0) { g_syrk_probe = value; return 0; }
    if (strcmp(key, "syrkxmap") == 0) { g_syrk_xmap = value; return 0; }
    if (strcmp(key, "syrkring") == 0) { g_syrk_ring = value; return 0; }
    if (strcmp(key, "numcu") == 0 && value > 0) { g_num_cu = value; return 0; }
    if (strcmp(key, "pkdma") == 0) { g_pk_dma = value; return 0; }
    if (strcmp(key, "pkdiag") == 0) { g_pk_diag = value; return 0; }
    if (strcmp(key, "e2merge") == 0) { g_e2_merge = value; return 0; }
    if (strcmp(key, "e2wide") == 0) { g_e2_wide = value; return 0; }
    if (strcmp(key, "e2xmap") == 0) { g_e2_xmap = value; return 0; }
    if (strcmp(key, "sqshift") == 0) { g_sq_shift = value; return 0; }
    if (strcmp(key, "pairtail") == 0) { g_pair_tail = value; return 0; }
    if (strcmp(key, "j2wg") == 0 && value >= 0) { g_j2_maxwg = value; return 0; }
    if (strcmp(key, "j2wide") == 0) { g_j2_wide = value; return 0; }
    if (strcmp(key, "mfmaprio") == 0) { g_mfma_prio = value; return 0; }
    if (strcmp(key, "e2prio") == 0) { g_e2_prio = value; return 0; }
    if (strcmp(key, "e2mt") == 0 && value >= 1 && value <= 10) { g_e2_mtmax = value; return 0; }
    return pamd::set_error(-3, "unknown tuning key", __FILE__, __LINE__);
}

long PAMD_df_vj_pass1_worksize(long npair, int naux, int nset)
{
    return (long)nset * naux * ceil_div(npair, J1_CHUNK2);
}

int PAMD_df_vj_pass1(const double *d_cderi, long npair, int naux, const double *d_dmtril, int nset,
                     double *d_rho, double *d_work, void *stream)
{
    PAMD_REQUIRE(nset >= 1 && nset <= MAX_NSET, "nset must be 1..4 per call");
    if (naux == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int nchunk = ceil_div(npair, J1_CHUNK2);
    dim3 grid(nchunk, ceil_div(naux, J1R));
    switch (nset) {
    case 1: vj_pass1_rows_kernel<1><<<grid, J1_THREADS, 0, st>>>(d_cderi, npair, naux, d_dmtril, d_work, nchunk); break;
    case 2: vj_pass1_rows_kernel<2><<<grid, J1_THREADS, 0, st>>>(d_cderi, npair, naux, d_dmtril, d_work, nchunk); break;
    case 3: vj_pass1_rows_kernel<3><<<grid, J1_THREADS, 0, st>>>(d_cderi, npair, naux, d_dmtril, d_work, nchunk); break;
    default: vj_pass1_rows_kernel<4><<<grid, J1_THREADS, 0, st>>>(d_cderi, npair, naux, d_dmtril, d_work, nchunk); break;
    }
    PAMD_CHECK_LAUNCH();
    int n = nset * naux;
    vj_pass1_reduce_kernel<false><<<ceil_div(n, 4), 256, 0, st>>>(d_work, d_rho, n, nchunk);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_df_vj_pass2(const double *d_cderi, long npair, int naux, const double *d_rho, int nset,
                     double *d_vjtril, void *stream)
{
    PAMD_REQUIRE(nset >= 1 && nset <= MAX_NSET, "nset must be 1..4 per call");
    if (naux == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (g_j2_wide && npair % 2 == 0 && ((uintptr_t)d_cderi % 16 == 0)) {
        int grid = ceil_div(npair / 2, 256);
        if (g_j2_maxwg > 0 && grid > g_j2_maxwg) grid = g_j2_maxwg;
        switch (nset) {
        case 1: vj_pass2_wide_kernel<1><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
        case 2: vj_pass2_wide_kernel<2><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
        case 3: vj_pass2_wide_kernel<3><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
        default: vj_pass2_wide_kernel<4><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
        }
        PAMD_CHECK_LAUNCH();
        return 0;
    }
    int grid = ceil_div(npair, 256);
    if (g_j2_maxwg > 0 && grid > g_j2_maxwg) grid = g_j2_maxwg;
    switch (nset) {
    case 1: vj_pass2_kernel<1><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
    case 2: vj_pass2_kernel<2><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
    case 3: vj_pass2_kernel<3><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
    default: vj_pass2_kernel<4><<<grid, 256, 0, st>>>(d_cderi, npair, naux, d_rho, d_vjtril); break;
    }
    PAMD_CHECK_LAUNCH();
    return 0;
}

// r06: the J passes of PAMD_df_vj_pass1 / _pass2 on the SQUARE layout d_sq[naux][rows][ld] (lstride = rows * ld doubles per aux
// row; the p >= q runs are read, dmtril / vjtril stay packed).  d_work as PAMD_df_vj_pass1_worksize(npair, naux, nset).
int PAMD_df_vj_pass1_sq(const double *d_sq, long lstride, int ld, int nao, int naux, const double *d_dmtril, int nset,
                        double *d_rho, double *d_work, void *stream)
{
    PAMD_REQUIRE(nset >= 1 && nset <= MAX_NSET, "nset must be 1..4 per call");
    PAMD_REQUIRE(ld >= nao && lstride >= (long)nao * ld, "PAMD_df_vj_pass1_sq: leading dimensions");
    if (naux == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const long npair = (long)nao * (nao + 1) / 2;
    int nchunk = ceil_div(npair, J1_CHUNK2);
    dim3 grid(nchunk, ceil_div(naux, J1R));
    switch (nset) {
    case 1: vj_pass1_sq_kernel<1><<<grid, J1_THREADS, 0, st>>>(d_sq, lstride, ld, npair, naux, d_dmtril, d_work, nchunk); break;
    case 2: vj_pass1_sq_kernel<2><<<grid, J1_THREADS, 0, st>>>(d_sq, lstride, ld, npair, naux, d_dmtril, d_work, nchunk); break;
    case 3: vj_pass1_sq_kernel<3><<<grid, J1_THREADS, 0, st>>>(d_sq, lstride, ld, npair, naux, d_dmtril, d_work, nchunk); break;
    default: vj_pass1_sq_kernel<4><<<grid, J1_THREADS, 0, st>>>(d_sq, lstride, ld, npair, naux, d_dmtril, d_work, nchunk); break;
    }
    PAMD_CHECK_LAUNCH();
    int n = nset * naux;
    vj_pass1_reduce_kernel<false><<<ceil_div(n, 4), 256, 0, st>>>(d_work, d_rho, n, nchunk);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_df_vj_pass2_sq(const double *d_sq, long lstride, int ld, int nao, int naux, const double *d_rho, int nset,
                        double *d_vjtril, void *stream)
{
    PAMD_REQUIRE(nset >= 1 && nset <= MAX_NSET, "nset must be 1..4 per call");
    PAMD_REQUIRE(ld >= nao && lstride >= (long)nao * ld, "PAMD_df_vj_pass2_sq: leading dimensions");
    if (naux == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    long grid = sq_units(nao);
    if (g_j2_maxwg > 0 && grid > g_j2_maxwg) grid = g_j2_maxwg;
    switch (nset) {
    case 1: vj_pass2_sq_kernel<1><<<(unsigned)grid, 256, 0, st>>>(d_sq, lstride, ld, nao, naux, d_rho, d_vjtril); break;
    case 2: vj_pass2_sq_kernel<2><<<(unsigned)grid, 256, 0, st>>>(d_sq, lstride, ld, nao, naux, d_rho, d_vjtril); break;
    case 3: vj_pass2_sq_kernel<3><<<(unsigned)grid, 256, 0, st>>>(d_sq, lstride, ld, nao, naux, d_rho, d_vjtril); break;
    default: vj_pass2_sq_kernel<4><<<(unsigned)grid, 256, 0, st>>>(d_sq, lstride, ld, nao, naux, d_rho, d_vjtril); break;
    }
    PAMD_CHECK_LAUNCH();
    return 0;
}

// One column slab of the build into the square layout: d_slab[nL][ncol] holds the packed columns [r0, r0 + ncol) = the AO rows
// [p0, p1) of every aux row (what PAMD_cderi_solve writes with ldc = ncol); both triangles of d_sq[nL][rows][ld] are filled.
int PAMD_unpack_tril_slab(const double *d_slab, long ncol, int nL, int p0, int p1, double *d_sq, int ld, long lstride, void *stream)
{
    PAMD_REQUIRE(p0 >= 0 && p1 >= p0 && ld >= p1 && ncol >= (long)p1 * (p1 + 1) / 2 - (long)p0 * (p0 + 1) / 2, "PAMD_unpack_tril_slab: bad slab");
    if (nL == 0 || p1 == p0) return 0;
    const long r0 = (long)p0 * (p0 + 1) / 2;
    dim3 grid(ceil_div(p1, 32), ceil_div(p1 - p0, 32), nL);
    unpack_slab_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_slab, ncol, r0, p0, p1, d_sq, ld, lstride);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// d_tril[count][nao_pair] (the reference's packed `_cderi` rows, pyscf/df/df.py:59-72) from d_sq[count][rows][ld]
int PAMD_pack_tril_rows(const double *d_sq, long lstride, int ld, int nao, int count, double *d_tril, void *stream)
{
    if (count == 0) return 0;
    const long npair = (long)nao * (nao + 1) / 2;
    dim3 grid(ceil_div(nao, 256), nao, count);
    pack_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_sq, lstride, ld, nao, d_tril, npair);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// Device analogue of AO2MOnr_e2_drv(ftrans=AO2MOtranse2_nr_s2, fmmm=AO2MOmmm_bra_nr_s2)
// (pyscf/lib/ao2mo/nr_ao2mo.c:1240-1266): out[L][i][p] = sum_q unpack(cderi[L])[p][q] orb[q][i].
//   d_orb   [orb_rows][ldo] row-major, columns >= norb zero up to ldo (PAMD_e2_orb_ld(nocc_pad) columns: whole kernel chunks);
//           orb_rows >= nao allocated rows (rows >= nao zero) - round_up(nao,16) enables the LDS-DMA kernel
//   d_out   [nL][nocc_pad][ldx]: nocc_pad rows per aux index - any value >= norb (r04: no longer a multiple of 16, so that the
//           K = X^T X that follows contracts nL * norb rows, not nL * round_up(norb, 16): 6 % of the SYRK at norb = 226)
//   d_rho   (nullable) [nL]: d_rho[L] += sum_{i,p} X[L][i][p] orb[p][i] (first J pass of the density orb orb^T)
// Doubles of workspace the fused first J pass needs (d_rho_work): one partial per wave of every workgroup of a row.
// Orbital tiling of the v2 half-transform kernels: chunks of 160 (NA = 5) or 128 (NA = 4) orbitals, whichever pads
// nocc_pad less (ties: 160, the shape with the better MFMA : fragment-load ratio); *waste = padded / exact - 1.
static void v2_tile(int nocc_pad, int *na, int *nchunk, double *waste)
{
    const int c160 = ceil_div(nocc_pad, 160) * 160, c128 = ceil_div(nocc_pad, 128) * 128;
    if (c128 < c160) { *na = 4; *nchunk = c128 / 128; *waste = (double)c128 / nocc_pad - 1.0; }
    else             { *na = 5; *nchunk = c160 / 160; *waste = (double)c160 / nocc_pad - 1.0; }
}
// r04: 128-orbital chunks with a LAST chunk of wm < 8 tiles in the 1 x 4 wave arrangement (e2_sq2 / e2_pk, WM > 0).  Returns
// true when that costs fewer MFMA tiles than the best uniform tiling; a last chunk is never narrower than 4 tiles (below
// that the panel DMA per MFMA doubles again and the exact-tile kernels are the better choice).
static bool v2_wide(int nocc_pad, int *nchunk, int *wm, double *waste)
{
    const int T = ceil_div(nocc_pad, 16), nch = ceil_div(T, 8), r = T - 8 * (nch - 1);
    if (!g_e2_wide || r == 8) return false;
    const int w = r < 4 ? 4 : r, cost = 8 * (nch - 1) + w;
    const int c160 = ceil_div(nocc_pad, 160) * 10, c128 = ceil_div(nocc_pad, 128) * 8;
    if (cost >= (c128 < c160 ? c128 : c160)) return false;
    *nchunk = nch; *wm = w; *waste = (double)cost * 16 / nocc_pad - 1.0;
    return true;
}
// padded MFMA work a v2 kernel may spend before the exact-tile kernels (lower matrix-pipe efficiency) are the better choice
static const double V2_WASTE_SQUARE = 0.13, V2_WASTE_PACKED = 0.30;

// r06 (first step): the ONE copy of the K = X^T X plan - tile shape and k splits.  Both orchestrations call it
// (df_jk.syrk_plan of the Python layer, df_get_jk_impl of the C handle); before, each carried its own transcription of the rule
// and the two had already drifted apart for small matrices with an odd block count.
//   flags_in < 0: the default - the RE-TILED triangle (flag 8: work items of four live 64 x 64 wave blocks) with the BALANCED k
//   split (flag 4) when the matrix has an odd number >= 5 of 64-column blocks, else 128 x 128 tiles; nsplit_in > 0 wins.
//   Balanced split: n full pieces + one 1/m-length remainder piece per work item such that units (n + ceil(1/m)) fits the 512
//   workgroup slots minus `reserve` (slots left to a co-running second J pass); needs >= 32 work items, else 4 uniform splits.
// Returns flags (bit 0 lower triangle, bit 1 LDS-DMA operands always set) and the number of partial-sum slabs.
int PAMD_syrk_item_count(int nao)
{
    const int nb = ceil_div(nao, 64);
    if (nb % 2 == 0 || nb < 5) return 0;
    const int nt = nb / 2;
    return nt * (nt - 1) / 2 + nt + ceil_div(nt, 3);
}

int PAMD_syrk_plan(int nao, int reserve, int flags_in, int nsplit_in, int *flags_out, int *nsplit_out)
{
    PAMD_REQUIRE(flags_out && nsplit_out && nao > 0, "PAMD_syrk_plan: arguments");
    const int items = PAMD_syrk_item_count(nao);
    const int flags = flags_in < 0 ? (items ? 12 : 0) : flags_in;
    const int base = 1 | 2 | flags;
    if (nsplit_in > 0) { *flags_out = base; *nsplit_out = nsplit_in; return 0; }
    if (flags & 4) {
        const int nt = ceil_div(nao, 128);
        const int units = ((flags & 8) && items) ? items : nt * (nt + 1) / 2;
        double best = 0;
        int bn = 0;
        for (int n = 1; n < 8; n++)
            for (int m = 1; m < 9; m++)
                if (units * n + ceil_div(units, m) <= 512 - reserve && n + 1.0 / m > best) { best = n + 1.0 / m; bn = n; }
        if (bn && units >= 32) { *flags_out = base; *nsplit_out = bn + 1; return 0; }
    }
    *flags_out = base & ~4;
    *nsplit_out = 4;
    return 0;
}

// Which layout the tensor gets, from what each candidate needs and what is free (bytes; each host layer prices its own work
// space and reserve): 2 = packed rows + a FULL square image while all three copies fit (`need_packed_image`, 0 = not on offer) - the
// fastest second J pass beside the co-running SYRK, measured; 1 = the square rows alone when that fits both while it is built and
// afterwards (`need_square_build`, `need_square_after`; 0 = not on offer); 0 = packed rows and whatever partial image is left.
// The ORDER of preference in one place for df.DF._choose_layout and the C handle's build_rows (r06).
int PAMD_df_layout_pick(long long need_packed_image, long long need_square_build, long long need_square_after, long long free_bytes,
                        int prefer_image)
{
    if (prefer_image && need_packed_image > 0 && need_packed_image <= free_bytes) return 2;
    if (need_square_build > 0 && need_square_build <= free_bytes && need_square_after <= free_bytes) return 1;
    return 0;
}

// Which schedule of the second J pass to keep, from the best time of each candidate (ms[0] overlap: side stream beside the SYRK,
// ms[1] serial: in line before the re-tiled SYRK, ms[2] fused: inside the SYRK kernel; ncand = 2 without the last): a challenger
// must win by 1 % (the trials are noisy at that level and 'overlap' is the schedule the kernels were tuned beside).  One rule for
// df_jk.get_jk_device's trials and the C handle's (r06).
int PAMD_j2_schedule_pick(const double *ms, int ncand)
{
    if (!ms || ncand < 2) return 0;
    int best = ms[1] < 0.99 * ms[0] ? 1 : 0;
    if (ncand > 2 && ms[2] < 0.99 * ms[best]) best = 2;
    return best;
}

// Aux rows per half-transform block of the MO branch: the X block ([blk][rows_per_aux][ldx] doubles) sized against `budget_bytes`
// of HBM like the reference's `blksize` against max_memory (pyscf/df/df_jk.py:359-360), then EQUAL blocks (each block's second J
// pass hides behind one SYRK).  The one rule of both host layers (r06).
long PAMD_k_block_rows(long naux, int rows_per_aux, int ldx, long long budget_bytes)
{
    const long n = naux > 1 ? naux : 1;
    long blk = (long)(budget_bytes / ((long long)(rows_per_aux > 0 ? rows_per_aux : 1) * (ldx > 0 ? ldx : 1) * 8));
    blk = blk < 1 ? 1 : (blk > n ? n : blk);
    const long nblk = (n + blk - 1) / blk;
    return (n + nblk - 1) / nblk;
}

// Leading dimension (columns, zero beyond the orbitals) the half-transform kernels want for nocc_pad orbital columns: whole
// chunks of the exact-tile kernels, of the uniform v2 tilings and - r04 - of the 128-column chunks with a wide last chunk.
int PAMD_e2_orb_ld(int nocc_pad)
{
    if (nocc_pad <= 0) return 0;
    const int p16 = ceil_div(nocc_pad, 16) * 16;
    int ldo = p16 > 160 ? ceil_div(p16, 160) * 160 : p16;
    const int mt = p16 / 16, nchunk = ceil_div(mt, 10);
    ldo = max(ldo, nchunk * ceil_div(ceil_div(mt, nchunk), 2) * 32);
    ldo = max(ldo, min(ceil_div(p16, 160) * 160, ceil_div(p16, 128) * 128));
    int nchw, wmw;
    double wastew;
    if (v2_wide(p16, &nchw, &wmw, &wastew) && wastew <= V2_WASTE_PACKED) ldo = max(ldo, nchw * 128);
    return ldo;
}

long PAMD_nr_e2_rho_worksize(int nL, int ldx, int nocc_pad)
{
    int nchunk = ceil_div(ceil_div(nocc_pad, 16), g_e2_mtmax);      // g_e2_mtmax <= 10 = the square kernel's chunking
    if (ceil_div(nocc_pad, 128) > nchunk) nchunk = ceil_div(nocc_pad, 128);
    return (long)nL * ceil_div(ldx, NT) * nchunk * 4;
}

static int reduce_rho_partials(const double *d_work, double *d_rho, int nL, int nslot, hipStream_t st)
{
    vj_pass1_reduce_kernel<true><<<ceil_div(nL, 4), 256, 0, st>>>(d_work, d_rho, nL, nslot);
    PAMD_CHECK_LAUNCH();
    return 0;
}

static int nr_e2_symm_impl(const double *d_cderi, long npair, int nL, int nao, const double *d_orb, int ldo,
                           int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work, void *stream,
                           const double *d_diag)
{
    PAMD_REQUIRE(d_rho == nullptr || d_rho_work != nullptr, "d_rho needs d_rho_work (PAMD_nr_e2_rho_worksize doubles)");
    PAMD_REQUIRE(nocc_pad >= 0 && nocc_pad <= ldo, "nocc_pad (rows of d_out per aux index) must be <= ldo");
    PAMD_REQUIRE(ldx >= nao, "ldx < nao");
    if (nL == 0 || nocc_pad == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int mt_total = ceil_div(nocc_pad, 16);
    {
        // all-DMA packed-operand kernel: 160- or 128-orbital chunks (as PAMD_nr_e2_square), q rows padded to a multiple of 16
        int na, nch, wm = 0;
        double waste;
        v2_tile(nocc_pad, &na, &nch, &waste);
        {
            int nchw, wmw;
            double wastew;
            if (v2_wide(nocc_pad, &nchw, &wmw, &wastew) && ldo >= nchw * 128) { na = 4; nch = nchw; wm = wmw; waste = wastew; }
        }
        const int kdim = ceil_div(nao, KB) * KB;
        if (g_pk_dma && waste <= V2_WASTE_PACKED && ldo >= nch * na * 32 && orb_rows >= kdim && ldo % 2 == 0 && ldx <= kdim &&
            (long)kdim * (kdim + 1) / 2 * 8 + 4096 < (1L << 31) && (long)orb_rows * ldo * 8 < (1L << 32) &&
            ((uintptr_t)d_orb % 16 == 0) && ((uintptr_t)d_cderi % 8 == 0)) {
            dim3 gpk(ceil_div(ldx, NT) * nch, nL);
            double *rw = d_rho ? d_rho_work : nullptr;
            const int ntile_p = ceil_div(ldx, NT);
#define LAUNCH_PK(NAV, RHOF, DG) e2_pk_kernel<NAV, RHOF, DG><<<gpk, 256, 0, st>>>(d_cderi, npair, nL, kdim, d_orb, ldo, d_out, nocc_pad, ldx, rw, nch, nao, d_diag, ntile_p, g_e2_xmap)
#define LAUNCH_PKW(WMV, RHOF, DG) e2_pk_kernel<4, RHOF, DG, WMV><<<gpk, 256, 0, st>>>(d_cderi, npair, nL, kdim, d_orb, ldo, d_out, nocc_pad, ldx, rw, nch, nao, d_diag, ntile_p, g_e2_xmap)
#define LAUNCH_PKW_ALL(WMV)                                                                                   \
            do {                                                                                              \
                if (d_diag && g_pk_diag) { if (d_rho) LAUNCH_PKW(WMV, true, true); else LAUNCH_PKW(WMV, false, true); }     \
                else                     { if (d_rho) LAUNCH_PKW(WMV, true, false); else LAUNCH_PKW(WMV, false, false); }   \
            } while (0)
            if (wm) {
                switch (wm) {
                case 4: LAUNCH_PKW_ALL(4); break;
                case 5: LAUNCH_PKW_ALL(5); break;
                case 6: LAUNCH_PKW_ALL(6); break;
                default: LAUNCH_PKW_ALL(7); break;
                }
            } else
            if (d_diag && g_pk_diag) {
                if (na == 5) { if (d_rho) LAUNCH_PK(5, true, true); else LAUNCH_PK(5, false, true); }
                else         { if (d_rho) LAUNCH_PK(4, true, true); else LAUNCH_PK(4, false, true); }
            } else {
                if (na == 5) { if (d_rho) LAUNCH_PK(5, true, false); else LAUNCH_PK(5, false, false); }
                else         { if (d_rho) LAUNCH_PK(4, true, false); else LAUNCH_PK(4, false, false); }
            }
#undef LAUNCH_PK
#undef LAUNCH_PKW
#undef LAUNCH_PKW_ALL
            PAMD_CHECK_LAUNCH();
            if (d_rho) return reduce_rho_partials(d_rho_work, d_rho, nL, (int)(gpk.x * 4), st);
            return 0;
        }
    }
    int nchunk = ceil_div(mt_total, g_e2_mtmax);
    int mt = ceil_div(mt_total, nchunk);
    // the kernel writes rows i < nocc_pad only; chunks are mt*16 wide
    dim3 grid(ceil_div(ldx, NT), nL, nchunk);
#define LAUNCH_E2(MT)                                                                         \
    e2_symm_kernel<MT, false><<<grid, 256, 0, st>>>(d_cderi, npair, nao, d_orb, ldo, d_out, nocc_pad, ldx, 0, 0, nullptr, d_rho ? d_rho_work : nullptr)
    // orbital tile reads m0+i < ldo must stay in bounds: require ldo >= nchunk*mt*16
    PAMD_REQUIRE(ldo >= nchunk * mt * 16, "orbital leading dimension too small for tile padding");
    switch (mt) {
    case 1: LAUNCH_E2(1); break;
    case 2: LAUNCH_E2(2); break;
    case 3: LAUNCH_E2(3); break;
    case 4: LAUNCH_E2(4); break;
    case 5: LAUNCH_E2(5); break;
    case 6: LAUNCH_E2(6); break;
    case 7: LAUNCH_E2(7); break;
    case 8: LAUNCH_E2(8); break;
    case 9: LAUNCH_E2(9); break;
    default: LAUNCH_E2(10); break;
    }
#undef LAUNCH_E2
    PAMD_CHECK_LAUNCH();
    // the partial layout needs the MT actually launched to cover the chunk count used by the worksize query
    if (d_rho) return reduce_rho_partials(d_rho_work, d_rho, nL, (int)(grid.x * grid.z * 4), st);
    return 0;
}

int PAMD_nr_e2_symm(const double *d_cderi, long npair, int nL, int nao, const double *d_orb, int ldo,
                    int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work, void *stream)
{
    return nr_e2_symm_impl(d_cderi, npair, nL, nao, d_orb, ldo, orb_rows, nocc_pad, d_out, ldx, d_rho, d_rho_work, stream, nullptr);
}

// Same, with the diagonal-block side image of the same aux rows (PAMD_e2_diag_blocks): d_diag[nL][ceil(ldx/128)][128][128].
// ldx must be the value the image was built for (its block count is part of the layout).
int PAMD_nr_e2_symm_diag(const double *d_cderi, long npair, int nL, int nao, const double *d_orb, int ldo,
                         int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work,
                         const double *d_diag, void *stream)
{
    PAMD_REQUIRE(d_diag == nullptr || (uintptr_t)d_diag % 16 == 0, "d_diag must be 16-byte aligned");
    return nr_e2_symm_impl(d_cderi, npair, nL, nao, d_orb, ldo, orb_rows, nocc_pad, d_out, ldx, d_rho, d_rho_work, stream, d_diag);
}

// diag[L][P][r][c] = B_L[128 P + r][128 P + c] (both triangles; 0 where an index is >= nao): one workgroup per (P, L),
// reads the 128 packed row pieces once (coalesced along c <= r) and writes both images through LDS
__global__ __launch_bounds__(256) void e2_diag_blocks_kernel(const double *__restrict__ cderi, long npair, int nao,
                                                             double *__restrict__ diag)
{
    __shared__ double t[NT][NT + 1];
    const int P = blockIdx.x;
    const long L = blockIdx.y;
    const double *src = cderi + L * npair;
    const int p0 = P * NT;
    for (int e = threadIdx.x; e < NT * NT; e += 256) {
        const int r = e >> 7, c = e & 127;
        const long pr = p0 + r, pc = p0 + c;
        double v = 0.0;
        if (c <= r && pr < nao) v = src[pr * (pr + 1) / 2 + pc];
        t[r][c] = v;
    }
    __syncthreads();
    double *dst = diag + (L * gridDim.x + P) * (long)(NT * NT);
    for (int e = threadIdx.x; e < NT * NT; e += 256) {
        const int r = e >> 7, c = e & 127;
        dst[e] = c <= r ? t[r][c] : t[c][r];
    }
}

long PAMD_e2_diag_size(int nL, int ldx) { return (long)nL * ceil_div(ldx, NT) * NT * NT; }

int PAMD_e2_diag_blocks(const double *d_cderi, long npair, int nL, int nao, int ldx, double *d_diag, void *stream)
{
    PAMD_REQUIRE(ldx >= nao, "ldx < nao");
    if (nL == 0) return 0;
    e2_diag_blocks_kernel<<<dim3(ceil_div(ldx, NT), nL), 256, 0, (hipStream_t)stream>>>(d_cderi, npair, nao, d_diag);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// X[L][i][p] = sum_q orb[q][i] sq[L][q][p] on the unpacked (square) image sq[nL][rows][ld] of the tensor
// (PAMD_unpack_tril with rows = round_up(nao, 16) into a zero-initialised buffer): the same contraction as
// PAMD_nr_e2_symm (AO2MOnr_e2_drv + dsymm, pyscf/df/df_jk.py:373-379), LDS-DMA on both operands.
//   d_sq   [nL][rows][ld], rows % 16 == 0, rows >= nao, ld even, ld >= nao, 256 doubles of slack after the buffer
//   d_orb  [orb_rows >= rows][ldo] zero rows beyond nao, ldo even, ldo >= chunks * tile width (see PAMD_e2_sq_ldo)
//   d_rho (nullable) [nL]: d_rho[L] += sum_{i,p} X[L][i][p] orb[p][i] = sum_pq B_L[p][q] (orb orb^T)[p][q], the first J pass
//   of the density the orbitals stand for, taken from the accumulators in the epilogue
// r06: `lstride` doubles between consecutive aux rows of d_sq (>= rows * ld, even).  The square LAYOUT pads it: rows * ld * 8 is a
// multiple of 32 KB at nao = 1856 / 2228 and of 8 MB at nao = 3072, so that the same (p, q) of consecutive aux rows - what the J
// passes and the workgroups of one column tile read at the same time - would all sit on one HBM channel.
int PAMD_nr_e2_square_ls(const double *d_sq, long ld, int rows, long lstride, int nL, int nao, const double *d_orb, int ldo,
                         int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work, void *stream)
{
    PAMD_REQUIRE(lstride >= (long)rows * ld && lstride % 2 == 0, "PAMD_nr_e2_square_ls: lstride");
    PAMD_REQUIRE(d_rho == nullptr || d_rho_work != nullptr, "d_rho needs d_rho_work (PAMD_nr_e2_rho_worksize doubles)");
    PAMD_REQUIRE(nocc_pad >= 0 && nocc_pad <= ldo, "nocc_pad (rows of d_out per aux index) must be <= ldo");
    PAMD_REQUIRE(rows % KB == 0 && rows >= nao && orb_rows >= rows, "q rows must be padded to a multiple of 16");
    PAMD_REQUIRE(ld % 2 == 0 && ldo % 2 == 0 && ld >= nao && ldx >= nao, "leading dimensions");
    PAMD_REQUIRE(((uintptr_t)d_sq | (uintptr_t)d_orb) % 16 == 0, "16-byte aligned operands");
    if (nL == 0 || nocc_pad == 0) return 0;
    d_sq += g_sq_shift;            // (probe: misaligned DMA source; results are then meaningless)
    hipStream_t st = (hipStream_t)stream;
    const int mt_total = ceil_div(nocc_pad, 16);               // orbital MFMA tiles
    const int nchunk = ceil_div(mt_total, 10);
    const int wa = ceil_div(ceil_div(mt_total, nchunk), 2);   // MFMA tiles per wave row; workgroup covers 2 wa tiles
    PAMD_REQUIRE(ldo >= nchunk * wa * 32, "orbital leading dimension too small for tile padding");
    dim3 grid(ceil_div(ldx, NT) * nchunk, nL);
    PAMD_REQUIRE((long)rows * ld * 8 < (1L << 31) && (long)orb_rows * ldo * 8 < (1L << 32), "panel offsets exceed 32 bits");
    int na2, nch2, wm2 = 0;
    double waste2;
    v2_tile(nocc_pad, &na2, &nch2, &waste2);
    {
        int nchw, wmw;
        double wastew;
        if (v2_wide(nocc_pad, &nchw, &wmw, &wastew) && ldo >= nchw * 128) { na2 = 4; nch2 = nchw; wm2 = wmw; waste2 = wastew; }
    }
    if (g_dma_v2 && waste2 <= V2_WASTE_SQUARE && ldo >= nch2 * na2 * 32) {
        // 160- or 128-orbital chunks: v2 kernel.  A last column tile that is at most half full runs as a second launch over
        // pairs of aux rows (PAIR instance); rho partials of both launches share one [nL][nslot][4] layout.
        const int nchunk = nch2;
        const int ptiles = ceil_div(ldx, NT);
        const int nslot = ptiles * nchunk;
        const int last_valid = nao - (ptiles - 1) * NT;
        const bool pair = g_pair_tail && ptiles > 1 && last_valid <= 64 && nL >= 2;
        const int pmain = pair ? ptiles - 1 : ptiles;
        dim3 gmain(pmain * nchunk, nL), gpair(nchunk, ceil_div(nL, 2));
        const bool merged = pair && g_e2_merge;
        dim3 gboth(pmain * nchunk, nL + ceil_div(nchunk * ceil_div(nL, 2), pmain * nchunk));
        double *rw = d_rho ? d_rho_work : nullptr;
#define LAUNCH_V2(NAV, RHOF)                                                                                        \
        do {                                                                                                         \
            if (merged) {                                                                                            \
                e2_sq2_kernel<NAV, RHOF, 2><<<gboth, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out, \
                                                                   nocc_pad, ldx, rw, nchunk, nao, ptiles - 1, nslot, nL, g_e2_prio, g_e2_xmap); \
                break;                                                                                               \
            }                                                                                                        \
            e2_sq2_kernel<NAV, RHOF, 0><<<gmain, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out,   \
                                                                   nocc_pad, ldx, rw, nchunk, nao, 0, nslot, nL, g_e2_prio, g_e2_xmap);    \
            if (pair)                                                                                                \
                e2_sq2_kernel<NAV, RHOF, 1><<<gpair, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out, \
                                                                      nocc_pad, ldx, rw, nchunk, nao, ptiles - 1, nslot, nL, g_e2_prio, g_e2_xmap); \
        } while (0)
#define LAUNCH_W(WMV, RHOF)                                                                                          \
        do {                                                                                                         \
            if (merged)                                                                                              \
                e2_sq2_kernel<4, RHOF, 2, WMV><<<gboth, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out, \
                                                                   nocc_pad, ldx, rw, nchunk, nao, ptiles - 1, nslot, nL, g_e2_prio, g_e2_xmap); \
            else                                                                                                     \
                e2_sq2_kernel<4, RHOF, 0, WMV><<<dim3(ptiles * nchunk, nL), 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, \
                                                                   d_out, nocc_pad, ldx, rw, nchunk, nao, 0, nslot, nL, g_e2_prio, g_e2_xmap); \
        } while (0)
        if (wm2) {                          // (without the merged pair rows every column tile, also a half-empty last one, is a main tile)
            switch (wm2) {
            case 4: if (d_rho) LAUNCH_W(4, true); else LAUNCH_W(4, false); break;
            case 5: if (d_rho) LAUNCH_W(5, true); else LAUNCH_W(5, false); break;
            case 6: if (d_rho) LAUNCH_W(6, true); else LAUNCH_W(6, false); break;
            default: if (d_rho) LAUNCH_W(7, true); else LAUNCH_W(7, false); break;
            }
        }
        else if (na2 == 5) { if (d_rho) LAUNCH_V2(5, true); else LAUNCH_V2(5, false); }
        else          { if (d_rho) LAUNCH_V2(4, true); else LAUNCH_V2(4, false); }
#undef LAUNCH_W
#undef LAUNCH_V2
        PAMD_CHECK_LAUNCH();
        if (d_rho) return reduce_rho_partials(d_rho_work, d_rho, nL, nslot * 4, st);
        return 0;
    }
#define LAUNCH_SQ(W)                                                                                            \
    do {                                                                                                        \
        if (d_rho)                                                                                              \
            e2_sq_kernel<W, true><<<grid, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out, nocc_pad, \
                                                        ldx, d_rho_work, nchunk);                                \
        else                                                                                                    \
            e2_sq_kernel<W, false><<<grid, 256, 0, st>>>(d_sq, ld, lstride, rows, d_orb, ldo, d_out, nocc_pad, \
                                                         ldx, nullptr, nchunk);                                  \
    } while (0)
    switch (wa) {
    case 1: LAUNCH_SQ(1); break;
    case 2: LAUNCH_SQ(2); break;
    case 3: LAUNCH_SQ(3); break;
    case 4: LAUNCH_SQ(4); break;
    default: LAUNCH_SQ(5); break;
    }
#undef LAUNCH_SQ
    PAMD_CHECK_LAUNCH();
    if (d_rho) return reduce_rho_partials(d_rho_work, d_rho, nL, (int)(grid.x * 4), st);
    return 0;
}

int PAMD_nr_e2_square(const double *d_sq, long ld, int rows, int nL, int nao, const double *d_orb, int ldo,
                      int orb_rows, int nocc_pad, double *d_out, int ldx, double *d_rho, double *d_rho_work, void *stream)
{
    return PAMD_nr_e2_square_ls(d_sq, ld, rows, (long)rows * ld, nL, nao, d_orb, ldo, orb_rows, nocc_pad, d_out, ldx, d_rho, d_rho_work, stream);
}

// out[y][i][n] = sum_k src_y[n][k] * orb[k][i]   (y < ny; src_y = d_src + y*src_stride, rows n of
// leading dimension lds, k contiguous).  Same MFMA kernel as PAMD_nr_e2_symm with a plain operand;
// numint's c = ao . C_occ (pyscf/dft/numint.py:328-469, VXCdot_ao_dm_sparse) in the [orbital][grid] layout.
// d_kmask (nullable): [ny][ceil(nrows/128)][ceil(kdim/16)] bytes, 0 = skip that 128 x 16 tile of src.
int PAMD_orb_dot_rows(const double *d_src, long lds, long src_stride, int ny, long nrows, int kdim,
                      const double *d_orb, int ldo, int nocc_pad, double *d_out, long ldout,
                      const unsigned char *d_kmask, void *stream)
{
    PAMD_REQUIRE(nocc_pad % 16 == 0 && nocc_pad <= ldo, "nocc_pad must be a multiple of 16 and <= ldo");
    if (ny == 0 || nrows == 0 || nocc_pad == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int mt_total = nocc_pad / 16;
    int nchunk = ceil_div(mt_total, 10);
    int mt = ceil_div(mt_total, nchunk);
    PAMD_REQUIRE(ldo >= nchunk * mt * 16, "orbital leading dimension too small for tile padding");
    dim3 grid(ceil_div(nrows, NT), ny, nchunk);
#define LAUNCH_P(MT)                                                                          \
    e2_symm_kernel<MT, true><<<grid, 256, 0, st>>>(d_src, lds, kdim, d_orb, ldo, d_out, nocc_pad, ldout, src_stride, nrows, d_kmask, nullptr)
    switch (mt) {
    case 1: LAUNCH_P(1); break;
    case 2: LAUNCH_P(2); break;
    case 3: LAUNCH_P(3); break;
    case 4: LAUNCH_P(4); break;
    case 5: LAUNCH_P(5); break;
    case 6: LAUNCH_P(6); break;
    case 7: LAUNCH_P(7); break;
    case 8: LAUNCH_P(8); break;
    case 9: LAUNCH_P(9); break;
    default: LAUNCH_P(10); break;
    }
#undef LAUNCH_P
    PAMD_CHECK_LAUNCH();
    return 0;
}

// C[s][m][ldc] += A[k][m]^T B[k][n] over the s-th k range; s in [0,nsplit).  Device analogue of
// lib.dot(buf1.T, buf1) (pyscf/df/df_jk.py:380; NPdgemm, pyscf/lib/np_helper/npdot.c:32).
// Work items of syrk_slots_kernel for an m x m lower triangle with an ODD number nb of 64-column blocks (see the kernel's
// comment); cached on the device per m.  16 ints per item: 4 slot column offsets, then per wave {a | b << 8 | live << 16, row
// block, column block}.  nitems = 0: even nb (or a tiny matrix) - the caller keeps the 2 x 2 tiling.
static int syrk_items(int m, const int **d_items, int *nitems, std::vector<int> *sorted = nullptr)
{
    // one table per (device, m): a process may hold handles on several GPUs and call from one host thread per device
    static std::map<std::pair<int, int>, std::pair<int *, int>> cache;
    static std::map<int, std::vector<int>> sorted_cache;        // per m: item indices in tile-row order (syrk_xcd_order)
    static std::mutex cache_mutex;
    int dev = 0;
    PAMD_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> guard(cache_mutex);
    const std::pair<int, int> key(dev, m);
    auto hit = cache.find(key);
    if (hit != cache.end()) {
        *d_items = hit->second.first; *nitems = hit->second.second;
        if (sorted) *sorted = sorted_cache[m];
        return 0;
    }
    const int nb = ceil_div(m, 64);
    std::vector<int> tab;
    if (nb % 2 == 1 && nb >= 5) {
        const int nt = nb / 2, r = nb - 1;                   // full 128-tiles per side, the odd last block row
        auto item = [&](int c0, int c1, int c2, int c3, const int w[4][4]) {    // w[wave] = {a, b, rb, cb} or rb < 0: dead
            const int cc[4] = {c0 * 64, c1 * 64, c2 * 64, c3 * 64};
            tab.insert(tab.end(), cc, cc + 4);
            for (int q = 0; q < 4; q++) {
                const bool lv = w[q][2] >= 0;
                tab.push_back((lv ? (w[q][0] | (w[q][1] << 8) | (1 << 16)) : 0));
                tab.push_back(lv ? w[q][2] : 0);
                tab.push_back(lv ? w[q][3] : 0);
            }
        };
        for (int I = 0; I < nt; I++)
            for (int J = 0; J < I; J++) {
                const int w[4][4] = {{0, 2, 2 * I, 2 * J}, {0, 3, 2 * I, 2 * J + 1}, {1, 2, 2 * I + 1, 2 * J}, {1, 3, 2 * I + 1, 2 * J + 1}};
                item(2 * I, 2 * I + 1, 2 * J, 2 * J + 1, w);
            }
        for (int I = 0; I < nt; I++) {
            const int w[4][4] = {{0, 0, 2 * I, 2 * I}, {1, 0, 2 * I + 1, 2 * I}, {1, 1, 2 * I + 1, 2 * I + 1}, {2, 0, r, 2 * I}};
            item(2 * I, 2 * I + 1, r, r, w);
        }
        std::vector<int> rest;                               // (r, 2I+1) for every I, then the corner (r, r)
        for (int I = 0; I < nt; I++) rest.push_back(2 * I + 1);
        size_t pos = 0;
        while (pos < rest.size()) {
            int cs[3] = {-1, -1, -1};
            for (int q = 0; q < 3 && pos < rest.size(); q++) cs[q] = rest[pos++];
            const bool corner = pos >= rest.size();          // the last item also takes the corner block
            const int w[4][4] = {{0, 1, cs[0] >= 0 ? r : -1, cs[0]}, {0, 2, cs[1] >= 0 ? r : -1, cs[1]},
                                 {0, 3, cs[2] >= 0 ? r : -1, cs[2]}, {0, 0, corner ? r : -1, r}};
            item(r, cs[0] >= 0 ? cs[0] : r, cs[1] >= 0 ? cs[1] : r, cs[2] >= 0 ? cs[2] : r, w);
            if (corner) break;
        }
    }
    int *d = nullptr;
    const int n = (int)(tab.size() / 16);
    if (n) {
        PAMD_CHECK_HIP(hipMalloc((void **)&d, tab.size() * sizeof(int)));
        PAMD_CHECK_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    {
        // tile-row order: by the first ("A") panel, then the "B" panel - the items of one tile row share their A panels
        std::vector<int> idx(n);
        for (int i = 0; i < n; i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
            const int ka = tab[(size_t)a * 16] * 4096 + tab[(size_t)a * 16 + 2] / 64, kb = tab[(size_t)b * 16] * 4096 + tab[(size_t)b * 16 + 2] / 64;
            return ka < kb;
        });
        sorted_cache[m] = idx;
        if (sorted) *sorted = idx;
    }
    cache[key] = {d, n};
    *d_items = d;
    *nitems = n;
    return 0;
}

// r06: XCD-aware dispatch order of the SYRK work items.  Workgroup b of a 1-D launch is placed on XCD b % 8
// (round-robin dispatch), every XCD has its own 4 MB L2 and 64 workgroup slots: the (item, split) units are laid out so that the
// units resident on one XCD at a time share X panels AND walk the same k rows in lockstep - their panel rows then hit in that L2.
//   mode 1  split-major: the full pieces, ordered (split, item sorted by tile row), are cut into 8 contiguous runs, one per XCD
//           (nao = 1856, 4 full pieces x 110 items: an XCD holds ONE k quarter of 55 items of 10-15 tile rows: 10.5-14.5 panels per
//           k-tile instead of 110); the short remainder pieces of the balanced split follow at the end of every XCD's list
//   mode 2  item-major: an XCD holds nitems / 8 consecutive items of every split (control: shared panels, no shared k range)
// sorted[] = the items in tile-row order; returns a device table order[8 * nmax] (item | split << 16, -1 = no work) cached per key.
static int syrk_xcd_order(int kind, int m, const std::vector<int> &sorted, int nsplit, bool last_short, int mode, const int **d_order,
                          int *nwg)
{
    static std::map<std::vector<int>, std::pair<int *, int>> cache;
    static std::mutex cache_mutex;
    int dev = 0;
    PAMD_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> guard(cache_mutex);
    const std::vector<int> key = {dev, kind, m, nsplit, last_short ? 1 : 0, mode, (int)sorted.size()};
    auto hit = cache.find(key);
    if (hit != cache.end()) { *d_order = hit->second.first; *nwg = hit->second.second; return 0; }
    const int nitems = (int)sorted.size(), nfull = last_short ? nsplit - 1 : nsplit, NX = 8;
    std::vector<std::vector<int>> lists(NX);
    const long units = (long)nitems * nfull;
    if (mode == 2) {
        for (int s = 0; s < nfull; s++)
            for (int i = 0; i < nitems; i++) lists[(long)i * NX / nitems].push_back(sorted[i] | (s << 16));
    } else {
        for (long u = 0; u < units; u++) {
            const int s = (int)(u / nitems), i = (int)(u % nitems);
            lists[u * NX / units].push_back(sorted[i] | (s << 16));
        }
    }
    if (last_short)
        for (int i = 0; i < nitems; i++) lists[(long)i * NX / nitems].push_back(sorted[i] | ((nsplit - 1) << 16));
    size_t nmax = 0;
    for (auto &l : lists) nmax = std::max(nmax, l.size());
    std::vector<int> ord(NX * nmax, -1);
    for (int x = 0; x < NX; x++)
        for (size_t i = 0; i < lists[x].size(); i++) ord[i * NX + x] = lists[x][i];
    int *d = nullptr;
    PAMD_CHECK_HIP(hipMalloc((void **)&d, ord.size() * sizeof(int)));
    PAMD_CHECK_HIP(hipMemcpy(d, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice));
    cache[key] = {d, (int)ord.size()};
    *d_order = d;
    *nwg = (int)ord.size();
    return 0;
}

static int dgemm_tn_impl(const double *d_A, int lda, const double *d_B, int ldb, double *d_C, int ldc,
                         int m, int n, long k, int lower_only, int nsplit, const unsigned char *d_maskA,
                         const unsigned char *d_maskB, void *stream, JStream *js = nullptr)
{
    PAMD_REQUIRE(nsplit >= 1, "nsplit >= 1");
    PAMD_REQUIRE(!(lower_only & 1) || m == n, "lower_only needs a square result");
    if (m == 0 || n == 0 || k == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    // flags bit 1 (value 2): caller guarantees 160 readable doubles from every row start and
    // k ranges that are multiples of 16 -> LDS-DMA kernel
    long kchunk = ((k + nsplit - 1) / nsplit + KB - 1) / KB * KB;
    const bool aligned = (lda % 2 == 0) && (ldb % 2 == 0) && (((uintptr_t)d_A | (uintptr_t)d_B) % 16 == 0) &&
                         (k % KB == 0) && (kchunk % KB == 0);
    const bool glds = (lower_only & 2) && aligned && (g_use_glds || d_maskA);
    // full (non-symmetric), unscreened products take 160 x 128 tiles: 80 instead of 64 MFMAs per wave between
    // barriers, and exactly 2 x 80 KiB of LDS per CU (the screened variant's k-tile list would not fit beside it)
    const bool wide = glds && !(lower_only & 1) && d_maskA == nullptr && m > 128 && g_gemm_wide;
    const int tile_m = wide ? 160 : NT;
    int tm = ceil_div(m, tile_m), tn = ceil_div(n, NT);
    int ntiles = (lower_only & 1) ? tm * (tm + 1) / 2 : tm * tn;
    dim3 grid(ntiles, nsplit);
    const bool v2 = glds && g_dma_v2 && !wide && d_maskA == nullptr &&
                    (kchunk + KB) * (long)((lda > ldb) ? lda : ldb) * 8 < (1L << 32);
    // SYRK variants (flags on top of 1 = lower triangle, 2 = LDS-DMA operands), both measured in r03 (profiles/r03):
    //   4  balanced k split: nsplit - 1 FULL pieces of kchunk rows per tile plus one SHORT remainder piece of <= kchunk / mfrac
    //      rows, dispatched last (blockIdx.y = nsplit - 1), so that the U short pieces run mfrac-deep in the 2 x 256 - U (nsplit
    //      - 1) workgroup slots the full pieces leave free: 4.25 instead of 4 effective splits at nao = 1856
    //   8  re-tiled triangle (syrk_slots_kernel): no dead 64 x 64 wave blocks when the matrix has an odd number of 64-blocks
    auto balanced_chunk = [&](long units) {
        if (!(lower_only & 4) || !g_syrk_frac || nsplit < 2) return kchunk;
        // workgroup slots left free for a co-running kernel (J pass 2): bits 8-15 of the flags word in units of 4, else the tuning key
        const int reserve = ((lower_only >> 8) & 0xff) ? ((lower_only >> 8) & 0xff) * 4 : g_syrk_reserve;
        const long slots = 2L * g_num_cu - reserve, full = units * (nsplit - 1);
        long mfrac = 0;                          // smallest depth that fits = longest admissible short piece
        for (long mm = 1; mm <= 64; mm++)
            if (full + (units + mm - 1) / mm <= slots) { mfrac = mm; break; }
        if (mfrac < 1) return kchunk;
        const long kt = (k + KB - 1) / KB;
        const long ct = (kt * mfrac + ((long)(nsplit - 1) * mfrac + 1) - 1) / ((long)(nsplit - 1) * mfrac + 1);
        // the FULL pieces of the balanced split are longer than the uniform ones the v2 test above was made with: the kernels'
        // 32-bit row offsets (k0 + k) * ld * 8 must stay inside the 4 GiB buffer window for them as well, else uniform pieces
        if ((ct * KB + KB) * (long)((lda > ldb) ? lda : ldb) * 8 >= (1L << 32)) return kchunk;
        return ct * KB;
    };
    if (v2 && (lower_only & 1) && (lower_only & 8) && g_syrk_slots && d_A == d_B && lda == ldb) {
        const int *d_items = nullptr;
        int nitems = 0;
        std::vector<int> sorted;
        int rc = syrk_items(m, &d_items, &nitems, &sorted);
        if (rc) return rc;
        if (nitems > 0) {
            dim3 g2(nitems, nsplit);
            const long kc = balanced_chunk(nitems);
            if (js) {
                // k-tiles of the whole launch, exactly as the kernel counts them; fuse only when a k-tile has room for its share
                long steps = 0;
                for (int y = 0; y < nsplit; y++) {
                    long kb = (long)y * kc;
                    if (kb > k) kb = k;
                    const long ke = (kb + kc < k && y + 1 < nsplit) ? kb + kc : k;
                    steps += (ke - kb + KB - 1) / KB;
                }
                js->total_steps = steps * nitems;
                js->total_rows = ceil_div(js->npair, 512L) * js->nb;
                if (js->total_steps <= 0 || js->npair % 2 || ((uintptr_t)js->B % 16) ||
                    (double)js->total_rows > 3.9 * (double)js->total_steps)
                    return 1;                                  // not a shape for the fused pass: the caller runs the pass by itself
                syrk_slots_kernel<0, true><<<g2, 256, 0, st>>>(d_A, lda, d_C, ldc, m, k, d_items, kc, g_mfma_prio, *js, nullptr, nsplit);
                PAMD_CHECK_LAUNCH();
                return 0;
            }
            JStream none;
            memset(&none, 0, sizeof(none));
            const int *d_order = nullptr;
            if (g_syrk_xmap && nitems < 65536 && nsplit < 32768) {
                int nwg = 0;
                if ((rc = syrk_xcd_order(1, m, sorted, nsplit, kc != kchunk, g_syrk_xmap, &d_order, &nwg))) return rc;
                g2 = dim3(nwg, 1);
            }
            if (g_syrk_ring && !g_syrk_probe) {
                syrk_ring4_kernel<<<g2, 256, 0, st>>>(d_A, lda, d_C, ldc, m, k, d_items, kc, d_order, nsplit);
                PAMD_CHECK_LAUNCH();
                return 0;
            }
            if (g_syrk_probe) syrk_slots_kernel<1, false><<<g2, 256, 0, st>>>(d_A, lda, d_C, ldc, m, k, d_items, kc, g_mfma_prio, none, d_order, nsplit);
            else syrk_slots_kernel<0, false><<<g2, 256, 0, st>>>(d_A, lda, d_C, ldc, m, k, d_items, kc, g_mfma_prio, none, d_order, nsplit);
            PAMD_CHECK_LAUNCH();
            return 0;
        }
    }
    if (js) return 1;                                          // no re-tiled SYRK for this shape: nothing launched
    if (v2) {
        const long kuni = kchunk;
        if (lower_only & 1) kchunk = balanced_chunk(ntiles);
        const int *d_order = nullptr;
        if (g_syrk_xmap && (lower_only & 1) && ntiles < 65536 && nsplit < 32768) {
            // lower-triangular tiles t = tm (tm + 1) / 2 + tn are in tile-row order already
            std::vector<int> sorted(ntiles);
            for (int t = 0; t < ntiles; t++) sorted[t] = t;
            int nwg = 0;
            int rc = syrk_xcd_order(0, m, sorted, nsplit, kchunk != kuni, g_syrk_xmap, &d_order, &nwg);
            if (rc) return rc;
            grid = dim3(nwg, 1);
        }
        gemm_tn_glds2_kernel<<<grid, 256, 0, st>>>(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, lower_only & 1, tn, kchunk, g_mfma_prio, d_order, nsplit);
    }
    else if (glds)
    {
        if (wide)
            gemm_tn_glds_kernel<5><<<grid, 256, 0, st>>>(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, 0, tn, d_maskA, d_maskB, tm, nsplit);
        else
            gemm_tn_glds_kernel<4><<<grid, 256, 0, st>>>(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, lower_only & 1, tn,
                                                         d_maskA, d_maskB, tm, nsplit);
    }
    else {
        PAMD_REQUIRE(d_maskA == nullptr, "masked dgemm_tn needs the aligned LDS-DMA path (flag 2, 16-byte aligned, k % 16 == 0)");
        gemm_tn_kernel<<<grid, 256, 0, st>>>(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, lower_only & 1, tn);
    }
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_dgemm_tn(const double *d_A, int lda, const double *d_B, int ldb, double *d_C, int ldc,
                  int m, int n, long k, int lower_only, int nsplit, void *stream)
{
    return dgemm_tn_impl(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, lower_only, nsplit, nullptr, nullptr, stream);
}

// Screened variant (numint._dot_ao_ao_sparse, pyscf/dft/numint.py:836-874 / VXCdot_ao_ao_sparse):
// d_maskA[ceil(k/16)][ceil(m/128)], d_maskB[ceil(k/16)][ceil(n/128)] bytes; a k-tile is skipped for an output
// tile when either panel tile is flagged 0.  Requires the aligned-operand contract of flag 2.
// r05: K[split] += X^T X (the re-tiled SYRK of PAMD_dgemm_tn, flags as there) WITH the second J pass of the same tensor rows inside
// the kernel: d_vj[pq] += sum_L d_rho[L] d_B[L][pq] for the nb packed rows d_B (syrk_slots_kernel<JF>).  Returns 0 when the fused
// kernel ran, 1 when the shape has no fused form (no re-tiled triangle, odd nao_pair, more than ~4 row-loads per k-tile): NOTHING
// was launched then and the caller issues PAMD_df_vj_pass2 + PAMD_dgemm_tn itself.  Replaces the pair `vj += rho . eri1` /
// `vk += lib.dot(buf1.T, buf1)` of pyscf/df/df_jk.py:367,380 by one launch.
int PAMD_syrk_jfused(const double *d_X, int ldx, double *d_C, int ldc, int m, long k, int flags, int nsplit, const double *d_B,
                     long npair, int nb, const double *d_rho, double *d_vj, void *stream)
{
    PAMD_REQUIRE(d_X && d_C && d_B && d_rho && d_vj && nb > 0 && npair > 0, "PAMD_syrk_jfused: bad arguments");
    JStream js;
    memset(&js, 0, sizeof(js));
    js.B = d_B;
    js.rho = d_rho;
    js.vj = d_vj;
    js.npair = npair;
    js.nb = nb;
    return dgemm_tn_impl(d_X, ldx, d_X, ldx, d_C, ldc, m, m, k, flags, nsplit, nullptr, nullptr, stream, &js);
}

int PAMD_dgemm_tn_masked(const double *d_A, int lda, const double *d_B, int ldb, double *d_C, int ldc,
                         int m, int n, long k, int nsplit, const unsigned char *d_maskA,
                         const unsigned char *d_maskB, void *stream)
{
    PAMD_REQUIRE(d_maskA && d_maskB, "masks required");
    return dgemm_tn_impl(d_A, lda, d_B, ldb, d_C, ldc, m, n, k, 2, nsplit, d_maskA, d_maskB, stream);
}

// d_flags[ceil(nrows/16)][ceil(ld/16)]: 1 where the 16 x 16 tile of src[nrows][ld] has an element above thr
// (the role of GTO_screen_index / make_mask, pyscf/lib/gto/grid_ao_drv.c:32-123, computed from the values)
int PAMD_tile_mask(const double *d_src, long ld, long nrows, double thr, unsigned char *d_flags, void *stream)
{
    if (nrows == 0) return 0;
    int nct = ceil_div(ld, 16);
    tile_mask_kernel<<<ceil_div(nrows, 16), 256, 0, (hipStream_t)stream>>>(d_src, ld, nrows, thr, d_flags, nct);
    PAMD_CHECK_LAUNCH();
    return 0;
}

int PAMD_reduce_splits(const double *d_part, int nsplit, int m, int ldc, double *d_out, int ldo,
                       int symmetrize, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(m, 256), m);
    reduce_splits_kernel<<<grid, 256, 0, st>>>(d_part, nsplit, m, ldc, d_out, ldo, symmetrize);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// NPdunpack_tril_2d analogue (pyscf/lib/np_helper/pack_tril.c:150-273), hermitian fill.
int PAMD_unpack_tril(const double *d_tril, long npair, int count, int nao, double *d_full, int ld,
                     int rows, void *stream)
{
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(ceil_div(ld, 256), nao, count);
    unpack_tril_kernel<<<grid, 256, 0, st>>>(d_tril, npair, nao, d_full, ld, rows);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// dmtril = pack_tril(dm + dm^T), diagonal halved (pyscf/df/df_jk.py:329-332)
int PAMD_pack_dm_tril(const double *d_dm, int nset, int nao, double *d_tril, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    long npair = (long)nao * (nao + 1) / 2;
    dim3 grid(ceil_div(nao, 256), nao, nset);
    pack_dm_kernel<<<grid, 256, 0, st>>>(d_dm, nao, d_tril, npair);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
