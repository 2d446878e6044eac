// FP64-MFMA half-transform body shared by the DF K path (df_jk.hip) and the block-sparse XC path (xc_sparse.hip).
#pragma once
#include "common.h"

namespace pamd {

// LDS row strides (in doubles).  A fragment read is ds_read_b64 with lanes 0-15 on row k and
// lanes 16-31 on row k+1: conflict-free when the row stride is == 16 (mod 32) doubles.
constexpr int KB = 16;          // k-depth of one LDS tile
constexpr int NT = 128;         // columns per workgroup tile
constexpr int LDN = NT + 16;    // 144 == 16 mod 32
constexpr int LDT = KB + 1;     // transposed tile [n][k], odd stride -> conflict-free b64 reads

// X[L][i][p] = sum_q Bsym_L[q][p] * orb[q][i]
//   grid: x = p tile (128 cols), y = L, z = chunk of MT*16 orbitals
//   MFMA roles: m = orbital i (A operand from orb), n = AO index p (B operand from cderi row)
//   The packed row is read directly: tiles below the diagonal (q >= p) are row-contiguous,
//   tiles above it are read through the transposed element row[p(p+1)/2+q] (q-contiguous) and
//   kept transposed in LDS.  Next tile is prefetched into registers during the MFMA phase.
//   PLAIN = true: the same MFMA structure for a plain operand src[n][k] (k contiguous, leading
//   dimension npair): out[y][i][n] = sum_k src_y[n][k] orb[k][i]  (used for c = C_occ^T ao^T in nr_rks);
//   then `nao` is the k extent, `ncols` the n extent and `npair` doubles as the row stride of src.
//   GATHER = true (PLAIN only): the orbital row of k index q is orb[kidx[q]] - the compact AO subset of a grid tile
//   picks its rows of the full coefficient matrix (block-sparse XC, VXCdot_ao_dm_sparse's shell lists,
//   pyscf/lib/dft/nr_numint_sparse.c:226-304).
// Device body: `row` = this workgroup's operand (packed aux row / plain src block), `out` = its output block,
// `km` = its k-tile mask row (nullable), `rho_slot` = its 4 per-wave partials of the fused first J pass (nullable).
template <int MT, bool PLAIN, bool GATHER>
__device__ __forceinline__ void e2_symm_body(
    const double *__restrict__ row, long npair, int nao, const double *__restrict__ orb, int ldo,
    double *__restrict__ out, int nocc_pad, long ldx, long ncols, const unsigned char *__restrict__ km,
    double *__restrict__ rho_slot, const int *__restrict__ kidx, const int p0, const int m0)
{
    constexpr int MW = MT * 16;                         // orbitals per workgroup
    constexpr int LDA = MW + ((MW % 32 == 16) ? 0 : 16);  // == 16 mod 32
    __shared__ double sA[KB * LDA];
    __shared__ double sB[(NT * LDT > KB * LDN) ? NT * LDT : KB * LDN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    double4_t acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    const int fk = lane >> 4, fn = lane & 15;
    const int sk = tid >> 4, sc = tid & 15;             // staging coordinates (row-major tiles)
    const int tn = tid >> 1, tk = (tid & 1) * 8;        // staging coordinates (transposed tiles)
    double ra[MT], rb[8];

    auto tile_above = [&](int q0) { return PLAIN || q0 + KB - 1 <= p0; };
    auto fetch = [&](int q0) {
        const int q = q0 + sk;
        const long qrow = (GATHER && q < nao) ? kidx[q] : q;
        const double *orow = orb + qrow * ldo + m0 + sc;
#pragma unroll
        for (int j = 0; j < MT; j++) ra[j] = (q < nao) ? orow[16 * j] : 0.0;
        if (PLAIN) {
            const long p = p0 + tn;
            const double *src = row + p * npair + q0 + tk;
#pragma unroll
            for (int j = 0; j < 8; j++) rb[j] = (p < ncols && q0 + tk + j < nao) ? src[j] : 0.0;
        } else if (tile_above(q0)) {
            const long p = p0 + tn;
            const double *src = row + p * (p + 1) / 2 + q0 + tk;
#pragma unroll
            for (int j = 0; j < 8; j++) rb[j] = (p < nao && q0 + tk + j < nao) ? src[j] : 0.0;
        } else if (q0 >= p0 + NT - 1) {
            const double *src = row + (long)q * (q + 1) / 2 + p0 + sc;
#pragma unroll
            for (int j = 0; j < 8; j++) rb[j] = (q < nao && p0 + sc + 16 * j < nao) ? src[16 * j] : 0.0;
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const long p = p0 + sc + 16 * j, qq = q;
                double v = 0.0;
                if (p < nao && qq < nao) v = (qq >= p) ? row[qq * (qq + 1) / 2 + p] : row[p * (p + 1) / 2 + qq];
                rb[j] = v;
            }
        }
    };

    // optional screening (PLAIN mode): kmask[(y * ntiles + tile) * nk + k-tile] == 0 -> the 128 x 16 operand
    // tile is negligible and its k-tile is skipped (numint's non0tab idea, pyscf/gto/eval_gto.py:146+)
    auto next_active = [&](int q) {
        if (km) while (q < nao && !km[q / KB]) q += KB;
        return q;
    };
    int q0 = next_active(0);
    if (q0 < nao) fetch(q0);
    while (q0 < nao) {
        const bool above = tile_above(q0);
#pragma unroll
        for (int j = 0; j < MT; j++) sA[sk * LDA + sc + 16 * j] = ra[j];
        if (above) {
#pragma unroll
            for (int j = 0; j < 8; j++) sB[tn * LDT + tk + j] = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) sB[sk * LDN + sc + 16 * j] = rb[j];
        }
        __syncthreads();
        const int qn = next_active(q0 + KB);
        if (qn < nao) fetch(qn);
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double bf[2];
#pragma unroll
            for (int b = 0; b < 2; b++) {
                int n = wave * 32 + b * 16 + fn;
                bf[b] = above ? sB[n * LDT + kk + fk] : sB[(kk + fk) * LDN + n];
            }
#pragma unroll
            for (int a = 0; a < MT; a++) {
                double af = sA[(kk + fk) * LDA + a * 16 + fn];
#pragma unroll
                for (int b = 0; b < 2; b++) acc[a][b] = mfma_f64_16x16x4(af, bf[b], acc[a][b]);
            }
        }
        __syncthreads();
        q0 = qn;
    }
    // ---- store: D[m = (lane>>4)+4r][n = lane&15]; rho (symmetric mode, nullable): rho[L] += sum_{i,p} X[L][i][p] orb[p][i],
    // the first J pass of the density orb orb^T (see e2_sq_kernel); here `rho` is the per-wave partial buffer
    const bool do_rho = !PLAIN && rho_slot != nullptr;
    double rho_acc = 0;
#pragma unroll
    for (int a = 0; a < MT; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            long p = p0 + wave * 32 + b * 16 + fn;
            if (p >= (PLAIN ? ncols : ldx)) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                int i = m0 + a * 16 + fk + 4 * r;
                if (i < nocc_pad) {
                    out[(long)i * ldx + p] = acc[a][b][r];
                    if (do_rho && p < nao) rho_acc += acc[a][b][r] * orb[p * ldo + i];
                }
            }
        }
    if (do_rho) {
        // one partial per wave, summed in a fixed order by vj_pass1_reduce_kernel: J is bitwise reproducible
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rho_acc += __shfl_xor(rho_acc, off, 64);
        if (lane == 0) rho_slot[wave] = rho_acc;
    }
}

}  // namespace pamd
