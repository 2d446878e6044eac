// Block-sparse XC contractions on compact AO subsets (gfx950).
//
// Reference idea: numint's screened contractions - every grid block carries the list of AO shells that are
// non-negligible on it (`non0tab` / `screen_index`, pyscf/dft/numint.py:2845, pyscf/lib/gto/grid_ao_drv.c:32-123
// GTO_screen_index) and the products run over those shells only (VXCdot_ao_dm_sparse,
// pyscf/lib/dft/nr_numint_sparse.c:226-304; VXCdot_ao_ao_sparse :890-973).  MI355X form: the grid is cut into tiles of
// G consecutive (box-sorted) points; for each tile the AO values of its active shells are stored COMPACTED,
//     ao_c[tile] = [comp][G][ld_t]        ld_t = round_up(#active functions, 16), compact AO index fastest,
// (cached in HBM across SCF iterations: at (H2O)_32 cc-pVTZ the compact image is 14 GB where the dense one is 64 GB),
// and every contraction is an FP64-MFMA GEMM on the compact operand, batched over the tiles in ONE launch:
//
//   sub_orb_dot   cmo[c][i][g]   = sum_{mu in tile} C[idx[mu]][i] ao_c[c][g][mu]     (orbital rows gathered through idx)
//   sub_scale     aow_c[g][mu]   = sum_c wv[c][g] ao_c[c][g][mu]
//   sub_vmat      M[idx[mu]][idx[nu]] += sum_{g in tile} ao_c[0][g][mu] aow_c[g][nu]  (LDS-DMA GEMM, scatter-add)
//
// idx[tile][ld_t] maps a compact column to its AO index (>= nao for padding columns).
#include "common.h"
#include "mfma_e2.h"

using namespace pamd;

namespace {

struct SubTiles {
    const long *ao_off;     // [ntile] offset (doubles) of ao_c[tile]
    const long *aow_off;    // [ntile] offset (doubles) of aow_c[tile] ([G][ld_t])
    const long *idx_off;    // [ntile] offset into idx
    const int *ld;          // [ntile] ld_t
    const int *idx;         // compact column -> AO index
};

// grid: x = 128-point slice of the tile, y = component, z = tile * nchunk + orbital chunk
template <int MT>
__global__ __launch_bounds__(256, 2) void sub_orb_dot_kernel(const double *__restrict__ ao_c, SubTiles tl, int G, int nchunk,
                                                             const double *__restrict__ orb, int ldo, int nocc_pad,
                                                             double *__restrict__ cmo, long comp_stride, long ldc)
{
    const int t = blockIdx.z / nchunk, chunk = blockIdx.z - t * nchunk;
    const int ld = tl.ld[t];
    const double *src = ao_c + tl.ao_off[t] + (long)blockIdx.y * G * ld;
    double *out = cmo + (long)blockIdx.y * comp_stride + (long)t * G;
    e2_symm_body<MT, true, true>(src, ld, ld, orb, ldo, out, nocc_pad, ldc, G, nullptr, nullptr, tl.idx + tl.idx_off[t],
                                 blockIdx.x * NT, chunk * (MT * 16));
}

// aow_c[tile][g][mu] = sum_c wv[c][tile*G + g] ao_c[tile][c][g][mu];  grid: x = column chunk, y = g, z = tile
__global__ __launch_bounds__(256) void sub_scale_kernel(const double *__restrict__ ao_c, SubTiles tl, int G, int ncomp,
                                                        const double *__restrict__ wv, long ldg, double *__restrict__ aow_c)
{
    const int t = blockIdx.z;
    const int ld = tl.ld[t];
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= ld) return;
    const long g = blockIdx.y;
    const double *a = ao_c + tl.ao_off[t] + g * ld + m;
    const long cs = (long)G * ld;
    const long gg = (long)t * G + g;
    double v = wv[gg] * a[0];
    if (ncomp == 4) v += wv[ldg + gg] * a[cs] + wv[2 * ldg + gg] * a[2 * cs] + wv[3 * ldg + gg] * a[3 * cs];
    aow_c[tl.aow_off[t] + g * ld + m] = v;
}

// One 128 x 128 block of  ao_c[0]^T aow_c  of one tile per workgroup (work item = {tile, tm, tn}), k = the tile's G grid
// points; both panels stream HBM/L2 -> LDS by LDS-DMA, double-buffered, one barrier per 16-point k-tile (the loop of
// gemm_tn_glds_kernel).  MFMA tiles that lie entirely beyond the tile's ld_t columns are skipped (wave-uniform
// predicates), so the padding of ld_t to the 128-wide block costs loads but no matrix-pipe time.  Epilogue: FP64
// no-return atomics into M[idx[mu]][idx[nu]].
__global__ __launch_bounds__(256, 2) void sub_vmat_kernel(const double *__restrict__ ao_c, const double *__restrict__ aow_c,
                                                          SubTiles tl, const int *__restrict__ work, int G, int nao,
                                                          double *__restrict__ vmat, long ldv)
{
    __shared__ double sb0[2 * KB * LDN];
    __shared__ double sb1[2 * KB * LDN];
    constexpr int PA = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = work[3 * blockIdx.x], tm = work[3 * blockIdx.x + 1], tn = work[3 * blockIdx.x + 2];
    const int ld = tl.ld[t];
    const int p0 = tm * NT, q0 = tn * NT;
    const double *A = ao_c + tl.ao_off[t] + p0 + lane * 2;
    const double *B = aow_c + tl.aow_off[t] + q0 + lane * 2;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;

    bool va[4], vb[4];
#pragma unroll
    for (int a = 0; a < 4; a++) va[a] = p0 + wr * 64 + a * 16 < ld;
#pragma unroll
    for (int b = 0; b < 4; b++) vb[b] = q0 + wc * 64 + b * 16 < ld;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage = [&](int k0, double *dst) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = wave * 4 + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(A + (long)(k0 + k) * ld),
                                             (__attribute__((address_space(3))) void *)(dst + k * LDN), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(B + (long)(k0 + k) * ld),
                                             (__attribute__((address_space(3))) void *)(dst + PA + k * LDN), 16, 0, 0);
        }
    };
    auto step = [&](const double *cur, double *nxt, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (k0 + KB < G) stage(k0 + KB, nxt);
        const double *sP = cur, *sQ = cur + PA;
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
                for (int b = 0; b < 4; b++)
                    if (va[a] && vb[b]) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
    stage(0, sb0);
    for (int k0 = 0; k0 < G; k0 += 2 * KB) {
        step(sb0, sb1, k0);
        if (k0 + KB < G) step(sb1, sb0, k0 + KB);
    }
    const int *idx = tl.idx + tl.idx_off[t];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        const int col = q0 + wc * 64 + b * 16 + fn;
        if (col >= ld) continue;
        const int aj = idx[col];
        if (aj >= nao) continue;
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rowi = p0 + wr * 64 + a * 16 + fk + 4 * r;
                if (rowi >= ld) continue;
                const int ai = idx[rowi];
                if (ai < nao) unsafeAtomicAdd(vmat + (long)ai * ldv + aj, acc[a][b][r]);
            }
    }
}

// ao_c[tile][c][g][mu] = dense[c][g0 + g][idx[mu]] (0 for padding columns and for rows beyond the dense block):
// fills the compact image from a dense PAMD_eval_ao block.  grid: x = column chunk, y = g, z = tile (of this call)
__global__ __launch_bounds__(256) void sub_gather_kernel(const double *__restrict__ dense, long dense_rows, int ldao,
                                                         int ncomp, long row0, long nrows_valid, SubTiles tl, int G,
                                                         int nao, double *__restrict__ ao_c)
{
    const int t = blockIdx.z;
    const int ld = tl.ld[t];
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= ld) return;
    const long g = blockIdx.y;
    const long row = row0 + (long)t * G + g;              // row of the dense block
    const int mu = tl.idx[tl.idx_off[t] + m];
    const bool ok = row < nrows_valid && mu < nao;
    for (int c = 0; c < ncomp; c++)
        ao_c[tl.ao_off[t] + ((long)c * G + g) * ld + m] = ok ? dense[((long)c * dense_rows + row) * ldao + mu] : 0.0;
}

}  // namespace

extern "C" {

// Tile tables (device): d_ao_off / d_aow_off / d_idx_off [ntile] (doubles / doubles / ints), d_ld [ntile], d_idx.
// The tile arrays passed to each call are those of the `ntile` tiles the call works on (tile t of the call = entry t).

// cmo[c][i][t*G + g] = sum_mu orb[idx_t[mu]][i] ao_c[t][c][g][mu]   (numint.eval_rho2's c = ao . C_occ on the active
// shells, pyscf/dft/numint.py:328-469 / VXCdot_ao_dm_sparse); orb rows >= nao must be zero rows (padding columns of a
// tile point at row nao).  ldc >= ntile * G.
int PAMD_sub_orb_dot(const double *d_ao_c, const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx,
                     int ntile, int G, int ncomp, const double *d_orb, int ldo, int nocc_pad, double *d_cmo,
                     long comp_stride, long ldc, void *stream)
{
    PAMD_REQUIRE(nocc_pad % 16 == 0 && nocc_pad <= ldo, "nocc_pad must be a multiple of 16 and <= ldo");
    PAMD_REQUIRE(G % NT == 0, "tile size must be a multiple of 128");
    if (ntile == 0 || nocc_pad == 0) return 0;
    const int mt_total = nocc_pad / 16;
    const int nchunk = ceil_div(mt_total, 10);
    const int mt = ceil_div(mt_total, nchunk);
    PAMD_REQUIRE(ldo >= nchunk * mt * 16, "orbital leading dimension too small for tile padding");
    PAMD_REQUIRE((long)ntile * nchunk < 65536, "too many tiles per call");
    SubTiles tl{d_ao_off, nullptr, d_idx_off, d_ld, d_idx};
    dim3 grid(G / NT, ncomp, ntile * nchunk);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH_S(MT) sub_orb_dot_kernel<MT><<<grid, 256, 0, st>>>(d_ao_c, tl, G, nchunk, d_orb, ldo, nocc_pad, d_cmo, comp_stride, ldc)
    switch (mt) {
    case 1: LAUNCH_S(1); break;
    case 2: LAUNCH_S(2); break;
    case 3: LAUNCH_S(3); break;
    case 4: LAUNCH_S(4); break;
    case 5: LAUNCH_S(5); break;
    case 6: LAUNCH_S(6); break;
    case 7: LAUNCH_S(7); break;
    case 8: LAUNCH_S(8); break;
    case 9: LAUNCH_S(9); break;
    default: LAUNCH_S(10); break;
    }
#undef LAUNCH_S
    PAMD_CHECK_LAUNCH();
    return 0;
}

// aow_c[t][g][mu] = sum_c wv[c][t*G + g] ao_c[t][c][g][mu]   (numint._scale_ao, pyscf/dft/numint.py:803-834)
int PAMD_sub_scale_ao(const double *d_ao_c, const long *d_ao_off, const long *d_aow_off, const int *d_ld, int ntile, int G,
                      int ncomp, int ld_max, const double *d_wv, long ldg, double *d_aow_c, void *stream)
{
    if (ntile == 0) return 0;
    PAMD_REQUIRE(ntile < 65536 && G < 65536, "grid limits");
    SubTiles tl{d_ao_off, d_aow_off, nullptr, d_ld, nullptr};
    dim3 grid(ceil_div(ld_max, 256), G, ntile);
    sub_scale_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_ao_c, tl, G, ncomp, d_wv, ldg, d_aow_c);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// vmat[idx_t[mu]][idx_t[nu]] += sum_g ao_c[t][0][g][mu] aow_c[t][g][nu] for the nwork work items {tile, tm, tn} of
// d_work[3 nwork] (numint._dot_ao_ao_sparse / VXCdot_ao_ao_sparse, pyscf/lib/dft/nr_numint_sparse.c:890-973, with the
// nao_sub^2 block scattered into the full matrix).  Both buffers need 256 doubles of slack behind the last tile.
int PAMD_sub_vmat(const double *d_ao_c, const long *d_ao_off, const double *d_aow_c, const long *d_aow_off,
                  const long *d_idx_off, const int *d_ld, const int *d_idx, const int *d_work, int nwork, int G, int nao,
                  double *d_vmat, long ldv, void *stream)
{
    PAMD_REQUIRE(G % (2 * KB) == 0 || G % KB == 0, "tile size must be a multiple of 16");
    PAMD_REQUIRE(((uintptr_t)d_ao_c | (uintptr_t)d_aow_c) % 16 == 0, "16-byte aligned operands");
    if (nwork == 0) return 0;
    SubTiles tl{d_ao_off, d_aow_off, d_idx_off, d_ld, d_idx};
    sub_vmat_kernel<<<nwork, 256, 0, (hipStream_t)stream>>>(d_ao_c, d_aow_c, tl, d_work, G, nao, d_vmat, ldv);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// Fill ao_c for `ntile` tiles from a dense AO block d_dense[ncomp][dense_rows][ldao] (PAMD_eval_ao): tile t of the call
// covers dense rows [row0 + t*G, row0 + (t+1)*G); rows >= nrows_valid are zero-filled (ragged last tile).
int PAMD_sub_gather_ao(const double *d_dense, long dense_rows, int ldao, int ncomp, long row0, long nrows_valid,
                       const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx, int ntile, int G,
                       int ld_max, int nao, double *d_ao_c, void *stream)
{
    if (ntile == 0) return 0;
    PAMD_REQUIRE(ntile < 65536 && G < 65536, "grid limits");
    SubTiles tl{d_ao_off, nullptr, d_idx_off, d_ld, d_idx};
    dim3 grid(ceil_div(ld_max, 256), G, ntile);
    sub_gather_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_dense, dense_rows, ldao, ncomp, row0, nrows_valid, tl, G, nao,
                                                           d_ao_c);
    PAMD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
