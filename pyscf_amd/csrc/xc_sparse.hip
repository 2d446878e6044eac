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
#include <algorithm>
#include <type_traits>
#include <vector>
#include "common.h"
#include "mfma_e2.h"

using namespace pamd;

static int g_orb_dot_dma = 1;
static int g_orb_rho_128 = 1;    // r06: 128-orbital chunks in PAMD_sub_orb_rho when they need fewer MFMA tiles than 160-wide ones ("orbrho128")
static int g_orb_rho_fused = 1;  // GGA: rho / grad rho in the orbital product's epilogue (PAMD_sub_orb_rho); A/B switch "orbrho"
static int g_vmat_probe = 0;   // benchmarking probes of sub_vmat_sym ("vmatprobe", see the kernel)
static int g_vmat_burst = 0;   // sub_vmat_sym: DMA rows of the next k-tile in one burst behind the first MFMA group ("vmatburst")
static int g_vmat_even = 1;    // PAMD_sub_vmat_work: pieces of an even number of 16-column groups ("vmateven"; 0: nearly equal pieces, r04)
static int g_vmat_flip = 0;    // sub_vmat_sym: which workgroups swap the roles of their waves ("vmatflip", see the kernel)
static int g_vmat_xcd = 1;     // sub_vmat*: work items of one tile on ONE XCD (its L2 then serves the panel re-reads); A/B switch "vmatxcd"

namespace {

struct SubTiles {
    const long *ao_off;     // [ntile] offset (doubles) of ao_c[tile]
    const long *aow_off;    // [ntile] offset (doubles) of aow_c[tile] ([G][ld_t])
    const long *idx_off;    // [ntile] offset into idx
    const int *ld;          // [ntile] ld_t
    const int *idx;         // compact column -> AO index
};


// Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2.  A work list ordered by tile
// therefore spreads the blocks of one tile - which read the same two panels - over all eight L2s, and every one of them fetches
// the panels from HBM again (r03 counters: 4 x the algorithmic bytes).  This maps block b to the list position that keeps
// consecutive items on one XCD: XCD x works through its own contiguous stretch of the list (speed only - any dispatch order is
// correct).
__device__ __forceinline__ int xcd_contiguous(int b, int n, int enabled)
{
    if (!enabled || n < 16) return b;
    const int q = n >> 3, r = n & 7, x = b & 7, slot = b >> 3;
    return x < r ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
}

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

// The same product for 160-orbital chunks with every operand by buffer-resource LDS-DMA (the scheme of e2_sq2 / e2_pk in
// df_jk.hip).  MFMA roles: m = orbital (A: gathered rows orb[idx[mu]], 128 columns by one row DMA + the 32 remainder
// columns of the wave's four rows by one DMA with per-lane row addresses), n = grid point, k = compact AO index.  The B
// operand ao_c[c][g][mu] is contiguous in k for a fixed n - the "transposed" case of e2_pk: one DMA moves eight grid
// rows x 16 mu with per-lane source addresses into XOR-swizzled 16-byte chunks.  grid: x = 128-point slice, y = component,
// z = tile * nchunk + chunk.
__global__ __launch_bounds__(256, 2) void sub_orb_dot2_kernel(const double *__restrict__ ao_c, SubTiles tl, int G, int nchunk,
                                                              const double *__restrict__ orb, int ldo, int nocc_pad,
                                                              double *__restrict__ cmo, long comp_stride, long ldc)
{
    __shared__ double sa0[KB * LDN + KB * 32];
    __shared__ double sa1[KB * LDN + KB * 32];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    constexpr int RB = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.z / nchunk, chunk = blockIdx.z - t * nchunk;
    const int ld = tl.ld[t];
    const int n0 = blockIdx.x * NT, m0 = chunk * 160;
    const int *idx = tl.idx + tl.idx_off[t];
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(ao_c + tl.ao_off[t] + ((long)blockIdx.y * G + n0) * ld), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_orb = __builtin_amdgcn_make_buffer_rsrc((void *)(orb + m0), 0, 0xffffffff, 0x00020000);
    const int ldo8 = ldo * 8;
    const int voff = lane * 16;
    const int rrow = lane >> 4;
    const int voff_remcol = (128 + ((((lane & 15) * 2) - 16 * (rrow & 1)) & 31)) * 8;      // + idx[row] * ldo8 per k-tile
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn;
    const int offr = RB + fk * 32 + ((wr * 16 + fn + 16 * (fk & 1)) & 31);
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
        voff_tr[j] = ((blk * 8 + pl_s) * ld + 2 * kp) * 8;
    }

    double4_t acc[5][4];
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *da, double *db, int j) {
        const int k = wave * 4 + j;
        const int row = idx[k0 + k];                                   // wave-uniform: a scalar load
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_orb, (__attribute__((address_space(3))) void *)(da + k * LDN), 16, voff, row * ldo8, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_b, (__attribute__((address_space(3))) void *)(db + k * 128), 16, voff_tr[j], k0 * 8, 0, 0);
        if (j == 0) {
            const int rowl = idx[k0 + wave * 4 + rrow];               // per 16-lane group: its own gathered row
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_orb, (__attribute__((address_space(3))) void *)(da + RB + wave * 128), 16,
                                                     rowl * ldo8 + voff_remcol, 0, 0, 0);
        }
    };
    auto step = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < ld) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[5], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = ca[offa + kk * LDN + a * 16];
            af[4] = ca[offr + kk * 32];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cb[atr[kk >> 2] + b * 256];
            stage_row(kn, na, nb, kk >> 2);
#pragma unroll
            for (int a = 0; a < 5; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; j++) stage_row(0, sa0, sq0, j);
    for (int k0 = 0; k0 < ld; k0 += 2 * KB) {
        step(sa0, sq0, sa1, sq1, k0);
        if (k0 + KB < ld) step(sa1, sq1, sa0, sq0, k0 + KB);
    }
    double *out = cmo + (long)blockIdx.y * comp_stride + (long)t * G;
#pragma unroll
    for (int a = 0; a < 5; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int n = n0 + wc * 64 + b * 16 + fn;
            if (n >= G) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = m0 + (a < 4 ? wr * 64 + a * 16 : 128 + wr * 16) + fk + 4 * r;
                if (i < nocc_pad) out[(long)i * ldc + n] = acc[a][b][r];
            }
        }
}

// r04: the orbital product fused with the density evaluation (GGA): rho, grad rho straight from the accumulators, the
// 5.5 GB c[comp][i][g] buffer of config 3 never exists (PAMD_sub_orb_dot + PAMD_rho_from_mo in one kernel; numint.py:328-469
// eval_rho2 on the active shells).  Same k-loop, operands and LDS image as sub_orb_dot2_kernel; what changes is which rows of
// the compact image make up the 128 B columns of a workgroup: 32 grid points x the FOUR components, laid out so that wave column
// wc owns points [16 wc, 16 wc + 16) and its b-th MFMA column tile is component b.  Lane (fk, fn), register r of acc[a][b] then
// holds c_b[i][g] for one orbital i = tile row and one point g = fn for ALL four b: rho += s_i c_0^2, grad rho_x += 2 s_i c_0 c_x
// are lane-local products; the sums over the orbitals run over (a, r) in the lane, over fk by two wave shuffles, over the two
// wave rows through 2 KB of LDS.  Orbital chunks beyond the first (nocc_pad > 160) add up by FP64 atomics into the zeroed rho.
// grid: x = 32-point slice of the tile, z = tile * nchunk + chunk.
// r06: REM = false is the 128-orbital chunk (no remainder block: 4 instead of 5 MFMA tiles per wave row, no remainder DMA) for
// orbital counts that several BALANCED chunks cover with less padding than 160-wide ones (nocc = 226 -> 15 tiles: 8 + 7 in two
// 128-chunks = 16 tiles of work instead of 10 + 10 = 20: the taxol shape's ao . C product loses its 20 % of zero columns).
template <bool REM>
__global__ __launch_bounds__(256, 2) void sub_orb_rho_kernel(const double *__restrict__ ao_c, SubTiles tl, int G, int nchunk,
                                                             const double *__restrict__ orb, int ldo, int nocc,
                                                             const double *__restrict__ sign, double *__restrict__ rho, long ldg)
{
    constexpr int NA = REM ? 5 : 4, CW = REM ? 160 : 128;
    __shared__ double sa0[KB * LDN + KB * 32];
    __shared__ double sa1[KB * LDN + KB * 32];
    __shared__ double sq0[KB * LDN];
    __shared__ double sq1[KB * LDN];
    constexpr int RB = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = blockIdx.z / nchunk, chunk = blockIdx.z - t * nchunk;
    const int ld = tl.ld[t];
    const int g0 = blockIdx.x * 32, m0 = chunk * CW;
    const int *idx = tl.idx + tl.idx_off[t];
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc((void *)(ao_c + tl.ao_off[t] + (long)g0 * ld), 0, 0xffffffff,
                                                                        0x00020000);
    const __amdgpu_buffer_rsrc_t r_orb = __builtin_amdgcn_make_buffer_rsrc((void *)(orb + m0), 0, 0xffffffff, 0x00020000);
    const int ldo8 = ldo * 8;
    const int voff = lane * 16;
    const int rrow = lane >> 4;
    const int voff_remcol = (128 + ((((lane & 15) * 2) - 16 * (rrow & 1)) & 31)) * 8;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn;
    const int offr = RB + fk * 32 + ((wr * 16 + fn + 16 * (fk & 1)) & 31);
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
        const int n = blk * 8 + pl_s;                              // B column of the workgroup: wave column n >> 6, its tile (n >> 4) & 3
        const int comp = (n >> 4) & 3, pt = ((n >> 6) << 4) | (n & 15);
        voff_tr[j] = (int)((((long)comp * G + pt) * ld + 2 * kp) * 8);
    }

    double4_t acc[NA][4];
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *da, double *db, int j) {
        const int k = wave * 4 + j;
        const int row = idx[k0 + k];
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_orb, (__attribute__((address_space(3))) void *)(da + k * LDN), 16, voff, row * ldo8, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_b, (__attribute__((address_space(3))) void *)(db + k * 128), 16, voff_tr[j], k0 * 8, 0, 0);
        if (REM && j == 0) {
            const int rowl = idx[k0 + wave * 4 + rrow];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_orb, (__attribute__((address_space(3))) void *)(da + RB + wave * 128), 16,
                                                     rowl * ldo8 + voff_remcol, 0, 0, 0);
        }
    };
    auto step = [&](const double *ca, const double *cb, double *na, double *nb, int k0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kn = (k0 + KB < ld) ? k0 + KB : k0;
#pragma unroll
        for (int kk = 0; kk < KB; kk += 4) {
            double af[NA], bf[4];
#pragma unroll
            for (int a = 0; a < 4; a++) af[a] = ca[offa + kk * LDN + a * 16];
            if (REM) af[NA - 1] = ca[offr + kk * 32];
#pragma unroll
            for (int b = 0; b < 4; b++) bf[b] = cb[atr[kk >> 2] + b * 256];
            stage_row(kn, na, nb, kk >> 2);
#pragma unroll
            for (int a = 0; a < NA; a++)
#pragma unroll
                for (int b = 0; b < 4; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; j++) stage_row(0, sa0, sq0, j);
    for (int k0 = 0; k0 < ld; k0 += 2 * KB) {
        step(sa0, sq0, sa1, sq1, k0);
        if (k0 + KB < ld) step(sa1, sq1, sa0, sq0, k0 + KB);
    }
    // ---- rho, grad rho of the wave's 16 points over its 80 orbitals
    double p0 = 0, px = 0, py = 0, pz = 0;
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = m0 + (a < 4 ? wr * 64 + a * 16 : 128 + wr * 16) + fk + 4 * r;
            const double c0 = acc[a][0][r];
            const double w = (sign != nullptr && i < nocc) ? sign[i] * c0 : c0;      // rows i >= nocc: zero orbital columns, c0 = 0
            p0 += w * c0;
            px += w * acc[a][1][r];
            py += w * acc[a][2][r];
            pz += w * acc[a][3][r];
        }
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
        p0 += __shfl_xor(p0, off, 64);
        px += __shfl_xor(px, off, 64);
        py += __shfl_xor(py, off, 64);
        pz += __shfl_xor(pz, off, 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the last k-tile re-loaded itself: let it land before the LDS is reused
    __syncthreads();
    double *part = sa0;                                      // [wave][4][16]
    if (fk == 0) {
        part[(wave * 4 + 0) * 16 + fn] = p0;
        part[(wave * 4 + 1) * 16 + fn] = 2.0 * px;
        part[(wave * 4 + 2) * 16 + fn] = 2.0 * py;
        part[(wave * 4 + 3) * 16 + fn] = 2.0 * pz;
    }
    __syncthreads();
    if (tid < 128) {
        const int c = tid >> 5, pt = tid & 31, wcol = pt >> 4, f = pt & 15;
        const double v = part[((0 * 2 + wcol) * 4 + c) * 16 + f] + part[((1 * 2 + wcol) * 4 + c) * 16 + f];
        double *dst = rho + (long)c * ldg + (long)t * G + g0 + pt;
        if (nchunk == 1) *dst = v;
        else unsafeAtomicAdd(dst, v);
    }
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
// points; both panels stream HBM/L2 -> LDS by buffer-resource LDS-DMA, double-buffered, one barrier per 16-point k-tile
// (the loop of gemm_tn_glds2_kernel).  MFMA tiles beyond the tile's ld_t columns are skipped at 16-column granularity
// (one loop instance per live-group count).  Epilogue: FP64 no-return atomics into M[idx[mu]][idx[nu]].
__global__ __launch_bounds__(256, 2) void sub_vmat_kernel(const double *__restrict__ ao_c, const double *__restrict__ aow_c,
                                                          SubTiles tl, const int *__restrict__ work, int G, int nao,
                                                          double *__restrict__ vmat, long ldv, int xcd)
{
    __shared__ double sb0[2 * KB * LDN];
    __shared__ double sb1[2 * KB * LDN];
    constexpr int PA = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int item = xcd_contiguous(blockIdx.x, gridDim.x, xcd);
    const int t = work[3 * item], tm = work[3 * item + 1], tn = work[3 * item + 2];
    const int ld = tl.ld[t];
    const int p0 = tm * NT, q0 = tn * NT;
    // buffer-resource LDS-DMA as in gemm_tn_glds2 (df_jk.hip): scalar row offsets, no per-lane address arithmetic, the
    // rows of k-tile t+1 issued between the MFMA groups of tile t, branch-free k-tile body
    const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc((void *)(ao_c + tl.ao_off[t] + p0), 0, 0xffffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc((void *)(aow_c + tl.aow_off[t] + q0), 0, 0xffffffff, 0x00020000);
    const int voff = lane * 16, ld8 = ld * 8;
    const int wr = wave >> 1, wc = wave & 1;
    const int fk = lane >> 4, fn = lane & 15;
    const int offa = fk * LDN + wr * 64 + fn, offb = PA + fk * LDN + wc * 64 + fn;

    // MFMA tiles entirely beyond the tile's ld_t columns are skipped at 16-column granularity: the wave's numbers of live
    // row / column groups (na, nb in 0..4) are wave-uniform and fixed for the whole k-loop, so the loop is instantiated for
    // each (na, nb) and chosen once - every instance keeps its k-tile body a single basic block.
    int na = (ld - (p0 + wr * 64) + 15) / 16, nb = (ld - (q0 + wc * 64) + 15) / 16;
    na = na < 0 ? 0 : (na > 4 ? 4 : na);
    nb = nb < 0 ? 0 : (nb > 4 ? 4 : nb);

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto stage_row = [&](int k0, double *dst, int j) {
        const int k = wave * 4 + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (__attribute__((address_space(3))) void *)(dst + k * LDN), 16, voff, (k0 + k) * ld8, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_b, (__attribute__((address_space(3))) void *)(dst + PA + k * LDN), 16, voff, (k0 + k) * ld8, 0, 0);
    };
    auto kloop = [&](auto NAc, auto NBc) {
        constexpr int NA = decltype(NAc)::value, NB = decltype(NBc)::value;
        auto step = [&](const double *cur, double *nxt, int k0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const int kn = (k0 + KB < G) ? k0 + KB : k0;
#pragma unroll
            for (int kk = 0; kk < KB; kk += 4) {
                double af[4], bf[4];
#pragma unroll
                for (int a = 0; a < NA; a++) af[a] = cur[offa + kk * LDN + a * 16];
#pragma unroll
                for (int b = 0; b < NB; b++) bf[b] = cur[offb + kk * LDN + b * 16];
                stage_row(kn, nxt, kk >> 2);
#pragma unroll
                for (int a = 0; a < NA; a++)
#pragma unroll
                    for (int b = 0; b < NB; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
            }
        };
        for (int k0 = 0; k0 < G; k0 += 2 * KB) {
            step(sb0, sb1, k0);
            if (k0 + KB < G) step(sb1, sb0, k0 + KB);
        }
    };
#pragma unroll
    for (int j = 0; j < 4; j++) stage_row(0, sb0, j);
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    auto pick_b = [&](auto NAc) {
        switch (nb) {
        case 1: kloop(NAc, I1{}); break;
        case 2: kloop(NAc, I2{}); break;
        case 3: kloop(NAc, I3{}); break;
        default: kloop(NAc, I4{}); break;
        }
    };
    if (na == 0 || nb == 0) {
        kloop(I0{}, I0{});                 // nothing to compute: stage the DMA rows only
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    switch (na) {
    case 1: pick_b(I1{}); break;
    case 2: pick_b(I2{}); break;
    case 3: pick_b(I3{}); break;
    default: pick_b(I4{}); break;
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

// r04: the LOWER triangle of V = M + M^T (numint.py:1157), M = ao_c[0]^T aow_c, on BALANCED blocks.
// The r03 kernel above cuts the ld_t x ld_t product of a tile into 128 x 128 blocks from the left: at the mean ld_t of config 3
// (400 = 3 x 128 + 16) 7 of its 16 blocks are edge blocks with 1/8 of the MFMA work and the full 32-k-tile DMA / barrier pipeline
// (matrix pipe busy 0.53).  Here the host cuts the 16-column groups of a tile into ceil(g / 8) pieces of nearly equal size
// (400 -> 7 + 6 + 6 + 6 groups), a work item is one pair of pieces (i >= j) = {tile, p0, gp, q0, gq, diag}, its two wave rows /
// columns split their piece evenly (ceil(gp / 2) | the rest), and the block accumulates BOTH products of the symmetrised matrix,
//     V[i][j] = A_i^T W_j + W_i^T A_j        (A = ao_c[0], W = aow_c),
// in one accumulator pass (two k-loops over the tile's G points: 2 x 32 k-tiles per prologue / epilogue instead of 32).  Only
// blocks on and below the diagonal exist; a diagonal block computes M_ii once and folds M_ii + M_ii^T in its epilogue (entry
// (r, c) is added at [max][min], twice when r == c): exactly the MFMA work of r03 and 5/8 of its FP64 atomics.
// idx[] ascends inside a tile, so compact (mu >= nu) is AO (idx[mu] >= idx[nu]): vmat receives exactly its lower triangle;
// PAMD_mirror_tril completes V.
// PROBE (benchmarking only, results meaningless unless 0): 1 = no scatter-add, 2 = no DMA inside the k-loop (stale tiles), 3 = no MFMAs
template <bool BURST, int PROBE>  // BURST: all DMA rows of k-tile t + 1 behind the FIRST MFMA group of tile t instead of one row per group
__global__ __launch_bounds__(256, 2) void sub_vmat_sym_kernel(const double *__restrict__ ao_c, const double *__restrict__ aow_c,
                                                              SubTiles tl, const int *__restrict__ work, int G, int nao,
                                                              double *__restrict__ vmat, long ldv, int flip)
{
    __shared__ double sb0[2 * KB * LDN];
    __shared__ double sb1[2 * KB * LDN];
    constexpr int PA = KB * LDN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int *w = work + 6 * (long)blockIdx.x;       // the list is in dispatch order (PAMD_sub_vmat_work)
    const int t = w[0], p0 = w[1], gp = w[2], q0 = w[3], gq = w[4], diag = w[5];
    if (gp == 0) return;                              // padding item of a shorter XCD queue (uniform for the workgroup)
    const int ld = tl.ld[t];
    const int voff = lane * 16, ld8 = ld * 8;
    // An odd piece (7 groups = 4 + 3) gives wave (0, 0) 16 MFMA tiles per k-group and wave (1, 1) 9.  Wave w of every workgroup sits
    // on SIMD w % 4, so with the same roles in both co-resident workgroups SIMD 0 carries 16 + 16 and SIMD 3 9 + 9.  Workgroups whose
    // `flip` bit is set hand the large share to wave 3 instead: a pair with different bits loads every SIMD with 16 + 9 | 12 + 12.
    // flip: 0 none, 1 by the parity of the XCD-queue slot (blockIdx.x >> 3), 2 by the parity of the 256-workgroup round
    const int role = wave ^ ((flip == 1 ? (blockIdx.x >> 3) & 1 : flip == 2 ? (blockIdx.x >> 8) & 1 : 0) ? 3 : 0);
    const int wr = role >> 1, wc = role & 1;
    const int fk = lane >> 4, fn = lane & 15;
    // even split of the piece between the two wave rows / columns
    const int ga0 = (gp + 1) >> 1, gb0 = (gq + 1) >> 1;
    const int na = wr == 0 ? ga0 : gp - ga0, nb = wc == 0 ? gb0 : gq - gb0;
    const int ra = wr * ga0 * 16, cb = wc * gb0 * 16;
    const int offa = fk * LDN + ra + fn, offb = PA + fk * LDN + cb + fn;
    const double *A = ao_c + tl.ao_off[t], *W = aow_c + tl.aow_off[t];
    // a diagonal block needs M_ii only: (M + M^T)_ii is folded in the epilogue (entry (r, c) goes to [max][min], twice on r == c)
    const int nphase = diag ? 1 : 2;

    double4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) acc[a][b] = double4_t{0, 0, 0, 0};

    auto kloop = [&](auto NAc, auto NBc) {
        constexpr int NA = decltype(NAc)::value, NB = decltype(NBc)::value;
        for (int ph = 0; ph < nphase; ph++) {
            // phase 0: rows of A (columns p0..) x rows of W (columns q0..); phase 1: W (p0..) x A (q0..)
            const __amdgpu_buffer_rsrc_t r_a = __builtin_amdgcn_make_buffer_rsrc((void *)((ph ? W : A) + p0), 0, 0xffffffff, 0x00020000);
            const __amdgpu_buffer_rsrc_t r_b = __builtin_amdgcn_make_buffer_rsrc((void *)((ph ? A : W) + q0), 0, 0xffffffff, 0x00020000);
            auto stage_row = [&](int k0, double *dst, int j) {
                const int k = wave * 4 + j;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_a, (__attribute__((address_space(3))) void *)(dst + k * LDN), 16, voff, (k0 + k) * ld8, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_b, (__attribute__((address_space(3))) void *)(dst + PA + k * LDN), 16, voff, (k0 + k) * ld8, 0, 0);
            };
            auto step = [&](const double *cur, double *nxt, int k0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                const int kn = (k0 + KB < G) ? k0 + KB : k0;
#pragma unroll
                for (int kk = 0; kk < KB; kk += 4) {
                    double af[4], bf[4];
#pragma unroll
                    for (int a = 0; a < NA; a++) af[a] = cur[offa + kk * LDN + a * 16];
#pragma unroll
                    for (int b = 0; b < NB; b++) bf[b] = cur[offb + kk * LDN + b * 16];
                    if (PROBE == 2) {
                    } else if (BURST) {
                        if (kk == 0) {
#pragma unroll
                            for (int j = 0; j < 4; j++) stage_row(kn, nxt, j);
                        }
                    } else {
                        stage_row(kn, nxt, kk >> 2);
                    }
                    if (PROBE == 3) {
                        acc[0][0][0] += af[0] + bf[0];
                        continue;
                    }
#pragma unroll
                    for (int a = 0; a < NA; a++)
#pragma unroll
                        for (int b = 0; b < NB; b++) acc[a][b] = mfma_f64_16x16x4(af[a], bf[b], acc[a][b]);
                }
            };
            __syncthreads();                   // every wave is done reading the buffers of the previous phase
#pragma unroll
            for (int j = 0; j < 4; j++) stage_row(0, sb0, j);
            for (int k0 = 0; k0 < G; k0 += 2 * KB) {
                step(sb0, sb1, k0);
                if (k0 + KB < G) step(sb1, sb0, k0 + KB);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last tile re-loaded itself: let it land before the buffers are reused
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    auto pick_b = [&](auto NAc) {
        switch (nb) {
        case 1: kloop(NAc, I1{}); break;
        case 2: kloop(NAc, I2{}); break;
        case 3: kloop(NAc, I3{}); break;
        default: kloop(NAc, I4{}); break;
        }
    };
    if (na <= 0 || nb <= 0) {
        kloop(I0{}, I0{});                 // nothing to compute: stage the DMA rows only (the barriers are the workgroup's)
        return;
    }
    switch (na) {
    case 1: pick_b(I1{}); break;
    case 2: pick_b(I2{}); break;
    case 3: pick_b(I3{}); break;
    default: pick_b(I4{}); break;
    }
    if (PROBE == 1) {
        double sum = 0;                                    // keeps every accumulator alive
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (sum == 1.2345e300) vmat[0] = 1;
        return;
    }
    const int *idx = tl.idx + tl.idx_off[t];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        if (b >= nb) continue;
        const int col = q0 + cb + b * 16 + fn;
        if (col >= ld) continue;
        const int aj = idx[col];
        if (aj >= nao) continue;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            if (a >= na) continue;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int rowi = p0 + ra + a * 16 + fk + 4 * r;
                if (rowi >= ld) continue;
                const int ai = idx[rowi];
                if (ai >= nao) continue;
                if (!diag) unsafeAtomicAdd(vmat + (long)ai * ldv + aj, acc[a][b][r]);
                else if (rowi > col) unsafeAtomicAdd(vmat + (long)ai * ldv + aj, acc[a][b][r]);
                else if (rowi < col) unsafeAtomicAdd(vmat + (long)aj * ldv + ai, acc[a][b][r]);
                else unsafeAtomicAdd(vmat + (long)ai * ldv + aj, 2.0 * acc[a][b][r]);
            }
        }
    }
}

// out[i][j] = out[j][i] = part[max(i, j)][min(i, j)]: completes a matrix accumulated on its lower triangle
__global__ void mirror_tril_kernel(const double *__restrict__ part, int m, int ldc, double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= m) return;
    out[(long)i * m + j] = i >= j ? part[(long)i * ldc + j] : part[(long)j * ldc + i];
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

int PAMD_set_tuning_xc(const char *key, int value)
{
    if (strcmp(key, "orbdotdma") == 0) { g_orb_dot_dma = value; return 0; }
    if (strcmp(key, "orbrho") == 0) { g_orb_rho_fused = value; return 0; }
    if (strcmp(key, "orbrho128") == 0) { g_orb_rho_128 = value; return 0; }
    if (strcmp(key, "vmatxcd") == 0) { g_vmat_xcd = value; return 0; }
    if (strcmp(key, "vmatflip") == 0) { g_vmat_flip = value; return 0; }
    if (strcmp(key, "vmateven") == 0) { g_vmat_even = value; return 0; }
    if (strcmp(key, "vmatburst") == 0) { g_vmat_burst = value; return 0; }
    if (strcmp(key, "vmatprobe") == 0) { g_vmat_probe = value; return 0; }
    return pamd::set_error(-3, "unknown tuning key", __FILE__, __LINE__);
}

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
    if (g_orb_dot_dma && mt >= 8 && ldo >= nchunk * 160 && ldo % 2 == 0 && (uintptr_t)d_orb % 16 == 0 &&
        (uintptr_t)d_ao_c % 16 == 0) {
        // 160-orbital chunks: all-DMA kernel (padding columns of a tile gather the zero row of d_orb: row index nao, which
        // must stay below 2^31 / ldo8 - checked by the caller's allocation of d_orb)
        sub_orb_dot2_kernel<<<grid, 256, 0, st>>>(d_ao_c, tl, G, nchunk, d_orb, ldo, nocc_pad, d_cmo, comp_stride, ldc);
        PAMD_CHECK_LAUNCH();
        return 0;
    }
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

// r04: rho[4][ldg] (rho, grad rho; entry t*G + g) of the density sum_i sign_i c_i c_i^T on the GGA compact image in ONE kernel
// (PAMD_sub_orb_dot + PAMD_rho_from_mo without the c[comp][i][g] buffer; numint.py:328-469).  Returns 1 - nothing launched -
// when the shape has no fused kernel (fewer than 128 orbitals per 160-chunk, unaligned operands): the caller then runs the two
// separate calls.  d_sign nullable [nocc].  With more than one orbital chunk (nocc_pad > 160) d_rho must be zeroed by the
// caller (the chunks add up by atomics); with one chunk every entry of the ntile * G points is written.
int PAMD_sub_orb_rho(const double *d_ao_c, const long *d_ao_off, const long *d_idx_off, const int *d_ld, const int *d_idx,
                     int ntile, int G, const double *d_orb, int ldo, int nocc, int nocc_pad, const double *d_sign, double *d_rho,
                     long ldg, void *stream)
{
    PAMD_REQUIRE(nocc_pad % 16 == 0 && nocc_pad <= ldo, "nocc_pad must be a multiple of 16 and <= ldo");
    PAMD_REQUIRE(G % NT == 0, "tile size must be a multiple of 128");
    if (ntile == 0 || nocc_pad == 0) return 0;
    const int mt_total = nocc_pad / 16;
    int nchunk = ceil_div(mt_total, 10);
    const int mt = ceil_div(mt_total, nchunk);
    // r06: 128-orbital chunks when they cover the orbitals with fewer MFMA tiles than 160-wide ones (nocc_pad = 240: 2 x 8 = 16
    // tiles instead of 2 x 10 = 20); tuning key "orbrho128" = 0 keeps the 160-wide chunks
    const int n128 = ceil_div(mt_total, 8);
    const bool narrow = g_orb_rho_128 && n128 * 8 < nchunk * 10 && ldo >= n128 * 128 + 32;
    if (narrow) nchunk = n128;
    if (!(g_orb_rho_fused && g_orb_dot_dma && mt >= 8 && ldo >= nchunk * (narrow ? 128 : 160) && ldo % 2 == 0 && (uintptr_t)d_orb % 16 == 0 &&
          (uintptr_t)d_ao_c % 16 == 0 && (long)ntile * nchunk < 65536))
        return 1;
    SubTiles tl{d_ao_off, nullptr, d_idx_off, d_ld, d_idx};
    dim3 grid(G / 32, 1, ntile * nchunk);
    if (narrow) sub_orb_rho_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(d_ao_c, tl, G, nchunk, d_orb, ldo, nocc, d_sign, d_rho, ldg);
    else sub_orb_rho_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(d_ao_c, tl, G, nchunk, d_orb, ldo, nocc, d_sign, d_rho, ldg);
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
    sub_vmat_kernel<<<nwork, 256, 0, (hipStream_t)stream>>>(d_ao_c, d_aow_c, tl, d_work, G, nao, d_vmat, ldv, g_vmat_flip);
    PAMD_CHECK_LAUNCH();
    return 0;
}

// r04: vmat (lower triangle) += the lower triangle of  M + M^T,  M = sum_tiles scatter(ao_c[t][0]^T aow_c[t]),  for the nwork
// work items {tile, p0, gp, q0, gq, diag} of d_work[6 nwork]: piece pairs (i >= j) of a balanced cut of the tile's 16-column
// groups (p0 / q0 first compact column, gp / gq <= 8 groups, diag = 1 for i == j); PAMD_sub_vmat_work builds the list on the
// host.  Complete the matrix with PAMD_mirror_tril.  Replaces PAMD_sub_vmat + PAMD_reduce_sym of r03.
int PAMD_sub_vmat_sym(const double *d_ao_c, const long *d_ao_off, const double *d_aow_c, const long *d_aow_off,
                      const long *d_idx_off, const int *d_ld, const int *d_idx, const int *d_work, int nwork, int G, int nao,
                      double *d_vmat, long ldv, void *stream)
{
    PAMD_REQUIRE(G % KB == 0, "tile size must be a multiple of 16");
    PAMD_REQUIRE(((uintptr_t)d_ao_c | (uintptr_t)d_aow_c) % 16 == 0, "16-byte aligned operands");
    if (nwork == 0) return 0;
    SubTiles tl{d_ao_off, d_aow_off, d_idx_off, d_ld, d_idx};
    #define LAUNCH_VS(B, P) sub_vmat_sym_kernel<B, P><<<nwork, 256, 0, (hipStream_t)stream>>>(d_ao_c, d_aow_c, tl, d_work, G, nao, d_vmat, ldv, g_vmat_flip)
    if (g_vmat_probe == 1) LAUNCH_VS(false, 1);
    else if (g_vmat_probe == 2) LAUNCH_VS(false, 2);
    else if (g_vmat_probe == 3) LAUNCH_VS(false, 3);
    else if (g_vmat_burst) LAUNCH_VS(true, 0);
    else LAUNCH_VS(false, 0);
#undef LAUNCH_VS
    PAMD_CHECK_LAUNCH();
    return 0;
}

// Host helper: the work list of PAMD_sub_vmat_sym for `ntile` tiles with leading dimensions ld[] (multiples of 16).
// work == NULL: returns the number of items only.  Returns the item count (6 ints each).
// The list is in DISPATCH order: workgroup b runs on XCD b % 8 (round-robin dispatch), so position b holds the next item of
// that XCD's own queue.  Tiles are dealt to the eight queues largest first, each to the queue with the least MFMA work so far
// (cost of an item = ceil(gp / 2) ceil(gq / 2) tile-MFMAs per k-step, twice that off the diagonal), and all items of a tile sit in ONE queue: the blocks
// that read the same two panels run on one XCD at about the same time and share its L2 (r03: every XCD fetched the panels from
// HBM again, 4 x the algorithmic bytes).  Shorter queues are padded with no-op items (gp = 0).  "vmatxcd" = 0: plain list.
long PAMD_sub_vmat_work(const int *ld, int ntile, int *work)
{
    std::vector<int> order(ntile);
    for (int i = 0; i < ntile; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return ld[a] > ld[b]; });
    const int nq = g_vmat_xcd ? 8 : 1;
    std::vector<std::vector<int>> queue(nq);
    std::vector<double> load(nq, 0.0);
    for (int t : order) {
        const int g = ld[t] / 16, np = (g + 7) / 8;
        if (np == 0) continue;
        std::vector<int> first(np + 1, 0);
        if (g_vmat_even) {
            // r06: pieces of an EVEN number of groups wherever possible.  A piece is split between two wave rows (columns); an odd
            // piece (7 = 4 + 3) leaves the workgroup waiting at every k-tile barrier for its 4 x 4 wave while the 3 x 3 wave idles
            // (model over config 3's plan: useful / (4 x slowest wave) = 0.859 with nearly equal pieces, 0.963 with these).  The
            // (g + 1) / 2 pairs of groups are dealt nearly equally; an odd g takes its one group off a largest piece (8 -> 7).
            const int h = (g + 1) / 2, base = h / np, rem = h % np;
            for (int i = 0; i < np; i++) first[i + 1] = first[i] + 2 * (base + (i < rem ? 1 : 0)) - (i == 0 && (g & 1) ? 1 : 0);
        } else {
            const int base = g / np, rem = g % np;
            for (int i = 0; i < np; i++) first[i + 1] = first[i] + base + (i < rem ? 1 : 0);
        }
        int qi = 0;
        for (int k = 1; k < nq; k++) if (load[k] < load[qi]) qi = k;
        for (int i = 0; i < np; i++)
            for (int j = 0; j <= i; j++) {
                const int gp = first[i + 1] - first[i], gq = first[j + 1] - first[j];
                const int w[6] = {t, first[i] * 16, gp, first[j] * 16, gq, i == j};
                queue[qi].insert(queue[qi].end(), w, w + 6);
                load[qi] += ((gp + 1) / 2) * ((gq + 1) / 2) * (i == j ? 1 : 2);
            }
    }
    size_t depth = 0;
    for (auto &q : queue) depth = std::max(depth, q.size() / 6);
    const long n = (long)depth * nq;
    if (work) {
        for (size_t d = 0; d < depth; d++)
            for (int k = 0; k < nq; k++) {
                int *w = work + 6 * (d * nq + k);
                if (d * 6 < queue[k].size()) memcpy(w, &queue[k][d * 6], 6 * sizeof(int));
                else { w[0] = 0; w[1] = 0; w[2] = 0; w[3] = 0; w[4] = 0; w[5] = 0; }
            }
    }
    return n;
}

int PAMD_mirror_tril(const double *d_part, int m, int ldc, double *d_out, void *stream)
{
    dim3 grid(ceil_div(m, 256), m);
    mirror_tril_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(d_part, m, ldc, d_out);
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
