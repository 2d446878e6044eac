// Host-array, opaque-handle entry points of the XC quadrature (mi_xc_build_grids / mi_nr_rks).
//
// The reference's numint reaches C with raw host pointers (pyscf/dft/numint.py:1074-1190 nr_rks -> eval_ao / eval_rho / libxc /
// _dot_ao_ao, every buffer a numpy array of the caller); the PAMD_sub_* / PAMD_eval_* launchers of this library take DEVICE
// pointers.  The functions below are the host-array form for callers without a device runtime (no torch):
//
//   PAMD_grid_weights_host  Becke / Stratmann / LKO weights of one atom's points: VXCgen_grid (lib/dft/grid_basis.c:32-101) +
//                           the normalisation of gen_grid.get_partition (pyscf/dft/gen_grid.py:341-419), host arrays in / out
//   PAMD_xc_create          libcint-format tables of the molecule + the quadrature (coords, weights as a Grids object holds them,
//                           pyscf/dft/gen_grid.py:487-744) -> handle owning the block-sparse plan of pyscf_amd/dft/sparse_grid.py:
//                           grid tiles of 512 points, per tile the AO shells with a value above 1e-13 (the role of GTO_screen_index,
//                           lib/gto/grid_ao_drv.c:32-123, computed from the values), their AO values compacted and cached in HBM
//   PAMD_xc_nr_rks          numint.nr_rks (numint.py:1074-1190) for densities given by orbital factors D = sum_i s_i c_i c_i^T
//                           (occupied orbitals scaled by sqrt(occ), s = +1; or a signed eigen-factorisation made by the caller):
//                           nelec, exc, vmat (host)
//   PAMD_xc_nr_uks          numint.nr_uks (numint.py:1192-1324), two spin densities
//
// The xc description is parsed on the host side of the binding exactly as in the reference (libxc.parse_xc, dft/libxc.py:496-720
// -> LIBXC_eval_xc(ids, facs), lib/dft/libxc_itrf.c:968-1024): this boundary takes the component weights fac[PAMD_XC_NFAC].
// Kernels: PAMD_eval_ao, PAMD_sub_gather_ao, PAMD_sub_orb_dot, PAMD_rho_from_mo, PAMD_eval_xc(_pol), PAMD_sub_scale_ao,
// PAMD_sub_vmat_sym, PAMD_mirror_tril - the same launch sequence as NumInt._sparse_xc (pyscf_amd/dft/numint.py).
#include <algorithm>
#include <cmath>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#include "host_tables.h"
#include "../../include/pyscf_amd.h"

using namespace pamd;
using namespace pamd::host;

namespace {

constexpr int XC_G = 512;                       // grid points per tile (NumInt.sparse_tile)
constexpr double XC_CUTOFF = 1e-13;             // NumInt.sparse_cutoff (the reference's `cutoff = CUTOFF * 1e2`, numint.py:2845)
constexpr size_t XC_BLOCK_BYTES = 6ul << 30;    // dense evaluation block / orbital-product work space

struct XcChunk {                                // a launch group of consecutive tiles
    int t0 = 0, nt = 0, ld_max = 0;
    long aow_base = 0, aow_size = 0, nwork = 0;
    int *d_work = nullptr;
};

struct XcPlan {
    bool built = false;
    int ncomp = 1, ntile = 0;
    std::vector<int> ld;
    std::vector<long> ao_off, aow_off, idx_off;
    long *d_ao_off = nullptr, *d_aow_off = nullptr, *d_idx_off = nullptr;
    int *d_ld = nullptr, *d_idx = nullptr;
    double *d_ao_c = nullptr;
    long ao_total = 0, max_aow = 0;
    std::vector<XcChunk> chunks;
    double density = 0;
};

// smax[tile][shell] = max over the tile's rows, the shell's functions and the components of |ao|
__global__ __launch_bounds__(256) void shell_tile_max_kernel(const double *__restrict__ ao, long comp_stride, int ncomp, int ldao, int G,
                                                             long nvalid, const int *__restrict__ sh_ao0, const int *__restrict__ sh_l,
                                                             int nsh, double *__restrict__ smax)
{
    const int t = blockIdx.y, s = blockIdx.x;
    const int f0 = sh_ao0[s], nf = 2 * sh_l[s] + 1;
    double m = 0;
    for (int e = threadIdx.x; e < G * nf; e += 256) {
        const int g = e / nf, f = e - g * nf;
        const long row = (long)t * G + g;
        if (row >= nvalid) continue;
        for (int c = 0; c < ncomp; c++) m = fmax(m, fabs(ao[c * comp_stride + row * ldao + f0 + f]));
    }
    __shared__ double red[256];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) smax[(long)t * nsh + s] = red[0];
}

// w[g] = vol[g] * pb[ia][g] / sum_a pb[a][g]
__global__ void becke_normalise_kernel(const double *__restrict__ pb, int natm, long ng, int ia, const double *__restrict__ vol,
                                       double *__restrict__ w)
{
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ng) return;
    double s = 0;
    for (int a = 0; a < natm; a++) s += pb[(long)a * ng + g];
    w[g] = vol[g] * pb[(long)ia * ng + g] / s;
}

}  // namespace

struct PAMD_xc {
    int device = 0, nao = 0, nsh = 0, ldao = 0;
    long ngrids = 0;
    int ntile = 0;
    hipStream_t st = nullptr;
    DevPool pool;
    Shells sh;
    int *d_l = nullptr, *d_ao0 = nullptr, *d_prim0 = nullptr, *d_nprim = nullptr, *d_fn2sh = nullptr, *d_c2s_off = nullptr;
    double *d_xyz = nullptr, *d_exps = nullptr, *d_coefs = nullptr, *d_c2s = nullptr;
    double *d_coords = nullptr, *d_weights = nullptr;            // weights zero-padded to ntile * G
    XcPlan plan[2];                                              // LDA (1 component), GGA (4)
    std::vector<PAMD_xc *> parts;                                // multi-device handle: grid tiles dealt round-robin (owned)
    int peer_ok = 0;
    // HIP events around the kernels of the last contraction, on the launch stream (PAMD_xc_last_timing; r05):
    // kinds 0 orbital product (+ rho), 1 eval_xc, 2 scale, 3 vmat
    std::vector<hipEvent_t> tev;
    double t_ms[4] = {0, 0, 0, 0};
    double last_sum_ld = 0, last_sum_ld2 = 0;
    int last_ncomp = 1, last_nocc_pad = 0;
    hipEvent_t timing_event(size_t i)
    {
        while (tev.size() <= i) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            tev.push_back(e);
        }
        return tev[i];
    }
    std::map<std::string, std::pair<double *, size_t>> ws;
    double *workspace(const std::string &name, size_t ndoubles, int *rc)
    {
        auto it = ws.find(name);
        *rc = 0;
        if (it != ws.end() && it->second.second >= ndoubles) return it->second.first;
        if (it != ws.end()) pool.release(it->second.first);
        double *p = nullptr;
        *rc = pool.alloc((void **)&p, (ndoubles + 256) * 8);
        if (*rc) return nullptr;
        (void)hipMemsetAsync(p, 0, (ndoubles + 256) * 8, st);
        ws[name] = {p, ndoubles};
        return p;
    }
    ~PAMD_xc()
    {
        for (PAMD_xc *p : parts) {
            (void)hipSetDevice(p->device);
            delete p;
        }
        if (!parts.empty()) (void)hipSetDevice(device);
        for (hipEvent_t e : tev) (void)hipEventDestroy(e);
        if (st) (void)hipStreamDestroy(st);
    }
};

namespace {

int build_plan(PAMD_xc *h, int gga)
{
    XcPlan &pl = h->plan[gga ? 1 : 0];
    if (pl.built) return 0;
    int rc;
    const int G = XC_G, nao = h->nao, nsh = h->nsh, ldao = h->ldao;
    const int ncomp = gga ? 4 : 1;
    pl.ncomp = ncomp;
    pl.ntile = h->ntile;
    const int ntile = h->ntile;
    // dense evaluation passes over runs of tiles
    const int run = (int)std::max<size_t>(1, XC_BLOCK_BYTES / ((size_t)ncomp * ldao * 8 * G));
    const long rows = (long)std::min(run, ntile) * G;
    double *d_dense = nullptr, *d_smax = nullptr;
    if ((rc = h->pool.alloc((void **)&d_dense, (size_t)ncomp * rows * ldao * 8))) return rc;
    if ((rc = h->pool.alloc((void **)&d_smax, (size_t)std::max(ntile, 1) * nsh * 8))) return rc;
    PAMD_CHECK_HIP(hipMemsetAsync(d_dense, 0, (size_t)ncomp * rows * ldao * 8, h->st));
    auto eval_run = [&](int t0, int cnt, long *ng_out) -> int {
        const long g0 = (long)t0 * G, ng = std::min<long>((long)cnt * G, h->ngrids - g0);
        *ng_out = ng;
        return PAMD_eval_ao(gga ? 1 : 0, h->d_l, h->d_ao0, h->d_prim0, h->d_nprim, h->d_xyz, h->d_exps, h->d_coefs, nsh, h->d_fn2sh, nao,
                            h->d_coords, g0, ng, h->d_c2s, h->d_c2s_off, d_dense, rows, ldao, 0.0, nullptr, h->st);
    };
    // 1. screen: active[tile][shell]
    for (int t0 = 0; t0 < ntile; t0 += run) {
        const int cnt = std::min(run, ntile - t0);
        long ng;
        if ((rc = eval_run(t0, cnt, &ng))) return rc;
        shell_tile_max_kernel<<<dim3(nsh, cnt), 256, 0, h->st>>>(d_dense, rows * ldao, ncomp, ldao, G, ng, h->d_ao0, h->d_l, nsh,
                                                                d_smax + (size_t)t0 * nsh);
        PAMD_CHECK_LAUNCH();
    }
    std::vector<double> smax((size_t)ntile * nsh);
    PAMD_CHECK_HIP(hipMemcpyAsync(smax.data(), d_smax, smax.size() * 8, hipMemcpyDeviceToHost, h->st));
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    h->pool.release(d_smax);
    // 2. tables (sparse_grid.SparsePlan._tables): compact column -> AO index, ascending; padding columns point at row nao
    pl.ld.assign(ntile, 16);
    pl.idx_off.assign(ntile, 0);
    pl.ao_off.assign(ntile, 0);
    pl.aow_off.assign(ntile, 0);
    std::vector<int> idx;
    double dens = 0;
    for (int t = 0; t < ntile; t++) {
        std::vector<int> fns;
        for (int s = 0; s < nsh; s++)
            if (smax[(size_t)t * nsh + s] > XC_CUTOFF)
                for (int f = 0; f < 2 * h->sh.l[s] + 1; f++) fns.push_back(h->sh.ao0[s] + f);
        const int l = (int)round_up(std::max<long>((long)fns.size(), 1), 16);
        pl.ld[t] = l;
        pl.idx_off[t] = (long)idx.size();
        idx.insert(idx.end(), fns.begin(), fns.end());
        idx.insert(idx.end(), l - fns.size(), nao);
        dens += (double)fns.size() / nao;
    }
    pl.density = ntile ? dens / ntile : 0;
    long off = 0;
    for (int t = 0; t < ntile; t++) { pl.ao_off[t] = off; off += (long)ncomp * G * pl.ld[t]; }
    pl.ao_total = off;
    // 3. launch groups bounded by the orbital-product work space; aow offsets restart in every group
    const int per = (int)std::max<size_t>(1, XC_BLOCK_BYTES / ((size_t)ncomp * 8 * G * 256));
    for (int t0 = 0; t0 < ntile; t0 += per) {
        XcChunk ch;
        ch.t0 = t0;
        ch.nt = std::min(per, ntile - t0);
        long a = 0;
        for (int t = t0; t < t0 + ch.nt; t++) {
            pl.aow_off[t] = a;
            a += (long)G * pl.ld[t];
            ch.ld_max = std::max(ch.ld_max, pl.ld[t]);
        }
        ch.aow_size = a;
        pl.max_aow = std::max(pl.max_aow, a);
        ch.nwork = PAMD_sub_vmat_work(pl.ld.data() + t0, ch.nt, nullptr);
        std::vector<int> work((size_t)std::max<long>(ch.nwork, 1) * 6, 0);
        PAMD_sub_vmat_work(pl.ld.data() + t0, ch.nt, work.data());
        if ((rc = upload(h->pool, work, &ch.d_work))) return rc;
        pl.chunks.push_back(ch);
    }
    if ((rc = upload(h->pool, pl.ld, &pl.d_ld)) || (rc = upload(h->pool, pl.ao_off, &pl.d_ao_off)) ||
        (rc = upload(h->pool, pl.aow_off, &pl.d_aow_off)) || (rc = upload(h->pool, pl.idx_off, &pl.d_idx_off)) ||
        (rc = upload(h->pool, idx, &pl.d_idx)))
        return rc;
    // 4. the compact image, cached for the life of the handle
    size_t free_b = 0, total_b = 0;
    PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    if (((size_t)pl.ao_total + 256) * 8 + (8ul << 30) > free_b) {
        snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_xc: the compact AO image (%.1f GB) does not fit the %.1f GB of free HBM",
                 pl.ao_total * 8e-9, free_b * 1e-9);
        return -2;
    }
    if ((rc = h->pool.alloc((void **)&pl.d_ao_c, ((size_t)pl.ao_total + 256) * 8))) return rc;
    PAMD_CHECK_HIP(hipMemsetAsync(pl.d_ao_c + pl.ao_total, 0, 256 * 8, h->st));
    for (int t0 = 0; t0 < ntile; t0 += run) {
        const int cnt = std::min(run, ntile - t0);
        long ng;
        if ((rc = eval_run(t0, cnt, &ng))) return rc;
        int ldm = 16;
        for (int t = t0; t < t0 + cnt; t++) ldm = std::max(ldm, pl.ld[t]);
        if ((rc = PAMD_sub_gather_ao(d_dense, rows, ldao, ncomp, 0, ng, pl.d_ao_off + t0, pl.d_idx_off + t0, pl.d_ld + t0, pl.d_idx, cnt, G,
                                     ldm, nao, pl.d_ao_c, h->st)))
            return rc;
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    h->pool.release(d_dense);
    pl.built = true;
    return 0;
}

struct OrbOp { double *d_orb = nullptr, *d_sign = nullptr; int nocc = 0, nocc_pad = 16; long ldo = 16; };

// orbital factors of one density -> device operand of PAMD_sub_orb_dot: rows = round_up(nao + 1, 16), row nao and beyond zero
int upload_orbitals(PAMD_xc *h, const char *tag, const double *orb, int nocc, const double *sign, OrbOp *o)
{
    int rc;
    const int nao = h->nao;
    o->nocc = nocc;
    o->nocc_pad = (int)round_up(std::max(nocc, 1), 16);
    o->ldo = o->nocc_pad > 160 ? round_up(o->nocc_pad, 160) : o->nocc_pad;
    const long rows = round_up(nao + 1, 16);
    std::vector<double> oh((size_t)rows * o->ldo, 0.0);
    for (int p = 0; p < nao; p++)
        for (int i = 0; i < nocc; i++) oh[(size_t)p * o->ldo + i] = orb[(size_t)p * nocc + i];
    o->d_orb = h->workspace(std::string("orb") + tag, oh.size(), &rc);
    if (rc) return rc;
    PAMD_CHECK_HIP(hipMemcpyAsync(o->d_orb, oh.data(), oh.size() * 8, hipMemcpyHostToDevice, h->st));
    o->d_sign = nullptr;
    if (sign && nocc) {
        bool neg = false;
        for (int i = 0; i < nocc; i++) neg = neg || sign[i] < 0;
        if (neg) {
            o->d_sign = h->workspace(std::string("sign") + tag, (size_t)nocc, &rc);
            if (rc) return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(o->d_sign, sign, (size_t)nocc * 8, hipMemcpyHostToDevice, h->st));
        }
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));                    // oh goes out of scope
    return 0;
}

// nelec / exc / vmat of one closed-shell density (spin = 0, ops[0]) or of a spin pair (spin = 1, ops[0], ops[1])
// vmat == NULL: the matrices stay on the device (work space "V", [nset][nao][nao]) for the multi-device sum
int xc_contract(PAMD_xc *h, const double *fac, int gga, int spin, OrbOp *ops, double *acc_h, double *vmat)
{
    int rc;
    if ((rc = build_plan(h, gga))) return rc;
    XcPlan &pl = h->plan[gga ? 1 : 0];
    const int G = XC_G, nao = h->nao, ncomp = pl.ncomp, nset = spin ? 2 : 1;
    const size_t n2 = (size_t)nao * nao;
    hipStream_t st = h->st;
    int nt_max = 0, nocc_pad_max = 16;
    for (const XcChunk &ch : pl.chunks) nt_max = std::max(nt_max, ch.nt);
    for (int s = 0; s < nset; s++) nocc_pad_max = std::max(nocc_pad_max, ops[s].nocc_pad);
    const long ldg = (long)std::max(nt_max, 1) * G;
    double *d_M = h->workspace("M", (size_t)nset * n2, &rc);
    if (rc) return rc;
    double *d_V = h->workspace("V", (size_t)nset * n2, &rc);
    if (rc) return rc;
    double *d_acc = h->workspace("acc", 4, &rc);
    if (rc) return rc;
    double *d_rho = h->workspace("rho", (size_t)nset * 4 * ldg, &rc);
    if (rc) return rc;
    double *d_wv = h->workspace("wv", (size_t)nset * 4 * ldg, &rc);
    if (rc) return rc;
    double *d_cmo = nullptr;                  // only the unfused orbital product needs it (LDA, fewer than 128 orbitals per chunk)
    double *d_aow = h->workspace("aow", (size_t)pl.max_aow, &rc);
    if (rc) return rc;
    PAMD_CHECK_HIP(hipMemsetAsync(d_M, 0, (size_t)nset * n2 * 8, st));
    PAMD_CHECK_HIP(hipMemsetAsync(d_acc, 0, 4 * 8, st));
    std::vector<int> mark_kind;                 // event i opens (kind) / event i + 1 closes it
    auto mark = [&](int kind) {
        if (mark_kind.size() >= 1024) return;
        hipEvent_t e = h->timing_event(mark_kind.size());
        if (!e) return;
        (void)hipEventRecord(e, st);
        mark_kind.push_back(kind);
    };
    for (const XcChunk &ch : pl.chunks) {
        const int t0 = ch.t0, nt = ch.nt;
        const long npts = (long)nt * G;
        const long *ao_off = pl.d_ao_off + t0, *aow_off = pl.d_aow_off + t0, *idx_off = pl.d_idx_off + t0;
        const int *ld = pl.d_ld + t0;
        const double *w_ch = h->d_weights + (size_t)t0 * G;
        for (int s = 0; s < nset; s++) {
            double *rho_s = d_rho + (size_t)s * 4 * ldg;
            if (ops[s].nocc == 0) {
                PAMD_CHECK_HIP(hipMemsetAsync(rho_s, 0, (size_t)4 * ldg * 8, st));
                continue;
            }
            const long cs = (long)ops[s].nocc_pad * npts;
            mark(0);
            if (gga) {
                // rho / grad rho in the orbital product's epilogue (no c[comp][i][g] buffer); 1 = no fused kernel for this shape
                if (ops[s].nocc_pad > 160) PAMD_CHECK_HIP(hipMemsetAsync(rho_s, 0, (size_t)4 * ldg * 8, st));
                rc = PAMD_sub_orb_rho(pl.d_ao_c, ao_off, idx_off, ld, pl.d_idx, nt, G, ops[s].d_orb, (int)ops[s].ldo, ops[s].nocc,
                                      ops[s].nocc_pad, ops[s].d_sign, rho_s, ldg, st);
                if (rc < 0) return rc;
                if (rc == 0) { mark(-1); continue; }
            }
            if (!d_cmo) {
                d_cmo = h->workspace("cmo", (size_t)ncomp * nocc_pad_max * ldg, &rc);
                if (rc) return rc;
            }
            if ((rc = PAMD_sub_orb_dot(pl.d_ao_c, ao_off, idx_off, ld, pl.d_idx, nt, G, ncomp, ops[s].d_orb, (int)ops[s].ldo, ops[s].nocc_pad,
                                       d_cmo, cs, npts, st)))
                return rc;
            if ((rc = PAMD_rho_from_mo(d_cmo, cs, npts, ops[s].nocc, ncomp, npts, rho_s, ldg, ops[s].d_sign, st))) return rc;
            mark(-1);
        }
        mark(1);
        if (spin)
            rc = PAMD_eval_xc_pol(fac, gga, d_rho, d_rho + (size_t)4 * ldg, w_ch, npts, ldg, d_wv, d_wv + (size_t)4 * ldg, d_acc, nullptr, st);
        else
            rc = PAMD_eval_xc(fac, gga, d_rho, w_ch, npts, ldg, d_wv, nullptr, d_acc, st);
        if (rc) return rc;
        mark(-1);
        for (int s = 0; s < nset; s++) {
            mark(2);
            if ((rc = PAMD_sub_scale_ao(pl.d_ao_c, ao_off, aow_off, ld, nt, G, ncomp, ch.ld_max, d_wv + (size_t)s * 4 * ldg, ldg, d_aow, st)))
                return rc;
            mark(3);
            if ((rc = PAMD_sub_vmat_sym(pl.d_ao_c, ao_off, d_aow, aow_off, idx_off, ld, pl.d_idx, ch.d_work, (int)ch.nwork, G, nao,
                                        d_M + (size_t)s * n2, nao, st)))
                return rc;
            mark(-1);
        }
    }
    for (int s = 0; s < nset; s++)
        if ((rc = PAMD_mirror_tril(d_M + (size_t)s * n2, nao, nao, d_V + (size_t)s * n2, st))) return rc;
    PAMD_CHECK_HIP(hipMemcpyAsync(acc_h, d_acc, 4 * 8, hipMemcpyDeviceToHost, st));
    if (vmat) PAMD_CHECK_HIP(hipMemcpyAsync(vmat, d_V, (size_t)nset * n2 * 8, hipMemcpyDeviceToHost, st));
    PAMD_CHECK_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < 4; k++) h->t_ms[k] = 0;
    for (size_t i = 0; i + 1 < mark_kind.size(); i++)
        if (mark_kind[i] >= 0) {               // (a scale mark is closed by the vmat mark that follows it)
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->tev[i], h->tev[i + 1]) == hipSuccess) h->t_ms[mark_kind[i]] += ms;
        }
    (void)hipGetLastError();
    h->last_sum_ld = h->last_sum_ld2 = 0;
    for (int v : pl.ld) { h->last_sum_ld += v; h->last_sum_ld2 += (double)v * v; }
    h->last_ncomp = ncomp;
    h->last_nocc_pad = nocc_pad_max;
    return 0;
}

__global__ void xc_sum_parts_kernel(const double *__restrict__ in, size_t stride, int nparts, double *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double a = 0;
        for (int p = 0; p < nparts; p++) a += in[(size_t)p * stride + i];
        out[i] = a;
    }
}

std::mutex g_xc_dev_mutex[64];

// one call on every part (one host thread per part), then nelec / exc summed on the host and the matrices on parts[0]'s device
// (peer copies where the devices can reach each other, host bounce otherwise; fixed order)
int xc_multi(PAMD_xc *m, const double *fac, int gga, int spin, const double *orbs, const int *nocc, const double *signs, double *acc_out,
             double *vmat)
{
    const int np = (int)m->parts.size(), nset = spin ? 2 : 1, nao = m->nao;
    const size_t n2 = (size_t)nao * nao, len = (size_t)nset * n2;
    std::vector<int> rcs(np, 0);
    std::vector<std::string> msgs(np);
    std::vector<double> accs((size_t)np * 4, 0.0);
    PAMD_xc *h0 = m->parts[0];
    PAMD_CHECK_HIP(hipSetDevice(h0->device));
    int rc0;
    double *gather = h0->workspace("gather", (size_t)np * len, &rc0);
    if (rc0) return rc0;
    PAMD_CHECK_HIP(hipStreamSynchronize(h0->st));
    std::vector<std::thread> th;
    for (int p = 0; p < np; p++)
        th.emplace_back([&, p]() {
            PAMD_xc *h = m->parts[p];
            auto body = [&]() -> int {
                PAMD_CHECK_HIP(hipSetDevice(h->device));
                std::lock_guard<std::mutex> lock(g_xc_dev_mutex[h->device & 63]);     // parts that share a device take turns
                OrbOp ops[2];
                int rc;
                if ((rc = upload_orbitals(h, "0", orbs, nocc[0], signs, &ops[0]))) return rc;
                if (spin && (rc = upload_orbitals(h, "1", orbs + (size_t)nao * nocc[0], nocc[1], signs ? signs + nocc[0] : nullptr, &ops[1])))
                    return rc;
                if ((rc = xc_contract(h, fac, gga, spin, ops, &accs[(size_t)p * 4], nullptr))) return rc;
                // push this part's matrices into its slot of the gather buffer on parts[0]'s device, from this thread and on this
                // part's stream: the copies of different parts run concurrently (one xGMI link each)
                const double *src = h->ws["V"].first;
                double *dst = gather + (size_t)p * len;
                if (h->device == h0->device) {
                    PAMD_CHECK_HIP(hipMemcpyAsync(dst, src, len * 8, hipMemcpyDeviceToDevice, h->st));
                } else if (m->peer_ok) {
                    PAMD_CHECK_HIP(hipMemcpyPeerAsync(dst, h0->device, src, h->device, len * 8, h->st));
                } else {
                    std::vector<double> bounce(len);
                    PAMD_CHECK_HIP(hipMemcpy(bounce.data(), src, len * 8, hipMemcpyDeviceToHost));
                    PAMD_CHECK_HIP(hipSetDevice(h0->device));
                    PAMD_CHECK_HIP(hipMemcpy(dst, bounce.data(), len * 8, hipMemcpyHostToDevice));
                    PAMD_CHECK_HIP(hipSetDevice(h->device));
                }
                PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
                return 0;
            };
            rcs[p] = body();
            if (rcs[p]) msgs[p] = g_errmsg;
        });
    for (auto &t : th) t.join();
    for (int p = 0; p < np; p++)
        if (rcs[p]) {
            snprintf(g_errmsg, sizeof(g_errmsg), "device %d (part %d): %s", m->parts[p]->device, p, msgs[p].c_str());
            return rcs[p];
        }
    for (int k = 0; k < 4; k++) {
        acc_out[k] = 0;
        for (int p = 0; p < np; p++) acc_out[k] += accs[(size_t)p * 4 + k];
    }
    PAMD_CHECK_HIP(hipSetDevice(h0->device));
    int rc;
    double *total = h0->workspace("total", len, &rc);
    if (rc) return rc;
    xc_sum_parts_kernel<<<1024, 256, 0, h0->st>>>(gather, len, np, total, len);
    PAMD_CHECK_LAUNCH();
    PAMD_CHECK_HIP(hipMemcpyAsync(vmat, total, len * 8, hipMemcpyDeviceToHost, h0->st));
    PAMD_CHECK_HIP(hipStreamSynchronize(h0->st));
    return 0;
}

}  // namespace

extern "C" {

int PAMD_grid_weights_host(const double *coords, long ngrids, const double *atm_coords, int natm, const double *radii_table, int scheme,
                           int ia, const double *vol, int device, double *weights)
{
    PAMD_REQUIRE(coords && atm_coords && vol && weights && ngrids >= 0 && natm > 0 && ia >= 0 && ia < natm, "PAMD_grid_weights_host: bad arguments");
    if (ngrids == 0) return 0;
    PAMD_CHECK_HIP(hipSetDevice(device));
    DevPool tmp;
    int rc;
    double *d_c = nullptr, *d_a = nullptr, *d_rt = nullptr, *d_pb = nullptr, *d_vol = nullptr, *d_w = nullptr;
    if ((rc = tmp.alloc((void **)&d_c, (size_t)ngrids * 3 * 8)) || (rc = tmp.alloc((void **)&d_a, (size_t)natm * 3 * 8)) ||
        (rc = tmp.alloc((void **)&d_pb, (size_t)natm * ngrids * 8)) || (rc = tmp.alloc((void **)&d_vol, (size_t)ngrids * 8)) ||
        (rc = tmp.alloc((void **)&d_w, (size_t)ngrids * 8)))
        return rc;
    PAMD_CHECK_HIP(hipMemcpy(d_c, coords, (size_t)ngrids * 3 * 8, hipMemcpyHostToDevice));
    PAMD_CHECK_HIP(hipMemcpy(d_a, atm_coords, (size_t)natm * 3 * 8, hipMemcpyHostToDevice));
    PAMD_CHECK_HIP(hipMemcpy(d_vol, vol, (size_t)ngrids * 8, hipMemcpyHostToDevice));
    if (radii_table) {
        if ((rc = tmp.alloc((void **)&d_rt, (size_t)natm * natm * 8))) return rc;
        PAMD_CHECK_HIP(hipMemcpy(d_rt, radii_table, (size_t)natm * natm * 8, hipMemcpyHostToDevice));
    }
    if ((rc = PAMD_grid_partition(d_pb, d_c, d_a, d_rt, natm, ngrids, scheme, nullptr))) return rc;
    becke_normalise_kernel<<<ceil_div(ngrids, 256), 256>>>(d_pb, natm, ngrids, ia, d_vol, d_w);
    PAMD_CHECK_LAUNCH();
    PAMD_CHECK_HIP(hipMemcpy(weights, d_w, (size_t)ngrids * 8, hipMemcpyDeviceToHost));
    return 0;
}

// devices[ndev]: the grid TILES (512 consecutive points of the caller's - box-sorted - order) are dealt round-robin over the
// parts, as the torch path deals them over ranks (dft/sparse_grid.py); nelec, exc and vmat are sums over grid points, so every
// part is an ordinary handle on its sub-grid.  A device may be listed more than once.
int PAMD_xc_create_multi(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, const double *coords,
                         const double *weights, long ngrids, const int *devices, int ndev, PAMD_xc **out)
{
    PAMD_REQUIRE(devices && ndev > 0 && ndev <= 64 && out && coords && weights && ngrids > 0, "PAMD_xc_create_multi: bad arguments");
    *out = nullptr;
    PAMD_xc *m = new PAMD_xc;
    struct Guard { PAMD_xc *p; ~Guard() { delete p; } } guard{m};
    m->device = devices[0];
    m->ngrids = ngrids;
    const long ntile = (ngrids + XC_G - 1) / XC_G;
    for (int p = 0; p < ndev; p++) {
        std::vector<double> c, w;
        for (long t = p; t < ntile; t += ndev) {             // (only the grid's last tile can be short, and it ends its part)
            const long g0 = t * XC_G, g1 = std::min<long>(g0 + XC_G, ngrids);
            c.insert(c.end(), coords + 3 * g0, coords + 3 * g1);
            w.insert(w.end(), weights + g0, weights + g1);
        }
        if (w.empty()) { c.assign(coords, coords + 3); w.assign(1, 0.0); }      // more parts than tiles: a weightless point
        PAMD_xc *part = nullptr;
        int rc = PAMD_xc_create(atm, natm, bas, nbas, env, nenv, c.data(), w.data(), (long)w.size(), devices[p], &part);
        if (rc) return rc;
        m->parts.push_back(part);
    }
    m->nao = m->parts[0]->nao;
    m->peer_ok = 1;
    PAMD_CHECK_HIP(hipSetDevice(devices[0]));
    for (int p = 1; p < ndev && m->peer_ok; p++) {
        if (devices[p] == devices[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[0], devices[p]) != hipSuccess || !can) { (void)hipGetLastError(); m->peer_ok = 0; break; }
        hipError_t e = hipDeviceEnablePeerAccess(devices[p], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) m->peer_ok = 0;
        (void)hipGetLastError();
        // ... and the other direction: part p pushes into part 0's gather buffer from its own thread and stream (as
        // PAMD_df_create_ex does; without it the copy may be staged through the host - ADVICE r04)
        int can2 = 0;
        if (hipDeviceCanAccessPeer(&can2, devices[p], devices[0]) != hipSuccess || !can2) { (void)hipGetLastError(); m->peer_ok = 0; break; }
        PAMD_CHECK_HIP(hipSetDevice(devices[p]));
        e = hipDeviceEnablePeerAccess(devices[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) m->peer_ok = 0;
        (void)hipGetLastError();
        PAMD_CHECK_HIP(hipSetDevice(devices[0]));
    }
    guard.p = nullptr;
    *out = m;
    return 0;
}

int PAMD_xc_create(const int *atm, int natm, const int *bas, int nbas, const double *env, int nenv, const double *coords,
                   const double *weights, long ngrids, int device, PAMD_xc **out)
{
    PAMD_REQUIRE(atm && bas && env && coords && weights && out && natm > 0 && nbas > 0 && nenv > 0 && ngrids > 0, "PAMD_xc_create: bad arguments");
    *out = nullptr;
    PAMD_CHECK_HIP(hipSetDevice(device));
    PAMD_xc *h = new PAMD_xc;
    struct Guard { PAMD_xc *p; ~Guard() { delete p; } } guard{h};
    h->device = device;
    PAMD_CHECK_HIP(hipStreamCreate(&h->st));
    h->sh = make_shells(atm, bas, 0, nbas, env);
    const Shells &sh = h->sh;
    for (int l : sh.l) PAMD_REQUIRE(l <= 4, "PAMD_xc_create: AO angular momentum beyond g");
    h->nao = sh.nao;
    h->nsh = sh.n;
    h->ldao = (int)round_up(sh.nao, 16);
    h->ngrids = ngrids;
    h->ntile = (int)((ngrids + XC_G - 1) / XC_G);
    std::vector<int> prim0(sh.n), nprim(sh.n), fn2sh;
    std::vector<double> exps, coefs, c2s;
    std::vector<int> c2s_off;
    for (int i = 0; i < sh.n; i++) {
        prim0[i] = (int)exps.size();
        nprim[i] = (int)sh.exps[i].size();
        exps.insert(exps.end(), sh.exps[i].begin(), sh.exps[i].end());
        coefs.insert(coefs.end(), sh.coefs[i].begin(), sh.coefs[i].end());
        fn2sh.insert(fn2sh.end(), 2 * sh.l[i] + 1, i);
    }
    for (int l = 0; l <= LMAX_TAB; l++) {
        c2s_off.push_back((int)c2s.size());
        std::vector<double> m = c2s_matrix(l);
        c2s.insert(c2s.end(), m.begin(), m.end());
    }
    int rc;
    if ((rc = upload(h->pool, sh.l, &h->d_l)) || (rc = upload(h->pool, sh.ao0, &h->d_ao0)) || (rc = upload(h->pool, prim0, &h->d_prim0)) ||
        (rc = upload(h->pool, nprim, &h->d_nprim)) || (rc = upload(h->pool, fn2sh, &h->d_fn2sh)) || (rc = upload(h->pool, sh.xyz, &h->d_xyz)) ||
        (rc = upload(h->pool, exps, &h->d_exps)) || (rc = upload(h->pool, coefs, &h->d_coefs)) || (rc = upload(h->pool, c2s, &h->d_c2s)) ||
        (rc = upload(h->pool, c2s_off, &h->d_c2s_off)))
        return rc;
    if ((rc = h->pool.alloc((void **)&h->d_coords, (size_t)ngrids * 3 * 8))) return rc;
    PAMD_CHECK_HIP(hipMemcpy(h->d_coords, coords, (size_t)ngrids * 3 * 8, hipMemcpyHostToDevice));
    const size_t wlen = (size_t)h->ntile * XC_G;
    if ((rc = h->pool.alloc((void **)&h->d_weights, wlen * 8))) return rc;
    PAMD_CHECK_HIP(hipMemset(h->d_weights, 0, wlen * 8));
    PAMD_CHECK_HIP(hipMemcpy(h->d_weights, weights, (size_t)ngrids * 8, hipMemcpyHostToDevice));
    guard.p = nullptr;
    *out = h;
    return 0;
}

void PAMD_xc_destroy(PAMD_xc *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    delete h;
}

int PAMD_xc_nao(const PAMD_xc *h, int *nao)
{
    PAMD_REQUIRE(h && nao, "null handle");
    *nao = h->nao;
    return 0;
}

// info[0] = tiles, [1] = mean fraction of AO functions active on a tile, [2] = GB of the cached compact image (of the plan built
// for xctype: 0 LDA, 1 GGA; builds it when needed)
int PAMD_xc_plan_info(PAMD_xc *h, int xctype, double *info)
{
    PAMD_REQUIRE(h && info, "null handle");
    if (!h->parts.empty()) {
        info[0] = info[1] = info[2] = 0;
        for (PAMD_xc *p : h->parts) {
            double pi[3];
            int rc = PAMD_xc_plan_info(p, xctype, pi);
            if (rc) return rc;
            info[0] += pi[0];
            info[1] += pi[1] * pi[0];
            info[2] += pi[2];
        }
        if (info[0] > 0) info[1] /= info[0];
        return 0;
    }
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    int rc = build_plan(h, xctype);
    if (rc) return rc;
    const XcPlan &pl = h->plan[xctype ? 1 : 0];
    info[0] = pl.ntile;
    info[1] = pl.density;
    info[2] = pl.ao_total * 8e-9;
    return 0;
}

// HIP-event timings of the kernels of the LAST PAMD_xc_nr_rks / _nr_uks (bench.py --single-process `xc_path.roofline`; r05):
// out[0] = parts; then the SLOWEST part's ms of {orbital product (+ densities), eval_xc, scale, vmat} in out[1..4]; out[5] = sum over
// ALL tiles of all parts of the compact widths ld_t, out[6] = of ld_t^2 (the flops the two MFMA products execute: 2 ncomp G sum(ld)
// nocc_pad and 2 G sum(ld^2)), out[7] = points per tile G, out[8] = components, out[9] = padded orbital count.
int PAMD_xc_last_timing(const PAMD_xc *h, double *out, int nout)
{
    PAMD_REQUIRE(h && out && nout >= 10, "PAMD_xc_last_timing: out[10]");
    std::vector<const PAMD_xc *> ps;
    if (h->parts.empty()) ps.push_back(h);
    for (const PAMD_xc *p : h->parts) ps.push_back(p);
    for (int i = 0; i < 10; i++) out[i] = 0;
    out[0] = (double)ps.size();
    double worst = -1;
    for (const PAMD_xc *p : ps) {
        const double tot = p->t_ms[0] + p->t_ms[1] + p->t_ms[2] + p->t_ms[3];
        if (tot > worst) {
            worst = tot;
            for (int k = 0; k < 4; k++) out[1 + k] = p->t_ms[k];
        }
        out[5] += p->last_sum_ld;
        out[6] += p->last_sum_ld2;
        out[8] = p->last_ncomp;
        out[9] = p->last_nocc_pad;
    }
    out[7] = XC_G;
    return 0;
}

int PAMD_xc_nr_rks(PAMD_xc *h, const double *fac, int xctype, int nset, const double *orbs, const int *nocc, const double *signs,
                   double *nelec, double *exc, double *vmat)
{
    PAMD_REQUIRE(h && fac && nset > 0 && orbs && nocc && nelec && exc && vmat, "PAMD_xc_nr_rks: bad arguments");
    PAMD_REQUIRE(xctype == 0 || xctype == 1, "PAMD_xc_nr_rks: xctype 0 (LDA) or 1 (GGA)");
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    const size_t n2 = (size_t)h->nao * h->nao;
    const double *o = orbs, *sg = signs;
    for (int s = 0; s < nset; s++) {
        OrbOp op;
        int rc;
        double acc[4];
        if (!h->parts.empty()) {
            if ((rc = xc_multi(h, fac, xctype, 0, o, nocc + s, sg, acc, vmat + (size_t)s * n2))) return rc;
        } else {
            if ((rc = upload_orbitals(h, "0", o, nocc[s], sg, &op))) return rc;
            if ((rc = xc_contract(h, fac, xctype, 0, &op, acc, vmat + (size_t)s * n2))) return rc;
        }
        nelec[s] = acc[0];
        exc[s] = acc[1];
        o += (size_t)h->nao * nocc[s];
        if (sg) sg += nocc[s];
    }
    return 0;
}

int PAMD_xc_nr_uks(PAMD_xc *h, const double *fac, int xctype, const double *orbs, const int *nocc, const double *signs, double *nelec,
                   double *exc, double *vmat)
{
    PAMD_REQUIRE(h && fac && orbs && nocc && nelec && exc && vmat, "PAMD_xc_nr_uks: bad arguments");
    PAMD_REQUIRE(xctype == 0 || xctype == 1, "PAMD_xc_nr_uks: xctype 0 (LDA) or 1 (GGA)");
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    OrbOp ops[2];
    int rc;
    double acc[4];
    if (!h->parts.empty()) {
        if ((rc = xc_multi(h, fac, xctype, 1, orbs, nocc, signs, acc, vmat))) return rc;
    } else {
        if ((rc = upload_orbitals(h, "0", orbs, nocc[0], signs, &ops[0]))) return rc;
        if ((rc = upload_orbitals(h, "1", orbs + (size_t)h->nao * nocc[0], nocc[1], signs ? signs + nocc[0] : nullptr, &ops[1]))) return rc;
        if ((rc = xc_contract(h, fac, xctype, 1, ops, acc, vmat))) return rc;
    }
    nelec[0] = acc[0];
    nelec[1] = acc[1];
    *exc = acc[2];
    return 0;
}

}  // extern "C"
