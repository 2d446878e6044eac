// Host-array, opaque-handle entry points of the DF J/K path.
//
// The reference's C entry points take raw host pointers and ints; the CALLER OWNS ALL BUFFERS (numpy arrays allocated in
// Python and passed by pointer, pyscf/df/df_jk.py:373-379, pyscf/gto/moleintor.py:590-596).  The PAMD_* kernel launchers of
// this library take DEVICE pointers, which is right for a runtime that already lives on the GPU (pyscf_amd.df.DF keeps its
// tensors in torch) but leaves a numpy-only caller without an allocator.  The functions below close that gap:
//
//   PAMD_df_create        libcint-format tables (atm, bas = AO rows then aux rows as gto.conc_env makes them, env) ->
//                         handle owning the 3-index tensor in HBM: what DF.build does (pyscf/df/df.py:147-199 ->
//                         df/incore.py:129-220 cholesky_eri: j2c, Cholesky, per AO-row slab L^-1 (Q|pq))
//   PAMD_df_get_jk        host dm (+ optional occupied orbitals) -> host vj, vk: df_jk.get_jk (pyscf/df/df_jk.py:280-413),
//                         MO branch :339-381 when orbitals are given, general-DM branch :382-408 otherwise
//   PAMD_df_export_cderi  rows [l0, l1) of `_cderi` to a host array (DF.loop, pyscf/df/df.py:214-242)
//   PAMD_df_naux / _nao / _destroy
//
// All device memory is hipMalloc'ed and owned by the handle; calls are synchronous with respect to the host on return; one
// handle per thread at a time (not re-entrant).  Host work here is the one-time table preparation that
// pyscf_amd/gto/moleintor.py does in numpy (segmented shells, primitive-pair records, cart->sph matrices); the numerical work
// is the same kernel family (PAMD_int3c2e_class, PAMD_cderi_solve, PAMD_nr_e2_*, PAMD_dgemm_tn, PAMD_df_vj_pass*).
// The metric factorisation (the reference hands it to LAPACK, df/incore.py:154) is a right-looking blocked Cholesky + block
// forward substitution for L^-1 on this library's own FP64-MFMA GEMMs (PAMD_dgemm_nt / _tn), 256 x 256 diagonal blocks on the
// host; only a linearly dependent metric (`lindep`, df/incore.py:263-270) needs rocSOLVER's syevd, resolved with dlopen on that
// rare path, so that libpyscf_amd.so itself links nothing new and a cold start does not pay for loading rocBLAS / rocSOLVER.
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <map>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "common.h"
#include "host_tables.h"
#include <cstdlib>
#include <chrono>
#include "../../include/pyscf_amd.h"

using namespace pamd;

using namespace pamd::host;

namespace {

struct PairClass {
    int li = 0, lj = 0, n = 0;
    std::vector<int> rowshell;
    int *d_ish = nullptr, *d_jsh = nullptr, *d_pp0 = nullptr, *d_npp = nullptr;
    double *d_pp = nullptr;
    void subrange(int sh0, int sh1, int *i0, int *i1) const
    {
        *i0 = (int)(std::lower_bound(rowshell.begin(), rowshell.end(), sh0) - rowshell.begin());
        *i1 = (int)(std::lower_bound(rowshell.begin(), rowshell.end(), sh1) - rowshell.begin());
    }
};

struct AuxClass {
    int l = 0, n = 0, npk = 0;
    int *d_f0 = nullptr;
    double *d_xyz = nullptr, *d_exp = nullptr, *d_coef = nullptr;
};

// all shell pairs (a, b) with l_a = li >= l_b = lj (a >= b when li == lj), primitive pairs screened by exp(-mu R^2) < e^-60,
// sorted by the row shell so that an AO-row slab is a contiguous range (moleintor._PairClass)
int make_pair_class(DevPool &pool, const Shells &s, int li, int lj, PairClass *pc)
{
    pc->li = li;
    pc->lj = lj;
    std::vector<int> ish, jsh, npp, pp0;
    std::vector<double> pp;
    for (int a = 0; a < s.n; a++) {
        if (s.l[a] != li) continue;
        for (int b = 0; b < s.n; b++) {
            if (s.l[b] != lj || (li == lj && a < b)) continue;
            const double *A = &s.xyz[3 * a], *B = &s.xyz[3 * b];
            const double r2 = (A[0] - B[0]) * (A[0] - B[0]) + (A[1] - B[1]) * (A[1] - B[1]) + (A[2] - B[2]) * (A[2] - B[2]);
            int cnt = 0;
            const int first = (int)(pp.size() / 8);
            for (size_t p = 0; p < s.exps[a].size(); p++)
                for (size_t q = 0; q < s.exps[b].size(); q++) {
                    const double ea = s.exps[a][p], eb = s.exps[b][q], z = ea + eb, arg = ea * eb / z * r2;
                    if (!(arg < EXPCUTOFF)) continue;
                    const double wa = ea / z;
                    double P[3];
                    for (int x = 0; x < 3; x++) P[x] = wa * A[x] + (1 - wa) * B[x];
                    const double rec[8] = {z, P[0], P[1], P[2], std::exp(-arg) * s.coefs[a][p] * s.coefs[b][q],
                                           P[0] - A[0], P[1] - A[1], P[2] - A[2]};
                    pp.insert(pp.end(), rec, rec + 8);
                    cnt++;
                }
            if (cnt) { ish.push_back(a); jsh.push_back(b); npp.push_back(cnt); pp0.push_back(first); }
        }
    }
    const int n = (int)ish.size();
    pc->n = n;
    if (n == 0) return 0;
    std::vector<int> order(n);
    for (int i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(),
                     [&](int x, int y) { return std::max(ish[x], jsh[x]) < std::max(ish[y], jsh[y]); });
    std::vector<int> o_ish(n), o_jsh(n), o_npp(n), o_pp0(n);
    pc->rowshell.resize(n);
    for (int i = 0; i < n; i++) {
        const int k = order[i];
        o_ish[i] = ish[k]; o_jsh[i] = jsh[k]; o_npp[i] = npp[k]; o_pp0[i] = pp0[k];
        pc->rowshell[i] = std::max(ish[k], jsh[k]);
    }
    int rc;
    if ((rc = upload(pool, o_ish, &pc->d_ish)) || (rc = upload(pool, o_jsh, &pc->d_jsh)) ||
        (rc = upload(pool, o_pp0, &pc->d_pp0)) || (rc = upload(pool, o_npp, &pc->d_npp)) || (rc = upload(pool, pp, &pc->d_pp)))
        return rc;
    return 0;
}

// (P, unit s function on the same centre): the 2-centre family (moleintor._PairClass2c)
int make_pair_class_2c(DevPool &pool, const Shells &s, int li, double s_factor, PairClass *pc)
{
    pc->li = li;
    pc->lj = 0;
    std::vector<int> ia, pp0, npp;
    std::vector<double> pp;
    for (int a = 0; a < s.n; a++) {
        if (s.l[a] != li) continue;
        ia.push_back(a);
        pp0.push_back((int)(pp.size() / 8));
        npp.push_back((int)s.exps[a].size());
        for (size_t p = 0; p < s.exps[a].size(); p++) {
            const double rec[8] = {s.exps[a][p], s.xyz[3 * a], s.xyz[3 * a + 1], s.xyz[3 * a + 2], s.coefs[a][p] * s_factor, 0, 0, 0};
            pp.insert(pp.end(), rec, rec + 8);
        }
    }
    pc->n = (int)ia.size();
    if (pc->n == 0) return 0;
    pc->rowshell = ia;
    int rc;
    if ((rc = upload(pool, ia, &pc->d_ish)) || (rc = upload(pool, ia, &pc->d_jsh)) || (rc = upload(pool, pp0, &pc->d_pp0)) ||
        (rc = upload(pool, npp, &pc->d_npp)) || (rc = upload(pool, pp, &pc->d_pp)))
        return rc;
    return 0;
}

int make_aux_class(DevPool &pool, const Shells &s, int l, AuxClass *ac)
{
    std::vector<int> idx;
    for (int i = 0; i < s.n; i++) if (s.l[i] == l) idx.push_back(i);
    ac->l = l;
    ac->n = (int)idx.size();
    if (ac->n == 0) return 0;
    size_t npk = 0;
    for (int i : idx) npk = std::max(npk, s.exps[i].size());
    ac->npk = (int)npk;
    std::vector<int> f0(ac->n);
    std::vector<double> xyz(3 * ac->n), ex(ac->n * npk, 1.0), co(ac->n * npk, 0.0);
    for (int j = 0; j < ac->n; j++) {
        const int i = idx[j];
        f0[j] = s.ao0[i];
        for (int x = 0; x < 3; x++) xyz[3 * j + x] = s.xyz[3 * i + x];
        for (size_t k = 0; k < s.exps[i].size(); k++) { ex[j * npk + k] = s.exps[i][k]; co[j * npk + k] = s.coefs[i][k]; }
    }
    int rc;
    if ((rc = upload(pool, f0, &ac->d_f0)) || (rc = upload(pool, xyz, &ac->d_xyz)) || (rc = upload(pool, ex, &ac->d_exp)) ||
        (rc = upload(pool, co, &ac->d_coef)))
        return rc;
    return 0;
}

// ------------------------------------------------------------------------------------------------ rocSOLVER / rocBLAS (dlopen)
struct RocLib {
    void *hblas = nullptr, *hsolver = nullptr, *handle = nullptr;
    int (*create)(void **) = nullptr;
    int (*destroy)(void *) = nullptr;
    int (*set_stream)(void *, hipStream_t) = nullptr;
    int (*dgemm)(void *, int, int, int, int, int, const double *, const double *, int, const double *, int, const double *,
                 double *, int) = nullptr;
    int (*dpotrf)(void *, int, int, double *, int, int *) = nullptr;
    int (*dsyevd)(void *, int, int, int, double *, int, double *, double *, int *) = nullptr;
    int open()
    {
        if (handle) return 0;
        hblas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        if (!hblas) hblas = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        hsolver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (!hsolver) hsolver = dlopen("/opt/rocm/lib/librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        PAMD_REQUIRE(hblas && hsolver, "librocblas.so / librocsolver.so not found (metric factorisation of PAMD_df_create)");
        create = (decltype(create))dlsym(hblas, "rocblas_create_handle");
        destroy = (decltype(destroy))dlsym(hblas, "rocblas_destroy_handle");
        set_stream = (decltype(set_stream))dlsym(hblas, "rocblas_set_stream");
        dgemm = (decltype(dgemm))dlsym(hblas, "rocblas_dgemm");
        dpotrf = (decltype(dpotrf))dlsym(hsolver, "rocsolver_dpotrf");
        dsyevd = (decltype(dsyevd))dlsym(hsolver, "rocsolver_dsyevd");
        PAMD_REQUIRE(create && destroy && set_stream && dsyevd, "rocBLAS / rocSOLVER symbols missing");
        PAMD_REQUIRE(create(&handle) == 0, "rocblas_create_handle failed");
        return 0;
    }
    // (the process-wide instance is never destroyed: at exit the HIP runtime may already be gone)
};
constexpr int ROC_OP_N = 111, ROC_FILL_UPPER = 121, ROC_EVECT_ORIGINAL = 211;

}  // namespace

// ------------------------------------------------------------------------------------------------ the handle
// One PAMD_df is either a SHARD (rows [l0, l0 + nL) of the tensor on one device: what PAMD_df_create makes, with l0 = 0 and all
// rows) or, when `parts` is not empty, a multi-device handle whose parts are shards on the listed devices.
// A shard keeps its rows in HBM as far as they fit (`n_res` rows, d_cderi) and the rest in page-locked host memory (h_cderi),
// streamed through two staging buffers during every J/K build (the out-of-core twin of the reference, pyscf/df/outcore.py:109-232,
// pyscf/df/df.py:167,214-242: there blocks of the HDF5 file, here blocks of pinned RAM under double-buffered H2D copies).
// One persistent host thread per part of a multi-device handle (r05; r04 spawned a std::thread per part per PAMD_df_get_jk): the
// thread binds its device once, keeps its HIP thread state, and runs the jobs the caller's thread posts to it.
struct PartWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    std::string msg;
    void start(int device);
    void post(std::function<int()> f)
    {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(f);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return rc;
    }
    ~PartWorker()
    {
        if (th.joinable()) {
            { std::lock_guard<std::mutex> lk(m); quit = true; cv.notify_all(); }
            th.join();
        }
    }
};

struct PAMD_df {
    int device = 0;
    int nao = 0, naux = 0;                  // AO functions, auxiliary functions
    int nL = 0;                             // tensor rows this handle answers for (a shard: its rows; multi: all rows)
    int l0 = 0, nL_total = 0;               // first global row of a shard, rows of the whole tensor
    long npair = 0;
    int rows = 0;                           // round_up(nao, 16)
    double omega = 0.0;                     // 0 Coulomb, > 0 erf(omega r12)/r12, < 0 erfc(|omega| r12)/r12
    DevPool pool;
    hipStream_t st = nullptr, side = nullptr, copy = nullptr;
    hipEvent_t ev = nullptr, ev_j = nullptr, ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    double *d_cderi = nullptr, *d_sq = nullptr, *d_diag = nullptr;
    int square = 0;                         // r06: 1 = d_sq[nL][rows][rows] is the ONLY copy of the rows (d_cderi is null): 2x, not 3x
    // doubles between consecutive aux rows of d_sq (an explicit argument of every kernel that walks them).  A pad of 288 doubles was
    // measured for the square LAYOUT (rows * rows * 8 is a multiple of 32 KB .. 8 MB at the named configurations; suspicion: the equal
    // (p, q) of consecutive rows share one HBM channel) - no difference (profiles/r06/kbench_square_layout.log), so none is applied
    static constexpr long SQ_STRIDE_PAD = 0;
    long sq_ls() const { return (long)rows * rows + (square ? SQ_STRIDE_PAD : 0); }
    int n_res = 0;                          // rows [0, n_res) resident in HBM, rows [n_res, nL) in h_cderi
    double *h_cderi = nullptr;              // host rows, (nL - n_res) x npair: page-locked memory of the handle, or (h_borrowed) the caller's
    int h_borrowed = 0, h_registered = 0;   // PAMD_df_create_from_rows: rows stay in the caller's array (e.g. an mmap of a `_cderi` file)
    double *d_stage[2] = {nullptr, nullptr};
    int stage_rows = 0;
    std::map<long, int> j2_policy;          // (nset, occupied counts) -> 0 overlap / 1 serial second J pass (PAMD_df_get_jk)
    struct J2Trial { int calls = 0, n[3] = {0, 0, 0}; double ms[3] = {1e30, 1e30, 1e30}; };
    std::map<long, J2Trial> j2_trial;       // r06: schedules still being timed on the caller's own calls (PAMD_DF_J2_TUNE != eager)
    std::vector<PAMD_df *> parts;           // multi-device handle: the shards (owned)
    std::vector<PartWorker *> workers;      // multi: one persistent host thread per part
    int peer_ok = 0;                        // multi: partial results reach part 0 by direct peer copies
    int partial = 0;                        // 1: one rank's shard of a multi-process job (PAMD_df_options.part / nparts): PARTIAL J/K
    double *h_orb = nullptr;                // page-locked staging of the padded orbitals (persistent: no per-call allocation / sync)
    size_t h_orb_len = 0;
    // timings of the last PAMD_df_get_jk (PAMD_df_last_timing): per part contraction and push into the gather buffer, sum + download
    double t_compute_ms = 0, t_push_ms = 0, t_sum_ms = 0;
    size_t push_bytes = 0;
    // HIP events around the half-transform and SYRK launches of the last build, on the stream they are launched on
    std::vector<hipEvent_t> tev;
    double t_e2_ms = 0, t_syrk_ms = 0;
    double last_mismatch = 0;               // r06: result of the in-call tag probe (PAMD_df_get_jk flags bit 1)
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
        *rc = pool.alloc((void **)&p, (ndoubles + 256) * 8);       // +256: the LDS-DMA kernels read whole 128-column panel rows
        if (*rc) return nullptr;
        (void)hipMemsetAsync(p, 0, (ndoubles + 256) * 8, st);
        ws[name] = {p, ndoubles};
        return p;
    }
    ~PAMD_df()
    {
        for (PartWorker *w : workers) delete w;
        for (PAMD_df *p : parts) {
            (void)hipSetDevice(p->device);
            delete p;
        }
        if (!parts.empty()) (void)hipSetDevice(device);
        if (h_cderi && h_registered) (void)hipHostUnregister(h_cderi);
        if (h_cderi && !h_borrowed) (void)hipHostFree(h_cderi);
        if (h_orb) (void)hipHostFree(h_orb);
        for (int k = 0; k < 2; k++) {
            if (ev_ready[k]) (void)hipEventDestroy(ev_ready[k]);
            if (ev_free[k]) (void)hipEventDestroy(ev_free[k]);
        }
        if (ev) (void)hipEventDestroy(ev);
        if (ev_j) (void)hipEventDestroy(ev_j);
        for (hipEvent_t e : tev) (void)hipEventDestroy(e);
        if (copy) (void)hipStreamDestroy(copy);
        if (side) (void)hipStreamDestroy(side);
        if (st) (void)hipStreamDestroy(st);
    }
};

void PartWorker::start(int device)
{
    th = std::thread([this, device]() {
        (void)hipSetDevice(device);
        for (;;) {
            std::function<int()> f;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                f = std::move(job);
                has_job = false;
            }
            const int r = f();
            std::lock_guard<std::mutex> lk(m);
            rc = r;
            msg = r ? g_errmsg : "";            // g_errmsg is thread local: carry it over to the caller's thread
            done = true;
            cv.notify_all();
        }
    });
}

namespace {

struct Tables {                            // host-side tables of one (AO basis, aux basis) pair, shared by all parts
    Shells ao, aux;
    int lmax_ao = 0, lmax_aux = 0;
    std::vector<double> c2s;
    std::vector<int> c2s_off;
};

struct Metric {                            // M^T with cderi = M (Q|pq), host copy shared by all parts (decompose_metric)
    std::vector<double> mt;
    int nrow = 0, lda = 0, tri = 0;
};

struct Engine {                            // device tables of one part, alive during the build
    const Tables *t = nullptr;
    std::vector<PairClass> pcs, pcs2c;
    std::vector<AuxClass> acs;
    double *d_rys = nullptr, *d_c2s = nullptr, *d_ao_xyz = nullptr, *d_aux_xyz = nullptr;
    int *d_c2s_off = nullptr, *d_ao_ao0 = nullptr, *d_aux_ao0 = nullptr;
};

std::mutex g_dev_mutex[64];                // builds on the same device are serialised (a devices list may repeat a device)

// Bytes this process may still take before it hits its container's memory limit (cgroup v2 `memory.max` - `memory.current`, else
// v1), or SIZE_MAX when there is no limit.  Page-locked host memory is charged to the cgroup like any other: r05 lost two GPU boxes
// to the kernel's OOM killer when the 301 GB of host rows of config 5 met a 300 GiB container limit (/proc/meminfo showed 3 TB).
size_t container_memory_left()
{
    auto read_num = [](const char *path, bool *is_max) -> long long {
        FILE *f = fopen(path, "r");
        if (!f) return -1;
        char buf[64] = {0};
        const bool ok = fgets(buf, sizeof(buf), f) != nullptr;
        fclose(f);
        if (!ok) return -1;
        if (is_max && !strncmp(buf, "max", 3)) { *is_max = true; return 0; }
        return atoll(buf);
    };
    bool unlimited = false;
    long long lim = read_num("/sys/fs/cgroup/memory.max", &unlimited), cur = read_num("/sys/fs/cgroup/memory.current", nullptr);
    if (lim < 0 && !unlimited) {
        lim = read_num("/sys/fs/cgroup/memory/memory.limit_in_bytes", nullptr);
        cur = read_num("/sys/fs/cgroup/memory/memory.usage_in_bytes", nullptr);
    }
    if (unlimited || lim <= 0 || lim > (1LL << 60) || cur < 0) return ~(size_t)0;
    // file cache is charged too but the kernel reclaims it before it kills anything: count it as available
    long long cache = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/memory.stat", "r")) {
        char key[64];
        long long val;
        while (fscanf(f, "%63s %lld", key, &val) == 2)
            if (!strcmp(key, "inactive_file") || !strcmp(key, "active_file")) cache += val;
        fclose(f);
    }
    if (cache > 0 && cache <= cur) cur -= cache;
    return lim > cur ? (size_t)(lim - cur) : 0;
}

int launch_class(const Engine &e, const PairClass &pc, int i0, int i1, const AuxClass &ac, double *T, long ldT, long row_offset,
                 int tril, const double *shell_xyz, const int *shell_ao0, double omega, hipStream_t st)
{
    if (i1 <= i0) return 0;
    PAMD_int3c2e_args a;
    memset(&a, 0, sizeof(a));
    a.pair_ish = pc.d_ish + i0;
    a.pair_jsh = pc.d_jsh + i0;
    a.pair_pp0 = pc.d_pp0 + i0;
    a.pair_npp = pc.d_npp + i0;
    a.pp = pc.d_pp;
    a.shell_xyz = shell_xyz;
    a.shell_ao0 = shell_ao0;
    a.aux_f0 = ac.d_f0;
    a.aux_xyz = ac.d_xyz;
    a.aux_exp = ac.d_exp;
    a.aux_coef = ac.d_coef;
    a.naux_cls = ac.n;
    a.npk = ac.npk;
    a.rys_table = e.d_rys;
    a.c2s = e.d_c2s;
    a.c2s_off = e.d_c2s_off;
    a.T = T;
    a.ldT = ldT;
    a.row_offset = row_offset;
    a.tril = tril;
    a.npairs = i1 - i0;
    a.omega = omega;
    return PAMD_int3c2e_class(pc.li, pc.lj, ac.l, &a, st);
}

void slab_rows(const Shells &ao, int sh0, int sh1, long *r0, long *r1)
{
    const long p0 = ao.ao0[sh0], p1 = sh1 < ao.n ? ao.ao0[sh1] : ao.nao;
    *r0 = p0 * (p0 + 1) / 2;
    *r1 = p1 * (p1 + 1) / 2;
}

__global__ void sub_inplace_kernel(double *__restrict__ a, const double *__restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] -= b[i];
}

// Cyclic Jacobi eigen-solver for a small symmetric matrix (row-major a[n][n], destroyed): w[k] eigenvalues, v[k][n] the
// eigenvector of w[k] (rows).  Used for linearly dependent metrics up to HOST_EIG_MAX functions; larger ones go to rocSOLVER.
constexpr int HOST_EIG_MAX = 768;
void jacobi_eig(int n, std::vector<double> &a, std::vector<double> &w, std::vector<double> &v)
{
    v.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; i++) v[(size_t)i * n + i] = 1.0;          // v holds V^T: row k = eigenvector k
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0, diag = 0;
        for (int i = 0; i < n; i++) {
            diag += a[(size_t)i * n + i] * a[(size_t)i * n + i];
            for (int j = 0; j < i; j++) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
        }
        if (off <= 1e-30 * (diag + off) || off == 0.0) break;
        for (int p = 0; p < n - 1; p++)
            for (int q = p + 1; q < n; q++) {
                const double apq = a[(size_t)p * n + q];
                if (apq == 0.0) continue;
                const double app = a[(size_t)p * n + p], aqq = a[(size_t)q * n + q];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; k++) {                        // columns p, q of A
                    const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
                    a[(size_t)k * n + p] = c * akp - sn * akq;
                    a[(size_t)k * n + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; k++) {                        // rows p, q of A and of V^T
                    const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
                    a[(size_t)p * n + k] = c * apk - sn * aqk;
                    a[(size_t)q * n + k] = sn * apk + c * aqk;
                    const double vpk = v[(size_t)p * n + k], vqk = v[(size_t)q * n + k];
                    v[(size_t)p * n + k] = c * vpk - sn * vqk;
                    v[(size_t)q * n + k] = sn * vpk + c * vqk;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; i++) w[i] = a[(size_t)i * n + i];
}

__global__ void negate_copy_kernel(const double *__restrict__ src, long lds, double *__restrict__ dst, long ldd, long rows, int cols)
{
    const long r = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x)
        if (r < rows) dst[r * ldd + c] = -src[r * lds + c];
}

// Lower Cholesky factor of the symmetric row-major matrix d_a (n x n), in place, right-looking blocked: the diagonal block is
// factorised and inverted on the host (nbk^3, negligible), the panel solve and the trailing update are FP64-MFMA GEMMs of this
// library (PAMD_dgemm_nt).  *info = 0 on success, else the 1-based column of the first non-positive pivot (the matrix is then
// left half-factorised: the caller falls back to the eigen-decomposition from a fresh copy).  The strict upper triangle is junk.
int chol_blocked(PAMD_df *h, double *d_a, int n, int *info)
{
    const int nbk = 256;
    int rc;
    *info = 0;
    double *d_p = nullptr, *d_n = nullptr, *d_dinv = nullptr;
    if ((rc = h->pool.alloc((void **)&d_p, (size_t)n * nbk * 8)) || (rc = h->pool.alloc((void **)&d_n, (size_t)n * nbk * 8)) ||
        (rc = h->pool.alloc((void **)&d_dinv, (size_t)nbk * nbk * 8)))
        return rc;
    std::vector<double> d((size_t)nbk * nbk), dinv((size_t)nbk * nbk);
    for (int j0 = 0; j0 < n && *info == 0; j0 += nbk) {
        const int bj = std::min(nbk, n - j0), j1 = j0 + bj;
        PAMD_CHECK_HIP(hipMemcpy2DAsync(d.data(), (size_t)bj * 8, d_a + (size_t)j0 * n + j0, (size_t)n * 8, (size_t)bj * 8, bj,
                                        hipMemcpyDeviceToHost, h->st));
        PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
        for (int c = 0; c < bj && *info == 0; c++) {                 // unblocked Cholesky of the diagonal block (lower, row-major)
            double s = d[(size_t)c * bj + c];
            for (int k = 0; k < c; k++) s -= d[(size_t)c * bj + k] * d[(size_t)c * bj + k];
            if (!(s > 0.0)) { *info = j0 + c + 1; break; }
            const double lcc = std::sqrt(s);
            d[(size_t)c * bj + c] = lcc;
            for (int r = c + 1; r < bj; r++) {
                double t = d[(size_t)r * bj + c];
                for (int k = 0; k < c; k++) t -= d[(size_t)r * bj + k] * d[(size_t)c * bj + k];
                d[(size_t)r * bj + c] = t / lcc;
            }
            for (int k = c + 1; k < bj; k++) d[(size_t)c * bj + k] = 0.0;
        }
        if (*info) break;
        std::fill(dinv.begin(), dinv.end(), 0.0);
        for (int c = 0; c < bj; c++) {                               // inverse of the lower block by forward substitution
            dinv[(size_t)c * bj + c] = 1.0 / d[(size_t)c * bj + c];
            for (int r = c + 1; r < bj; r++) {
                double t = 0;
                for (int k = c; k < r; k++) t += d[(size_t)r * bj + k] * dinv[(size_t)k * bj + c];
                dinv[(size_t)r * bj + c] = -t / d[(size_t)r * bj + r];
            }
        }
        PAMD_CHECK_HIP(hipMemcpy2DAsync(d_a + (size_t)j0 * n + j0, (size_t)n * 8, d.data(), (size_t)bj * 8, (size_t)bj * 8, bj,
                                        hipMemcpyHostToDevice, h->st));
        if (j1 < n) {
            const long mrem = n - j1;
            PAMD_CHECK_HIP(hipMemcpyAsync(d_dinv, dinv.data(), (size_t)bj * bj * 8, hipMemcpyHostToDevice, h->st));
            PAMD_CHECK_HIP(hipMemsetAsync(d_p, 0, (size_t)mrem * bj * 8, h->st));
            // P = A21 L11^-T:  P[m][c] = sum_k A21[m][k] Dinv[c][k]
            if ((rc = PAMD_dgemm_nt(d_a + (size_t)j1 * n + j0, n, d_dinv, bj, d_p, bj, (int)mrem, bj, bj, 1, h->st))) return rc;
            PAMD_CHECK_HIP(hipMemcpy2DAsync(d_a + (size_t)j1 * n + j0, (size_t)n * 8, d_p, (size_t)bj * 8, (size_t)bj * 8, mrem,
                                            hipMemcpyDeviceToDevice, h->st));
            negate_copy_kernel<<<(unsigned)mrem, 256, 0, h->st>>>(d_p, bj, d_n, bj, mrem, bj);
            PAMD_CHECK_LAUNCH();
            // trailing update A22 -= P P^T (the full square: the upper half is junk anyway)
            if ((rc = PAMD_dgemm_nt(d_n, bj, d_p, bj, d_a + (size_t)j1 * n + j1, n, (int)mrem, (int)mrem, bj, 1, h->st))) return rc;
        }
        PAMD_CHECK_HIP(hipStreamSynchronize(h->st));                 // d / dinv (host) are reused by the next block column
    }
    h->pool.release(d_p); h->pool.release(d_n); h->pool.release(d_dinv);
    return 0;
}

// M with cderi = M (Q|pq): rows of L^-1 (Cholesky) or (V / sqrt(w))^T over the eigenvalues > lindep (df/incore.py:153-158,
// 263-270).  Returns M^T as mt[naux][lda] on the host (lda = round_up(nrow, 16), zero padded) and the triangular flag.
int decompose_metric(PAMD_df *h, double *d_j2c, int naux, double lindep, std::vector<double> *mt, int *nrow, int *lda, int *tri,
                     bool force_ed = false)
{
    int rc;
    const size_t n2 = (size_t)naux * naux;
    double *d_a = nullptr;
    if ((rc = h->pool.alloc((void **)&d_a, n2 * 8))) return rc;
    PAMD_CHECK_HIP(hipMemcpyAsync(d_a, d_j2c, n2 * 8, hipMemcpyDeviceToDevice, h->st));
    int info = 1;
    if (!force_ed && (rc = chol_blocked(h, d_a, naux, &info))) return rc;
    if (info == 0) {
        // W = (L^-1)^T (upper triangular, row-major) by block forward substitution (the scheme of pyscf_amd/df/incore.py:
        // _decompose_j2c of r02): W[i, i] = Dinv_i^T on the host, W[:i0, i] = -(L[i, :i0] Linv[:i0, :i0])^T Dinv_i^T as two GEMMs of
        // this library per block row:  T = L[i,:i0] W[:i0,:i0]^T (NT),  W[:i0, i] += T^T (-Dinv_i^T) (TN)
        const int blk = 256;
        double *d_w = nullptr, *d_t = nullptr, *d_dt = nullptr;
        if ((rc = h->pool.alloc((void **)&d_w, n2 * 8)) || (rc = h->pool.alloc((void **)&d_t, (size_t)blk * naux * 8)) ||
            (rc = h->pool.alloc((void **)&d_dt, (size_t)blk * blk * 8)))
            return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(d_w, 0, n2 * 8, h->st));
        std::vector<double> dblk((size_t)blk * blk), dinv((size_t)blk * blk), dneg((size_t)blk * blk);
        for (int i0 = 0; i0 < naux; i0 += blk) {
            const int bi = std::min(blk, naux - i0);
            PAMD_CHECK_HIP(hipMemcpy2DAsync(dblk.data(), (size_t)bi * 8, d_a + (size_t)i0 * naux + i0, (size_t)naux * 8,
                                            (size_t)bi * 8, bi, hipMemcpyDeviceToHost, h->st));
            PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
            std::fill(dinv.begin(), dinv.end(), 0.0);
            for (int c = 0; c < bi; c++) {
                dinv[(size_t)c * bi + c] = 1.0 / dblk[(size_t)c * bi + c];
                for (int r = c + 1; r < bi; r++) {
                    double t = 0;
                    for (int k = c; k < r; k++) t += dblk[(size_t)r * bi + k] * dinv[(size_t)k * bi + c];
                    dinv[(size_t)r * bi + c] = -t / dblk[(size_t)r * bi + r];
                }
            }
            // diagonal block of W: Dinv^T; operand of the second GEMM: B[k][n] = -Dinv[n][k]
            for (int r = 0; r < bi; r++)
                for (int c = 0; c < bi; c++) { dblk[(size_t)r * bi + c] = dinv[(size_t)c * bi + r]; dneg[(size_t)r * bi + c] = -dinv[(size_t)c * bi + r]; }
            PAMD_CHECK_HIP(hipMemcpy2DAsync(d_w + (size_t)i0 * naux + i0, (size_t)naux * 8, dblk.data(), (size_t)bi * 8, (size_t)bi * 8,
                                            bi, hipMemcpyHostToDevice, h->st));
            if (i0) {
                PAMD_CHECK_HIP(hipMemcpyAsync(d_dt, dneg.data(), (size_t)bi * bi * 8, hipMemcpyHostToDevice, h->st));
                PAMD_CHECK_HIP(hipMemsetAsync(d_t, 0, (size_t)bi * i0 * 8, h->st));
                // T[m][c] = sum_k L[i0 + m][k] Linv[k][c] = sum_k L[i0 + m][k] W[c][k]      (m < bi, c < i0, k < i0)
                if ((rc = PAMD_dgemm_nt(d_a + (size_t)i0 * naux, naux, d_w, naux, d_t, i0, bi, i0, i0, 1, h->st))) return rc;
                // W[c][i0 + n] += sum_k T[k][c] (-Dinv[n][k])                                   (c < i0, n < bi, k < bi)
                if ((rc = PAMD_dgemm_tn(d_t, i0, d_dt, bi, d_w + i0, naux, i0, bi, bi, 0, 1, h->st))) return rc;
            }
            PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
        }
        *nrow = naux;
        *lda = (int)std::max<long>(round_up(naux, 16), 16);
        mt->assign((size_t)naux * *lda, 0.0);
        PAMD_CHECK_HIP(hipMemcpy2D(mt->data(), (size_t)*lda * 8, d_w, (size_t)naux * 8, (size_t)naux * 8, naux, hipMemcpyDeviceToHost));
        *tri = 1;
        h->pool.release(d_w); h->pool.release(d_t); h->pool.release(d_dt);
    } else {
        // metric not positive definite: eigen-decomposition, keep w > lindep (pyscf/df/incore.py:263-270).  Small metrics: a
        // Jacobi solver on the host (no library start-up: the first use of the system rocSOLVER costs minutes on a cold box);
        // larger ones: rocSOLVER's syevd, resolved with dlopen only here.
        std::vector<double> w, v;                                 // row m of v = eigenvector m
        if (naux <= HOST_EIG_MAX) {
            std::vector<double> aj(n2);
            PAMD_CHECK_HIP(hipMemcpy(aj.data(), d_j2c, n2 * 8, hipMemcpyDeviceToHost));
            jacobi_eig(naux, aj, w, v);
        } else {
            static RocLib roc;             // one rocBLAS handle per process (creating one loads the library's kernels: seconds)
            if ((rc = roc.open())) return rc;
            roc.set_stream(roc.handle, h->st);
            double *d_w = nullptr, *d_e = nullptr;
            int *d_info = nullptr;
            if ((rc = h->pool.alloc((void **)&d_w, (size_t)naux * 8)) || (rc = h->pool.alloc((void **)&d_e, (size_t)naux * 8)) ||
                (rc = h->pool.alloc((void **)&d_info, 64)))
                return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(d_a, d_j2c, n2 * 8, hipMemcpyDeviceToDevice, h->st));
            PAMD_REQUIRE(roc.dsyevd(roc.handle, ROC_EVECT_ORIGINAL, ROC_FILL_UPPER, naux, d_a, naux, d_w, d_e, d_info) == 0,
                         "rocsolver_dsyevd failed");
            PAMD_CHECK_HIP(hipMemcpyAsync(&info, d_info, 4, hipMemcpyDeviceToHost, h->st));
            PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
            PAMD_REQUIRE(info == 0, "rocsolver_dsyevd did not converge");
            w.resize(naux);
            v.resize(n2);                                         // column-major eigenvectors: row m of the buffer = vector m
            PAMD_CHECK_HIP(hipMemcpy(w.data(), d_w, (size_t)naux * 8, hipMemcpyDeviceToHost));
            PAMD_CHECK_HIP(hipMemcpy(v.data(), d_a, n2 * 8, hipMemcpyDeviceToHost));
            h->pool.release(d_w); h->pool.release(d_e); h->pool.release(d_info);
        }
        std::vector<int> keep;
        for (int m = 0; m < naux; m++) if (w[m] > lindep) keep.push_back(m);
        *nrow = (int)keep.size();
        *lda = (int)std::max<long>(round_up(*nrow, 16), 16);
        mt->assign((size_t)naux * *lda, 0.0);
        for (int j = 0; j < *nrow; j++) {
            const int m = keep[j];
            const double f = 1.0 / std::sqrt(w[m]);
            for (int q = 0; q < naux; q++) (*mt)[(size_t)q * *lda + j] = v[(size_t)m * naux + q] * f;
        }
        *tri = 0;
    }
    h->pool.release(d_a);
    return 0;
}


// ------------------------------------------------------------------------------------------------ build
int make_tables(const int *atm, const int *bas, int nbas_ao, int nbas_aux, const double *env, Tables *t)
{
    t->ao = make_shells(atm, bas, 0, nbas_ao, env);
    t->aux = make_shells(atm, bas, nbas_ao, nbas_ao + nbas_aux, env);
    for (int l : t->ao.l) t->lmax_ao = std::max(t->lmax_ao, l);
    for (int l : t->aux.l) t->lmax_aux = std::max(t->lmax_aux, l);
    PAMD_REQUIRE(t->lmax_ao <= 4 && t->lmax_aux <= LMAX_TAB && !(t->lmax_aux > 5 && t->lmax_ao > 3),
                 "angular momentum beyond the instantiated kernels");
    for (int l = 0; l <= LMAX_TAB; l++) {
        t->c2s_off.push_back((int)t->c2s.size());
        std::vector<double> m = c2s_matrix(l);
        t->c2s.insert(t->c2s.end(), m.begin(), m.end());
    }
    return 0;
}

int init_streams(PAMD_df *h);

int init_shard(PAMD_df *h, const Tables &t, int device, double omega)
{
    h->device = device;
    h->omega = omega;
    h->nao = t.ao.nao;
    h->naux = t.aux.nao;
    h->npair = (long)h->nao * (h->nao + 1) / 2;
    h->rows = (int)round_up(h->nao, 16);
    return init_streams(h);
}

int init_streams(PAMD_df *h)
{
    {
        // r06: the MFMA kernels' stream at the device's HIGHEST queue priority, the side stream (second J pass) at the LOWEST: the
        // dispatcher then places the SYRK's one-round grid before the pass's 6700 long-running workgroups instead of racing them
        // for wave slots (bimodal 45 / 50 ms SYRK in the square layout, profiles/r06/native_percall_stream_priority.log).
        // PAMD_DF_STREAM_PRIO=0: plain hipStreamCreate as in r03-r05.
        const char *e = getenv("PAMD_DF_STREAM_PRIO");
        int lo = 0, hi = 0;
        if (!(e && e[0] == '0') && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && lo != hi) {
            PAMD_CHECK_HIP(hipStreamCreateWithPriority(&h->st, hipStreamDefault, hi));
            PAMD_CHECK_HIP(hipStreamCreateWithPriority(&h->side, hipStreamDefault, lo));
        } else {
            (void)hipGetLastError();
            PAMD_CHECK_HIP(hipStreamCreate(&h->st));
            PAMD_CHECK_HIP(hipStreamCreate(&h->side));
        }
    }
    PAMD_CHECK_HIP(hipStreamCreate(&h->copy));
    PAMD_CHECK_HIP(hipEventCreateWithFlags(&h->ev, hipEventDisableTiming));
    PAMD_CHECK_HIP(hipEventCreateWithFlags(&h->ev_j, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
        PAMD_CHECK_HIP(hipEventCreateWithFlags(&h->ev_ready[k], hipEventDisableTiming));
        PAMD_CHECK_HIP(hipEventCreateWithFlags(&h->ev_free[k], hipEventDisableTiming));
    }
    return 0;
}

// Rys table, cart->sph matrices, shell coordinates, shell-pair and aux-class tables on the handle's device
int prepare_engine(PAMD_df *h, const Tables &t, Engine &e, DevPool &tmp)
{
    int rc;
    e.t = &t;
    if ((rc = tmp.alloc((void **)&e.d_rys, (size_t)PAMD_rys_table_len() * 8))) return rc;
    if ((rc = PAMD_rys_table_upload(e.d_rys, h->st))) return rc;
    if ((rc = upload(tmp, t.c2s, &e.d_c2s)) || (rc = upload(tmp, t.c2s_off, &e.d_c2s_off)) || (rc = upload(tmp, t.ao.xyz, &e.d_ao_xyz)) ||
        (rc = upload(tmp, t.ao.ao0, &e.d_ao_ao0)) || (rc = upload(tmp, t.aux.xyz, &e.d_aux_xyz)) ||
        (rc = upload(tmp, t.aux.ao0, &e.d_aux_ao0)))
        return rc;
    for (int li = 0; li <= t.lmax_ao; li++)
        for (int lj = 0; lj <= li; lj++) {
            PairClass pc;
            if ((rc = make_pair_class(tmp, t.ao, li, lj, &pc))) return rc;
            if (pc.n) e.pcs.push_back(pc);
        }
    for (int l = 0; l <= t.lmax_aux; l++) {
        AuxClass ac;
        if ((rc = make_aux_class(tmp, t.aux, l, &ac))) return rc;
        if (ac.n) e.acs.push_back(ac);
    }
    return 0;
}

// d_dst = integrals of the operator selected by h->omega; fill(omega') runs all class launches of one pass into its argument.
// omega < 0 (short range, erfc): Coulomb pass minus the long-range pass at |omega| (pyscf_amd/gto/moleintor.py:_short_range)
template <class Fill>
int fill_operator(PAMD_df *h, double *d_dst, double *d_scratch, size_t n, Fill fill)
{
    int rc;
    PAMD_CHECK_HIP(hipMemsetAsync(d_dst, 0, n * 8, h->st));
    if ((rc = fill(d_dst, h->omega < 0 ? 0.0 : h->omega))) return rc;
    if (h->omega < 0) {
        PAMD_REQUIRE(d_scratch, "short-range pass needs a scratch buffer");
        PAMD_CHECK_HIP(hipMemsetAsync(d_scratch, 0, n * 8, h->st));
        if ((rc = fill(d_scratch, -h->omega))) return rc;
        sub_inplace_kernel<<<2048, 256, 0, h->st>>>(d_dst, d_scratch, n);
        PAMD_CHECK_LAUNCH();
    }
    return 0;
}

// metric (P|Q) -> Metric (host): df/incore.py:150-158 (+ :263-270)
int compute_metric(PAMD_df *h, const Engine &e, DevPool &tmp, double lindep, Metric *m)
{
    int rc;
    const Tables &t = *e.t;
    const int naux = h->naux;
    std::vector<PairClass> pcs2c;
    for (int l = 0; l <= t.lmax_aux; l++) {
        PairClass pc;
        if ((rc = make_pair_class_2c(tmp, t.aux, l, 1.0 / t.c2s[t.c2s_off[0]], &pc))) return rc;
        if (pc.n) pcs2c.push_back(pc);
    }
    const size_t n2 = (size_t)naux * naux;
    double *d_j2c = nullptr, *d_scr = nullptr;
    if ((rc = tmp.alloc((void **)&d_j2c, n2 * 8))) return rc;
    if (h->omega < 0 && (rc = tmp.alloc((void **)&d_scr, n2 * 8))) return rc;
    rc = fill_operator(h, d_j2c, d_scr, n2, [&](double *dst, double om) {
        for (const PairClass &pc : pcs2c)
            for (const AuxClass &ac : e.acs) {
                const int r = launch_class(e, pc, 0, pc.n, ac, dst, naux, 0, 0, e.d_aux_xyz, e.d_aux_ao0, om, h->st);
                if (r) return r;
            }
        return 0;
    });
    if (rc) return rc;
    {
        // symmetrise on the host: (j2c + j2c^T) / 2 as pyscf_amd/df/incore.py does before the factorisation
        std::vector<double> j(n2);
        PAMD_CHECK_HIP(hipMemcpyAsync(j.data(), d_j2c, n2 * 8, hipMemcpyDeviceToHost, h->st));
        PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
        for (int a = 0; a < naux; a++)
            for (int b = 0; b < a; b++) {
                const double v = 0.5 * (j[(size_t)a * naux + b] + j[(size_t)b * naux + a]);
                j[(size_t)a * naux + b] = j[(size_t)b * naux + a] = v;
            }
        PAMD_CHECK_HIP(hipMemcpy(d_j2c, j.data(), n2 * 8, hipMemcpyHostToDevice));
    }
    if ((rc = decompose_metric(h, d_j2c, naux, lindep, &m->mt, &m->nrow, &m->lda, &m->tri))) return rc;
    tmp.release(d_j2c);
    if (d_scr) tmp.release(d_scr);
    return 0;
}

int build_square_image(PAMD_df *h, size_t cap_left, size_t reserve)
{
    if (h->square) return 0;
    // K path on the unpacked image when HBM allows (DF.k_square = 'auto': 48 GB must stay free afterwards);
    // PAMD_DF_SQUARE=0 in the environment = DF.k_square = False (tests, memory-constrained callers)
    const char *env = getenv("PAMD_DF_SQUARE");
    if (env && env[0] == '0') return 0;
    size_t free_b = 0, total_b = 0;
    PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t need = ((size_t)h->nL * h->rows * h->rows + 256) * 8;
    if (h->nL == 0 || h->n_res < h->nL || need + (48ul << 30) + reserve > free_b || need > cap_left) return 0;
    int rc = h->pool.alloc((void **)&h->d_sq, need);
    if (rc) { h->d_sq = nullptr; return 0; }
    PAMD_CHECK_HIP(hipMemsetAsync(h->d_sq, 0, need, h->st));
    return PAMD_unpack_tril(h->d_cderi, h->npair, h->nL, h->nao, h->d_sq, h->rows, h->rows, h->st);
}

// no room for the square image: keep at least the 128 x 128 diagonal blocks unpacked (14 % of the packed size at nao 1856), so
// that the packed-operand half transform reads the k-tiles crossing the diagonal once and unmasked (DF.diag_image)
int build_diag_image(PAMD_df *h, size_t cap_left)
{
    if (h->square || h->d_sq || h->nL == 0 || h->n_res < h->nL || h->nao < 128) return 0;
    size_t free_b = 0, total_b = 0;
    PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    const size_t need = (size_t)PAMD_e2_diag_size(h->nL, h->rows) * 8;
    if (need + (48ul << 30) > free_b || need > cap_left) return 0;
    int rc = h->pool.alloc((void **)&h->d_diag, need);
    if (rc) { h->d_diag = nullptr; return 0; }
    return PAMD_e2_diag_blocks(h->d_cderi, h->npair, h->nL, h->nao, h->rows, h->d_diag, h->st);
}

// Rows [h->l0, h->l0 + h->nL) of the tensor, AO-row slab by slab (df/incore.py:189-217): into HBM as far as `max_device_bytes`
// (0: the device's free memory) allows, the remaining rows into page-locked host memory through a staging buffer.
// r06 - one layout, one budget: when 2x the packed rows, the build's work slabs and what the rest of the
// calculation needs afterwards (half-transform block 12 GB + 4 GB of J/K work space + `reserve`, the XC leg's compact AO image and
// work buffers as the caller states them in PAMD_df_options.reserve_bytes) fit the device, the rows are built STRAIGHT INTO the
// square layout d_sq[nL][rows][rows] - every column slab is solved into a work slab and scattered into both triangles
// (PAMD_unpack_tril_slab) - and no packed copy exists.  PAMD_DF_LAYOUT=packed|square overrides; PAMD_DF_SQUARE=0 = packed.
int build_rows(PAMD_df *h, const Engine &e, DevPool &tmp, const Metric &m, size_t max_device_bytes, size_t reserve)
{
    int rc;
    const Tables &t = *e.t;
    const int naux = h->naux, nL = h->nL;
    const long npair = h->npair;
    double *d_mt = nullptr;
    if ((rc = upload(tmp, m.mt, &d_mt))) return rc;
    size_t free_b = 0, total_b = 0;
    PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    const char *envcap = getenv("PAMD_DF_DEVICE_BYTES");            // tests / memory-constrained callers
    if (envcap && atof(envcap) > 0) max_device_bytes = (size_t)atof(envcap);
    const size_t cap = max_device_bytes ? std::min(free_b, max_device_bytes) : free_b;
    const size_t row_b = (size_t)npair * 8, tensor_b = (size_t)nL * row_b;
    const int npass = h->omega < 0 ? 2 : 1;
    size_t slab_bytes = std::min<size_t>(24ul << 30, (size_t)npair * naux * 8);
    const size_t margin = std::min<size_t>(1ul << 30, cap / 16);
    size_t stage_b = 0;
    {
        const char *lay = getenv("PAMD_DF_LAYOUT"), *sqenv = getenv("PAMD_DF_SQUARE");
        const size_t sq_b = ((size_t)nL * ((size_t)h->rows * h->rows + PAMD_df::SQ_STRIDE_PAD) + 256) * 8;
        const size_t slab12 = std::min<size_t>(12ul << 30, slab_bytes);
        const size_t out_b = std::min<size_t>(slab12, (size_t)((double)slab12 * std::max(nL, 1) / std::max(naux, 1)) + (1ul << 20));
        const size_t after = sq_b + (12ul << 30) + (4ul << 30) + reserve;
        // packed rows + a FULL image are preferred while all three copies fit the budget (48 GB + reserve stay free after the image,
        // build_square_image): beside the co-running SYRK the packed second J pass costs less than the square one (measured,
        // profiles/r06/README.md).  The square layout is for the tensors whose 3x does not fit (taxol on one GPU).  The order of
        // preference is the library's (PAMD_df_layout_pick, shared with df.DF._choose_layout); the byte counts are this layer's.
        const char *pref = getenv("PAMD_DF_PREFER_IMAGE");
        const bool sq_ok = nL > 0 && h->nao >= 128;
        const long long need_pi = tensor_b + npass * slab_bytes + margin <= cap ? (long long)(tensor_b + sq_b + (48ul << 30) + reserve + margin) : 0;
        bool want = PAMD_df_layout_pick(need_pi, sq_ok ? (long long)(sq_b + npass * slab12 + out_b + margin) : 0, (long long)after, (long long)cap,
                                        !(pref && pref[0] == '0')) == 1;
        if (lay && lay[0] == 'p') want = false;
        if (sqenv && sqenv[0] == '0') want = false;
        if (lay && lay[0] == 's' && nL > 0 && sq_b + npass * slab12 + out_b + margin <= cap) want = true;
        if (want) { h->square = 1; slab_bytes = slab12; }
    }
    if (h->square) {
        h->n_res = nL;
    } else if (tensor_b + npass * slab_bytes + margin <= cap) {
        h->n_res = nL;
    } else {
        // out of core: a share of the cap each for the slab work space, the two staging buffers and the J/K work space
        // (r05: smaller reserves - 3 GB slabs, 2 GB staging buffers, 12 GB of J/K work space instead of 6 / 4 / 20 - keep ~15 GB
        // more of the tensor resident: that much less page-locked host memory, which is what a container limit is charged for)
        slab_bytes = std::min<size_t>(slab_bytes, std::min<size_t>(3ul << 30, cap / (4 * npass)));
        stage_b = std::min<size_t>(2ul << 30, cap / 8);
        const size_t work_b = std::min<size_t>(12ul << 30, cap / 4);
        // head room for the other clients of the device in this process (a torch context created AFTER the handle found no
        // memory at config 5 - "no HIP device" in the SCF driver): 4 GB or 1/16 of the cap stay unclaimed
        const size_t used = npass * slab_bytes + 2 * stage_b + work_b + margin + std::min<size_t>(4ul << 30, cap / 16);
        h->stage_rows = (int)std::min<size_t>(stage_b / row_b, (size_t)nL);
        if (h->stage_rows < 1) {
            snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create: %.3f GB of device memory cannot stage one tensor row (%.3f GB)",
                     cap * 1e-9, row_b * 1e-9);
            return -2;
        }
        h->n_res = cap > used ? (int)std::min<size_t>((cap - used) / row_b, (size_t)nL) : 0;
        const size_t host_b = (size_t)(nL - h->n_res) * row_b;
        const size_t left = container_memory_left();
        if (left != ~(size_t)0 && host_b + (12ul << 30) > left) {
            // refuse BEFORE the kernel's OOM killer does: the process (and on a GPU box the whole container) would be killed
            snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create: %.1f GB of the tensor do not fit the device and the %.1f GB of page-locked "
                     "host memory they need exceed what the container's memory limit leaves (%.1f GB incl. a 12 GB margin): use more "
                     "devices / ranks", tensor_b * 1e-9, host_b * 1e-9, left * 1e-9);
            return -3;
        }
        if (hipHostMalloc((void **)&h->h_cderi, std::max<size_t>(host_b, 8), hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->h_cderi = nullptr;
            snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create: %.1f GB of the tensor do not fit the device and %.1f GB of page-locked "
                     "host memory could not be allocated", tensor_b * 1e-9, host_b * 1e-9);
            return -2;
        }
        for (int k = 0; k < 2; k++)
            if ((rc = h->pool.alloc((void **)&h->d_stage[k], (size_t)h->stage_rows * row_b + 256 * 8))) return rc;
    }
    if (h->square) {
        const size_t sq_b = ((size_t)nL * h->sq_ls() + 256) * 8;
        if ((rc = h->pool.alloc((void **)&h->d_sq, sq_b))) return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(h->d_sq, 0, sq_b, h->st));             // pads (rows / columns >= nao) and the slack stay zero
    } else if (h->n_res && (rc = h->pool.alloc((void **)&h->d_cderi, (size_t)h->n_res * row_b + 256 * 8))) return rc;
    const long max_rows = std::max<long>((long)(slab_bytes / ((size_t)naux * 8)), 1);
    std::vector<std::pair<int, int>> slabs;
    long bufrows = 0;
    for (int sh0 = 0; sh0 < t.ao.n;) {
        int sh1 = sh0 + 1;
        long r0, r1;
        while (sh1 < t.ao.n) {
            slab_rows(t.ao, sh0, sh1 + 1, &r0, &r1);
            if (r1 - r0 > max_rows) break;
            sh1++;
        }
        slab_rows(t.ao, sh0, sh1, &r0, &r1);
        bufrows = std::max(bufrows, r1 - r0);
        slabs.push_back({sh0, sh1});
        sh0 = sh1;
    }
    double *d_T = nullptr, *d_T2 = nullptr, *d_slab = nullptr;
    if (h->square && (rc = tmp.alloc((void **)&d_slab, (size_t)std::max(nL, 1) * bufrows * 8))) return rc;
    if ((rc = tmp.alloc((void **)&d_T, (size_t)bufrows * naux * 8))) return rc;
    if (npass == 2 && (rc = tmp.alloc((void **)&d_T2, (size_t)bufrows * naux * 8))) return rc;
    for (auto &sl : slabs) {
        long r0, r1;
        slab_rows(t.ao, sl.first, sl.second, &r0, &r1);
        const long ncol = r1 - r0;
        rc = fill_operator(h, d_T, d_T2, (size_t)ncol * naux, [&](double *dst, double om) {
            for (const PairClass &pc : e.pcs) {
                int i0, i1;
                pc.subrange(sl.first, sl.second, &i0, &i1);
                for (const AuxClass &ac : e.acs) {
                    const int r = launch_class(e, pc, i0, i1, ac, dst, naux, r0, 1, e.d_ao_xyz, e.d_ao_ao0, om, h->st);
                    if (r) return r;
                }
            }
            return 0;
        });
        if (rc) return rc;
        if (h->square) {
            const int p0 = t.ao.ao0[sl.first], p1 = sl.second < t.ao.n ? t.ao.ao0[sl.second] : t.ao.nao;
            if ((rc = PAMD_cderi_solve(d_mt + h->l0, m.lda, d_T, naux, d_slab, ncol, nL, ncol, naux, h->l0, m.tri, h->st))) return rc;
            if ((rc = PAMD_unpack_tril_slab(d_slab, ncol, nL, p0, p1, h->d_sq, h->rows, h->sq_ls(), h->st))) return rc;
            continue;
        }
        if (h->n_res &&
            (rc = PAMD_cderi_solve(d_mt + h->l0, m.lda, d_T, naux, h->d_cderi + r0, npair, h->n_res, ncol, naux, h->l0, m.tri, h->st)))
            return rc;
        for (int rb0 = h->n_res; rb0 < nL; rb0 += h->stage_rows) {
            // host rows of this column slab: solve into the staging buffer ([rows][ncol]), strided copy into the pinned tensor
            const int nb = std::min(h->stage_rows, nL - rb0);
            if ((rc = PAMD_cderi_solve(d_mt + h->l0 + rb0, m.lda, d_T, naux, h->d_stage[0], ncol, nb, ncol, naux, h->l0 + rb0, m.tri, h->st)))
                return rc;
            PAMD_CHECK_HIP(hipMemcpy2DAsync(h->h_cderi + (size_t)(rb0 - h->n_res) * npair + r0, row_b, h->d_stage[0], (size_t)ncol * 8,
                                            (size_t)ncol * 8, nb, hipMemcpyDeviceToHost, h->st));
        }
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    tmp.release(d_T);
    if (d_T2) tmp.release(d_T2);
    if (d_slab) tmp.release(d_slab);
    tmp.release(d_mt);
    const size_t held = (size_t)h->n_res * row_b + 2 * stage_b;
    const size_t cap_left = max_device_bytes ? (cap > held ? cap - held : 0) : ~(size_t)0;
    if ((rc = build_square_image(h, cap_left, reserve))) return rc;
    if ((rc = build_diag_image(h, cap_left))) return rc;
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    return 0;
}

// one shard, start to finish, on the calling thread: `metric` is computed here (and returned) when m->nrow == 0 on entry
int create_shard(const Tables &t, int device, double omega, double lindep, size_t max_device_bytes, Metric *m, int part, int nparts,
                 PAMD_df **out, size_t reserve = 0)
{
    *out = nullptr;
    PAMD_CHECK_HIP(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_dev_mutex[device & 63]);
    PAMD_df *h = new PAMD_df;
    struct Guard { PAMD_df *p; ~Guard() { delete p; } } guard{h};
    int rc;
    if ((rc = init_shard(h, t, device, omega))) return rc;
    Engine e;
    DevPool tmp;                           // tables and scratch that die with this call
    if ((rc = prepare_engine(h, t, e, tmp))) return rc;
    if (m->nrow == 0 && m->mt.empty() && (rc = compute_metric(h, e, tmp, lindep, m))) return rc;
    // contiguous, row-balanced shards (DF.shard_range)
    const int base = m->nrow / nparts, rem = m->nrow % nparts;
    h->nL_total = m->nrow;
    h->l0 = part * base + std::min(part, rem);
    h->nL = base + (part < rem ? 1 : 0);
    if ((rc = build_rows(h, e, tmp, *m, max_device_bytes, reserve))) return rc;
    guard.p = nullptr;
    *out = h;
    return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ J/K of one shard
struct Seg {                               // a run of tensor rows as the kernels see it
    const double *rows;                    // packed rows on the device
    int n, row0;                           // count, first (shard-local) row
    const double *sq, *diag;               // unpacked image / diagonal-block image of these rows (nullable)
    int stage;                             // -1: resident; 0 / 1: arrives in that staging buffer
};

struct OrbSet {                            // one density's occupied orbitals on the device (df_jk.pad_orbitals)
    double *d_orb = nullptr;
    int no = 0, nocc_pad = 0;
    long ldo = 0;
};

// dm   [nset][nao][nao] host, f64, C order.
// orbo nullable.  Not NULL: the occupied orbitals scaled by sqrt(occ), set after set, each (nao, nocc[s]) C order (row = AO) -
//      K comes from the MO branch (df_jk.py:339-381).  NULL: general-DM branch (df_jk.py:382-408), hermi is then ignored for K.
// flags bit 0: the caller guarantees dm[s] = orbo_s orbo_s^T (what make_rdm1 builds) - the first J pass then comes out of the
//      half transform's epilogue instead of a pass over the tensor.
// vj, vk caller-owned [nset][nao][nao] (NULL with with_j / with_k = 0).
// serial_j2: the second J pass of the fused path in line before a re-tiled SYRK (1), on the side stream beside a plain one (0), or -
// r05 - inside the re-tiled SYRK kernel itself (2: PAMD_syrk_jfused; shapes without a fused form fall back to 1)
// download = 0: the results stay on the device (work spaces "vjtril": packed J~ [nset][npair], "vk": [nset][nao][nao]) for the
//      multi-device reduction; the stream is synchronised either way.
// Rows in host memory (out-of-core shard) arrive block by block in two staging buffers, the copy of block b + 1 under the kernels
// of block b; every contraction is a sum over aux rows, so each block is contracted completely (both J passes, half transform,
// SYRK into the split-K partials) while it is on the device - one sweep over the host rows per build.
static const int SYRK_RESERVE = 16;          // DF.k_syrk_reserve of the torch path

// r06 - the tag probe INSIDE the call (PAMD_df_get_jk flags bit 1).  "Is dm[s] == orbo_s orbo_s^T?" decides whether the first J pass
// may come out of the half transform's epilogue; the binding used to answer it on the host BEFORE the blocking call (one D v product
// per density: with the probe now on the FULL matrix - ADVICE r05 - 5-7 ms of a 256-thread host's BLAS start-up per call at config
// 3).  Here it runs on the calling thread AFTER the kernels have been queued and before the final synchronisation: the GPU works
// ~100 ms, the probe ~3 ms - hidden.  max_s |D_s v - C_s (C_s^T v)| / max(1, |D_s v|) for one fixed pseudo-random vector.
static double host_dm_mismatch(const double *dm, const double *orbo, const int *nocc, int nset, int nao)
{
    std::vector<double> v(nao), t, u(nao);
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < nao; i++) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;                         // xorshift64: any fixed vector will do
        v[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
    }
    double worst = 0;
    const double *c = orbo;
    for (int s = 0; s < nset; s++) {
        const int no = nocc[s];
        t.assign(std::max(no, 1), 0.0);
        for (int p = 0; p < nao; p++) {
            const double *row = c + (size_t)p * no;
            for (int i = 0; i < no; i++) t[i] += row[i] * v[p];
        }
        double dmax = 1.0, emax = 0.0;
        const double *d = dm + (size_t)s * nao * nao;
        for (int p = 0; p < nao; p++) {
            const double *drow = d + (size_t)p * nao, *row = c + (size_t)p * no;
            double a = 0, b = 0;
            for (int q = 0; q < nao; q++) a += drow[q] * v[q];
            for (int i = 0; i < no; i++) b += row[i] * t[i];
            u[p] = a - b;
            dmax = std::max(dmax, std::fabs(a));
            emax = std::max(emax, std::fabs(u[p]));
        }
        worst = std::max(worst, emax / dmax);
        c += (size_t)nao * no;
    }
    return worst;
}

static int df_get_jk_impl(PAMD_df *h, const double *dm, const double *orbo, const int *nocc, int nset, int nao, int hermi, int with_j,
                          int with_k, int flags, double *vj, double *vk, int serial_j2, int download)
{
    (void)hermi;
    // r06: flags bit 3 - dm, orbo, vj and vk are DEVICE pointers on this handle's device (the HBM-resident SCF
    // loop over a handle-held tensor: nothing crosses PCIe).  The tag cannot be probed from the host then: it counts as promised
    // (bit 0) or not given at all (J from the matrix).
    const bool dev_io = (flags & 8) != 0;
    if (dev_io) flags &= ~2;
    const hipMemcpyKind k_in = dev_io ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const hipMemcpyKind k_out = dev_io ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    PAMD_REQUIRE(h && dm && nset > 0 && nao == h->nao, "PAMD_df_get_jk: bad arguments (nao must equal the handle's)");
    PAMD_REQUIRE(!download || ((!with_j || vj) && (!with_k || vk)), "PAMD_df_get_jk: output pointers");
    PAMD_REQUIRE(with_j || with_k, "PAMD_df_get_jk: nothing to do");
    PAMD_REQUIRE(!orbo || nocc, "PAMD_df_get_jk: orbo needs nocc[nset]");
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    hipStream_t st = h->st;
    int rc;
    const long npair = h->npair;
    const int nL = h->nL, ldx = h->rows, rows = h->rows;
    const size_t n2 = (size_t)nao * nao;
    const bool streamed = h->n_res < nL;
    if (streamed) serial_j2 = 1;           // the staged rows are released when the main stream is done with them
    if (h->square && serial_j2 == 2) serial_j2 = 0;     // the in-SYRK pass streams PACKED rows: not a schedule of the square layout
    const long lstride = h->sq_ls();
    static const bool j2_before = [] { const char *e = getenv("PAMD_DF_J2_ORDER"); return e && e[0] == 'b'; }();
    bool j2_deferred = false;
    // the J passes over aux rows [b0, b0 + nb) of a segment, from whichever layout holds them (square: the p >= q runs)
    auto j_pass1 = [&](const double *rows_pk, const double *rows_sq, int nb, const double *dt, int ns, double *rho, double *work, hipStream_t s_) {
        return rows_pk ? PAMD_df_vj_pass1(rows_pk, npair, nb, dt, ns, rho, work, s_)
                       : PAMD_df_vj_pass1_sq(rows_sq, lstride, h->rows, nao, nb, dt, ns, rho, work, s_);
    };
    auto j_pass2 = [&](const double *rows_pk, const double *rows_sq, int nb, const double *rho, int ns, double *vjt, hipStream_t s_) {
        return rows_pk ? PAMD_df_vj_pass2(rows_pk, npair, nb, rho, ns, vjt, s_)
                       : PAMD_df_vj_pass2_sq(rows_sq, lstride, h->rows, nao, nb, rho, ns, vjt, s_);
    };
    double *d_vjt = nullptr, *d_vk = nullptr, *d_rho = nullptr, *d_dt = nullptr, *d_w1 = nullptr, *d_part = nullptr;
    const bool fused = with_j && with_k && orbo && (flags & 3) && nL > 0;       // bit 0: promised; bit 1: verified below, beside the kernels
    // the matrix itself is needed on the device for J from the matrix and for the general-DM K branch only: with J taken from the
    // orbitals (fused) the 8 nao^2-byte upload of a pageable caller array per set (per part of a device list) is skipped
    const bool need_dm = (with_j && !fused) || (with_k && !orbo);
    double *d_dm = nullptr;
    if (need_dm) {
        d_dm = h->workspace("dm", (size_t)nset * n2, &rc);
        if (rc) return rc;
        PAMD_CHECK_HIP(hipMemcpyAsync(d_dm, dm, (size_t)nset * n2 * 8, k_in, st));
    }
    double *d_vjfull = nullptr;
    if (with_j && download) {
        // (claimed before any kernel is queued: a fresh work space is zeroed on `st`, and the unpack below runs on the side stream)
        d_vjfull = h->workspace("vjfull", (size_t)nset * n2, &rc);
        if (rc) return rc;
    }
    bool j_on_st = false;
    if (with_j) {
        d_vjt = h->workspace("vjtril", (size_t)nset * npair, &rc);
        if (rc) return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(d_vjt, 0, (size_t)nset * npair * 8, st));
        d_rho = h->workspace("rho", (size_t)nset * std::max(nL, 1), &rc);
        if (rc) return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(d_rho, 0, (size_t)nset * std::max(nL, 1) * 8, st));
        if (!fused && nL > 0) {
            d_dt = h->workspace("dmtril", (size_t)nset * npair, &rc);
            if (rc) return rc;
            if ((rc = PAMD_pack_dm_tril(d_dm, nset, nao, d_dt, st))) return rc;
        }
    }
    // ---- rows as the kernels will see them
    std::vector<Seg> segs;
    if (h->n_res > 0) segs.push_back({h->d_cderi, h->n_res, 0, h->d_sq, h->d_diag, -1});     // (square layout: rows = null, sq = the rows)
    for (int b0 = h->n_res, k = 0; b0 < nL; b0 += h->stage_rows, k ^= 1)
        segs.push_back({h->d_stage[k], std::min(h->stage_rows, nL - b0), b0, nullptr, nullptr, k});
    int max_seg = 0;
    for (const Seg &sg : segs) max_seg = std::max(max_seg, sg.n);
    if (with_j && !fused && nL > 0) {
        d_w1 = h->workspace("vj1work", (size_t)std::max<long>(PAMD_df_vj_pass1_worksize(npair, max_seg, std::min(nset, 4)), 1), &rc);
        if (rc) return rc;
    }
    // ---- K set-up: SYRK plan, orbitals of every set, split-K partials of every set
    int nsplit = 4, syrk_flags = 1 | 2;
    std::vector<OrbSet> orbs(nset);
    size_t budget = 12ul << 30;                                 // DF.k_block_bytes
    if (streamed) {
        size_t free_b = 0, total_b = 0;
        PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
        budget = std::min(budget, std::max<size_t>(free_b / 3, 1ul << 20));
    }
    if (with_k) {
        d_vk = h->workspace("vk", (size_t)nset * n2, &rc);
        if (rc) return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(d_vk, 0, (size_t)nset * n2 * 8, st));
        // tile shape and k splits of K = X^T X: PAMD_syrk_plan, the one rule of both orchestrations (r06).  Beside a co-running J pass 2
        // the balanced schedule leaves SYRK_RESERVE of the 512 workgroup slots to the pass (flag bits 8-15 of PAMD_dgemm_tn)
        const int reserve = (fused && serial_j2 == 0) ? SYRK_RESERVE : 0;
        if ((rc = PAMD_syrk_plan(nao, reserve, orbo ? -1 : 0, 0, &syrk_flags, &nsplit))) return rc;
        if (reserve && (syrk_flags & 4)) syrk_flags |= (reserve / 4) << 8;
        d_part = h->workspace("kpart", (size_t)nset * nsplit * n2, &rc);
        if (rc) return rc;
        PAMD_CHECK_HIP(hipMemsetAsync(d_part, 0, (size_t)nset * nsplit * n2 * 8, st));
        const double *op = orbo;
        size_t orb_stage_off = 0;
        for (int s = 0; s < nset && orbo; s++) {
            OrbSet &o = orbs[s];
            o.no = nocc[s];
            const double *o_s = op;
            op += (size_t)nao * o.no;
            if (o.no == 0) continue;
            o.nocc_pad = (int)round_up(o.no, 16);
            const long ldo = PAMD_e2_orb_ld(o.nocc_pad);       // the library's one rule (df_jk.pad_orbitals asks the same function; r06: the
            o.ldo = ldo;                                         // handle's own transcription of its first three terms is gone)
            const size_t olen = (size_t)rows * ldo;
            if (dev_io) {
                // device orbitals [nao][no] -> the zero-padded operand [rows][ldo], on the stream
                char name[32];
                snprintf(name, sizeof(name), "orb%d", s);
                o.d_orb = h->workspace(name, olen, &rc);
                if (rc) return rc;
                PAMD_CHECK_HIP(hipMemsetAsync(o.d_orb, 0, olen * 8, st));
                PAMD_CHECK_HIP(hipMemcpy2DAsync(o.d_orb, (size_t)ldo * 8, o_s, (size_t)o.no * 8, (size_t)o.no * 8, nao, hipMemcpyDeviceToDevice, st));
                continue;
            }
            // padded image in the handle's page-locked staging area (all sets side by side; the call ends with a stream
            // synchronisation, so the area is free again at the next call): no per-call allocation, no extra synchronisation
            if (orb_stage_off + olen > h->h_orb_len) {
                PAMD_CHECK_HIP(hipStreamSynchronize(st));                  // copies of the earlier sets still read the old area
                size_t want = std::max<size_t>(2 * (orb_stage_off + olen), (size_t)1 << 20);
                double *nb = nullptr;
                PAMD_CHECK_HIP(hipHostMalloc((void **)&nb, want * 8, hipHostMallocPortable));
                if (h->h_orb) (void)hipHostFree(h->h_orb);
                h->h_orb = nb;
                h->h_orb_len = want;
                orb_stage_off = 0;
            }
            double *oh = h->h_orb + orb_stage_off;
            orb_stage_off += olen;
            for (int p = 0; p < nao; p++) {
                double *row = oh + (size_t)p * ldo;
                memcpy(row, o_s + (size_t)p * o.no, (size_t)o.no * 8);
                if (ldo > o.no) memset(row + o.no, 0, (size_t)(ldo - o.no) * 8);
            }
            if (rows > nao) memset(oh + (size_t)nao * ldo, 0, (size_t)(rows - nao) * ldo * 8);
            char name[32];
            snprintf(name, sizeof(name), "orb%d", s);
            o.d_orb = h->workspace(name, olen, &rc);
            if (rc) return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(o.d_orb, oh, olen * 8, hipMemcpyHostToDevice, st));
        }
    }
    // general-DM branch: the density itself is the "orbital" operand
    double *d_orb_dm = nullptr;
    const long ldo_dm = std::max<long>(rows > 160 ? round_up(rows, 160) : round_up(rows, 32), PAMD_e2_orb_ld(rows));   // whole wave tiles
    if (with_k && !orbo && nL > 0) {
        d_orb_dm = h->workspace("orb_dm", (size_t)rows * ldo_dm, &rc);
        if (rc) return rc;
    }
    // ---- staging pipeline of the host rows
    for (int k = 0; k < 2 && streamed; k++) PAMD_CHECK_HIP(hipEventRecord(h->ev_free[k], st));
    auto issue_copy = [&](size_t i) -> int {
        if (i >= segs.size() || segs[i].stage < 0) return 0;
        const Seg &sg = segs[i];
        PAMD_CHECK_HIP(hipStreamWaitEvent(h->copy, h->ev_free[sg.stage], 0));
        PAMD_CHECK_HIP(hipMemcpyAsync(h->d_stage[sg.stage], h->h_cderi + (size_t)(sg.row0 - h->n_res) * npair, (size_t)sg.n * npair * 8,
                                      hipMemcpyHostToDevice, h->copy));
        PAMD_CHECK_HIP(hipEventRecord(h->ev_ready[sg.stage], h->copy));
        return 0;
    };
    // HIP events around the MFMA kernels of this build (kinds 0/1: half transform begin / end, 2/3: SYRK begin / end), read after
    // the final synchronisation: PAMD_df_last_timing -> bench.py --single-process `roofline`
    std::vector<int> mark_kind;
    auto mark = [&](int kind) {
        if (mark_kind.size() >= 512) return;
        hipEvent_t e = h->timing_event(mark_kind.size());
        if (!e) return;
        (void)hipEventRecord(e, st);
        mark_kind.push_back(kind);
    };
    for (size_t i = 0; i < segs.size(); i++)
        if (segs[i].stage >= 0) { if ((rc = issue_copy(i))) return rc; break; }     // prime the first staged block
    for (size_t si = 0; si < segs.size(); si++) {
        const Seg &sg = segs[si];
        if (sg.stage >= 0) {
            if ((rc = issue_copy(si + 1))) return rc;                  // the next block travels while this one is contracted
            PAMD_CHECK_HIP(hipStreamWaitEvent(st, h->ev_ready[sg.stage], 0));
        }
        if (with_j && !fused) {
            for (int s0 = 0; s0 < nset; s0 += 4) {
                const int ns = std::min(4, nset - s0);
                // rho of 4 sets at a time: [ns][n] contiguous for the kernels, then scattered into rho[s][row0 ..]
                double *d_rs = h->workspace("rho_seg", (size_t)4 * std::max(max_seg, 1), &rc);
                if (rc) return rc;
                if ((rc = j_pass1(sg.rows, sg.sq, sg.n, d_dt + (size_t)s0 * npair, ns, d_rs, d_w1, st))) return rc;
                if ((rc = j_pass2(sg.rows, sg.sq, sg.n, d_rs, ns, d_vjt + (size_t)s0 * npair, st))) return rc;
            }
            PAMD_CHECK_HIP(hipEventRecord(h->ev_j, st));
            j_on_st = true;
        }
        for (int s = 0; s < nset && with_k; s++) {
            double *part_s = d_part + (size_t)s * nsplit * n2;
            if (orbo) {
                // ---- MO branch: X_L = B_L C~, K += X^T X (df_jk.py:353-380)
                const OrbSet &o = orbs[s];
                if (o.no == 0) continue;
                const long blk = PAMD_k_block_rows(sg.n, o.nocc_pad, ldx, (long long)budget);      // (df_jk._k_blocksize asks the same function)
                // rows per aux index in X = the orbitals themselves (no padding to 16): the SYRK contracts nb * no rows, padded with
                // zero rows to a whole k-tile at the END of the block only
                const int xr = o.no;
                double *d_X = h->workspace("X", ((size_t)blk * xr + 16) * ldx, &rc);
                if (rc) return rc;
                double *d_rw = nullptr;
                if (fused) {
                    d_rw = h->workspace("rho_work", (size_t)std::max<long>(PAMD_nr_e2_rho_worksize((int)blk, ldx, o.nocc_pad), 1), &rc);
                    if (rc) return rc;
                }
                for (long b0 = 0; b0 < sg.n; b0 += blk) {
                    const int nb = (int)std::min<long>(blk, sg.n - b0);
                    const double *sub = sg.rows ? sg.rows + (size_t)b0 * npair : nullptr;
                    const double *sub_sq = sg.sq ? sg.sq + (size_t)b0 * lstride : nullptr;
                    double *rho_b = fused ? d_rho + (size_t)s * nL + sg.row0 + b0 : nullptr;
                    mark(0);
                    if (sg.sq)
                        rc = PAMD_nr_e2_square_ls(sub_sq, rows, rows, lstride, nb, nao, o.d_orb, (int)o.ldo, rows, xr,
                                                  d_X, ldx, rho_b, d_rw, st);
                    else
                        rc = PAMD_nr_e2_symm_diag(sub, npair, nb, nao, o.d_orb, (int)o.ldo, rows, xr, d_X, ldx, rho_b, d_rw,
                                                  sg.diag ? sg.diag + (size_t)PAMD_e2_diag_size((int)b0, ldx) : nullptr, st);
                    if (rc) return rc;
                    mark(1);
                    if (fused && serial_j2 != 2) {
                        // second J pass of this block: in line, or on the side stream beside the block's SYRK (HBM- beside MFMA-bound)
                        if (serial_j2) {
                            if ((rc = j_pass2(sub, sub_sq, nb, rho_b, 1, d_vjt + (size_t)s * npair, st))) return rc;
                            PAMD_CHECK_HIP(hipEventRecord(h->ev_j, st));
                            j_on_st = true;
                        } else {
                            PAMD_CHECK_HIP(hipEventRecord(h->ev, st));
                            PAMD_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev, 0));
                            // r06: the pass is ENQUEUED after the SYRK (below) unless PAMD_DF_J2_ORDER=before: launched first, its
                            // 6700 long-running workgroups sometimes take the CUs' wave slots before the SYRK's 496 workgroups
                            // arrive, and the balanced one-round SYRK then starts part of its grid late (bimodal: 45 or 50 ms)
                            if (j2_before && (rc = j_pass2(sub, sub_sq, nb, rho_b, 1, d_vjt + (size_t)s * npair, h->side))) return rc;
                            j2_deferred = !j2_before;
                        }
                    }
                    const long kx = (long)nb * xr, kx16 = round_up(kx, 16);
                    if (kx16 > kx) { PAMD_CHECK_HIP(hipMemsetAsync(d_X + (size_t)kx * ldx, 0, (size_t)(kx16 - kx) * ldx * 8, st)); }
                    mark(2);
                    if (fused && serial_j2 == 2) {
                        // r05: the second J pass of these rows INSIDE the SYRK kernel (PAMD_syrk_jfused); 1 = no fused form for the
                        // shape, nothing launched: the pass in line, then the plain SYRK
                        rc = PAMD_syrk_jfused(d_X, ldx, part_s, nao, nao, kx16, syrk_flags, nsplit, sub, npair, nb, rho_b,
                                              d_vjt + (size_t)s * npair, st);
                        if (rc < 0) return rc;
                        if (rc == 1) {
                            if ((rc = PAMD_df_vj_pass2(sub, npair, nb, rho_b, 1, d_vjt + (size_t)s * npair, st))) return rc;
                            if ((rc = PAMD_dgemm_tn(d_X, ldx, d_X, ldx, part_s, nao, nao, nao, kx16, syrk_flags, nsplit, st))) return rc;
                        }
                        PAMD_CHECK_HIP(hipEventRecord(h->ev_j, st));
                        j_on_st = true;
                    } else if ((rc = PAMD_dgemm_tn(d_X, ldx, d_X, ldx, part_s, nao, nao, nao, kx16, syrk_flags, nsplit, st))) return rc;
                    mark(3);
                    if (j2_deferred) {
                        if ((rc = j_pass2(sub, sub_sq, nb, rho_b, 1, d_vjt + (size_t)s * npair, h->side))) return rc;
                        j2_deferred = false;
                    }
                }
            } else {
                // ---- general-DM branch: T_L = B_L D, K = sum_L T_L^T B_L (df_jk.py:382-407)
                PAMD_CHECK_HIP(hipMemsetAsync(d_orb_dm, 0, (size_t)rows * ldo_dm * 8, st));
                PAMD_CHECK_HIP(hipMemcpy2DAsync(d_orb_dm, (size_t)ldo_dm * 8, d_dm + (size_t)s * n2, (size_t)nao * 8, (size_t)nao * 8, nao,
                                                hipMemcpyDeviceToDevice, st));
                long blk = std::max<long>(1, (long)(budget / ((size_t)rows * ldx * 8)));
                blk = std::min<long>(blk, sg.n);
                const long nblk = (sg.n + blk - 1) / blk;
                blk = std::max<long>(1, ((sg.n + nblk - 1) / nblk) / 2);
                double *d_X = h->workspace("X", (size_t)blk * rows * ldx, &rc);
                if (rc) return rc;
                // rows with an unpacked image: it IS the second operand (no per-block unpack) and feeds the square-image kernel
                double *d_full = nullptr;
                const bool padded = sg.sq && lstride != (long)rows * rows;      // square LAYOUT: the aux rows are not one tall matrix
                if (!sg.sq || padded) {
                    d_full = h->workspace("full", (size_t)blk * rows * ldx, &rc);
                    if (rc) return rc;
                    PAMD_CHECK_HIP(hipMemsetAsync(d_full, 0, (size_t)blk * rows * ldx * 8, st));
                }
                for (long b0 = 0; b0 < sg.n; b0 += blk) {
                    const int nb = (int)std::min<long>(blk, sg.n - b0);
                    const double *sub = sg.rows ? sg.rows + (size_t)b0 * npair : nullptr;
                    const double *second = d_full;
                    if (sg.sq) {
                        second = sg.sq + (size_t)b0 * lstride;
                        if ((rc = PAMD_nr_e2_square_ls(second, rows, rows, lstride, nb, nao, d_orb_dm, (int)ldo_dm, rows, rows, d_X, ldx, nullptr,
                                                       nullptr, st)))
                            return rc;
                        if (padded) {
                            // the product below reads consecutive aux rows as ONE tall matrix: a contiguous copy of the block (a device
                            // copy against 4 nb nao^3 flops; the general-DM branch is the rare one, df_jk.py:382-407)
                            PAMD_CHECK_HIP(hipMemcpy2DAsync(d_full, (size_t)rows * rows * 8, second, (size_t)lstride * 8, (size_t)rows * rows * 8, nb,
                                                            hipMemcpyDeviceToDevice, st));
                            second = d_full;
                        }
                    } else {
                        if ((rc = PAMD_nr_e2_symm_diag(sub, npair, nb, nao, d_orb_dm, (int)ldo_dm, rows, rows, d_X, ldx, nullptr, nullptr,
                                                       sg.diag ? sg.diag + (size_t)PAMD_e2_diag_size((int)b0, ldx) : nullptr, st))) return rc;
                        if ((rc = PAMD_unpack_tril(sub, npair, nb, nao, d_full, ldx, rows, st))) return rc;
                    }
                    if ((rc = PAMD_dgemm_tn(d_X, ldx, second, ldx, part_s, nao, nao, nao, (long)nb * rows, 0 | 2, nsplit, st))) return rc;
                }
            }
        }
        if (sg.stage >= 0) PAMD_CHECK_HIP(hipEventRecord(h->ev_free[sg.stage], st));
    }
    for (int s = 0; s < nset && with_k && nL > 0; s++)
        if (!orbo || orbs[s].no)
            if ((rc = PAMD_reduce_splits(d_part + (size_t)s * nsplit * n2, nsplit, nao, nao, d_vk + (size_t)s * n2, nao, orbo ? 1 : 0, st)))
                return rc;
    if (download) {
        // r05: J~ is complete as soon as its last second-pass kernel has run - before the SYRK of the last block ends.  Its unpack
        // and its device -> host copy go to the SIDE stream, under the tail of the K work on `st` (1-2 ms of the host-array call
        // at config 3); K follows on `st`.
        if (with_k) PAMD_CHECK_HIP(hipMemcpyAsync(vk, d_vk, (size_t)nset * n2 * 8, k_out, st));
        if (with_j) {
            if (j_on_st) PAMD_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_j, 0));
            if (nL == 0) {                          // nothing was queued: order the side stream behind the zero fill on `st`
                PAMD_CHECK_HIP(hipEventRecord(h->ev_j, st));
                PAMD_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_j, 0));
            }
            if ((rc = PAMD_unpack_tril(d_vjt, npair, nset, nao, d_vjfull, nao, nao, h->side))) return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(vj, d_vjfull, (size_t)nset * n2 * 8, k_out, h->side));
        }
    } else if (fused && serial_j2 == 0) {
        PAMD_CHECK_HIP(hipEventRecord(h->ev, h->side));
        PAMD_CHECK_HIP(hipStreamWaitEvent(st, h->ev, 0));
    }
    // r06: everything is queued - the tag probe runs now, on this thread, beside the kernels (flags bit 1 without bit 0)
    h->last_mismatch = 0;
    if (fused && (flags & 2) && !(flags & 1)) {
        static const int dbg = [] { const char *e = getenv("PAMD_DF_DEBUG_TIMING"); return (e && e[0] == '1') ? 1 : 0; }();
        const auto tp0 = std::chrono::steady_clock::now();
        h->last_mismatch = host_dm_mismatch(dm, orbo, nocc, nset, nao);
        if (dbg) {
            const double tp = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
            const hipError_t q = hipStreamQuery(st);
            (void)hipGetLastError();
            fprintf(stderr, "PAMD_df_get_jk: tag probe %.2f ms on the host; main stream %s when it ended\n", tp,
                    q == hipSuccess ? "already IDLE" : "still busy");
        }
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(st));
    PAMD_CHECK_HIP(hipStreamSynchronize(h->side));
    PAMD_CHECK_HIP(hipStreamSynchronize(h->copy));
    h->t_e2_ms = h->t_syrk_ms = 0;
    for (size_t i = 0; i + 1 < mark_kind.size(); i++)
        if ((mark_kind[i] == 0 && mark_kind[i + 1] == 1) || (mark_kind[i] == 2 && mark_kind[i + 1] == 3)) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->tev[i], h->tev[i + 1]) == hipSuccess) (mark_kind[i] == 0 ? h->t_e2_ms : h->t_syrk_ms) += ms;
        }
    (void)hipGetLastError();
    return 0;
}

// Schedule of the second J pass on the fused path (DF.j2_policy of the Python layer): which of the two is faster depends on the
// shape (config 3: overlapped, taxol on one GPU: in line or - r05 - inside the SYRK kernel), so a tensor of 4 GB and more gets all
// three timed once per (nset, occupied count) - three extra builds at the first call - and the choice is kept in the handle.
// PAMD_DF_J2 = overlap | serial | fused overrides.
static int shard_get_jk(PAMD_df *h, const double *dm, const double *orbo, const int *nocc, int nset, int nao, int hermi, int with_j,
                        int with_k, int flags, double *vj, double *vk, int download)
{
    const bool fused = with_j && with_k && orbo && (flags & 3) && h->nL > 0;
    int serial = 0;
    if (fused && h->n_res == h->nL) {
        const char *env = getenv("PAMD_DF_J2");
        if (env && (env[0] == 's' || env[0] == 'o' || env[0] == 'f')) {
            serial = env[0] == 's' ? 1 : (env[0] == 'f' ? 2 : 0);
        } else if ((size_t)h->nL * (size_t)h->npair * 8 >= (4ul << 30) && nocc) {
            long key = nset;
            for (int s = 0; s < nset; s++) key = key * 4099 + nocc[s];
            auto it = h->j2_policy.find(key);
            const char *tune = getenv("PAMD_DF_J2_TUNE");
            const int ncand = h->square ? 2 : 3;
            if (it == h->j2_policy.end() && !(tune && tune[0] == 'e')) {
                // r06, the default: no trial builds - the caller's OWN calls are the trials (as df_jk.get_jk_device, DF.j2_tune =
                // 'lazy').  First call of a shape: overlap, untimed (work spaces, page-locked buffers); each following call runs the
                // candidate with the fewest samples under the host clock; after two samples of each the best is kept.  Every
                // schedule returns the same J and K.  PAMD_DF_J2_TUNE=eager: the seven builds inside the first call (bench.py).
                PAMD_df::J2Trial &t = h->j2_trial[key];
                int sched = 0;
                if (t.calls > 0)
                    for (int c = 1; c < ncand; c++) if (t.n[c] < t.n[sched]) sched = c;
                if (t.calls > 0 && t.n[sched] >= 2) {
                    const int best = PAMD_j2_schedule_pick(t.ms, ncand);
                    h->j2_policy.emplace(key, best);
                    h->j2_trial.erase(key);
                    return df_get_jk_impl(h, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, flags, vj, vk, best, download);
                }
                const bool timed = t.calls++ > 0;
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = df_get_jk_impl(h, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, flags, vj, vk, sched, download);
                if (rc == 0 && timed) {
                    PAMD_df::J2Trial &t2 = h->j2_trial[key];
                    t2.ms[sched] = std::min(t2.ms[sched], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                    t2.n[sched]++;
                }
                return rc;
            }
            if (it == h->j2_policy.end()) {
                double ms[3] = {1e30, 1e30, 1e30};
                // overlap (priming, untimed), then overlap, serial, fused into the SYRK (packed rows only) - r06: the BEST of two runs
                // each (a single run now and then carries a 20 ms hiccup and used to pick the slower schedule for good)
                for (int trial = 0; trial < (h->square ? 5 : 7); trial++) {
                    const int sched = trial ? (trial - 1) / 2 : 0;
                    const auto t0 = std::chrono::steady_clock::now();
                    const int rc = df_get_jk_impl(h, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, flags, vj, vk, sched, download);
                    if (rc) return rc;
                    if (trial) ms[sched] = std::min(ms[sched], std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
                }
                const int best = PAMD_j2_schedule_pick(ms, ncand);
                it = h->j2_policy.emplace(key, best).first;
            }
            serial = it->second;
        }
    }
    return df_get_jk_impl(h, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, flags, vj, vk, serial, download);
}

namespace {

// out[i] = sum_p in[p * stride + i]  (fixed order: the multi-device sum is reproducible run to run)
__global__ void sum_parts_kernel(const double *__restrict__ in, size_t stride, int nparts, double *__restrict__ out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double a = 0;
        for (int p = 0; p < nparts; p++) a += in[(size_t)p * stride + i];
        out[i] = a;
    }
}

// lower triangle of a symmetric (nao, nao) matrix <-> packed rows (the K of the MO branch travels packed like J~)
__global__ void pack_lower_kernel(const double *__restrict__ full, int nao, double *__restrict__ tril)
{
    const int p = blockIdx.y;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q <= p; q += gridDim.x * blockDim.x)
        tril[(size_t)p * (p + 1) / 2 + q] = full[(size_t)p * nao + q];
}

// every part contracts its shard on its own PERSISTENT host thread (PartWorker); errors come back with their messages
struct PartResult { int rc = 0; std::string msg; };

template <class F>
int run_parts(PAMD_df *m, F f)
{
    const std::vector<PAMD_df *> &parts = m->parts;
    if (m->workers.size() != parts.size()) {
        for (PartWorker *w : m->workers) delete w;
        m->workers.clear();
        for (size_t p = 0; p < parts.size(); p++) {
            m->workers.push_back(new PartWorker);
            m->workers.back()->start(parts[p]->device);
        }
    }
    for (size_t p = 0; p < parts.size(); p++) m->workers[p]->post([&f, &parts, p]() { return f((int)p, parts[p]); });
    int first = -1;
    for (size_t p = 0; p < parts.size(); p++)
        if (m->workers[p]->wait() && first < 0) first = (int)p;
    if (first >= 0) {
        snprintf(g_errmsg, sizeof(g_errmsg), "device %d (part %d): %s", parts[first]->device, first, m->workers[first]->msg.c_str());
        return m->workers[first]->rc;
    }
    return 0;
}

// Partial [J~ | K] of the parts -> part 0, fixed-order sum there, one download.  K of the MO branch is symmetric and travels
// packed.  Every part PUSHES its message into its slot of the gather buffer on parts[0]'s device from its own host thread and on
// its own stream (multi_push): on a node with one xGMI link per pair of GPUs the ndev - 1 copies run concurrently, each on its own
// link (a gather issued from part 0's single stream would serialise them: 7 x 0.55 ms against a ~14 ms shard at 8 GPUs).
struct MultiMsg {
    int nset = 0, nao = 0, with_j = 0, with_k = 0;
    bool k_symmetric = false;
    size_t nj = 0, nk = 0, len = 0;
    double *gather = nullptr;                   // [nparts][len] on parts[0]'s device
};

int multi_prepare(PAMD_df *m, int nset, int nao, int with_j, int with_k, bool k_symmetric, MultiMsg *mm)
{
    PAMD_df *h0 = m->parts[0];
    const size_t n2 = (size_t)nao * nao;
    mm->nset = nset; mm->nao = nao; mm->with_j = with_j; mm->with_k = with_k; mm->k_symmetric = k_symmetric;
    mm->nj = with_j ? (size_t)nset * h0->npair : 0;
    mm->nk = with_k ? (size_t)nset * (k_symmetric ? (size_t)h0->npair : n2) : 0;
    mm->len = mm->nj + mm->nk;
    PAMD_CHECK_HIP(hipSetDevice(h0->device));
    int rc;
    mm->gather = h0->workspace("gather", m->parts.size() * mm->len, &rc);
    if (rc) return rc;
    PAMD_CHECK_HIP(hipStreamSynchronize(h0->st));         // (a fresh work space is zeroed on h0's stream)
    return 0;
}

// on the part's own thread, after its shard_get_jk: lay out [J~ | K (packed)] and copy it into the part's gather slot
int multi_push(PAMD_df *m, int p, const MultiMsg &mm)
{
    PAMD_df *h = m->parts[p], *h0 = m->parts[0];
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    const size_t n2 = (size_t)mm.nao * mm.nao;
    const long npair = h->npair;
    int rc;
    double *dst = mm.gather + (size_t)p * mm.len;
    const bool same = h->device == h0->device;
    // same device: straight into the slot; else stage the message here and push it in one copy
    double *msg = dst;
    if (!same) {
        msg = h->workspace("msg", mm.len, &rc);
        if (rc) return rc;
    }
    if (mm.with_j) PAMD_CHECK_HIP(hipMemcpyAsync(msg, h->ws["vjtril"].first, mm.nj * 8, hipMemcpyDeviceToDevice, h->st));
    if (mm.with_k) {
        const double *d_vk = h->ws["vk"].first;
        if (mm.k_symmetric) {
            for (int s = 0; s < mm.nset; s++) {
                pack_lower_kernel<<<dim3((mm.nao + 255) / 256, mm.nao), 256, 0, h->st>>>(d_vk + (size_t)s * n2, mm.nao,
                                                                                      msg + mm.nj + (size_t)s * npair);
                PAMD_CHECK_LAUNCH();
            }
        } else {
            PAMD_CHECK_HIP(hipMemcpyAsync(msg + mm.nj, d_vk, mm.nk * 8, hipMemcpyDeviceToDevice, h->st));
        }
    }
    if (!same) {
        if (m->peer_ok) {
            PAMD_CHECK_HIP(hipMemcpyPeerAsync(dst, h0->device, msg, h->device, mm.len * 8, h->st));
        } else {
            std::vector<double> bounce(mm.len);
            PAMD_CHECK_HIP(hipMemcpyAsync(bounce.data(), msg, mm.len * 8, hipMemcpyDeviceToHost, h->st));
            PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
            PAMD_CHECK_HIP(hipSetDevice(h0->device));
            PAMD_CHECK_HIP(hipMemcpy(dst, bounce.data(), mm.len * 8, hipMemcpyHostToDevice));
            PAMD_CHECK_HIP(hipSetDevice(h->device));
        }
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    return 0;
}

int multi_sum_download(PAMD_df *m, const MultiMsg &mm, double *vj, double *vk)
{
    PAMD_df *h0 = m->parts[0];
    const int np = (int)m->parts.size(), nset = mm.nset, nao = mm.nao;
    const long npair = h0->npair;
    const size_t n2 = (size_t)nao * nao;
    int rc;
    PAMD_CHECK_HIP(hipSetDevice(h0->device));
    double *total = h0->workspace("total", mm.len, &rc);
    if (rc) return rc;
    sum_parts_kernel<<<1024, 256, 0, h0->st>>>(mm.gather, mm.len, np, total, mm.len);
    PAMD_CHECK_LAUNCH();
    if (mm.with_j) {
        double *d_vj = h0->workspace("vjfull", (size_t)nset * n2, &rc);
        if (rc) return rc;
        if ((rc = PAMD_unpack_tril(total, npair, nset, nao, d_vj, nao, nao, h0->st))) return rc;
        PAMD_CHECK_HIP(hipMemcpyAsync(vj, d_vj, (size_t)nset * n2 * 8, hipMemcpyDeviceToHost, h0->st));
    }
    if (mm.with_k) {
        if (mm.k_symmetric) {
            double *d_vkf = h0->workspace("vkfull", (size_t)nset * n2, &rc);
            if (rc) return rc;
            if ((rc = PAMD_unpack_tril(total + mm.nj, npair, nset, nao, d_vkf, nao, nao, h0->st))) return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(vk, d_vkf, (size_t)nset * n2 * 8, hipMemcpyDeviceToHost, h0->st));
        } else {
            PAMD_CHECK_HIP(hipMemcpyAsync(vk, total + mm.nj, (size_t)nset * n2 * 8, hipMemcpyDeviceToHost, h0->st));
        }
    }
    PAMD_CHECK_HIP(hipStreamSynchronize(h0->st));
    return 0;
}

}  // namespace

extern "C" {

int PAMD_df_create_ex(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                      const PAMD_df_options *opt, PAMD_df **out)
{
    PAMD_REQUIRE(atm && bas && env && out && opt && natm > 0 && nbas_ao > 0 && nbas_aux > 0 && nenv > 0, "PAMD_df_create: bad arguments");
    *out = nullptr;
    const int ndev = opt->ndev > 0 ? opt->ndev : 1;
    PAMD_REQUIRE(opt->ndev <= 0 || opt->devices, "PAMD_df_create: devices[ndev]");
    PAMD_REQUIRE(ndev <= 64, "PAMD_df_create: at most 64 parts");
    std::vector<int> devs(ndev, 0);
    for (int i = 0; i < ndev && opt->devices; i++) devs[i] = opt->devices[i];
    int ndevice = 0;
    PAMD_CHECK_HIP(hipGetDeviceCount(&ndevice));
    for (int d : devs) PAMD_REQUIRE(d >= 0 && d < ndevice, "PAMD_df_create: device index out of range");
    Tables t;
    int rc;
    if ((rc = make_tables(atm, bas, nbas_ao, nbas_aux, env, &t))) return rc;
    const size_t cap = opt->max_device_bytes > 0 ? (size_t)opt->max_device_bytes : 0;
    // flags bit 2 (r06): reserve_bytes is valid - HBM the caller's calculation wants left free on every device besides the J/K work
    // space (the XC leg's compact AO image and work buffers): part of the layout decision of build_rows
    const size_t reserve = (opt->flags & 4) && opt->reserve_bytes > 0 ? (size_t)opt->reserve_bytes : 0;
    Metric m;
    if (opt->flags & 2) {
        // one rank's shard of a multi-process job: rows of part `part` of `nparts` (DF.shard_range) on devices[0], resident as
        // far as the device / max_device_bytes allows, the rest streamed from page-locked host memory; PAMD_df_get_jk then
        // returns this shard's PARTIAL J/K and the caller sums over the ranks (RCCL all-reduce in pyscf_amd.df.DF)
        PAMD_REQUIRE(opt->nparts > 0 && opt->part >= 0 && opt->part < opt->nparts, "PAMD_df_create: part / nparts");
        PAMD_df *h = nullptr;
        if ((rc = create_shard(t, devs[0], opt->omega, opt->lindep, cap, &m, opt->part, opt->nparts, &h, reserve))) return rc;
        h->partial = opt->nparts > 1;
        *out = h;
        return 0;
    }
    if (opt->ndev <= 0 || (ndev == 1 && !(opt->flags & 1))) {
        // the plain single-device handle
        PAMD_df *h = nullptr;
        if ((rc = create_shard(t, devs[0], opt->omega, opt->lindep, cap, &m, 0, 1, &h, reserve))) return rc;
        *out = h;
        return 0;
    }
    PAMD_df *mh = new PAMD_df;
    struct Guard { PAMD_df *p; ~Guard() { delete p; } } guard{mh};
    mh->device = devs[0];
    mh->omega = opt->omega;
    mh->parts.assign(ndev, nullptr);
    // part 0 factorises the metric on its device; the host copy of M^T is then shared by all parts, which build their row
    // ranges concurrently, one host thread per part (the raw (Q|pq) slabs are generated redundantly: ~0.1 s at config 3)
    if ((rc = create_shard(t, devs[0], opt->omega, opt->lindep, cap, &m, 0, ndev, &mh->parts[0], reserve))) return rc;
    std::vector<PAMD_df *> rest(ndev - 1, nullptr);
    {
        std::vector<PartResult> res(ndev - 1);
        std::vector<std::thread> th;
        for (int p = 1; p < ndev; p++)
            th.emplace_back([&, p]() {
                res[p - 1].rc = create_shard(t, devs[p], opt->omega, opt->lindep, cap, &m, p, ndev, &rest[p - 1], reserve);
                if (res[p - 1].rc) res[p - 1].msg = g_errmsg;
            });
        for (auto &x : th) x.join();
        for (int p = 1; p < ndev; p++) mh->parts[p] = rest[p - 1];
        for (int p = 1; p < ndev; p++)
            if (res[p - 1].rc) {
                snprintf(g_errmsg, sizeof(g_errmsg), "device %d (part %d): %s", devs[p], p, res[p - 1].msg.c_str());
                for (PAMD_df *&q : mh->parts) { if (q) { (void)hipSetDevice(q->device); delete q; q = nullptr; } }
                mh->parts.clear();
                return res[p - 1].rc;
            }
    }
    PAMD_df *h0 = mh->parts[0];
    mh->nao = h0->nao;
    mh->naux = h0->naux;
    mh->npair = h0->npair;
    mh->rows = h0->rows;
    mh->nL = mh->nL_total = h0->nL_total;
    // direct peer copies into part 0 where the devices can reach each other (xGMI); PAMD_DF_PEER=0 forces the host bounce
    mh->peer_ok = 1;
    const char *envp = getenv("PAMD_DF_PEER");
    if (envp && envp[0] == '0') mh->peer_ok = 0;
    PAMD_CHECK_HIP(hipSetDevice(devs[0]));
    for (int p = 1; p < ndev && mh->peer_ok; p++) {
        if (devs[p] == devs[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devs[0], devs[p]) != hipSuccess || !can) { (void)hipGetLastError(); mh->peer_ok = 0; break; }
        hipError_t e = hipDeviceEnablePeerAccess(devs[p], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) mh->peer_ok = 0;
        (void)hipGetLastError();
        // ... and the other direction: the parts push into part 0's gather buffer from their own streams
        int can2 = 0;
        if (hipDeviceCanAccessPeer(&can2, devs[p], devs[0]) != hipSuccess || !can2) { (void)hipGetLastError(); mh->peer_ok = 0; break; }
        PAMD_CHECK_HIP(hipSetDevice(devs[p]));
        e = hipDeviceEnablePeerAccess(devs[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) mh->peer_ok = 0;
        (void)hipGetLastError();
        PAMD_CHECK_HIP(hipSetDevice(devs[0]));
    }
    guard.p = nullptr;
    *out = mh;
    return 0;
}

// The metric factorisation of DF.build as a call of its own (df/incore.py:150-158, :263-270): j2c[naux][naux] (host, symmetric)
// -> m[nrow][naux] (host; caller provides naux x naux doubles) with cderi = m (Q|pq): rows of L^-1 (*tri = 1) or, when the Cholesky
// factorisation meets a non-positive pivot or force_ed is set (decompose_j2c = 'ED', df/grad/rhf.py:423-443), (V / sqrt(w))^T over
// the eigenvalues > lindep (*tri = 0).  The same blocked Cholesky / block forward substitution on this library's FP64-MFMA GEMMs
// that PAMD_df_create uses - pyscf_amd/df/incore.py calls it too, so there is ONE factorisation code path (r04).
int PAMD_metric_decompose(const double *j2c, int naux, double lindep, int force_ed, int device, double *m, int *nrow, int *tri)
{
    PAMD_REQUIRE(j2c && m && nrow && tri && naux > 0, "PAMD_metric_decompose: bad arguments");
    PAMD_CHECK_HIP(hipSetDevice(device));
    PAMD_df h;
    h.device = device;
    PAMD_CHECK_HIP(hipStreamCreate(&h.st));
    int rc;
    double *d_j = nullptr;
    if ((rc = h.pool.alloc((void **)&d_j, (size_t)naux * naux * 8))) return rc;
    PAMD_CHECK_HIP(hipMemcpy(d_j, j2c, (size_t)naux * naux * 8, hipMemcpyHostToDevice));
    std::vector<double> mt;
    int lda = 0;
    if ((rc = decompose_metric(&h, d_j, naux, lindep, &mt, nrow, &lda, tri, force_ed != 0))) return rc;
    for (int r = 0; r < *nrow; r++)
        for (int q = 0; q < naux; q++) m[(size_t)r * naux + q] = mt[(size_t)q * lda + r];
    return 0;
}

int PAMD_df_create(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                   double lindep, int device, PAMD_df **out)
{
    PAMD_df_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.lindep = lindep;
    opt.devices = &device;
    opt.ndev = 1;
    return PAMD_df_create_ex(atm, natm, bas, nbas_ao, nbas_aux, env, nenv, &opt, out);
}

int PAMD_df_create_multi(const int *atm, int natm, const int *bas, int nbas_ao, int nbas_aux, const double *env, int nenv,
                         double lindep, const int *devices, int ndev, PAMD_df **out)
{
    PAMD_REQUIRE(devices && ndev > 0, "PAMD_df_create_multi: devices[ndev]");
    PAMD_df_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.lindep = lindep;
    opt.devices = devices;
    opt.ndev = ndev;
    opt.flags = 1;                          // a one-entry list still goes through the sharded code path
    return PAMD_df_create_ex(atm, natm, bas, nbas_ao, nbas_aux, env, nenv, &opt, out);
}

// A handle over tensor rows the CALLER already has (r05; `DF._cderi = ndarray | 'file.h5'`, pyscf/df/df.py:153-155,214-242): rows
// [nrows][nao_pair] f64 in host memory - a numpy array, or an mmap of the contiguous 'j3c' dataset of PySCF's own HDF5 file.  What
// fits the device (or max_device_bytes) is uploaded once; the remaining rows are streamed block by block under the kernels in every
// PAMD_df_get_jk, exactly like the host rows of an out-of-core PAMD_df_create_ex handle - flags bit 0: straight from the caller's
// memory, which must then outlive the handle (no copy: a 560 GB file needs no 292 GB of page-locked RAM; the region is page-locked
// in place when the driver allows, else it travels as pageable memory); without the flag the rows are copied into page-locked
// memory of the handle.  No integrals, no metric: PAMD_df_naux = nrows.
int PAMD_df_create_from_rows(const double *rows, int nrows, int nao, int device, long long max_device_bytes, int flags, PAMD_df **out)
{
    PAMD_REQUIRE(rows && out && nrows > 0 && nao > 0, "PAMD_df_create_from_rows: bad arguments");
    *out = nullptr;
    int ndevice = 0;
    PAMD_CHECK_HIP(hipGetDeviceCount(&ndevice));
    PAMD_REQUIRE(device >= 0 && device < ndevice, "PAMD_df_create_from_rows: device index out of range");
    PAMD_CHECK_HIP(hipSetDevice(device));
    std::lock_guard<std::mutex> lock(g_dev_mutex[device & 63]);
    PAMD_df *h = new PAMD_df;
    struct Guard { PAMD_df *p; ~Guard() { delete p; } } guard{h};
    h->device = device;
    h->nao = nao;
    h->naux = nrows;
    h->npair = (long)nao * (nao + 1) / 2;
    h->rows = (int)round_up(nao, 16);
    h->nL = h->nL_total = nrows;
    int rc;
    if ((rc = init_streams(h))) return rc;
    const long npair = h->npair;
    const size_t row_b = (size_t)npair * 8, tensor_b = (size_t)nrows * row_b;
    size_t free_b = 0, total_b = 0;
    PAMD_CHECK_HIP(hipMemGetInfo(&free_b, &total_b));
    size_t cap_in = max_device_bytes > 0 ? (size_t)max_device_bytes : 0;
    const char *envcap = getenv("PAMD_DF_DEVICE_BYTES");
    if (envcap && atof(envcap) > 0) cap_in = (size_t)atof(envcap);
    const size_t cap = cap_in ? std::min(free_b, cap_in) : free_b;
    const size_t margin = std::min<size_t>(1ul << 30, cap / 16);
    size_t stage_b = 0;
    if (tensor_b + margin <= cap) {
        h->n_res = nrows;
    } else {
        stage_b = std::min<size_t>(4ul << 30, cap / 8);
        const size_t work_b = std::min<size_t>(20ul << 30, cap / 4);
        const size_t used = 2 * stage_b + work_b + margin + std::min<size_t>(4ul << 30, cap / 16);
        h->stage_rows = (int)std::min<size_t>(stage_b / row_b, (size_t)nrows);
        if (h->stage_rows < 1) {
            snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create_from_rows: %.3f GB of device memory cannot stage one tensor row (%.3f GB)",
                     cap * 1e-9, row_b * 1e-9);
            return -2;
        }
        h->n_res = cap > used ? (int)std::min<size_t>((cap - used) / row_b, (size_t)nrows) : 0;
        for (int k = 0; k < 2; k++)
            if ((rc = h->pool.alloc((void **)&h->d_stage[k], (size_t)h->stage_rows * row_b + 256 * 8))) return rc;
    }
    if (h->n_res) {
        if ((rc = h->pool.alloc((void **)&h->d_cderi, (size_t)h->n_res * row_b + 256 * 8))) return rc;
        const int step = (int)std::max<size_t>(1, (1ul << 30) / row_b);
        for (int r0 = 0; r0 < h->n_res; r0 += step) {
            const int nb = std::min(step, h->n_res - r0);
            PAMD_CHECK_HIP(hipMemcpy(h->d_cderi + (size_t)r0 * npair, rows + (size_t)r0 * npair, (size_t)nb * row_b, hipMemcpyHostToDevice));
        }
    }
    if (h->n_res < nrows) {
        const size_t host_b = (size_t)(nrows - h->n_res) * row_b;
        const double *src = rows + (size_t)h->n_res * npair;
        if (flags & 1) {
            h->h_cderi = const_cast<double *>(src);
            h->h_borrowed = 1;
            if (hipHostRegister((void *)h->h_cderi, host_b, hipHostRegisterDefault) == hipSuccess) h->h_registered = 1;
            (void)hipGetLastError();              // not registrable (a read-only file mapping, a locked-memory limit): pageable copies
        } else {
            const size_t left = container_memory_left();
            if (left != ~(size_t)0 && host_b + (12ul << 30) > left) {
                snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create_from_rows: a page-locked copy of %.1f GB exceeds what the container's "
                         "memory limit leaves (%.1f GB): flags bit 0 streams from the caller's array instead", host_b * 1e-9, left * 1e-9);
                return -3;
            }
            if (hipHostMalloc((void **)&h->h_cderi, host_b, hipHostMallocDefault) != hipSuccess) {
                (void)hipGetLastError();
                h->h_cderi = nullptr;
                snprintf(g_errmsg, sizeof(g_errmsg), "PAMD_df_create_from_rows: %.1f GB of page-locked host memory could not be allocated "
                         "(flags bit 0 streams from the caller's array instead)", host_b * 1e-9);
                return -2;
            }
            memcpy(h->h_cderi, src, host_b);
        }
    }
    const size_t held = (size_t)h->n_res * row_b + 2 * stage_b;
    const size_t cap_left = cap_in ? (cap > held ? cap - held : 0) : ~(size_t)0;
    if ((rc = build_square_image(h, cap_left, 0))) return rc;
    if ((rc = build_diag_image(h, cap_left))) return rc;
    PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
    guard.p = nullptr;
    *out = h;
    return 0;
}

// Page-locked host memory for callers that want their result / input arrays copied at the PCIe rate (a pageable 8 nao^2-byte
// array moves at ~10 GB/s and faults its pages in on first touch; page-locked memory at ~50 GB/s): NativeDF / NativeNumInt hand
// out J, K and vxc in arrays made of it.  Portable: every device of the process sees it as pinned.
int PAMD_host_alloc(long long nbytes, void **out)
{
    PAMD_REQUIRE(out, "PAMD_host_alloc: null output pointer");
    *out = nullptr;
    PAMD_REQUIRE(nbytes >= 0, "PAMD_host_alloc: negative size");
    PAMD_CHECK_HIP(hipHostMalloc(out, nbytes > 8 ? (size_t)nbytes : 8, hipHostMallocPortable));
    return 0;
}

int PAMD_host_free(void *p)
{
    if (p) PAMD_CHECK_HIP(hipHostFree(p));
    return 0;
}

void PAMD_df_destroy(PAMD_df *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    delete h;
}

int PAMD_df_naux(const PAMD_df *h, int *naux)
{
    PAMD_REQUIRE(h && naux, "null handle");
    *naux = h->nL;
    return 0;
}

int PAMD_df_nao(const PAMD_df *h, int *nao)
{
    PAMD_REQUIRE(h && nao, "null handle");
    *nao = h->nao;
    return 0;
}

// layout[0] parts, [1] rows resident in HBM (all parts), [2] rows in page-locked host memory, [3] rows with a square image,
// [4] direct peer copies between the parts (0 / 1); part_rows (nullable) [nparts] rows per part
int PAMD_df_layout(const PAMD_df *h, long *layout, int *part_rows)
{
    PAMD_REQUIRE(h && layout, "null handle");
    std::vector<const PAMD_df *> ps;
    if (h->parts.empty()) ps.push_back(h);
    for (const PAMD_df *p : h->parts) ps.push_back(p);
    layout[0] = (long)ps.size();
    layout[1] = layout[2] = layout[3] = 0;
    layout[4] = h->parts.empty() ? 0 : h->peer_ok;
    for (size_t i = 0; i < ps.size(); i++) {
        layout[1] += ps[i]->n_res;
        layout[2] += ps[i]->nL - ps[i]->n_res;
        layout[3] += ps[i]->d_sq ? ps[i]->nL : 0;
        if (part_rows) part_rows[i] = ps[i]->nL;
    }
    return 0;
}

int PAMD_df_tensor_layout(const PAMD_df *h)
{
    if (!h) return 0;
    if (h->parts.empty()) return h->square;
    for (const PAMD_df *p : h->parts) if (!p->square) return 0;
    return 1;
}

static int shard_export(PAMD_df *h, int l0, int l1, double *out)      // shard-local rows
{
    PAMD_CHECK_HIP(hipSetDevice(h->device));
    if (h->square) {
        // the reference's packed rows (pyscf/df/df.py:59-72) out of the square layout: packed on the device block by block
        const int blk = (int)std::max<size_t>(1, (1ul << 30) / ((size_t)h->npair * 8));
        int rc;
        double *d_tmp = h->workspace("export", (size_t)std::min(blk, std::max(l1 - l0, 1)) * h->npair, &rc);
        if (rc) return rc;
        for (int b0 = l0; b0 < l1; b0 += blk) {
            const int nb = std::min(blk, l1 - b0);
            if ((rc = PAMD_pack_tril_rows(h->d_sq + (size_t)b0 * h->sq_ls(), h->sq_ls(), h->rows, h->nao, nb, d_tmp, h->st)))
                return rc;
            PAMD_CHECK_HIP(hipMemcpyAsync(out + (size_t)(b0 - l0) * h->npair, d_tmp, (size_t)nb * h->npair * 8, hipMemcpyDeviceToHost, h->st));
            PAMD_CHECK_HIP(hipStreamSynchronize(h->st));
        }
        return 0;
    }
    const int r1 = std::min(l1, h->n_res);
    if (l0 < r1)
        PAMD_CHECK_HIP(hipMemcpy(out, h->d_cderi + (size_t)l0 * h->npair, (size_t)(r1 - l0) * h->npair * 8, hipMemcpyDeviceToHost));
    const int a = std::max(l0, h->n_res);
    if (a < l1)
        memcpy(out + (size_t)(a - l0) * h->npair, h->h_cderi + (size_t)(a - h->n_res) * h->npair, (size_t)(l1 - a) * h->npair * 8);
    return 0;
}

int PAMD_df_export_cderi(PAMD_df *h, int l0, int l1, double *out)
{
    PAMD_REQUIRE(h && out && 0 <= l0 && l0 <= l1 && l1 <= h->nL, "PAMD_df_export_cderi: bad row range");
    if (h->parts.empty()) return shard_export(h, l0, l1, out);
    for (PAMD_df *p : h->parts) {
        const int a = std::max(l0, p->l0), b = std::min(l1, p->l0 + p->nL);
        if (a < b) {
            const int rc = shard_export(p, a - p->l0, b - p->l0, out + (size_t)(a - l0) * h->npair);
            if (rc) return rc;
        }
    }
    return 0;
}

int PAMD_df_get_jk(PAMD_df *h, const double *dm, const double *orbo, const int *nocc, int nset, int nao, int hermi, int with_j,
                   int with_k, int flags, double *vj, double *vk)
{
    PAMD_REQUIRE(h, "PAMD_df_get_jk: null handle");
    PAMD_REQUIRE((!with_j || vj) && (!with_k || vk) && (with_j || with_k), "PAMD_df_get_jk: output pointers");
    typedef std::chrono::steady_clock clk;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    if (h->parts.empty() || h->parts.size() == 1) {
        // one shard: no gather, no sum - straight into the caller's arrays (r05: a one-entry device list used to pay the pack /
        // push / sum / unpack round trip and a thread start per call: 122.7 against 110.5 ms at config 3)
        PAMD_df *p = h->parts.empty() ? h : h->parts[0];
        const auto t0 = clk::now();
        int r = shard_get_jk(p, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, flags, vj, vk, 1);
        const double mis = p->last_mismatch;
        if (!r && with_j && with_k && orbo && (flags & 2) && !(flags & 1) && mis > 1e-10)
            // the tag did not describe the matrix: J from the matrix itself (the K of the MO branch follows the tag, as in the
            // reference, df_jk.py:340) - the rare path pays a second, J-only call
            r = shard_get_jk(p, dm, nullptr, nullptr, nset, nao, hermi, 1, 0, 0, vj, nullptr, 1);
        p->last_mismatch = mis;
        p->t_compute_ms = ms_since(t0);
        p->t_push_ms = 0;
        p->push_bytes = 0;
        h->t_sum_ms = 0;
        h->last_mismatch = p->last_mismatch;
        return r;
    }
    PAMD_REQUIRE(dm && nset > 0 && nao == h->nao, "PAMD_df_get_jk: bad arguments (nao must equal the handle's)");
    PAMD_REQUIRE(!(flags & 8), "PAMD_df_get_jk: device pointers (flags bit 3) need a one-part handle");
    // one host thread per device contracts that device's shard (the serial decomposition this replaces: df_jk.py:362-381);
    // the partial [J~ | K] are summed on part 0's device and leave in one download
    MultiMsg mm;
    int rc = multi_prepare(h, nset, nao, with_j, with_k, orbo != nullptr, &mm);
    if (rc) return rc;
    rc = run_parts(h, [&](int ip, PAMD_df *p) -> int {
        const auto t0 = clk::now();
        // (the tag probe of flags bit 1 runs once, on part 0's thread; the other parts take the tag as promised)
        const int fl = (ip == 0 || !(flags & 2)) ? flags : ((flags & ~2) | 1);
        const int r = shard_get_jk(p, dm, orbo, nocc, nset, nao, hermi, with_j, with_k, fl, nullptr, nullptr, 0);
        p->t_compute_ms = ms_since(t0);
        if (r) return r;
        const auto t1 = clk::now();
        const int r2 = multi_push(h, ip, mm);
        p->t_push_ms = ms_since(t1);
        p->push_bytes = p->device == h->parts[0]->device ? 0 : mm.len * 8;
        return r2;
    });
    if (rc) return rc;
    const auto t2 = clk::now();
    rc = multi_sum_download(h, mm, vj, vk);
    h->t_sum_ms = ms_since(t2);
    const double mis = h->parts[0]->last_mismatch;
    if (!rc && with_j && with_k && orbo && (flags & 2) && !(flags & 1) && mis > 1e-10)
        rc = PAMD_df_get_jk(h, dm, nullptr, nullptr, nset, nao, hermi, 1, 0, 0, vj, nullptr);     // J from the matrix (see the one-part path)
    h->last_mismatch = mis;
    return rc;
}

// max_s |D_s v - C_s (C_s^T v)| / max(1, |D_s v|) found by the tag probe of the last PAMD_df_get_jk (flags bit 1); 0 when none ran
double PAMD_df_last_mismatch(const PAMD_df *h) { return h ? h->last_mismatch : 0.0; }

// max_s |D_s v - C_s (C_s^T v)| / max(1, |D_s v|) on HOST arrays for one fixed pseudo-random vector: the tag probe of BOTH host layers
// (r06: pyscf_amd.lib.dm_orbital_mismatch calls this; PAMD_df_get_jk runs the same loops beside its queued kernels, flags bit 1).
// dm [nset][nao][nao]; orbo: the nset blocks [nao][nocc[s]] one after the other.
int PAMD_dm_orbital_mismatch(const double *dm, const double *orbo, const int *nocc, int nset, int nao, double *out)
{
    PAMD_REQUIRE(dm && orbo && nocc && out && nset > 0 && nao > 0, "PAMD_dm_orbital_mismatch: arguments");
    *out = host_dm_mismatch(dm, orbo, nocc, nset, nao);
    return 0;
}

// Timings of the LAST PAMD_df_get_jk on this handle (bench.py --single-process `roofline` / `comm`): out[0] = parts, out[1] = host ms
// of the fixed-order sum + unpack + download on part 0, out[2] = 1 when the partial results travelled by direct peer copies (xGMI),
// 0 for the host bounce or a single device; then per part p: out[3 + 5p] = host ms of the shard's contraction (kernels +
// synchronisation), out[4 + 5p] = host ms of laying out and pushing [J~ | K] into part 0's gather buffer, out[5 + 5p] = bytes that
// crossed devices, out[6 + 5p] / out[7 + 5p] = HIP-event ms of the half-transform / SYRK launches (MO branch) on their stream.
int PAMD_df_last_timing(const PAMD_df *h, double *out, int nout)
{
    PAMD_REQUIRE(h && out, "PAMD_df_last_timing: null argument");
    std::vector<const PAMD_df *> ps;
    if (h->parts.empty()) ps.push_back(h);
    for (const PAMD_df *p : h->parts) ps.push_back(p);
    PAMD_REQUIRE(nout >= 3 + 5 * (int)ps.size(), "PAMD_df_last_timing: out[3 + 5 * parts]");
    bool cross = false;
    for (const PAMD_df *p : ps) cross = cross || p->device != ps[0]->device;
    out[0] = (double)ps.size();
    out[1] = h->t_sum_ms;
    out[2] = (cross && h->peer_ok) ? 1.0 : 0.0;
    for (size_t i = 0; i < ps.size(); i++) {
        out[3 + 5 * i] = ps[i]->t_compute_ms;
        out[4 + 5 * i] = ps[i]->t_push_ms;
        out[5 + 5 * i] = (double)ps[i]->push_bytes;
        out[6 + 5 * i] = ps[i]->t_e2_ms;
        out[7 + 5 * i] = ps[i]->t_syrk_ms;
    }
    return 0;
}

// {first global row of this handle's rows, rows it holds, rows of the whole tensor, 1 when PAMD_df_get_jk returns PARTIAL sums}
int PAMD_df_shard_info(const PAMD_df *h, int *info)
{
    PAMD_REQUIRE(h && info, "PAMD_df_shard_info: null argument");
    info[0] = h->l0;
    info[1] = h->nL;
    info[2] = h->nL_total ? h->nL_total : h->nL;
    info[3] = h->partial;
    return 0;
}

}  // extern "C"
