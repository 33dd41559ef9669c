// ttx_engine.hip -- host orchestration + C-ABI (include/ttx.h) of the MI355X TT-cross engine.
//
// One engine = one dtt_dmrgg problem (reference lib/dmrgg.f90:11-1050).  The sweep is a stream-ordered
// chain of kernels with NO host round trip inside a sweep: the data-dependent control flow of the rook
// loop (lib/dmrgg.f90:516-582) lives in device-side step states that each kernel resolves from the
// previous kernel's partial arg-max records.  The host synchronises once per sweep to read the
// reference's per-sweep report line and to apply the stop rule (lib/dmrgg.f90:1010-1019).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include <atomic>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <rccl/rccl.h>   // types and enums only: the library is dlopen()ed by ttx_comm_init

#include "../../include/ttx.h"
#include "ttx_kernels.h"
#include "ttx_de.h"
#include "ttx_mvn.h"
#include "ttx_ttops.h"
#include "ttx_fused.h"
#include "ttx_cluster.h"

static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(TTX_EHIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" const char *ttx_last_error(void) { return g_err.c_str(); }
extern "C" int ttx_version(void) { return 1; }


// RCCL entry points, resolved at run time (single-GPU users never load librccl)
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load()
{
    if (g_rccl.lib) return TTX_OK;
    void *L = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!L) L = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!L) return fail(TTX_EHIP, "cannot load librccl.so: %s", dlerror());
#define SYM_(f, name) *(void **)(&g_rccl.f) = dlsym(L, name); if (!g_rccl.f) return fail(TTX_EHIP, "librccl.so lacks %s", name);
    SYM_(GetUniqueId, "ncclGetUniqueId") SYM_(CommInitRank, "ncclCommInitRank") SYM_(CommDestroy, "ncclCommDestroy")
    SYM_(Send, "ncclSend") SYM_(Recv, "ncclRecv") SYM_(AllReduce, "ncclAllReduce") SYM_(GroupStart, "ncclGroupStart")
    SYM_(GroupEnd, "ncclGroupEnd") SYM_(GetErrorString, "ncclGetErrorString")
#undef SYM_
    g_rccl.lib = L;
    return TTX_OK;
}
#define NCCLCHECK(x) do { ncclResult_t e_ = (x); if (e_ != ncclSuccess) return fail(TTX_EHIP, "%s failed: %s", #x, g_rccl.GetErrorString(e_)); } while (0)

struct ttx_engine {
    ttx_config cfg;
    std::vector<int32_t> n1;            // 1-based n, size d+2
    std::vector<double> par, aux, quadw;
    std::vector<int32_t> own;           // own[0..nproc]
    int d = 0, RM = 0, NM = 0, G = 0, NC = 0, nbmax = 0, H = 0, mode = 0;
    int g0 = 0;                         // first global group held by this process
    int W = 1, wrank = 0;               // processes (GPUs) of the job, my index
    DevProb P{};
    hipStream_t stream = nullptr;
    std::vector<void *> allocs;
    size_t SB = 0, QB = 0;              // doubles in the summary / quadrature gather buffers
    double *h_sum = nullptr;            // pinned [SB]
    char *h_msg = nullptr;              // pinned 4*MSZ (host-callback transport staging)
    double *h_tmp = nullptr;            // pinned max(QB, SB)
    char *recvL = nullptr, *recvR = nullptr;   // device receive buffers for remote neighbours
    // transports between GPUs: RCCL (device buffers, stream-ordered) or host callbacks (staged through pinned memory)
    ncclComm_t comm = nullptr;
    ttx_transport cb{};
    bool have_cb = false;
    int cb_error = 0;                   // set by a failed host-function transfer (checked at the next host synchronisation)
    // staging of host-transport all-reduces: a ring of pinned slots, one per reduction in flight
    struct RedJob { ttx_engine *h; double *buf; size_t count; int op; };
    static const int NRED = 16;
    RedJob red[NRED];
    double *red_mem = nullptr; size_t red_cap = 0; int red_next = 0;
    RedJob *red_slot(size_t count, int op)
    {
        if (!red_mem || count > red_cap) return nullptr;
        RedJob *j = &red[red_next];
        j->h = this; j->buf = red_mem + (size_t)red_next * red_cap; j->count = count; j->op = op;
        red_next = (red_next + 1) % NRED;
        return j;
    }
    struct ShmTransport *shm = nullptr; // built-in node-local transport (ttx_comm_init_shm)
    std::vector<ttx_sweep_rec> recs;
    std::vector<int32_t> tapes;         // [nsweeps-1][d+1][4]
    std::vector<int32_t> rfinal;
    std::vector<std::vector<uint8_t>> updhist;
    int64_t neval = 0;
    double seconds = 0.0;
    bool ran = false;
    // profiling
    bool profile = false;
    struct Ev { int kind, n; hipEvent_t a, b; };
    std::vector<Ev> evs;
    std::vector<hipEvent_t> evpool;
    int64_t k_launches[TTX_K_NKINDS] = {0};
    double k_ms[TTX_K_NKINDS] = {0}, k_bytes[TTX_K_NKINDS] = {0};
    // tt_lib utilities: compact work buffers, allocated on first use
    double *Wa = nullptr, *Wb = nullptr, *Wc = nullptr, *Wd = nullptr, *Sm = nullptr, *bak = nullptr;
    int *Si = nullptr;
    size_t lds_half = 0, lds_lot = 0, lds_par = 0;
    int half_vals = 0, lot_vals = 0;
    int de_v2 = 0;                      // Ising D/E: wave-per-pivot half-step kernel k_halfstep_de (ttx_de.h)
    int de_v5 = 0; size_t lds_de5 = 0;  // ... as a relay of four waves (k_halfstep_de5): the default where it fits
    int de5_fallbacks = 0;
    int de_slots = 0; size_t lds_de = 0;
    int de_team = 0, de_team_units = 256, det_fallbacks = 0; size_t lds_det = 0;
    int de_test_fault = 0;              // test hook (TTX_DE_TEST_FAULT=<sweep>): the team half-steps of that sweep get a grid of one unit
    int de_team6_units = 1024; size_t lds_det6 = 0;                                      // ... and by teams of 6 waves, several per CU, for the launches above that   // ... by a team of 14 waves per unit (k_halfstep_det) while the ranks are small
    int lot_rows = 0; size_t lds_der = 0;   // ... four candidates per wave, one per DPP row (k_lottery_eval_de_rows)
    double *h_svd = nullptr; double svd_seq = 0.0;   // pinned: [seq, rank, sweeps, sv...] of the core dtt_svd works on (k_svd_report)
    int de_lot_point = 0;                   // Ising D/E, unit-cut path: lottery candidates row-parallel without tables (k_lottery_eval_decp)
    int lot_wave = 0;                       // Ising D/E: lottery candidates and boundary corners by the row-wise wave evaluator (ttx_de.h)
    int mvn_v2 = 0; size_t lds_mvn = 0;     // mvn: wave-per-pivot half-step and wave-per-candidate lottery (ttx_mvn.h)
    int fast_cap = 0;                       // TTX_ARITH=fast: rows of each decay table the lottery kernel keeps in LDS
    bool want_fast = false;                 // fast arithmetic was asked for (ttx_config.arith / TTX_ARITH); P.arith says where it is effective
    int fused = 0;                      // whole-sweep kernel (ttx_fused.h) usable for this problem
    size_t lds_fused = 0;
    hipStream_t qstream = nullptr;      // forked per-sweep quadrature (single-process runs)
    hipEvent_t ev_sum[2] = {nullptr, nullptr}, ev_val[2] = {nullptr, nullptr};
    double *h_sum_base = nullptr;       // pinned [2][SB]: summaries of the two sweeps in flight
    double *h_val = nullptr;            // pinned [2]: per-sweep quadrature values
    int cluster_zkeep = 0;              // cluster kernel keeps the sorted pivot lists of all own bonds in LDS
    int cluster_ldsinv = 0;             // cluster kernel keeps the neighbour LU factors in LDS
    int cluster = 0;                    // workgroups per bond group of the cluster sweep kernel (ttx_cluster.h); 0: not used
    size_t lds_cluster = 0;
    int *h_abort = nullptr;             // pinned, device-visible: the cluster kernel's barrier-timeout flag
    int cluster_coop = 0;               // launch the cluster kernel with hipLaunchCooperativeKernel
    int cluster_fallbacks = 0;          // runs replayed on the chain path after a cluster abort
    bool cluster_aborted = false;
    // user integrand evaluated on the host (TTX_FUN_HOST): see DevProb::hostpass
    ttx_host_fun hfun = nullptr;
    const double *hfun_par = nullptr;   // the caller's par(*), passed through untouched
    size_t HS = 0;
    int64_t host_calls = 0;
    int64_t n_resid = 0;                // rook half-steps of the last run that took a residual (all groups)
};

// ---- worker threads for the host integrand (the reference evaluates `fun` inside !$OMP PARALLEL DO regions,
//      lib/dmrgg.f90:169,222,455,520,553; `fun` must be thread-safe there and here) ----------------------------
namespace {
class HostPool {
  public:
    static HostPool &get() { static HostPool p; return p; }
    int threads() const { return (int)workers.size() + 1; }
    // fn(i) for i in [0, n): the calling thread takes part; returns when all are done.  The pool is one per process:
    // engines driven from different host threads take turns on it, one batch at a time (`turn`).
    void run(size_t n, const std::function<void(size_t)> &fn)
    {
        if (n == 0) return;
        if (workers.empty() || n < 32) { for (size_t i = 0; i < n; i++) fn(i); return; }
        std::lock_guard<std::mutex> one_batch(turn);
        {
            std::lock_guard<std::mutex> lk(mu);
            job = &fn; total = n; next = 0; pending = workers.size(); gen++;
        }
        cv.notify_all();
        work();
        std::unique_lock<std::mutex> lk(mu);
        done.wait(lk, [&] { return pending == 0; });
        job = nullptr;
    }
  private:
    HostPool()
    {
        int nt = 0;
        if (const char *e = getenv("TTX_HOST_THREADS")) nt = atoi(e);
        else if (const char *e2 = getenv("OMP_NUM_THREADS")) nt = atoi(e2);
        if (nt <= 0) nt = (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
        for (int t = 1; t < nt; t++) workers.emplace_back([this] { loop(); });
    }
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu); quit = true; gen++; }
        cv.notify_all();
        for (auto &w : workers) w.join();
    }
    void work()
    {
        for (;;) {
            size_t lo, hi;
            {
                std::lock_guard<std::mutex> lk(mu);
                if (next >= total) return;
                lo = next; hi = std::min(total, lo + std::max<size_t>(1, total / (8 * (workers.size() + 1)))); next = hi;
            }
            for (size_t i = lo; i < hi; i++) (*job)(i);
        }
    }
    void loop()
    {
        unsigned long long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return gen != seen; });
                seen = gen;
                if (quit) return;
            }
            work();
            { std::lock_guard<std::mutex> lk(mu); if (--pending == 0) done.notify_all(); }
        }
    }
    std::vector<std::thread> workers;
    std::mutex mu, turn;
    std::condition_variable cv, done;
    const std::function<void(size_t)> *job = nullptr;
    size_t total = 0, next = 0, pending = 0;
    unsigned long long gen = 0;
    bool quit = false;
};
}

// after the index pass of a kernel: wait for it, call the user's function for every requested slot, clear the requests
static int host_eval(ttx_engine *h)
{
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipGetLastError());
    DevProb &P = h->P;
    const size_t nslot = (size_t)h->G * h->HS;
    std::vector<uint32_t> todo;
    for (size_t s = 0; s < nslot; s++) if (P.hreq[s]) { todo.push_back((uint32_t)s); P.hreq[s] = 0; }
    const int32_t d = h->d;
    const int32_t *nn = h->n1.data() + 1;
    HostPool::get().run(todo.size(), [&](size_t i) {
        int32_t ind[2048];
        const short *row = P.hidx + (size_t)todo[i] * d;
        for (int k = 0; k < d; k++) ind[k] = row[k];
        P.hval[todo[i]] = h->hfun(&d, ind, nn, h->hfun_par);
    });
    h->host_calls += (int64_t)todo.size();
    return TTX_OK;
}

template <class T>
static int dev_alloc(ttx_engine *h, T **p, size_t count)
{
    void *q = nullptr;
    HIPCHECK(hipMalloc(&q, count * sizeof(T) + 64));
    HIPCHECK(hipMemset(q, 0, count * sizeof(T) + 64));
    // the fill runs on the null stream, which is NOT ordered against the engine's non-blocking stream: finish it
    // here, or a buffer allocated on first use could be zeroed after the first kernel has written it
    HIPCHECK(hipDeviceSynchronize());
    h->allocs.push_back(q);
    *p = (T *)q;
    return TTX_OK;
}

// lib/default.f90:78-97 share(): own(p) = first + int(dble(last-first+1)*dble(p)/nproc)
static void share(int first, int last, int nproc, std::vector<int32_t> &own)
{
    own.assign(nproc + 1, 0);
    own[0] = first;
    for (int p = 1; p < nproc; p++) own[p] = first + (int)((double)(last - first + 1) * (double)p / nproc);
    own[nproc] = last + 1;
}

static double powi(double a, int b)
{
    double r = 1.0;
    for (;;) { if (b & 1) r *= a; b /= 2; if (b == 0) break; a *= a; }
    return r;
}

static int ensure_lds(const void *fn, size_t need, size_t &cur);

// nofun: an engine that only holds a tensor train (ttx_from_tt / ttx_read): no integrand, ttx_run refused
static int create_impl(ttx_engine **out, const ttx_config *cfg, bool nofun)
{
    if (!out || !cfg) return fail(TTX_EINVAL, "ttx_create: null argument");
    *out = nullptr;
    if (!cfg->n) return fail(TTX_EINVAL, "ttx_create: mode sizes missing");
    if (cfg->d < 2) return fail(TTX_EINVAL, "dtt_dmrgg: l,m: 1 %d", cfg->d);
    if (cfg->maxrank < 1 || cfg->maxrank > 128) return fail(TTX_EINVAL, "ttx_create: maxrank must be in 1..128 (got %d)", cfg->maxrank);
    if (cfg->pivoting < -1) return fail(TTX_EINVAL, "dtt_dmrgg: unknown pivoting: %d", cfg->pivoting);   // lib/dmrgg.f90:590-592
    if (2 * cfg->pivoting + 2 > TTX_MAXH) return fail(TTX_EINVAL, "dtt_dmrgg: pivoting %d too large", cfg->pivoting);
    if (!(nofun && cfg->fun_id == 0) && (cfg->fun_id < 1 || cfg->fun_id > 4)) return fail(TTX_EINVAL, "ttx_create: unknown fun_id %d", cfg->fun_id);
    if (cfg->npar < 0 || (cfg->npar > 0 && !cfg->par)) return fail(TTX_EINVAL, "ttx_create: par missing");
    if (cfg->fun_id == TTX_FUN_ISING && (cfg->npar < 2 * cfg->n[0] + 1)) return fail(TTX_EINVAL, "ttx_create: the Ising integrand needs par(1:2n+1) (nodes, weights, id)");
    if ((cfg->fun_id == TTX_FUN_STDNORM || cfg->fun_id == TTX_FUN_MVN) && cfg->npar < cfg->n[0]) return fail(TTX_EINVAL, "ttx_create: the integrand needs the nodes par(1:n)");
    // the built-in integrands address par(ind) (and the Ising weights par(n(1) + ind)): no mode may be larger than the first
    // (test_crs_ising.f90:181-183); with the reference this is the caller's business, here it would be a read outside the parameter vector
    if (cfg->fun_id != TTX_FUN_HOST && cfg->fun_id != 0)
        for (int k = 1; k < cfg->d; k++)
            if (cfg->n[k] > cfg->n[0]) return fail(TTX_EINVAL, "ttx_create: mode %d has %d points, more than the first mode (%d): the built-in integrands index par by n(1)", k + 1, cfg->n[k], cfg->n[0]);
    if (cfg->fun_id == TTX_FUN_HOST && cfg->d > 2048) return fail(TTX_EINVAL, "ttx_create: host integrand: at most 2048 dimensions (tt_size)");
    const int W = cfg->world_size < 1 ? 1 : cfg->world_size;
    const int nproc = std::max(cfg->nproc < 1 ? 1 : cfg->nproc, 1);
    if (nproc >= cfg->d) return fail(TTX_EINVAL, "nproc exceeds or equal dimension, cannot proceed");   // lib/dmrgg.f90:114-117
    if (nproc < W) return fail(TTX_EINVAL, "ttx_create: %d bond groups cannot be spread over %d GPUs", nproc, W);
    if (nproc > 256) return fail(TTX_EINVAL, "ttx_create: at most 256 bond groups");
    if (cfg->world_rank < 0 || cfg->world_rank >= W) return fail(TTX_EINVAL, "ttx_create: world_rank out of range");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "ttx_create: no HIP device (the engine has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(TTX_ENODEV, "ttx_create: device %d not present", cfg->device);
    HIPCHECK(hipSetDevice(cfg->device));

    ttx_engine *h = new ttx_engine();
    h->cfg = *cfg;
    h->cfg.nproc = nproc;
    h->W = W; h->wrank = cfg->world_rank;
    const int d = cfg->d;
    h->d = d; h->RM = cfg->maxrank;
    h->n1.assign(d + 2, 1);
    int NM = 1;
    for (int k = 1; k <= d; k++) { h->n1[k] = cfg->n[k - 1]; if (cfg->n[k - 1] < 1 || cfg->n[k - 1] > 32000) { delete h; return fail(TTX_EINVAL, "bad mode size"); } NM = std::max(NM, cfg->n[k - 1]); }
    h->NM = NM;
    if ((long long)h->RM * NM > (long long)TTX_MAXPART * TTX_BLK) { delete h; return fail(TTX_EINVAL, "maxrank*n too large"); }
    if (cfg->npar > 0) h->par.assign(cfg->par, cfg->par + cfg->npar);
    if (cfg->aux && cfg->naux > 0) h->aux.assign(cfg->aux, cfg->aux + cfg->naux);
    if (cfg->mybonds) h->own.assign(cfg->mybonds, cfg->mybonds + nproc + 1);
    else share(1, d - 1, nproc, h->own);                               // lib/dmrgg.f90:126-130
    for (int g = 0; g < nproc; g++) if (h->own[g + 1] <= h->own[g]) { delete h; return fail(TTX_EINVAL, "mybonds: empty group %d", g); }
    // the groups must tile the bonds 1 .. d-1 (own(0) = 1, own(nproc) = d, lib/default.f90:78-97): anything else would address
    // cores that do not exist or leave bonds without an owner
    if (h->own[0] != 1 || h->own[nproc] != d) { const int a = h->own[0], b = h->own[nproc]; delete h; return fail(TTX_EINVAL, "mybonds: must run from 1 to d = %d (got %d .. %d)", d, a, b); }
    // bond groups are dealt contiguously to the GPUs of the job
    h->g0 = (int)((long long)nproc * h->wrank / W);
    h->G = (int)((long long)nproc * (h->wrank + 1) / W) - h->g0;
    h->nbmax = 0;
    for (int g = 0; g < nproc; g++) h->nbmax = std::max(h->nbmax, h->own[g + 1] - h->own[g]);   // same launch shape on every GPU
    h->NC = h->nbmax + 1;
    h->mode = (cfg->pivoting == 0) ? 1 : (cfg->pivoting < 0) ? 2 : 0;
    h->H = (cfg->pivoting <= 0) ? 2 : 2 * cfg->pivoting;
    HIPCHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));

    DevProb &P = h->P;
    P.d = d; P.RM = h->RM; P.NM = NM; P.G = h->G; P.NC = h->NC; P.g0 = h->g0;
    P.fun_id = cfg->fun_id; P.piv = cfg->pivoting; P.npar = cfg->npar; P.nprocs = nproc;
    P.ising_id = (cfg->fun_id == TTX_FUN_ISING) ? (int)cfg->par[2 * cfg->n[0]] : 0;
    P.has_quad = cfg->quadw != nullptr;
    P.small_element = 10 * 2.220446049250313e-16; P.small_pivot = 1.e-5;   // lib/dmrgg.f90:70-71
    P.mvn_norm = 1.0;
    if (cfg->fun_id == TTX_FUN_MVN) {
        if (cfg->naux < d + d * d + 1) { delete h; return fail(TTX_EINVAL, "mvn: aux too short"); }
        P.mvn_norm = std::sqrt(powi(2.0 * 3.141592653589793, d) * cfg->aux[d + (size_t)d * d]);   // lib/mvn_pdf.f90:82
        if (!(P.mvn_norm > 0.0) || !std::isfinite(P.mvn_norm)) {
            const double nrm = P.mvn_norm, det = cfg->aux[d + (size_t)d * d];
            delete h;
            return fail(TTX_EINVAL, "ttx_create: mvn normalisation sqrt((2 pi)^d det) = %g is not a positive finite number (det = %g under- or overflows at d = %d)",
                        nrm, det, d);
        }
    }
    P.SS = (size_t)h->RM * NM; P.SW = (size_t)NM * h->RM; P.CS = (size_t)h->RM * NM * h->RM;
    const size_t G = h->G, NC = h->NC, RM = h->RM;
    int rc;
    int *dn; double *dpar, *daux = nullptr, *dq = nullptr;
#define A_(call) if ((rc = (call)) != TTX_OK) { ttx_destroy(h); return rc; }
    A_(dev_alloc(h, &dn, d + 2));
    A_(dev_alloc(h, &dpar, cfg->npar + 1));
    HIPCHECK(hipMemcpy(dn, h->n1.data(), sizeof(int) * (d + 2), hipMemcpyHostToDevice));
    if (cfg->npar > 0) HIPCHECK(hipMemcpy(dpar, h->par.data(), sizeof(double) * cfg->npar, hipMemcpyHostToDevice));
    if (!h->aux.empty()) { A_(dev_alloc(h, &daux, h->aux.size())); HIPCHECK(hipMemcpy(daux, h->aux.data(), sizeof(double) * h->aux.size(), hipMemcpyHostToDevice)); }
    if (cfg->quadw) {
        h->quadw.assign((size_t)(d + 1) * NM, 0.0);
        size_t off = 0;
        for (int k = 1; k <= d; k++) { for (int j = 0; j < h->n1[k]; j++) h->quadw[(size_t)k * NM + j] = cfg->quadw[off + j]; off += h->n1[k]; }
        A_(dev_alloc(h, &dq, h->quadw.size()));
        HIPCHECK(hipMemcpy(dq, h->quadw.data(), sizeof(double) * h->quadw.size(), hipMemcpyHostToDevice));
    }
    P.n = dn; P.par = dpar; P.aux = daux; P.quadw = dq;
    {   // TTX_ARITH: exact (default) or fast; fast is effective where a re-associated evaluator exists (ttx_fast.h)
        bool want = cfg->arith == TTX_ARITH_FAST;
        if (cfg->arith != TTX_ARITH_EXACT && cfg->arith != TTX_ARITH_FAST) { ttx_destroy(h); return fail(TTX_EINVAL, "ttx_create: arith must be TTX_ARITH_EXACT or TTX_ARITH_FAST (got %d)", cfg->arith); }
        if (const char *e = getenv("TTX_ARITH")) {
            const std::string v = e;
            if (v == "fast") want = true;
            else if (v == "exact") want = (cfg->arith == TTX_ARITH_FAST);
            else { ttx_destroy(h); return fail(TTX_EINVAL, "TTX_ARITH must be exact or fast (got %s)", e); }
        }
        h->want_fast = want && !nofun;
        // the integrand reads node par[ind - 1] for ind up to the LARGEST mode size (lib: nodes + ind), whatever n(1) is
        int nnode = 0; for (int k = 0; k < d; k++) nnode = std::max(nnode, (int)cfg->n[k]);
        nnode = std::min(nnode, (int)cfg->npar);
        bool unit = true;               // Ising: all nodes in [0,1] (every running product non-increasing: the cut at 2^-54 is valid)
        if (cfg->fun_id == TTX_FUN_ISING) for (int j = 0; j < nnode; j++) if (!(cfg->par[j] >= 0.0 && cfg->par[j] <= 1.0)) unit = false;
        P.arith = (want && !nofun && ((cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1 && unit) || cfg->fun_id == TTX_FUN_MVN)) ? 1 : 0;
        if (P.arith) {
            P.FD = d + 1;
            // tables per bond, kept for the whole run and extended incrementally (ttx_fast.h); TTX_FAST_PERSIST=0: mvn rebuilds the
            // tables of a bond step's two pivot sets with k_fast_tables instead (the first version, kept as a cross-check)
            P.fpersist = 1;
            if (cfg->fun_id == TTX_FUN_MVN && getenv("TTX_FAST_PERSIST") && atoi(getenv("TTX_FAST_PERSIST")) == 0) P.fpersist = 0;
            const size_t slots = P.fpersist ? G * NC : G;
            for (int sd = 0; sd < 2; sd++) {
                A_(dev_alloc(h, &P.fNear[sd], slots * (size_t)P.FD * RM));
                A_(dev_alloc(h, &P.fPiv[sd], slots * (size_t)TTX_FS * RM));
                if (cfg->fun_id == TTX_FUN_MVN) A_(dev_alloc(h, &P.fDv[sd], slots * (size_t)P.FD * RM));
            }
            if (cfg->fun_id == TTX_FUN_MVN) {
                std::vector<double> sy((size_t)d * d);
                const double *ic = h->aux.data() + d;
                for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) sy[i + (size_t)d * j] = 0.5 * (ic[i + (size_t)d * j] + ic[j + (size_t)d * i]);
                double *ds;
                A_(dev_alloc(h, &ds, sy.size()));
                HIPCHECK(hipMemcpy(ds, sy.data(), sizeof(double) * sy.size(), hipMemcpyHostToDevice));
                P.auxS = ds;
            }
        }
    }
    const bool de_lane = getenv("TTX_DE_LANE") && atoi(getenv("TTX_DE_LANE")) == 1;
    if (cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1 && !P.arith && de_lane) {
        // one fiber element per lane, every pair by division, rows ended at the unit cut (f_ising_de with `unit`): no tables, no teams
        P.de_unit = 1;
        for (int j = 0; j < std::min((int)cfg->npar, std::max((int)cfg->n[0], (int)NM)); j++) if (!(cfg->par[j] >= 0.0 && cfg->par[j] <= 1.0)) P.de_unit = 0;
    }
    if (cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1 && !P.arith && !de_lane && !(getenv("TTX_DE_TABLES") && atoi(getenv("TTX_DE_TABLES")) == 0)) {
        P.de_npair = d * (d + 1) / 2;
        A_(dev_alloc(h, &P.deTL, G * (size_t)P.de_npair * RM)); A_(dev_alloc(h, &P.deTR, G * (size_t)P.de_npair * RM));
        A_(dev_alloc(h, &P.deUL, G * (size_t)(d + 1) * RM));
        P.de_unit = 1;                  // nodes in [0,1]: every running product stays in [0,1] and fdiv_unit is exact
        for (int j = 0; j < std::min((int)cfg->npar, std::max((int)cfg->n[0], (int)NM)); j++) if (!(cfg->par[j] >= 0.0 && cfg->par[j] <= 1.0)) P.de_unit = 0;
        if (getenv("TTX_DE_FASTDIV") && atoi(getenv("TTX_DE_FASTDIV")) == 0) P.de_unit = 0;
        // nodes in [0,1]: compact tables, every row of the pair triangle ends at the unit cut (k_de_ctables, k_halfstep_dec; same bits).
        // TTX_DE_CUT=0: the full tables and the kernels of round 2 (wave teams, row-wise lottery)
        P.de_cut = (P.de_unit && !(getenv("TTX_DE_CUT") && atoi(getenv("TTX_DE_CUT")) == 0)) ? 1 : 0;
        if (P.de_cut) { A_(dev_alloc(h, &P.deCL, G * (size_t)(d + 1) * RM)); A_(dev_alloc(h, &P.deCR, G * (size_t)(d + 1) * RM)); }
        h->de_lot_point = d <= 160;
        if (const char *e = getenv("TTX_DE_LOT_POINT")) h->de_lot_point = atoi(e) != 0;
        h->de_slots = (int)RM * ((NM + 63) / 64);
        h->lds_de = sizeof(double) * (5 * (size_t)(((d + 7) & ~7) + 8) + 256) + (P.de_cut ? sizeof(int) * (size_t)(((d + 7) & ~7) + 8) : 0);
        h->de_v2 = cfg->pivoting >= 0 && h->de_slots <= TTX_MAXPART && h->lds_de <= 150 * 1024 &&
                   !(getenv("TTX_DE_V2") && atoi(getenv("TTX_DE_V2")) == 0);
        h->lds_det = sizeof(double) * det_lds_doubles(d, 3);
        h->lds_det6 = sizeof(double) * det_lds_doubles(d, 1);
        if (getenv("TTX_DE_TEAM6_UNITS")) h->de_team6_units = atoi(getenv("TTX_DE_TEAM6_UNITS"));
        h->de_team = h->de_v2 && !P.de_cut && h->lds_det <= 150 * 1024 && !(getenv("TTX_DE_TEAM") && atoi(getenv("TTX_DE_TEAM")) == 0);
        if (getenv("TTX_DE_TEAM_UNITS")) h->de_team_units = atoi(getenv("TTX_DE_TEAM_UNITS"));
        if (getenv("TTX_DE_TEST_FAULT")) h->de_test_fault = atoi(getenv("TTX_DE_TEST_FAULT"));
        h->lds_de5 = sizeof(double) * de5_lds_doubles(d);
        h->de_v5 = h->de_v2 && !P.de_cut && de5_fits(d) && h->lds_de5 <= 150 * 1024 && getenv("TTX_DE_V5") && atoi(getenv("TTX_DE_V5")) == 1;
    }
    if (cfg->fun_id == TTX_FUN_MVN) {
        std::vector<double> t((size_t)d * d);
        for (int i = 0; i < d; i++) for (int j = 0; j < d; j++) t[j + (size_t)d * i] = h->aux[d + i + (size_t)d * j];
        double *dt;
        A_(dev_alloc(h, &dt, t.size()));
        HIPCHECK(hipMemcpy(dt, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
        P.auxT = dt;
    }
    A_(dev_alloc(h, &P.arg, G * NC * P.CS)); A_(dev_alloc(h, &P.col, G * NC * P.CS)); A_(dev_alloc(h, &P.row, G * NC * P.CS));
    A_(dev_alloc(h, &P.inv, G * NC * RM * RM)); A_(dev_alloc(h, &P.vip, G * NC * 4 * RM));
    A_(dev_alloc(h, &P.L, G * NC * d * RM)); A_(dev_alloc(h, &P.R, G * NC * d * RM));
    A_(dev_alloc(h, &P.r, G * (d + 2))); A_(dev_alloc(h, &P.rr, G * (d + 2))); A_(dev_alloc(h, &P.upd, G * (d + 2))); A_(dev_alloc(h, &P.tape, G * (d + 2) * 4));
    A_(dev_alloc(h, &P.acol, G * RM * NM)); A_(dev_alloc(h, &P.arow, G * RM * NM));
    A_(dev_alloc(h, &P.Tq, G * NC * RM * RM));
    A_(dev_alloc(h, &P.ind0, d + 2)); A_(dev_alloc(h, &P.gs, G));
    {   // the bond range of every local group is fixed for the life of the engine (k_reset leaves it alone)
        GroupState *g0s = (GroupState *)calloc(1, sizeof(GroupState));
        for (size_t g = 0; g < G; g++) {
            g0s->first = h->own[h->g0 + g]; g0s->last = h->own[h->g0 + g + 1] - 1; g0s->gglobal = h->g0 + (int)g;
            hipError_t e = hipMemcpy(P.gs + g, g0s, offsetof(GroupState, S), hipMemcpyHostToDevice);
            if (e != hipSuccess) { free(g0s); ttx_destroy(h); return fail(TTX_EHIP, "ttx_create: %s", hipGetErrorString(e)); }
        }
        free(g0s);
    }
    P.nfb = (int)((RM * NM + TTX_BLK - 1) / TTX_BLK);
    if (cfg->pivoting < 0) A_(dev_alloc(h, &P.pfull, G * NM * RM * (size_t)P.nfb));
    if (cfg->pivoting < 0 && getenv("TTX_FULLPIV") && std::string(getenv("TTX_FULLPIV")) == "mfma" && RM <= 64 &&
        (long long)(RM * NM) * (long long)(RM * NM) < (1LL << 31)) {
        // dense full pivoting: the whole superblock resident (8 (RM NM)^2 bytes per group: 21 MB at r=32, n=51; 334 MB at r=64, n=101)
        const size_t side = RM * NM, tiles = ((side + 63) / 64) * ((side + 63) / 64);
        A_(dev_alloc(h, &P.sb, G * side * side));
        A_(dev_alloc(h, &P.pfull2, G * tiles));
        P.fp_mfma = 1; P.fp_tiles = (int)tiles;
    }
    // exchange buffers
    P.XD = RM * NM + RM * RM;
    P.IOFF = (sizeof(int) * (XH + d + 2) + 15) & ~(size_t)15;
    P.MSZ = (P.IOFF + sizeof(double) * P.XD + 15) & ~(size_t)15;
    A_(dev_alloc(h, &P.msgR, G * P.MSZ)); A_(dev_alloc(h, &P.msgL, G * P.MSZ));
    A_(dev_alloc(h, &h->recvL, P.MSZ)); A_(dev_alloc(h, &h->recvR, P.MSZ));
    A_(dev_alloc(h, &P.inL, G)); A_(dev_alloc(h, &P.inR, G));
    A_(dev_alloc(h, &P.red, G * 4)); A_(dev_alloc(h, &P.redsend, 4));
    h->QB = (size_t)nproc * RM * RM + 2 * nproc;
    h->SB = SUM_HDR + nproc + 5 * (size_t)(d + 1);
    A_(dev_alloc(h, &P.qsend, h->QB)); A_(dev_alloc(h, &P.qwork, ((size_t)nproc + 2) * RM * RM)); A_(dev_alloc(h, &P.sumsend, h->SB));
    P.qscr = nullptr;
    if (2 * sizeof(double) * ((size_t)RM * RM + 2) > 150 * 1024) A_(dev_alloc(h, &P.qscr, (size_t)G * 2 * RM * RM));
    if (W > 1) { A_(dev_alloc(h, &P.redrecv, 4)); A_(dev_alloc(h, &P.qall, h->QB)); A_(dev_alloc(h, &P.sumrecv, h->SB)); }
    else { P.redrecv = P.redsend; P.qall = P.qsend; P.sumrecv = P.sumsend; }     // one GPU: results alias the inputs
    {   // message routing: neighbour on this GPU -> its send buffer; on another GPU -> the receive buffer
        std::vector<char *> il(G, nullptr), ir(G, nullptr);
        for (size_t g = 0; g < G; g++) {
            if (g > 0) il[g] = P.msgR + (g - 1) * P.MSZ; else if (h->g0 > 0) il[g] = h->recvL;
            if (g + 1 < G) ir[g] = P.msgL + (g + 1) * P.MSZ; else if (h->g0 + (int)G < nproc) ir[g] = h->recvR;
        }
        HIPCHECK(hipMemcpy(P.inL, il.data(), sizeof(char *) * G, hipMemcpyHostToDevice));
        HIPCHECK(hipMemcpy(P.inR, ir.data(), sizeof(char *) * G, hipMemcpyHostToDevice));
    }
#undef A_
    {
        int *ctl, *rq;
        rc = dev_alloc(h, &ctl, (size_t)4); if (rc) { ttx_destroy(h); return rc; }
        rc = dev_alloc(h, &rq, (size_t)G * (d + 2)); if (rc) { ttx_destroy(h); return rc; }
        P.ctl = ctl; P.rq = rq; P.accuracy = cfg->accuracy; P.maxrank = cfg->maxrank;
    }
    HIPCHECK(hipHostMalloc((void **)&h->h_sum_base, sizeof(double) * 2 * h->SB));
    h->h_sum = h->h_sum_base;
    memset(h->h_sum_base, 0, sizeof(double) * 2 * h->SB);
    HIPCHECK(hipHostMalloc((void **)&h->h_val, sizeof(double) * 2));
    HIPCHECK(hipStreamCreateWithFlags(&h->qstream, hipStreamNonBlocking));
    for (int x = 0; x < 2; x++) { HIPCHECK(hipEventCreateWithFlags(&h->ev_sum[x], hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&h->ev_val[x], hipEventDisableTiming)); }
    HIPCHECK(hipHostMalloc((void **)&h->h_msg, 4 * P.MSZ));
    HIPCHECK(hipHostMalloc((void **)&h->h_tmp, sizeof(double) * std::max(h->QB, h->SB)));
    if (W > 1) {
        h->red_cap = std::max<size_t>(std::max(h->QB, h->SB), 8);
        HIPCHECK(hipHostMalloc((void **)&h->red_mem, sizeof(double) * h->red_cap * ttx_engine::NRED));
    }
    {   // lottery CDF segment tables for every K that can occur (K <= maxrank*n): pure function of K, see ttx_cdf.h
        const int kmax = h->RM * NM;
        if (kmax <= 16384) {
            std::vector<ttx_cdfseg> tab((size_t)(kmax + 1) * TTX_TABSEG);
            std::vector<int> ns(kmax + 1, 0);
            std::vector<ttx_cdfseg> tmp(TTX_MAXSEG);
            bool ok = true;
            for (int K = 1; K <= kmax && ok; K++) {
                int n_ = ttx_cdf_build(K, tmp.data());
                if (n_ > TTX_TABSEG) { ok = false; break; }
                ns[K] = n_;
                memcpy(&tab[(size_t)K * TTX_TABSEG], tmp.data(), sizeof(ttx_cdfseg) * n_);
            }
            if (ok) {
                ttx_cdfseg *dt; int *dn_;
                rc = dev_alloc(h, &dt, tab.size()); if (rc) { ttx_destroy(h); return rc; }
                rc = dev_alloc(h, &dn_, ns.size()); if (rc) { ttx_destroy(h); return rc; }
                HIPCHECK(hipMemcpy(dt, tab.data(), sizeof(ttx_cdfseg) * tab.size(), hipMemcpyHostToDevice));
                HIPCHECK(hipMemcpy(dn_, ns.data(), sizeof(int) * ns.size(), hipMemcpyHostToDevice));
                P.cdf_tab = dt; P.cdf_ns = dn_; P.cdf_kmax = kmax;
            }
        }
    }
    h->lds_par = sizeof(double) * (cfg->npar + 2);
    {   // half-step LDS: index rows always fit the limit checked below; value rows (Ising C fast path) if <= 96 KB
        const size_t VS = ((d + 7) & ~7) + 8;
        const size_t base = sizeof(double) * (cfg->npar + RM + 4);
        const size_t idx_bytes = sizeof(short) * ((RM + 1) * VS + 16), val_bytes = sizeof(double) * ((RM + 1) * 2 * VS + 4);
        h->half_vals = (cfg->fun_id == TTX_FUN_ISING && P.ising_id == 1 && base + val_bytes <= 100 * 1024) ? 1 : 0;
        h->lds_half = base + (h->half_vals ? val_bytes : idx_bytes);
        const size_t dif_bytes = sizeof(double) * ((RM + 1) * VS + 4);          // mvn: rows of differences x - mu
        if (cfg->fun_id == TTX_FUN_MVN && base + dif_bytes <= 140 * 1024) { h->half_vals = 1; h->lds_half = base + dif_bytes; }
        if (P.arith) h->lds_half = std::max(h->lds_half, base + sizeof(double) * ((size_t)std::max(P.FD, TTX_FNR) + 4 + RM + 2));   // far[] + cross terms
    }
    {
        const int nlotmax = 2 * h->RM + 2 * NM;
        const size_t VS = ((d + 7) & ~7) + 8;
        h->lds_lot = sizeof(double) * (cfg->npar + 4) + sizeof(int) * 4 * (nlotmax + 4) + sizeof(short) * (2 * RM * VS + 16);
        const size_t lot_dif = sizeof(double) * (cfg->npar + 4) + sizeof(int) * 4 * (nlotmax + 4) + sizeof(double) * (2 * RM * VS + 4);
        if (cfg->fun_id == TTX_FUN_MVN && lot_dif <= 120 * 1024) { h->lot_vals = 1; h->lds_lot = lot_dif; }
        if (P.arith && cfg->fun_id == TTX_FUN_ISING) {
            // fast mode: the leading rows of the two decay tables ([row][RM] doubles each) instead of the index rows
            const size_t hdr = sizeof(double) * (cfg->npar + 4) + sizeof(int) * 4 * (nlotmax + 4) + 32;
            const size_t rowb = 2 * sizeof(double) * RM;
            h->fast_cap = (int)std::min<size_t>(std::min<size_t>(40, (size_t)d + 1), hdr < 100 * 1024 ? (100 * 1024 - hdr) / rowb : 0);
            h->lds_lot = std::max(h->lds_lot, hdr + rowb * h->fast_cap);
        }
    }
    {   // whole-sweep kernels (Ising C): TTX_SWEEP = auto | chain | fused | cluster
        const size_t VS = ((d + 7) & ~7) + 8;
        const int nlotmax = 2 * h->RM + 2 * NM;
        const char *env = getenv("TTX_SWEEP");
        const std::string want = env ? env : "auto";
        const bool fastc = cfg->fun_id == TTX_FUN_ISING && P.ising_id == 1 && cfg->pivoting >= 0 && P.cdf_tab != nullptr;
        // one 1024-thread workgroup per group: everything of a bond step in one CU's LDS
        h->lds_fused = sizeof(double) * (cfg->npar + 4 + 4 * RM * VS + 2 * RM * NM + RM + 4) + sizeof(int) * 4 * (nlotmax + 4);
        const bool fused_ok = fastc && h->RM <= 64 && nlotmax <= FB && h->lds_fused <= 150 * 1024;
        // a cluster of NB 256-thread workgroups per group, all resident at once (G*NB <= number of CUs)
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, cfg->device));
        int NB = 8;
        if (const char *e = getenv("TTX_CLUSTER_NB")) NB = atoi(e);
        NB = std::max(1, std::min(NB, TTX_CLMAX));
        while (NB > 1 && h->G * NB > prop.multiProcessorCount / 2) NB--;        // leave room for other processes on the card
        const size_t SL = (size_t)RM * ((NM + NB - 1) / NB + 1);
        h->lds_cluster = sizeof(double) * (cfg->npar + 4 + 2 * RM * (2 * VS + 2) + 4 * SL + 2 * RM + 8) + sizeof(int) * 4 * (nlotmax + 4);
        if (h->lds_cluster + sizeof(double) * 2 * RM * RM <= 150 * 1024) { h->cluster_ldsinv = 1; h->lds_cluster += sizeof(double) * 2 * RM * RM; }
        {
            const size_t zk = sizeof(int) * ((size_t)h->nbmax * 2 * RM + 2 * h->nbmax + 4);
            if (h->lds_cluster + zk <= 150 * 1024) { h->cluster_zkeep = 1; h->lds_cluster += zk; }
        }
        bool cluster_ok = fastc && h->RM <= 64 && NB >= 2 && h->G * NB <= prop.multiProcessorCount && h->lds_cluster <= 150 * 1024;
        if (cluster_ok) {
            // residency: what the device can hold of THIS kernel with THIS much dynamic LDS; the grid may use half of it
            static size_t a_cl0 = 0;
            if ((rc = ensure_lds(reinterpret_cast<const void *>(k_sweep_cluster), h->lds_cluster, a_cl0))) { ttx_destroy(h); return rc; }
            int occ = 0;
            HIPCHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_sweep_cluster, CB, h->lds_cluster));
            const long long cap = (long long)occ * prop.multiProcessorCount;
            const long long grid = 8LL * NB * ((h->G + 7) / 8);
            if (grid * 2 > cap) cluster_ok = false;
            int coop = 0;
            HIPCHECK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, cfg->device));
            // plain launch by default: with the occupancy gate above every workgroup is placed as soon as the launch starts;
            // the cooperative launch (TTX_CLUSTER_COOP=1) adds the runtime's own refusal of oversized grids but costs
            // ~30 us per launch on this stack (C_64: 5.28 -> 5.80 ms per run, measured)
            h->cluster_coop = coop && getenv("TTX_CLUSTER_COOP") && atoi(getenv("TTX_CLUSTER_COOP")) == 1;
        }
        if (want == "cluster") { if (cluster_ok) h->cluster = NB; }
        else if (want == "fused") { if (fused_ok) h->fused = 1; }
        else if (want == "auto") { if (cluster_ok) h->cluster = NB; else if (fused_ok && h->G == 1) h->fused = 1; }
        else if (want != "chain") { ttx_destroy(h); return fail(TTX_EINVAL, "TTX_SWEEP must be auto, chain, fused or cluster (got %s)", want.c_str()); }
        // TTX_ARITH=fast for Ising C: a closed form inside the cluster kernel (f_ising_cfast); the other paths evaluate C exactly
        if (h->cluster && h->want_fast && cfg->fun_id == TTX_FUN_ISING && P.ising_id == 1) P.arith = 1;
        if (h->cluster) {
            unsigned *ctr; ClPart *cp;
            rc = dev_alloc(h, &ctr, (size_t)h->G); if (rc) { ttx_destroy(h); return rc; }
            rc = dev_alloc(h, &cp, (size_t)2 * h->G * TTX_CLREC); if (rc) { ttx_destroy(h); return rc; }
            P.cl_ctr = ctr; P.cl_part = cp;
#ifdef TTX_STAMPS
            if (getenv("TTX_DBG_WAVES")) { long long *dbg; rc = dev_alloc(h, &dbg, (size_t)8 * 64 * 8); if (rc) { ttx_destroy(h); return rc; } P.dbg = dbg; }
#endif
            HIPCHECK(hipHostMalloc((void **)&h->h_abort, sizeof(int)));
            *h->h_abort = 0;
            P.cl_abort = h->h_abort;
            if (const char *e = getenv("TTX_CLUSTER_TEST_ABORT")) P.cl_test_abort = atoi(e);
        }
    }
    if (h->lds_half > 160 * 1024 || h->lds_lot > 120 * 1024) { ttx_destroy(h); return fail(TTX_EINVAL, "problem too large for LDS staging (d*maxrank)"); }
    {   // lottery: one wave of candidates per workgroup where one evaluation is a long dependent chain (Ising D/E, mvn)
        const bool heavy = (cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1) || cfg->fun_id == TTX_FUN_MVN;
        const int nlotmax = 2 * h->RM + 2 * NM;
        P.lot_nb = 1;
        if (heavy && !(getenv("TTX_LOTTERY_NB") && atoi(getenv("TTX_LOTTERY_NB")) == 1)) P.lot_nb = std::min((nlotmax + 63) / 64, 64);
        if (P.lot_nb > 1 && (nlotmax + P.lot_nb - 1) / P.lot_nb > 64) P.lot_nb = 1;      // more candidates than 64 blocks x 64: keep one block
        LotPart *lp; unsigned *lc;
        rc = dev_alloc(h, &lp, (size_t)h->G * P.lot_nb); if (rc) { ttx_destroy(h); return rc; }
        rc = dev_alloc(h, &lc, (size_t)h->G); if (rc) { ttx_destroy(h); return rc; }
        P.lotp = lp; P.lot_ctr = lc;
        P.lot_max = nlotmax;
        if (cfg->fun_id == TTX_FUN_MVN && d <= 64 * MVN_MAXQ && cfg->pivoting >= 0 && (int)RM * ((NM + 63) / 64) <= TTX_MAXPART &&
            !(getenv("TTX_MVN_V2") && atoi(getenv("TTX_MVN_V2")) == 0)) {
            int *lcd; double *lf;
            rc = dev_alloc(h, &lcd, (size_t)h->G * nlotmax * 4); if (rc) { ttx_destroy(h); return rc; }
            rc = dev_alloc(h, &lf, (size_t)h->G * nlotmax); if (rc) { ttx_destroy(h); return rc; }
            P.lotc = lcd; P.lotf = lf;
            h->mvn_v2 = 1; h->de_slots = (int)RM * ((NM + 63) / 64);
            P.bnd_wave = 1;
            h->lds_mvn = sizeof(double) * (3 * (size_t)d + 8);
        }
        if (cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1 && P.arith) P.bnd_wave = 1;     // boundary corners by de_fast_point_wave
        if (cfg->fun_id == TTX_FUN_ISING && P.ising_id != 1 && P.deTL) {
            int *lcd; double *lf;
            rc = dev_alloc(h, &lcd, (size_t)h->G * nlotmax * 4); if (rc) { ttx_destroy(h); return rc; }
            rc = dev_alloc(h, &lf, (size_t)h->G * nlotmax); if (rc) { ttx_destroy(h); return rc; }
            P.lotc = lcd; P.lotf = lf;
            // candidates and boundary corners by the row-wise wave evaluator (ttx_de.h); TTX_LOTTERY_WAVE=0: one lane per element
            h->lds_der = sizeof(double) * de_rows_lds_doubles(d);
            h->lot_wave = h->de_v2 && !P.de_cut && h->lds_der <= 150 * 1024 && !(getenv("TTX_LOTTERY_WAVE") && atoi(getenv("TTX_LOTTERY_WAVE")) == 0);
            P.bnd_wave = h->lot_wave || (P.de_cut && h->de_v2 && h->lds_der <= 150 * 1024);    // boundary corners by one wave per corner
            h->lot_rows = h->lot_wave ? ((getenv("TTX_LOTTERY_ROWS") && atoi(getenv("TTX_LOTTERY_ROWS")) == 2) ? 2 : 1) : 0;
        }
    }
    if (cfg->fun_id == TTX_FUN_HOST) {
        // slots of one group: the largest point set any evaluating kernel asks for in one launch
        int nn = h->n1[1];
        for (int k = 2; k <= d; k++) nn = std::min(nn, h->n1[k]);
        const size_t snum = (size_t)std::max(8, nproc);
        h->HS = std::max<size_t>({(size_t)h->RM * NM, (size_t)nn * snum, (size_t)h->NC * NM, (size_t)2 * h->RM + 2 * NM, (size_t)2 * NM, (size_t)256});
        const size_t nslot = (size_t)h->G * h->HS;
        HIPCHECK(hipHostMalloc((void **)&P.hidx, sizeof(short) * nslot * d));
        HIPCHECK(hipHostMalloc((void **)&P.hval, sizeof(double) * nslot));
        HIPCHECK(hipHostMalloc((void **)&P.hreq, nslot));
        memset(P.hreq, 0, nslot); memset(P.hval, 0, sizeof(double) * nslot);
        P.HS = (int)h->HS; P.hostpass = 0;
    }
    *out = h;
    return TTX_OK;
}
extern "C" int ttx_create(ttx_engine **out, const ttx_config *cfg) { return create_impl(out, cfg, false); }

static void shm_close(ttx_engine *h);
extern "C" void ttx_destroy(ttx_engine *h)
{
    if (!h) return;
    if (h->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->comm);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->h_sum_base) (void)hipHostFree(h->h_sum_base);
    if (h->h_val) (void)hipHostFree(h->h_val);
    if (h->h_svd) (void)hipHostFree(h->h_svd);
    if (h->qstream) (void)hipStreamDestroy(h->qstream);
    for (int x = 0; x < 2; x++) { if (h->ev_sum[x]) (void)hipEventDestroy(h->ev_sum[x]); if (h->ev_val[x]) (void)hipEventDestroy(h->ev_val[x]); }
    if (h->h_msg) (void)hipHostFree(h->h_msg);
    if (h->h_tmp) (void)hipHostFree(h->h_tmp);
    if (h->red_mem) (void)hipHostFree(h->red_mem);
    shm_close(h);
    if (h->h_abort) (void)hipHostFree(h->h_abort);
    if (h->P.hidx) (void)hipHostFree(h->P.hidx);
    if (h->P.hval) (void)hipHostFree(h->P.hval);
    if (h->P.hreq) (void)hipHostFree(h->P.hreq);
    for (auto &e : h->evpool) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

// ---- transports between the GPUs of one job ---------------------------------------------------------
extern "C" int ttx_comm_unique_id(uint8_t id[128])
{
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    NCCLCHECK(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, 128);
    return TTX_OK;
}
extern "C" int ttx_comm_init(ttx_engine *h, const uint8_t id[128])
{
    if (!h) return fail(TTX_EINVAL, "ttx_comm_init: null handle");
    if (h->W == 1) return TTX_OK;
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHECK(hipSetDevice(h->cfg.device));
    ncclUniqueId u;
    memcpy(u.internal, id, 128);
    NCCLCHECK(g_rccl.CommInitRank(&h->comm, h->W, u, h->wrank));
    return TTX_OK;
}
// Loop-back self-test of the RCCL transport on ONE device: a communicator of one rank, then exactly the calls the multi-GPU data
// path makes -- a grouped ncclSend / ncclRecv pair of one packed neighbour message (ncclChar) on a non-blocking stream, the
// 3-double MAX all-reduce of the sweep maxima and a SUM all-reduce of doubles (quadrature partials / per-sweep summary) -- each
// followed by a byte comparison.  It exercises the dlopen'ed symbol table, the datatypes and the stream ordering on pools where
// a second GPU (and with it ttx_comm_init with world_size > 1) is not available.
extern "C" int ttx_k_rccl_selftest(int32_t device, int64_t msg_bytes, int32_t nsum)
{
    if (msg_bytes < 1 || nsum < 1) return fail(TTX_EINVAL, "ttx_k_rccl_selftest: bad sizes");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    int rc = rccl_load();
    if (rc) return rc;
    ncclUniqueId u;
    NCCLCHECK(g_rccl.GetUniqueId(&u));
    ncclComm_t comm = nullptr;
    NCCLCHECK(g_rccl.CommInitRank(&comm, 1, u, 0));
    hipStream_t st = nullptr;
    char *dsend = nullptr, *drecv = nullptr; double *dred = nullptr, *dout = nullptr;
    std::vector<char> hs((size_t)msg_bytes), hr((size_t)msg_bytes);
    std::vector<double> hd((size_t)nsum + 4), ho((size_t)nsum + 4);
    for (int64_t i = 0; i < msg_bytes; i++) hs[(size_t)i] = (char)((i * 131 + 7) & 0xff);
    for (size_t i = 0; i < hd.size(); i++) hd[i] = 1.0 / (double)(i + 3) - 0.25 * (double)(i % 5);
    int bad = 0;
    auto body = [&]() -> int {
        HIPCHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        HIPCHECK(hipMalloc((void **)&dsend, (size_t)msg_bytes)); HIPCHECK(hipMalloc((void **)&drecv, (size_t)msg_bytes));
        HIPCHECK(hipMalloc((void **)&dred, sizeof(double) * hd.size())); HIPCHECK(hipMalloc((void **)&dout, sizeof(double) * hd.size()));
        HIPCHECK(hipMemcpyAsync(dsend, hs.data(), (size_t)msg_bytes, hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemsetAsync(drecv, 0, (size_t)msg_bytes, st));
        HIPCHECK(hipMemcpyAsync(dred, hd.data(), sizeof(double) * hd.size(), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemsetAsync(dout, 0, sizeof(double) * hd.size(), st));
        NCCLCHECK(g_rccl.GroupStart());
        NCCLCHECK(g_rccl.Send(dsend, (size_t)msg_bytes, ncclChar, 0, comm, st));
        NCCLCHECK(g_rccl.Recv(drecv, (size_t)msg_bytes, ncclChar, 0, comm, st));
        NCCLCHECK(g_rccl.GroupEnd());
        NCCLCHECK(g_rccl.AllReduce(dred, dout, 4, ncclDouble, ncclMax, comm, st));                          // amax, pivotmax, -pivotmin (+ pad)
        NCCLCHECK(g_rccl.AllReduce(dred + 4, dout + 4, (size_t)nsum, ncclDouble, ncclSum, comm, st));        // summary / quadrature partials
        HIPCHECK(hipMemcpyAsync(hr.data(), drecv, (size_t)msg_bytes, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipMemcpyAsync(ho.data(), dout, sizeof(double) * ho.size(), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (memcmp(hs.data(), hr.data(), (size_t)msg_bytes) != 0) bad |= 1;
        if (memcmp(hd.data(), ho.data(), sizeof(double) * hd.size()) != 0) bad |= 2;
        return TTX_OK;
    };
    rc = body();
    if (dsend) (void)hipFree(dsend); if (drecv) (void)hipFree(drecv); if (dred) (void)hipFree(dred); if (dout) (void)hipFree(dout);
    if (st) (void)hipStreamDestroy(st);
    (void)g_rccl.CommDestroy(comm);
    if (rc) return rc;
    if (bad) return fail(TTX_EHIP, "ttx_k_rccl_selftest: %s came back different", bad == 1 ? "the point-to-point message" : bad == 2 ? "an all-reduce" : "message and all-reduce");
    return TTX_OK;
}

// ---- built-in node-local host transport over POSIX shared memory ------------------------------------------------
// For jobs whose processes share a node but cannot use RCCL (several ranks on ONE GPU -- RCCL refuses that -- or no
// librccl), and for launchers without MPI (the Fortran drop-in layer): the ttx_transport primitives implemented on a
// shared segment.  Point-to-point: one mailbox per (receiver, side) with a sequence / acknowledge pair; all-reduce: every
// rank deposits its vector, a sense-reversing barrier, every rank folds the W vectors in rank order (so all ranks get
// the identical bits), a second barrier before the slots are reused.  Waits are bounded (60 s) and report failure.
// Attaching is a handshake on a per-initialisation NONCE, so that a rank can never end up on a segment that rank 0 did not
// create in THIS call (a segment left by a crashed job, or the one of the previous dtt_dmrgg of the same job, still carries
// ready = 1 and old counters): rank 0 unlinks the name, creates a fresh segment and publishes a random nonce; rank r copies the
// nonce it sees into hello[r] and waits for go == that nonce, which rank 0 sets once every hello matches.  A segment whose go
// is already set before the rank said hello is stale by construction; while it waits, a rank re-checks that the NAME still
// leads to the inode it has mapped (rank 0's unlink + create changes it) and starts over if not.
struct ShmHeader {
    std::atomic<uint32_t> ready, arrived, sense;
    uint32_t W; uint64_t msz, redcap;
    std::atomic<uint64_t> nonce, go;
    std::atomic<uint64_t> hello[256];
};
static_assert(sizeof(ShmHeader) <= 4096, "the header shares the first page of the segment");
struct ShmBox { std::atomic<uint64_t> seq, ack; uint64_t bytes; };
struct ShmTransport {
    void *base = nullptr; size_t size = 0; std::string name; int rank = 0, W = 1; bool owner = false;
    ShmHeader *hd = nullptr;
    size_t msz = 0, redcap = 0;
    uint32_t my_sense = 0;
    ShmBox *box(int r, int side) const { return (ShmBox *)((char *)base + 4096 + ((size_t)r * 2 + side) * (64 + msz)); }   // side 0: from the left, 1: from the right
    char *boxdata(int r, int side) const { return (char *)box(r, side) + 64; }
    double *red(int r) const { return (double *)((char *)base + 4096 + (size_t)W * 2 * (64 + msz) + (size_t)r * redcap * sizeof(double)); }
};
static bool shm_wait(const std::function<bool()> &cond)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (cond()) return true;
        if ((spins & 1023u) == 1023u) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) return false;
            std::this_thread::yield();
        }
    }
}
static int shm_barrier(ShmTransport *T)
{
    T->my_sense ^= 1u;
    if (T->hd->arrived.fetch_add(1u, std::memory_order_acq_rel) + 1u == (uint32_t)T->W) {
        T->hd->arrived.store(0u, std::memory_order_relaxed);
        T->hd->sense.store(T->my_sense, std::memory_order_release);
        return 0;
    }
    return shm_wait([&] { return T->hd->sense.load(std::memory_order_acquire) == T->my_sense; }) ? 0 : 1;
}
static int shm_sendrecv(void *ctx, int to, const void *sbuf, int64_t ns, int from, void *rbuf, int64_t nr)
{
    ShmTransport *T = (ShmTransport *)ctx;
    if ((size_t)ns > T->msz || (size_t)nr > T->msz) return 1;
    if (to >= 0) {                                   // I am the left neighbour of `to` when to == rank + 1: its side-0 box
        const int side = (to == T->rank + 1) ? 0 : 1;
        ShmBox *b = T->box(to, side);
        if (!shm_wait([&] { return b->ack.load(std::memory_order_acquire) == b->seq.load(std::memory_order_relaxed); })) return 1;
        memcpy(T->boxdata(to, side), sbuf, (size_t)ns);
        b->bytes = (uint64_t)ns;
        b->seq.fetch_add(1u, std::memory_order_release);
    }
    if (from >= 0) {
        const int side = (from == T->rank - 1) ? 0 : 1;
        ShmBox *b = T->box(T->rank, side);
        if (!shm_wait([&] { return b->seq.load(std::memory_order_acquire) != b->ack.load(std::memory_order_relaxed); })) return 1;
        memcpy(rbuf, T->boxdata(T->rank, side), (size_t)std::min<uint64_t>((uint64_t)nr, b->bytes));
        b->ack.fetch_add(1u, std::memory_order_release);
    }
    return 0;
}
static int shm_allreduce(void *ctx, double *buf, int64_t count, int op)
{
    ShmTransport *T = (ShmTransport *)ctx;
    if ((size_t)count > T->redcap) return 1;
    memcpy(T->red(T->rank), buf, sizeof(double) * (size_t)count);
    if (shm_barrier(T)) return 1;
    for (int64_t i = 0; i < count; i++) {
        double a = T->red(0)[i];
        for (int r = 1; r < T->W; r++) a = op ? std::max(a, T->red(r)[i]) : a + T->red(r)[i];
        buf[i] = a;
    }
    return shm_barrier(T);
}
static void shm_close(ttx_engine *h)
{
    ShmTransport *T = h->shm;
    if (!T) return;
    if (T->base) munmap(T->base, T->size);
    if (T->owner) shm_unlink(T->name.c_str());
    delete T;
    h->shm = nullptr;
}
extern "C" int ttx_comm_init_shm(ttx_engine *h, const char *name)
{
    if (!h || !name || !*name) return fail(TTX_EINVAL, "ttx_comm_init_shm: null argument");
    if (h->W == 1) return TTX_OK;
    if (h->shm) return fail(TTX_ESTATE, "ttx_comm_init_shm: already initialised");
    ShmTransport *T = new ShmTransport();
    T->name = std::string(name[0] == '/' ? "" : "/") + name;
    T->rank = h->wrank; T->W = h->W;
    T->msz = (h->P.MSZ + 63) & ~(size_t)63;
    T->redcap = std::max<size_t>(std::max(h->QB, h->SB), 8);
    T->size = 4096 + (size_t)T->W * 2 * (64 + T->msz) + (size_t)T->W * T->redcap * sizeof(double);
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    if (T->rank == 0) {
        shm_unlink(T->name.c_str());
        int fd = shm_open(T->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)T->size) != 0) { if (fd >= 0) close(fd); delete T; return fail(TTX_EHIP, "ttx_comm_init_shm: cannot create %s", name); }
        T->owner = true;
        T->base = mmap(nullptr, T->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (T->base == MAP_FAILED) { T->base = nullptr; shm_unlink(T->name.c_str()); delete T; return fail(TTX_EHIP, "ttx_comm_init_shm: mmap failed"); }
        T->hd = (ShmHeader *)T->base;          // a fresh segment is zero-filled: sequence numbers, counters, sense, go and hello start at 0
        T->hd->W = (uint32_t)T->W; T->hd->msz = T->msz; T->hd->redcap = T->redcap;
        std::random_device rd;
        uint64_t nonce = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)getpid() << 17) ^ (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count();
        if (nonce == 0) nonce = 1;
        T->hd->nonce.store(nonce, std::memory_order_relaxed);
        T->hd->ready.store(1u, std::memory_order_release);
        const bool all = shm_wait([&] {
            for (int r = 1; r < T->W; r++) if (T->hd->hello[r].load(std::memory_order_acquire) != nonce) return false;
            return true;
        });
        if (!all) { munmap(T->base, T->size); shm_unlink(T->name.c_str()); delete T; return fail(TTX_EHIP, "ttx_comm_init_shm: not all %d ranks attached to %s", T->W, name); }
        T->hd->go.store(nonce, std::memory_order_release);
    } else {
        bool joined = false, mismatch = false;
        while (!joined && elapsed() < 60.0) {
            int fd = shm_open(T->name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd < 0 || fstat(fd, &st) != 0 || (size_t)st.st_size < T->size) { if (fd >= 0) close(fd); std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
            const ino_t ino = st.st_ino;
            void *base = mmap(nullptr, T->size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
            close(fd);
            if (base == MAP_FAILED) { delete T; return fail(TTX_EHIP, "ttx_comm_init_shm: mmap failed"); }
            ShmHeader *hd = (ShmHeader *)base;
            auto still_current = [&] {          // does the name still lead to the segment that is mapped here?
                int f2 = shm_open(T->name.c_str(), O_RDWR, 0600);
                struct stat s2;
                const bool same = f2 >= 0 && fstat(f2, &s2) == 0 && s2.st_ino == ino;
                if (f2 >= 0) close(f2);
                return same;
            };
            bool restart = false, said = false;
            uint64_t nonce = 0;
            auto t_chk = std::chrono::steady_clock::now();
            while (!restart && elapsed() < 60.0) {
                if (!said && hd->ready.load(std::memory_order_acquire) == 1u) {
                    mismatch = hd->W != (uint32_t)T->W || hd->msz != T->msz || hd->redcap != T->redcap;
                    nonce = hd->nonce.load(std::memory_order_relaxed);
                    if (mismatch || hd->go.load(std::memory_order_acquire) == nonce) {
                        // another problem's segment, or one whose initialisation is over: not ours -- wait for rank 0 to replace the name
                        while (elapsed() < 60.0 && still_current()) std::this_thread::sleep_for(std::chrono::milliseconds(2));
                        restart = true;
                        break;
                    }
                    hd->hello[T->rank].store(nonce, std::memory_order_release);
                    said = true;
                }
                if (said && hd->go.load(std::memory_order_acquire) == nonce) { joined = true; break; }
                if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_chk).count() > 0.02) {
                    t_chk = std::chrono::steady_clock::now();
                    if (!still_current()) restart = true;
                }
                std::this_thread::yield();
            }
            if (joined) { T->base = base; T->hd = hd; }
            else munmap(base, T->size);
        }
        if (!joined) {
            delete T;
            return mismatch ? fail(TTX_EINVAL, "ttx_comm_init_shm: the ranks disagree about the problem (or %s belongs to another job)", name)
                            : fail(TTX_EHIP, "ttx_comm_init_shm: rank 0 did not create %s (or never saw every rank)", name);
        }
    }
    h->shm = T;
    h->cb.ctx = T; h->cb.sendrecv = shm_sendrecv; h->cb.allreduce = shm_allreduce; h->have_cb = true;
    // everybody is attached before rank 0 may unlink the name at destroy time
    if (shm_barrier(T)) return fail(TTX_EHIP, "ttx_comm_init_shm: not all %d ranks attached", T->W);
    return TTX_OK;
}

extern "C" int ttx_set_transport(ttx_engine *h, const ttx_transport *t)
{
    if (!h || !t || !t->sendrecv || !t->allreduce) return fail(TTX_EINVAL, "ttx_set_transport: null argument");
    h->cb = *t; h->have_cb = true;
    return TTX_OK;
}

extern "C" int ttx_set_integrand_host(ttx_engine *h, ttx_host_fun fun, const double *par)
{
    if (!h || !fun) return fail(TTX_EINVAL, "ttx_set_integrand_host: null argument");
    if (h->cfg.fun_id != TTX_FUN_HOST) return fail(TTX_ESTATE, "ttx_set_integrand_host: the engine was not created with fun_id = TTX_FUN_HOST");
    h->hfun = fun; h->hfun_par = par;
    return TTX_OK;
}
extern "C" int64_t ttx_host_calls(const ttx_engine *h) { return h ? h->host_calls : 0; }

// host functions of the stream-ordered host transport (run on a runtime thread when the stream reaches them; no HIP calls)
static void hostfn_xfer(void *ud)
{
    ttx_engine *h = (ttx_engine *)ud;
    const DevProb &P = h->P;
    const int left = (h->g0 > 0) ? h->wrank - 1 : -1, right = (h->g0 + h->G < h->cfg.nproc) ? h->wrank + 1 : -1;
    char *sR = h->h_msg, *sL = h->h_msg + P.MSZ, *rL = h->h_msg + 2 * P.MSZ, *rR = h->h_msg + 3 * P.MSZ;
    if (h->cb.sendrecv(h->cb.ctx, right, sR, (int64_t)P.MSZ, left, rL, (int64_t)P.MSZ)) h->cb_error = 1;
    if (h->cb.sendrecv(h->cb.ctx, left, sL, (int64_t)P.MSZ, right, rR, (int64_t)P.MSZ)) h->cb_error = 1;
}
static void hostfn_allreduce(void *ud)
{
    ttx_engine::RedJob *j = (ttx_engine::RedJob *)ud;
    if (j->h->cb.allreduce(j->h->cb.ctx, j->buf, (int64_t)j->count, j->op)) j->h->cb_error = 1;
}

// messages of the boundary groups to the neighbouring GPUs (device buffers; after k_exch_pack)
static int xfer_neighbours(ttx_engine *h)
{
    if (h->W == 1) return TTX_OK;
    DevProb &P = h->P;
    const int left = (h->g0 > 0) ? h->wrank - 1 : -1, right = (h->g0 + h->G < h->cfg.nproc) ? h->wrank + 1 : -1;
    char *outR = P.msgR + (size_t)(h->G - 1) * P.MSZ, *outL = P.msgL;
    if (h->comm) {
        NCCLCHECK(g_rccl.GroupStart());
        if (right >= 0) { NCCLCHECK(g_rccl.Send(outR, P.MSZ, ncclChar, right, h->comm, h->stream)); NCCLCHECK(g_rccl.Recv(h->recvR, P.MSZ, ncclChar, right, h->comm, h->stream)); }
        if (left >= 0) { NCCLCHECK(g_rccl.Send(outL, P.MSZ, ncclChar, left, h->comm, h->stream)); NCCLCHECK(g_rccl.Recv(h->recvL, P.MSZ, ncclChar, left, h->comm, h->stream)); }
        NCCLCHECK(g_rccl.GroupEnd());
        return TTX_OK;
    }
    if (!h->have_cb) return fail(TTX_ESTATE, "world_size > 1 but neither ttx_comm_init nor ttx_set_transport was called");
    // Host transport, STREAM-ORDERED: device -> pinned copies, then a host function enqueued in the stream runs the two
    // sendrecv calls, then pinned -> device copies.  The host thread that called ttx_run does not wait: the control flow
    // (and therefore the pipelined sweep loop) is the same as with RCCL, only the primitive differs.
    char *sR = h->h_msg, *sL = h->h_msg + P.MSZ;
    if (right >= 0) HIPCHECK(hipMemcpyAsync(sR, outR, P.MSZ, hipMemcpyDeviceToHost, h->stream));
    if (left >= 0) HIPCHECK(hipMemcpyAsync(sL, outL, P.MSZ, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipLaunchHostFunc(h->stream, hostfn_xfer, h));
    if (left >= 0) HIPCHECK(hipMemcpyAsync(h->recvL, h->h_msg + 2 * P.MSZ, P.MSZ, hipMemcpyHostToDevice, h->stream));
    if (right >= 0) HIPCHECK(hipMemcpyAsync(h->recvR, h->h_msg + 3 * P.MSZ, P.MSZ, hipMemcpyHostToDevice, h->stream));
    return TTX_OK;
}
// all-reduce of `count` doubles from device buffer src into device buffer dst; op 0 = sum, 1 = max
static int allreduce_dev(ttx_engine *h, const double *src, double *dst, size_t count, int op)
{
    if (h->W == 1) return TTX_OK;                       // dst aliases src
    if (h->comm) { NCCLCHECK(g_rccl.AllReduce(src, dst, count, ncclDouble, op ? ncclMax : ncclSum, h->comm, h->stream)); return TTX_OK; }
    if (!h->have_cb) return fail(TTX_ESTATE, "world_size > 1 but neither ttx_comm_init nor ttx_set_transport was called");
    // each pending reduction gets its own pinned slot and descriptor (several may be enqueued before the first one runs)
    ttx_engine::RedJob *job = h->red_slot(count, op);
    if (!job) return fail(TTX_EHIP, "allreduce staging exhausted");
    HIPCHECK(hipMemcpyAsync(job->buf, src, sizeof(double) * count, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipLaunchHostFunc(h->stream, hostfn_allreduce, job));
    HIPCHECK(hipMemcpyAsync(dst, job->buf, sizeof(double) * count, hipMemcpyHostToDevice, h->stream));
    return TTX_OK;
}

// ---- launch helpers -------------------------------------------------------------------------------
static hipEvent_t ev_get(ttx_engine *h)
{
    if (!h->evpool.empty()) { hipEvent_t e = h->evpool.back(); h->evpool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
// brackets `n` consecutive launches of one kind with a single HIP event pair on the engine's stream
struct KScope {
    ttx_engine *h; int kind, n; hipEvent_t a = nullptr, b = nullptr;
    KScope(ttx_engine *h_, int kind_, int n_ = 1) : h(h_), kind(kind_), n(n_) { if (h->profile) { a = ev_get(h); b = ev_get(h); (void)hipEventRecord(a, h->stream); } }
    ~KScope() { if (h->profile) { (void)hipEventRecord(b, h->stream); h->evs.push_back({kind, n, a, b}); } else h->k_launches[kind] += n; }
};
static void ev_collect(ttx_engine *h)
{
    for (auto &e : h->evs) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e.a, e.b);
        h->k_ms[e.kind] += ms; h->k_launches[e.kind] += e.n;
        h->evpool.push_back(e.a); h->evpool.push_back(e.b);
    }
    h->evs.clear();
}

static double erank_host(const ttx_engine *h, const int32_t *r)
{
    // lib/tt.f90:1228-1245
    const int l = 1, m = h->d, d = m;
    if (d == 1) return 0.0;
    double s = 0.0;
    for (int i = l; i <= m; i++) s = s + r[i - 1] * h->n1[i] * r[i];
    if (s == 0.0) return s;
    int b = r[l - 1] * h->n1[l] + h->n1[m] * r[m];
    if (d == 2) return s / b;
    int a = 0;
    for (int i = l + 1; i <= m - 1; i++) a += h->n1[i];
    return (std::sqrt(b * b + 4.0 * a * s) - b) / (2.0 * a);
}

// Fortran Ew.d (the optional leading zero is dropped when the sign needs its place)
static std::string fmt_e(int w, int dgt, double v)
{
    char tmp[64], body[64];
    if (v != 0.0 && std::isfinite(v)) {
        snprintf(tmp, sizeof tmp, "%.*e", dgt - 1, v);
        char *e = strchr(tmp, 'e');
        int ex = atoi(e + 1) + 1;
        *e = 0;
        std::string digs;
        for (char *c = tmp; *c; c++) if (*c >= '0' && *c <= '9') digs.push_back(*c);
        snprintf(body, sizeof body, ".%sE%c%02d", digs.c_str(), ex < 0 ? '-' : '+', abs(ex));
    } else snprintf(body, sizeof body, ".%sE+00", std::string(dgt, '0').c_str());
    std::string s = body;
    const bool neg = std::signbit(v) && v != 0.0;
    if ((int)s.size() + 1 + (neg ? 1 : 0) <= w) s = "0" + s;
    if (neg) s = "-" + s;
    if ((int)s.size() < w) s = std::string(w - s.size(), ' ') + s;
    return s;
}

static void print_line(const ttx_engine *h, const ttx_sweep_rec &r, double val_prev)
{
    // lib/dmrgg.f90:971-1008 (printed by rank 0 only)
    if (h->wrank != 0) return;
    const char *sd = r.dir == 0 ? "::" : r.dir == 1 ? ">>" : "<<";
    printf("%3d%2s rank%5.1f time: %s n_evals: %10lld", r.it, sd, r.erank, fmt_e(9, 3, r.seconds).c_str(), (long long)r.neval);
    if (h->P.has_quad) {
        if (r.it == 0) printf(" val %s", fmt_e(20, 14, r.val).c_str());
        else if (h->cfg.has_tru) printf(" err %s val %s", fmt_e(8, 3, std::fabs(1.0 - r.val / h->cfg.tru)).c_str(), fmt_e(20, 14, r.val).c_str());
        else printf(" cnv %s val %s", fmt_e(8, 3, std::fabs(1.0 - r.val / val_prev)).c_str(), fmt_e(20, 14, r.val).c_str());
    }
    printf("\n");
    fflush(stdout);
}

// job-wide summary of the sweep -> h->h_sum (one all-reduce, one stream synchronisation)
static int readback(ttx_engine *h)
{
    hipLaunchKernelGGL(k_collect, dim3(1), dim3(256), 0, h->stream, h->P);
    {
        int rc = allreduce_dev(h, h->P.sumsend, h->P.sumrecv, h->SB, 0);
        if (rc) return rc;
        HIPCHECK(hipMemcpyAsync(h->h_sum, h->P.sumrecv, sizeof(double) * h->SB, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (h->cb_error) return fail(TTX_EHIP, "host transport: a sendrecv / allreduce callback failed");
    }
    if (h->profile) ev_collect(h);
    return TTX_OK;
}
static inline const double *sum_bond(const ttx_engine *h, int p) { return h->h_sum + SUM_HDR + h->cfg.nproc + 5 * (size_t)p; }

// bond ranks as the owning groups hold them (valid after readback)
static std::vector<int32_t> global_ranks(const ttx_engine *h)
{
    std::vector<int32_t> r(h->d + 1, 1);
    for (int p = 1; p < h->d; p++) r[p] = (int32_t)sum_bond(h, p)[0];
    return r;
}
// The reference prints erank(arg) of rank 0, whose knowledge of a bond owned by rank g lags g-1 sweeps behind
// (the pivot tape hops one rank per sweep, lib/dmrgg.f90:768-850).  The engine ships pivots only to direct
// neighbours, so the lag is reproduced here from the per-sweep history of accepted pivots.
static std::vector<int32_t> rank0_view(ttx_engine *h, int it)
{
    const int d = h->d;
    if (it == 0) { h->updhist.clear(); return std::vector<int32_t>(d + 1, 1); }
    std::vector<uint8_t> u(d + 1, 0);
    for (int p = 1; p < d; p++) u[p] = sum_bond(h, p)[1] > 0.0;
    h->updhist.push_back(u);
    std::vector<int32_t> r(d + 1, 1);
    for (int g = 0; g < h->cfg.nproc; g++) {
        int lag = std::max(0, g - 1), upto = it - lag;       // sweeps 1..upto of group g's bonds are known to rank 0
        for (int p = h->own[g]; p < h->own[g + 1]; p++)
            for (int t = 1; t <= upto; t++) r[p] += h->updhist[t - 1][p];
    }
    return r;
}

// quadrature of the current cores: per-core matrices, per-group chains, gather over GPUs, tree
static int launch_quad(ttx_engine *h, int mode, const double *w)
{
    DevProb &P = h->P;
    const size_t lds_q = sizeof(double) * ((size_t)h->RM * h->RM + 2);
    hipLaunchKernelGGL(k_quad_build, dim3(h->NC, h->G), dim3(256), lds_q, h->stream, P, mode, w);
    hipLaunchKernelGGL(k_quad_chain, dim3(h->G), dim3(256), P.qscr ? 0 : 2 * lds_q, h->stream, P);
    if (h->cfg.nproc > 1) {
        int rc = allreduce_dev(h, P.qsend, P.qall, h->QB, 0);
        if (rc) return rc;
        hipLaunchKernelGGL(k_quad_tree, dim3(1), dim3(256), 0, h->stream, P);
    }
    return TTX_OK;
}

// raise a kernel's dynamic-LDS ceiling to `need` bytes if no earlier launch asked for as much (`cur`: the caller's record)
static int ensure_lds(const void *fn, size_t need, size_t &cur)
{
    if (need <= cur) return TTX_OK;
    HIPCHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    cur = need;
    return TTX_OK;
}

// the whole-sweep cluster kernel: cooperative launch (the runtime guarantees -- or refuses -- co-residency of the grid)
static int launch_cluster(ttx_engine *h, int dir, int epoch)
{
    DevProb P = h->P;
    int nsteps = h->nbmax, NB = h->cluster, ldsinv = h->cluster_ldsinv, zkeep = h->cluster_zkeep;
    const dim3 grid(8 * h->cluster * ((h->G + 7) / 8)), block(CB);
    if (h->cluster_coop) {
        void *args[] = {&P, &dir, &nsteps, &NB, &ldsinv, &epoch, &zkeep};
        HIPCHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void *>(k_sweep_cluster), grid, block, args, (unsigned)h->lds_cluster, h->stream));
    } else
        hipLaunchKernelGGL(k_sweep_cluster, grid, block, h->lds_cluster, h->stream, P, dir, nsteps, NB, ldsinv, epoch, zkeep);
    return TTX_OK;
}

template <int FUN>
static int run_impl(ttx_engine *h)
{
    using clk = std::chrono::steady_clock;
    const auto t0 = clk::now();
    auto since = [&]() { return std::chrono::duration<double>(clk::now() - t0).count(); };
    DevProb &P = h->P;
    const int d = h->d, G = h->G, nproc = h->cfg.nproc;
    hipStream_t st = h->stream;
    h->recs.clear(); h->tapes.clear();
    int rc;

    // kernels that may stage more than the default 64 KB of dynamic LDS (160 KB per CU on gfx950)
    {   // the dynamic-LDS ceiling is a property of the FUNCTION: raise it when an engine needs more than any before it
        static size_t a_half = 0, a_lot = 0, a_fused = 0, a_cluster = 0;        // per instantiation (FUN) of this template
        if ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep<FUN>), h->lds_half, a_half))) return rc;
        if ((rc = ensure_lds(reinterpret_cast<const void *>(k_lottery<FUN>), h->lds_lot, a_lot))) return rc;
        if (h->fused && (rc = ensure_lds(reinterpret_cast<const void *>(k_sweep_fused), h->lds_fused, a_fused))) return rc;
        if (h->cluster && (rc = ensure_lds(reinterpret_cast<const void *>(k_sweep_cluster), h->lds_cluster, a_cluster))) return rc;
        static size_t a_de0 = 0, a_de1 = 0;
        static size_t a_dec = 0;
        static size_t a_dlc = 0, a_dlp = 0;
        if (h->de_v2 && P.de_cut && ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_dec), h->lds_de, a_dec)) ||
                                     (rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_dec), h->lds_de, a_dlc)) ||
                                     (rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_decp), sizeof(double) * (2 * (size_t)(d + 64) + 2048), a_dlp)))) return rc;
        if (h->de_v2 && ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_de<true>), h->lds_de, a_de0)) ||
                         (rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_de<false>), h->lds_de, a_de1)))) return rc;
        static size_t a_dt0 = 0, a_dt1 = 0;
        if (h->de_team && ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_det<true, 3>), h->lds_det, a_dt0)) ||
                           (rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_det<false, 3>), h->lds_det, a_dt1)))) return rc;
        static size_t a_dt2 = 0, a_dt3 = 0;
        if (h->de_team && ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_det<true, 1>), h->lds_det6, a_dt2)) ||
                           (rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_det<false, 1>), h->lds_det6, a_dt3)))) return rc;
        static size_t a_d50 = 0, a_d51 = 0;
        if (h->de_v5 && ((rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_de5<true>), h->lds_de5, a_d50)) ||
                         (rc = ensure_lds(reinterpret_cast<const void *>(k_halfstep_de5<false>), h->lds_de5, a_d51)))) return rc;
        static size_t a_lr0 = 0, a_lr1 = 0;
        static size_t a_lr2 = 0, a_lr3 = 0;
        if (h->lot_rows && ((rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_de_rows<true, false>), h->lds_der, a_lr0)) ||
                            (rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_de_rows<false, false>), h->lds_der, a_lr1)) ||
                            (rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_de_rows<true, true>), h->lds_der, a_lr2)) ||
                            (rc = ensure_lds(reinterpret_cast<const void *>(k_lottery_eval_de_rows<false, true>), h->lds_der, a_lr3)))) return rc;
    }
    if (h->cluster) *h->h_abort = 0;
    // an evaluating kernel: once with the device integrand; with a host integrand twice around the host's calls
    auto EV = [&](auto &&launch) -> int {
        if (FUN != FUN_HOST) { launch(P); return TTX_OK; }
        DevProb Q = P;
        Q.hostpass = 1; launch(Q);
        if (int rc_ = host_eval(h)) return rc_;
        Q.hostpass = 2; launch(Q);
        return TTX_OK;
    };
    // ---- reset state (lib/dmrgg.f90:96-100, 141-148, 279-288) ----
    hipLaunchKernelGGL(k_reset, dim3(64), dim3(256), 0, st, P, h->SB, h->QB);
    // ---- initial cross (:151-301) ----
    const int smin = 8, snum = std::max(smin, nproc);
    int nn = h->n1[1];
    for (int k = 2; k <= d; k++) nn = std::min(nn, h->n1[k]);
    {
        KScope ks(h, TTX_K_OTHER, 4);
        const size_t lds_s = h->lds_par + 16 + sizeof(short) * 256 * (size_t)(((d + 7) & ~7) + 8);
        const int srows = lds_s <= 150 * 1024 ? 1 : 0;
        static size_t a_samp = 0;
        if (srows && (rc = ensure_lds(reinterpret_cast<const void *>(k_init_samples<FUN>), lds_s, a_samp))) return rc;
        if ((rc = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_init_samples<FUN>, dim3(G), dim3(256), srows ? lds_s : h->lds_par, st, Q, snum, nn, FUN == FUN_HOST ? 0 : srows, 0); }))) return rc;
        if ((rc = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_init_fibers<FUN>, dim3(h->NC, G), dim3(256), h->lds_par, st, Q); }))) return rc;
        hipLaunchKernelGGL(k_init_factors, dim3(h->NC, G), dim3(256), (P.arith && P.fpersist) ? sizeof(double) * 4 * (d + 8) : 0, st, P);
        hipLaunchKernelGGL(k_init_final, dim3(G), dim3(256), 0, st, P);
    }
    // Single-process whole-sweep path: the first sweep kernel is enqueued right behind the initial cross, and the host
    // waits only for the copy of the initial summary (an event), not for the stream.
    const bool pipe0 = (h->cluster || h->fused) && h->W == 1 && !h->profile;
    bool head1 = false;
    if (pipe0) {
        hipLaunchKernelGGL(k_collect, dim3(1), dim3(256), 0, st, P);
        h->h_sum = h->h_sum_base;
        HIPCHECK(hipMemcpyAsync(h->h_sum, P.sumrecv, sizeof(double) * h->SB, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipEventRecord(h->ev_sum[0], st));
        if (1 < h->cfg.maxrank) {
            if (h->cluster) { if ((rc = launch_cluster(h, 1, 1))) return rc; }
            else hipLaunchKernelGGL(k_sweep_fused, dim3(G), dim3(FB), h->lds_fused, st, P, 1, h->nbmax);
            h->k_launches[TTX_K_HALFSTEP] += 1;
            head1 = true;
        }
        HIPCHECK(hipEventSynchronize(h->ev_sum[0]));
    } else if ((rc = readback(h))) return rc;
    HIPCHECK(hipGetLastError());
    double val = 1.0, val_prev = 1.0;
    for (int g = 0; g < nproc; g++) val = (g == 0) ? h->h_sum[SUM_HDR + g] : val * h->h_sum[SUM_HDR + g];   // :259-267 PROD
    if (!P.has_quad) val = 0.0;
    val_prev = val;
    {
        ttx_sweep_rec r{};
        r.it = 0; r.dir = 0; r.erank = erank_host(h, rank0_view(h, 0).data()); r.neval = (int64_t)h->h_sum[SUM_NEVAL]; r.val = val;
        r.amax = h->h_sum[SUM_AMAX]; r.pivotmax = -1; r.pivotmin = -1; r.seconds = since();
        h->recs.push_back(r);
        if (h->cfg.verbose) print_line(h, r, val_prev);
    }

    // ---- main loop (:309-1020) ----
    int it = 0, strike = 0;
    bool ready = (it + 1 >= h->cfg.maxrank);
    const int nfb = (h->RM * h->NM + TTX_BLK - 1) / TTX_BLK;
    const size_t lds_acc = sizeof(double) * std::max<size_t>(h->RM + 2, (size_t)std::min<int>(h->RM, 64) * std::min<int>(h->RM, 64));   // roles A/B: x[RM]; C/D: the staged LU
    // Pipelined mode (whole-sweep kernels, one process): the stopping rule also runs on the device (k_sweep_end), so
    // sweep it+1 is enqueued BEFORE the host has read the summary of sweep it -- the GPU never waits for the host.  When the rule fires, the one sweep that is already enqueued finds the stop flag and does nothing.
    // On a single GPU the per-sweep quadrature (only reported, never fed back) runs on its own stream next to the
    // following sweep: it reads a snapshot of the ranks and only slabs that already exist (appends are in place).
    // With several processes the exchange and the summary travel by stream-ordered collectives (RCCL, or the host transport's
    // host functions), so the same loop applies; only the fork of the quadrature is single-process (one communicator must
    // not be driven from two streams at once).
    const bool pipe = (h->cluster || h->fused) && !h->profile && !(getenv("TTX_PIPELINE") && atoi(getenv("TTX_PIPELINE")) == 0);
    const bool forkq = pipe && P.has_quad && h->W == 1;
    DevProb Pq = P;
    if (pipe) Pq.r = P.rq;

    // part 1: the sweep over the own bonds; part 2: exchange, end-of-sweep work, quadrature; 3: both
    auto enqueue_sweep = [&](int it_, int part) -> int {
        const int dir = 2 - it_ % 2, slot = it_ & 1;
        if (part & 1) {
        if (h->cluster) {
            KScope ks(h, TTX_K_HALFSTEP, 1);
            if (int rc_ = launch_cluster(h, dir, it_)) return rc_;
        } else if (h->fused) {
            KScope ks(h, TTX_K_HALFSTEP, 1);
            hipLaunchKernelGGL(k_sweep_fused, dim3(G), dim3(FB), h->lds_fused, st, P, dir, h->nbmax);
        }
        for (int pp = 1; pp <= h->nbmax && !h->fused && !h->cluster; pp++) {
            if (P.deTL) {   // Ising D/E: pair factors of this bond that do not span it (shared by all elements through a pivot)
                KScope ks(h, TTX_K_OTHER);
                if (P.de_cut) hipLaunchKernelGGL(k_de_ctables, dim3(2 * h->RM, G), dim3(256), sizeof(double) * (size_t)(d + 2), st, P, dir, pp);
                else hipLaunchKernelGGL(k_de_tables, dim3((2 * (d + 1) * h->RM + 255) / 256, G), dim3(256), 0, st, P, dir, pp);
            }
            const bool fastk = P.arith && (FUN == FUN_MVN || (FUN == FUN_ISING && P.ising_id != 1));
            if (fastk && !P.fpersist) {    // TTX_ARITH=fast, mvn: per-pivot tables of this bond step (ttx_fast.h; Ising D/E keeps its tables per bond)
                KScope ks(h, TTX_K_OTHER);
                hipLaunchKernelGGL(k_fast_tables<FUN>, dim3(2 * h->RM, G), dim3(64), sizeof(double) * 2 * (d + 8), st, P, dir, pp);
            }
            if (h->cfg.pivoting >= 0) {
                if (fastk) {
                    KScope ks(h, TTX_K_LOTTERY);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(P.lot_nb, G), dim3(P.lot_nb == 1 ? 512 : 256), h->lds_lot, st, P, dir, pp, h->fast_cap, 0);
                } else if (FUN == FUN_MVN && h->mvn_v2) {
                    KScope ks(h, TTX_K_LOTTERY, 3);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 1);
                    hipLaunchKernelGGL(k_lottery_eval_mvn, dim3(P.lot_max, G), dim3(64), h->lds_mvn, st, P);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 2);
                } else if (FUN == FUN_ISING && P.de_cut && h->de_v2 && P.lotc) {
                    // compact tables: one wave per candidate between the drawing and the scoring launch
                    KScope ks(h, TTX_K_LOTTERY, 3);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 1);
                    // one wave per candidate: row-parallel without tables up to d = 160 (measured: 23 % less lottery time at D_64, even at
                    // D_256), from the compact tables beyond (TTX_DE_LOT_POINT=0/1 forces one)
                    if (h->de_lot_point) hipLaunchKernelGGL(k_lottery_eval_decp, dim3(P.lot_max, G), dim3(64), sizeof(double) * (2 * (size_t)(d + 64) + 2048), st, P);
                    else hipLaunchKernelGGL(k_lottery_eval_dec, dim3(P.lot_max, G), dim3(64), h->lds_de, st, P);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 2);
                } else if (FUN == FUN_ISING && h->lot_wave) {
                    KScope ks(h, TTX_K_LOTTERY, 3);
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 1);
                    {
                        const dim3 gr((P.lot_max + 3) / 4, G);
                        if (h->lot_rows == 2) {         // TTX_LOTTERY_ROWS=2: with the pivots' factor tables (measured slower: HBM latency)
                            if (P.de_unit) hipLaunchKernelGGL((k_lottery_eval_de_rows<true, true>), gr, dim3(64), h->lds_der, st, P);
                            else hipLaunchKernelGGL((k_lottery_eval_de_rows<false, true>), gr, dim3(64), h->lds_der, st, P);
                        } else if (P.de_unit) hipLaunchKernelGGL((k_lottery_eval_de_rows<true, false>), gr, dim3(64), h->lds_der, st, P);
                        else hipLaunchKernelGGL((k_lottery_eval_de_rows<false, false>), gr, dim3(64), h->lds_der, st, P);
                    }
                    hipLaunchKernelGGL(k_lottery<FUN>, dim3(1, G), dim3(512), h->lds_lot, st, P, dir, pp, h->lot_vals, 2);
                } else
                { KScope ks(h, TTX_K_LOTTERY); if (int rc_ = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_lottery<FUN>, dim3(P.lot_nb, G), dim3(P.lot_nb == 1 ? 512 : 256), h->lds_lot, st, Q, dir, pp, h->lot_vals, 0); })) return rc_; }
                KScope ks(h, TTX_K_HALFSTEP, h->H);
                if (fastk) {
                    for (int hh = 0; hh < h->H; hh++) hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G), dim3(TTX_BLK), h->lds_half, st, P, hh, dir, h->mode, 0);
                } else if (FUN == FUN_MVN && h->mvn_v2) {
                    for (int hh = 0; hh < h->H; hh++) hipLaunchKernelGGL(k_halfstep_mvn, dim3(h->de_slots, G), dim3(64), h->lds_mvn, st, P, hh, dir, h->mode);
                } else if (FUN == FUN_ISING && h->de_v2) {
                    // while the ranks are small (at most it_ + 1 during sweep it_) a unit gets a team of 14 waves on a CU of its own, up to
                    // 1024 units a team of 6 waves (three such teams fit a CU), beyond that one wave
                    const int rb = std::min((int)h->RM, it_ + 1);
                    const int team_slots = (h->de_test_fault && it_ == h->de_test_fault) ? 1 : rb * ((h->NM + 63) / 64);
                    const bool team = h->de_team && !h->de_v5 && team_slots * G <= h->de_team_units;
                    const bool team6 = h->de_team && !h->de_v5 && !team && team_slots * G <= h->de_team6_units;
                    for (int hh = 0; hh < h->H; hh++) {
                        if (team) {
                            if (P.de_unit) hipLaunchKernelGGL((k_halfstep_det<true, 3>), dim3(team_slots, G), dim3(64 * 14), h->lds_det, st, P, hh, dir, h->mode);
                            else hipLaunchKernelGGL((k_halfstep_det<false, 3>), dim3(team_slots, G), dim3(64 * 14), h->lds_det, st, P, hh, dir, h->mode);
                        } else if (team6) {
                            if (P.de_unit) hipLaunchKernelGGL((k_halfstep_det<true, 1>), dim3(team_slots, G), dim3(64 * 6), h->lds_det6, st, P, hh, dir, h->mode);
                            else hipLaunchKernelGGL((k_halfstep_det<false, 1>), dim3(team_slots, G), dim3(64 * 6), h->lds_det6, st, P, hh, dir, h->mode);
                        } else if (h->de_v5) {
                            if (P.de_unit) hipLaunchKernelGGL(k_halfstep_de5<true>, dim3(h->de_slots, G), dim3(64 * DE5_W), h->lds_de5, st, P, hh, dir, h->mode);
                            else hipLaunchKernelGGL(k_halfstep_de5<false>, dim3(h->de_slots, G), dim3(64 * DE5_W), h->lds_de5, st, P, hh, dir, h->mode);
                        } else if (P.de_cut) hipLaunchKernelGGL(k_halfstep_dec, dim3(h->de_slots, G), dim3(64), h->lds_de, st, P, hh, dir, h->mode);
                        else if (P.de_unit) hipLaunchKernelGGL(k_halfstep_de<true>, dim3(h->de_slots, G), dim3(64), h->lds_de, st, P, hh, dir, h->mode);
                        else hipLaunchKernelGGL(k_halfstep_de<false>, dim3(h->de_slots, G), dim3(64), h->lds_de, st, P, hh, dir, h->mode);
                    }
                } else
                for (int hh = 0; hh < h->H; hh++)
                    if (int rc_ = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G), dim3(TTX_BLK), h->lds_half, st, Q, hh, dir, h->mode, h->half_vals); })) return rc_;
            } else {
                // full pivoting (:341-408): every superblock column through the half-step kernel, global arg-max,
                // then the cross through the winner (evaluation only)
                KScope ks(h, TTX_K_HALFSTEP, 5);
                hipLaunchKernelGGL(k_bond_begin, dim3(G), dim3(64), 0, st, P, dir, pp);
                if (P.fp_mfma && FUN != FUN_HOST) {
                    // one dense step: evaluate the superblock once, residual by fp64 MFMA fused with the arg-max
                    const int side = h->RM * h->NM, gx = (side + 63) / 64;
                    hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G, h->NM * h->RM), dim3(TTX_BLK), h->lds_half, st, P, 0, dir, 4, fastk ? 0 : h->half_vals);
                    hipLaunchKernelGGL(k_full_gemm_argmax, dim3(gx, gx, G), dim3(256), 0, st, P);
                    hipLaunchKernelGGL(k_full_resolve2, dim3(G), dim3(256), 0, st, P, gx, gx);
                } else if (FUN == FUN_HOST) {
                    // the user's `fun` with full pivoting (:341-408 works with any fun): one superblock column (k,q) per launch pair --
                    // pass 1 hands the column's multi-indices to the host, pass 2 takes the values, residual and partial arg-max; the
                    // ranks are at most it_ + 1 in sweep it_, columns beyond n2 r2 return at once
                    const int zmax = h->NM * std::min((int)h->RM, it_ + 1);
                    for (int z = 0; z < zmax; z++) {
                        DevProb Q = P;
                        Q.zbase = z;
                        Q.hostpass = 1;
                        hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G, 1), dim3(TTX_BLK), h->lds_half, st, Q, 0, dir, 3, 0);
                        if (int rc_ = host_eval(h)) return rc_;
                        Q.hostpass = 2;
                        hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G, 1), dim3(TTX_BLK), h->lds_half, st, Q, 0, dir, 3, 0);
                    }
                    hipLaunchKernelGGL(k_full_resolve, dim3(G), dim3(256), 0, st, P);
                } else {
                hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G, h->NM * h->RM), dim3(TTX_BLK), h->lds_half, st, P, 0, dir, 3, fastk ? 0 : h->half_vals);
                hipLaunchKernelGGL(k_full_resolve, dim3(G), dim3(256), 0, st, P);
                }
                if (int rc_ = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G), dim3(TTX_BLK), h->lds_half, st, Q, 0, dir, 2, fastk ? 0 : h->half_vals); })) return rc_;
                if (int rc_ = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_halfstep<FUN>, dim3(nfb, G), dim3(TTX_BLK), h->lds_half, st, Q, 1, dir, 2, fastk ? 0 : h->half_vals); })) return rc_;
            }
            { KScope ks(h, TTX_K_ACCEPT); hipLaunchKernelGGL(k_accept, dim3(2 * nfb + 2 * h->NM + 1, G), dim3(TTX_BLK), lds_acc, st, P, h->H, nfb); }
        }
        }
        if (!(part & 2)) return TTX_OK;
        if (forkq) HIPCHECK(hipStreamWaitEvent(st, h->ev_val[slot ^ 1], 0));   // the previous quadrature is done with the boundaries
        {   // per-sweep exchange between bond groups (:763-961)
            KScope ks(h, TTX_K_EXCHANGE, (h->W > 1 ? 4 : 2) + (nproc > 1 ? 1 : 0));
            if (!h->cluster) hipLaunchKernelGGL(k_exch_pack, dim3(G), dim3(256), 0, st, P);     // the cluster kernel packs at its end
            if (h->W > 1) {
                hipLaunchKernelGGL(k_exch_localmax, dim3(1), dim3(64), 0, st, P);
                if (int rc_ = xfer_neighbours(h)) return rc_;
                if (int rc_ = allreduce_dev(h, P.redsend, P.redrecv, 4, 1)) return rc_;
            }
            hipLaunchKernelGGL(k_exch_max_apply, dim3(G), dim3(256), (P.arith && P.fpersist) ? sizeof(double) * 4 * (d + 8) : 0, st, P, h->W > 1 ? 1 : 0, nproc > 1 ? 1 : 0);
            if (nproc > 1) {
                const size_t VSb = ((d + 7) & ~7) + 8;
                const size_t lds_b = h->lds_par + 16 + sizeof(short) * 2 * VSb + sizeof(double) * (64 * 64 + 4) +
                                     (P.bnd_wave ? sizeof(double) * (std::max<size_t>(2 * (size_t)de_rows_stride(d), 3 * (size_t)d + 2) + 8 + (P.de_cut ? 64 * 32 : 0)) : 0);    // D/E: two value rows; mvn: 3 d + 1 doubles
                static size_t a_bnd = 0;
                if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_exch_boundary<FUN>), lds_b, a_bnd)) return rc_;
                if (int rc_ = EV([&](const DevProb &Q) { hipLaunchKernelGGL(k_exch_boundary<FUN>, dim3(2 * h->NM, G), dim3(TTX_BLK), lds_b, st, Q); })) return rc_;
            }
        }
        if (pipe && h->W == 1) hipLaunchKernelGGL(k_sweep_end, dim3(1), dim3(256), 0, st, P, it_, h->h_sum_base + (size_t)slot * h->SB);
        if (pipe && h->W > 1) {      // this GPU's part -> SUM all-reduce -> pinned slot, all stream-ordered
            hipLaunchKernelGGL(k_sweep_end, dim3(1), dim3(256), 0, st, P, it_, P.sumsend);
            if (int rc_ = allreduce_dev(h, P.sumsend, P.sumrecv, h->SB, 0)) return rc_;
            HIPCHECK(hipMemcpyAsync(h->h_sum_base + (size_t)slot * h->SB, P.sumrecv, sizeof(double) * h->SB, hipMemcpyDeviceToHost, st));
        }
        if (pipe) HIPCHECK(hipEventRecord(h->ev_sum[slot], st));     // the summary of the sweep is in the pinned slot
        if (P.has_quad) {
            hipStream_t sq = forkq ? h->qstream : st;
            if (forkq) HIPCHECK(hipStreamWaitEvent(sq, h->ev_sum[slot], 0));    // fork point = end of the sweep's main-stream work
            KScope ks(h, TTX_K_QUAD, nproc > 1 ? 3 : 2);
            const size_t lds_q = sizeof(double) * ((size_t)h->RM * h->RM + 2);
            hipLaunchKernelGGL(k_quad_build, dim3(h->NC, G), dim3(256), lds_q, sq, pipe ? Pq : P, 0, P.quadw);
            hipLaunchKernelGGL(k_quad_chain, dim3(G), dim3(256), P.qscr ? 0 : 2 * lds_q, sq, pipe ? Pq : P);
            if (nproc > 1) {
                if (int rc_ = allreduce_dev(h, P.qsend, P.qall, h->QB, 0)) return rc_;     // W > 1: sq is the main stream
                hipLaunchKernelGGL(k_quad_tree, dim3(1), dim3(256), 0, sq, pipe ? Pq : P);
            }
            if (pipe) {
                HIPCHECK(hipMemcpyAsync(h->h_val + slot, &P.gs[0].val, sizeof(double), hipMemcpyDeviceToHost, sq));
                HIPCHECK(hipEventRecord(h->ev_val[slot], sq));
            }
            if (h->cb_error) return fail(TTX_EHIP, "host transport: a sendrecv / allreduce callback failed");
        }
        return TTX_OK;
    };
    // ---- finalise (:1029): dtt_lua shifts the rightmost inv of every group to its neighbour first.  Enqueued once; in the
    //      pipelined loop as soon as the stopping rule has fired on the host (after_val >= 0: the slot whose forked quadrature
    //      still reads the cores the finalisation overwrites -- a stream-side wait, the host does not idle) ----
    bool fin_enqueued = false;
    auto enqueue_final = [&](int after_val) -> int {
        if (fin_enqueued) return TTX_OK;
        fin_enqueued = true;
        if (after_val >= 0) HIPCHECK(hipStreamWaitEvent(st, h->ev_val[after_val], 0));
        HIPCHECK(hipMemsetAsync(P.ctl, 0, sizeof(int) * 3, st));      // ctl[3] (fault counter of the team / relay half-steps) is read by ttx_run afterwards
        KScope ks(h, TTX_K_OTHER, 2);
        if (nproc > 1) {
            hipLaunchKernelGGL(k_exch_pack, dim3(G), dim3(256), 0, st, P);
            if (int rc_ = xfer_neighbours(h)) return rc_;
            hipLaunchKernelGGL(k_exch_apply, dim3(G), dim3(256), (P.arith && P.fpersist) ? sizeof(double) * 4 * (d + 8) : 0, st, P);
        }
        // threads (= columns) per workgroup: as many as fit the LDS next to the LU panel (256, 128 or 64); the columns of a core are
        // spread over grid.z (at maxrank 64 the 256-thread staging did not fit: the kernels then ran out of L2 with one workgroup per
        // core, 6 ms at D_256)
        int ft = 256;
        while (ft > 64 && sizeof(double) * ((size_t)h->RM * h->RM + (size_t)ft * h->RM) > 96 * 1024) ft >>= 1;
        const size_t lds_f = sizeof(double) * ((size_t)h->RM * h->RM + (size_t)ft * h->RM);
        const int fl = lds_f <= 150 * 1024 ? 1 : 0;
        const int fz = std::max(1, std::min(64, (h->NM * (int)h->RM + ft - 1) / ft));
        if (fl) {
            static size_t a_luar = 0, a_lual = 0;
            if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_fin_luar), lds_f, a_luar)) return rc_;
            if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_fin_lual), lds_f, a_lual)) return rc_;
        }
        hipLaunchKernelGGL(k_fin_luar, dim3(h->NC, G, fz), dim3(ft), fl ? lds_f : 0, st, P, fl);
        hipLaunchKernelGGL(k_fin_lual, dim3(h->NC, G, fz), dim3(ft), fl ? lds_f : 0, st, P, fl);
        return TTX_OK;
    };
    // host side of a finished sweep: record, tapes, log line, stopping rule (identical to k_sweep_end)
    auto process_sweep = [&](int it_) -> int {
        const int dir = 2 - it_ % 2, slot = it_ & 1;
        if (pipe) {
            HIPCHECK(hipEventSynchronize(h->ev_sum[slot]));
            h->h_sum = h->h_sum_base + (size_t)slot * h->SB;
            {   // the rule needs only the summary: if it fires, the finalisation goes out before the host waits for the value
                bool rdy = (it_ + 1 >= h->cfg.maxrank);
                if (h->cfg.accuracy >= 0.0) rdy = rdy || (((h->h_sum[SUM_PMAX] <= h->cfg.accuracy * h->h_sum[SUM_AMAX]) ? strike + 1 : 0) >= 3);
                if (rdy) { if (int rc_ = enqueue_final(P.has_quad ? slot : -1)) return rc_; }
            }
            if (P.has_quad) { HIPCHECK(hipEventSynchronize(h->ev_val[slot])); val = h->h_val[slot]; }
        } else {
            if (int rc_ = readback(h)) return rc_;
            if (P.has_quad) val = h->h_sum[SUM_VAL];      // every GPU ran the same tree on the same gathered matrices
        }
        HIPCHECK(hipGetLastError());
        if (h->cluster && *(volatile int *)h->h_abort) { h->cluster_aborted = true; return fail(TTX_EHIP, "cluster sweep kernel: barrier timed out (workgroups of a bond group were not co-resident)"); }
        ttx_sweep_rec r{};
        r.it = it_; r.dir = dir; r.erank = erank_host(h, rank0_view(h, it_).data()); r.neval = (int64_t)h->h_sum[SUM_NEVAL]; r.val = val;
        r.amax = h->h_sum[SUM_AMAX]; r.pivotmax = h->h_sum[SUM_PMAX]; r.pivotmin = h->h_sum[SUM_PMIN]; r.seconds = since();
        h->recs.push_back(r);
        {
            size_t o = h->tapes.size();
            h->tapes.resize(o + (size_t)(d + 1) * 4, -1);
            for (int p = 1; p < d; p++) for (int x = 0; x < 4; x++) h->tapes[o + (size_t)p * 4 + x] = (int32_t)sum_bond(h, p)[1 + x];
        }
        if (h->cfg.verbose) print_line(h, r, val_prev);
        val_prev = val;
        ready = ready || (it_ + 1 >= h->cfg.maxrank);                             // :1011
        if (h->cfg.accuracy >= 0.0) {                                             // :1012-1019 (rank 0's amax, global pivotmax)
            if (r.pivotmax <= h->cfg.accuracy * r.amax) strike++; else strike = 0;
            ready = ready || (strike >= 3);
        }
        return TTX_OK;
    };
    if (!pipe) {
        while (!ready) {
            it++;
            if ((rc = enqueue_sweep(it, 3))) return rc;
            if ((rc = process_sweep(it))) return rc;
        }
    } else if (!ready) {
        // only the sweep kernel of it+1 is enqueued speculatively (it keeps the GPU busy while the host reads the summary of
        // sweep it); once the rule has fired that one launch finds the stop flag and exits, and its tail is never enqueued
        if (forkq) { HIPCHECK(hipEventRecord(h->ev_val[0], h->qstream)); HIPCHECK(hipEventRecord(h->ev_val[1], h->qstream)); }
        if (!head1 && (rc = enqueue_sweep(1, 1))) return rc;
        for (it = 1; !ready; it++) {
            if ((rc = enqueue_sweep(it, 2))) return rc;
            if (it + 1 < h->cfg.maxrank && (rc = enqueue_sweep(it + 1, 1))) return rc;
            if ((rc = process_sweep(it))) return rc;
        }
        it--;
    }
    if ((rc = enqueue_final(-1))) return rc;
    if ((rc = readback(h))) return rc;
    HIPCHECK(hipGetLastError());
#ifdef TTX_STAMPS
    if (P.dbg) {   // per-wave cycle stamps of one bond step of the cluster kernel (WST in ttx_cluster.h), relative to the earliest one
        std::vector<long long> hb(8 * 64 * 8);
        HIPCHECK(hipMemcpy(hb.data(), P.dbg, sizeof(long long) * hb.size(), hipMemcpyDeviceToHost));
        long long t0 = 0;
        for (long long v : hb) if (v && (!t0 || v < t0)) t0 = v;
        for (int hh = 0; hh < 8; hh++) {
            bool any = false;
            for (int w = 0; w < 64; w++) if (hb[((size_t)hh * 64 + w) * 8]) any = true;
            if (!any) continue;
            fprintf(stderr, "half-step %d (cycles since the first stamp; per wave: start eval-done reduced published polled reduced2)\n", hh);
            for (int w = 0; w < 64; w++) {
                const long long *e = &hb[((size_t)hh * 64 + w) * 8];
                if (!e[0]) continue;
                fprintf(stderr, "  blk %d wave %d: %7lld %7lld %7lld %7lld %7lld %7lld\n", w / 4, w % 4, e[0] - t0, e[1] - t0, e[2] - t0, e[3] - t0, e[4] - t0, e[5] - t0);
            }
        }
    }
    {   // debug build: average wall_clock64 ticks (10 ns) per phase of the lottery (0) and half-step (1) kernels
        GroupState g0s;
        HIPCHECK(hipMemcpy(&g0s, P.gs, sizeof(GroupState), hipMemcpyDeviceToHost));
        for (int k = 0; k < 2; k++) {
            fprintf(stderr, "stamps kernel %d (n=%lld):", k, g0s.nstamp[k]);
            for (int x = 0; x < 16; x++) fprintf(stderr, " %.2fus", g0s.nstamp[k] ? 0.01 * (double)g0s.stamp[k][x] / (double)g0s.nstamp[k] : 0.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    h->neval = (int64_t)h->h_sum[SUM_NEVAL];
    h->k_bytes[TTX_K_HALFSTEP] += h->h_sum[SUM_BYTES];
    h->n_resid = (int64_t)h->h_sum[SUM_NRESID];
    h->rfinal = global_ranks(h);
    h->seconds = since();
    h->ran = true;
    return TTX_OK;
}

extern "C" int ttx_run(ttx_engine *h)
{
    if (!h) return fail(TTX_EINVAL, "ttx_run: null handle");
    HIPCHECK(hipSetDevice(h->cfg.device));
    if (h->cfg.fun_id == 0) return fail(TTX_ESTATE, "ttx_run: this engine holds a loaded tensor train and has no integrand");
    for (int k = 0; k < TTX_K_NKINDS; k++) { h->k_launches[k] = 0; h->k_ms[k] = 0; h->k_bytes[k] = 0; }
    switch (h->cfg.fun_id) {
        case TTX_FUN_ISING: {
            h->cluster_aborted = false;
            int rc = run_impl<FUN_ISING>(h);
            if (rc && h->cluster_aborted) {
                // A wait inside the cluster kernel timed out.  Nothing of the run is kept: drain both streams (kernels behind
                // the aborted one see ctl[0] and do nothing), retire the cluster path for this engine and replay the run on
                // the multi-kernel chain -- all sweep implementations produce the identical result.
                (void)hipStreamSynchronize(h->stream);
                (void)hipStreamSynchronize(h->qstream);
                (void)hipGetLastError();
                *h->h_abort = 0;
                h->cluster = 0; h->cluster_aborted = false; h->cluster_fallbacks++;
                if (h->P.ising_id == 1) h->P.arith = 0;        // the closed form of Ising C lives in the cluster kernel only
                for (int k = 0; k < TTX_K_NKINDS; k++) { h->k_launches[k] = 0; h->k_ms[k] = 0; h->k_bytes[k] = 0; }
                rc = run_impl<FUN_ISING>(h);
            }
            if (h->de_team && !h->de_v5) {
                // k_halfstep_det found more units than the grid the host sized from its bound on the ranks (never expected):
                // nothing of the run is kept -- whatever it returned --, the teams are retired for this engine and the run is repeated.
                // (ctl[3] is cleared by k_reset at the start of a run only; the finalisation leaves it alone.)
                int faults = 0;
                (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->qstream);
                if (hipMemcpy(&faults, h->P.ctl + 3, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) faults = 0;
                if (faults) {
                    (void)hipGetLastError();
                    if (h->W > 1) return fail(TTX_EHIP, "ttx_run: k_halfstep_det met a rank above the host's bound; set TTX_DE_TEAM=0");
                    h->de_team = 0; h->det_fallbacks++;
                    for (int k = 0; k < TTX_K_NKINDS; k++) { h->k_launches[k] = 0; h->k_ms[k] = 0; h->k_bytes[k] = 0; }
                    rc = run_impl<FUN_ISING>(h);
                }
            }
            if (h->de_v5) {
                // the relay of k_halfstep_de5 reports a broken hand-over (bounded waits) in ctl[3]: nothing of such a run is
                // kept, the relay is retired for this engine and the run repeated with k_halfstep_de (identical results)
                int faults = 0;
                (void)hipStreamSynchronize(h->stream); (void)hipStreamSynchronize(h->qstream);
                if (hipMemcpy(&faults, h->P.ctl + 3, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) faults = 0;
                if (faults) {
                    (void)hipGetLastError();
                    if (h->W > 1) return fail(TTX_EHIP, "ttx_run: the wave relay of k_halfstep_de5 broke (%d hand-overs); set TTX_DE_V5=0", faults);
                    h->de_v5 = 0; h->de5_fallbacks++;
                    for (int k = 0; k < TTX_K_NKINDS; k++) { h->k_launches[k] = 0; h->k_ms[k] = 0; h->k_bytes[k] = 0; }
                    rc = run_impl<FUN_ISING>(h);
                }
            }
            return rc;
        }
        case TTX_FUN_STDNORM: return run_impl<FUN_STDNORM>(h);
        case TTX_FUN_HOST:
            if (!h->hfun) return fail(TTX_ESTATE, "ttx_run: call ttx_set_integrand_host first");
            h->host_calls = 0;
            return run_impl<FUN_HOST>(h);
        default: return run_impl<FUN_MVN>(h);
    }
}

extern "C" int ttx_num_sweeps(const ttx_engine *h) { return h ? (int)h->recs.size() : 0; }
extern "C" int ttx_get_sweeps(const ttx_engine *h, ttx_sweep_rec *out, int cap)
{
    if (!h || !out) return fail(TTX_EINVAL, "ttx_get_sweeps: null argument");
    int n = std::min(cap, (int)h->recs.size());
    memcpy(out, h->recs.data(), sizeof(ttx_sweep_rec) * n);
    return TTX_OK;
}
extern "C" int ttx_get_tapes(const ttx_engine *h, int32_t *out, int64_t cap)
{
    if (!h || !out) return fail(TTX_EINVAL, "ttx_get_tapes: null argument");
    int64_t n = std::min<int64_t>(cap, (int64_t)h->tapes.size());
    memcpy(out, h->tapes.data(), sizeof(int32_t) * n);
    return TTX_OK;
}
extern "C" int64_t ttx_neval(const ttx_engine *h) { return h ? h->neval : 0; }
extern "C" double ttx_seconds(const ttx_engine *h) { return h ? h->seconds : 0.0; }
extern "C" int ttx_get_ranks(const ttx_engine *h, int32_t *r)
{
    if (!h || !r || !h->ran) return fail(TTX_ESTATE, "ttx_get_ranks: run first");
    memcpy(r, h->rfinal.data(), sizeof(int32_t) * (h->d + 1));
    return TTX_OK;
}
static int owner_of_core(const ttx_engine *h, int k)
{
    for (int g = 0; g < h->G; g++) {
        int first = h->own[h->g0 + g], last = h->own[h->g0 + g + 1] - 1;
        bool lastg = (h->g0 + g == h->cfg.nproc - 1);
        if (k >= first && (k <= last || (lastg && k == h->d))) return g;
    }
    return -1;
}
static double *core_dev(const ttx_engine *h, int k)
{
    const int g = owner_of_core(h, k), first = h->own[h->g0 + g];
    return h->P.arg + ((size_t)g * h->NC + (k - first)) * h->P.CS;
}
static void push_ranks(ttx_engine *h)
{
    std::vector<int32_t> rr((size_t)h->G * (h->d + 2), 1);
    for (int g = 0; g < h->G; g++) for (int p = 0; p <= h->d; p++) rr[(size_t)g * (h->d + 2) + p] = h->rfinal[p];
    (void)hipMemcpyAsync(h->P.r, rr.data(), sizeof(int32_t) * rr.size(), hipMemcpyHostToDevice, h->stream);
    (void)hipStreamSynchronize(h->stream);
}
// all-reduce (sum) of a device buffer of any length over the processes of the job, in place: the host transports stage through a
// pinned slot of red_cap doubles, so the buffer goes in pieces
static int allreduce_big(ttx_engine *h, double *buf, size_t count)
{
    if (h->W == 1) return TTX_OK;
    const size_t piece = h->comm ? count : std::max<size_t>(1, h->red_cap);
    int inflight = 0;
    for (size_t o = 0; o < count; o += piece) {
        if (int rc = allreduce_dev(h, buf + o, buf + o, std::min(piece, count - o), 0)) return rc;
        // host transports: a reduction's descriptor and pinned slot are recycled after NRED enqueues -- drain before that
        if (!h->comm && ++inflight >= ttx_engine::NRED - 2) { HIPCHECK(hipStreamSynchronize(h->stream)); inflight = 0; }
    }
    return TTX_OK;
}
// The finalised train of a MULTI-PROCESS job on every process, as a new single-process engine with the same integrand (`out`):
// each process copies the cores it holds into its slots of the new engine's core array and a SUM all-reduce over the job's transport
// fills in the others (their slots hold -0.0 here, the neutral element of fp addition for every value).  dtt_accchk, norm, dot_product, ort, svd and dtt_write of the
// reference work on a `type(dtt)` that holds ALL cores (lib/tt.f90, lib/dmrgg.f90:1081-1166): on a multi-process engine they go
// through this copy.  Collective: every process of the job must call it.
extern "C" int ttx_replicate(ttx_engine *h, ttx_engine **out)
{
    if (!h || !out) return fail(TTX_EINVAL, "ttx_replicate: null argument");
    *out = nullptr;
    if (!h->ran) return fail(TTX_ESTATE, "ttx_replicate: run dtt_dmrgg first");
    HIPCHECK(hipSetDevice(h->cfg.device));
    const int d = h->d;
    ttx_config c = h->cfg;
    std::vector<int32_t> nn(h->n1.begin() + 1, h->n1.begin() + 1 + d);
    std::vector<double> qw;
    c.n = nn.data(); c.par = h->par.empty() ? nullptr : h->par.data(); c.aux = h->aux.empty() ? nullptr : h->aux.data(); c.naux = (int32_t)h->aux.size();
    c.quadw = nullptr;
    if (!h->quadw.empty()) { for (int k = 1; k <= d; k++) for (int j = 0; j < h->n1[k]; j++) qw.push_back(h->quadw[(size_t)k * h->NM + j]); c.quadw = qw.data(); }
    c.nproc = 1; c.mybonds = nullptr; c.world_rank = 0; c.world_size = 1; c.verbose = 0; c.arith = h->P.arith;
    ttx_engine *e = nullptr;
    int rc = create_impl(&e, &c, h->cfg.fun_id == 0);
    if (rc) return rc;
    e->hfun = h->hfun; e->hfun_par = h->hfun_par;
    if (e->RM != h->RM || e->NM != h->NM) { ttx_destroy(e); return fail(TTX_EHIP, "ttx_replicate: layout mismatch"); }
    const size_t CS = h->P.CS;
    // slots of the other processes' cores hold -0.0: x + (-0.0) = x for EVERY x, also for x = -0.0 (x + (+0.0) would turn it into +0.0)
    hipLaunchKernelGGL(k_fill_const, dim3(1024), dim3(256), 0, h->stream, (size_t)d * CS, e->P.arg, -0.0);
    for (int k = 1; k <= d; k++) {
        if (owner_of_core(h, k) < 0) continue;
        HIPCHECK(hipMemcpyAsync(e->P.arg + (size_t)(k - 1) * CS, core_dev(h, k), sizeof(double) * CS, hipMemcpyDeviceToDevice, h->stream));
    }
    if ((rc = allreduce_big(h, e->P.arg, (size_t)d * CS))) { ttx_destroy(e); return rc; }
    // the run-time RNG stream continues where rank 0 of the reference left it (dtt_accchk draws from it): the position of the job's
    // FIRST bond group, handed to every process through the same kind of all-reduce (an integer below 2^53 is exact in a double)
    unsigned long long rp = 0;
    HIPCHECK(hipMemcpyAsync(&rp, &h->P.gs[0].rngpos, sizeof(rp), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (h->W > 1) {
        double v = (h->g0 == 0) ? (double)rp : -0.0;
        double *dv = e->P.redsend;                      // a few doubles of the replica, not in use yet
        HIPCHECK(hipMemcpyAsync(dv, &v, sizeof(double), hipMemcpyHostToDevice, h->stream));
        if ((rc = allreduce_dev(h, dv, dv, 1, 0))) { ttx_destroy(e); return rc; }
        HIPCHECK(hipMemcpyAsync(&v, dv, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        rp = (unsigned long long)v;
    }
    if (h->cb_error) { ttx_destroy(e); return fail(TTX_EHIP, "host transport: a sendrecv / allreduce callback failed"); }
    HIPCHECK(hipMemcpy(&e->P.gs[0].rngpos, &rp, sizeof(rp), hipMemcpyHostToDevice));
    e->rfinal = h->rfinal; e->neval = h->neval; e->ran = true;
    push_ranks(e);
    *out = e;
    return TTX_OK;
}
// run `fn` on a replica of a multi-process engine's train
template <class FN>
static int with_replica(ttx_engine *h, FN fn)
{
    ttx_engine *e = nullptr;
    int rc = ttx_replicate(h, &e);
    if (rc) return rc;
    rc = fn(e);
    const std::string msg = g_err;
    ttx_destroy(e);
    if (rc) g_err = msg;
    return rc;
}

extern "C" int64_t ttx_core_size(const ttx_engine *h, int k)
{
    if (!h || !h->ran || k < 1 || k > h->d || owner_of_core(h, k) < 0) return 0;
    return (int64_t)h->rfinal[k - 1] * h->n1[k] * h->rfinal[k];
}
extern "C" int ttx_get_core(const ttx_engine *h, int k, double *buf)
{
    if (!h || !buf || !h->ran) return fail(TTX_ESTATE, "ttx_get_core: run first");
    if (k < 1 || k > h->d) return fail(TTX_EINVAL, "ttx_get_core: core %d out of range", k);
    int g = owner_of_core(h, k);
    if (g < 0) return fail(TTX_EINVAL, "ttx_get_core: core %d is not held by this process", k);
    const int r0 = h->rfinal[k - 1], r1 = h->rfinal[k], n = h->n1[k], first = h->own[h->g0 + g];
    const double *src = h->P.arg + ((size_t)g * h->NC + (k - first)) * h->P.CS;
    // strided device -> compact host: one 2D copy per right-rank slab (pure data movement)
    for (int s = 0; s < r1; s++)
        HIPCHECK(hipMemcpy2D(buf + (size_t)r0 * n * s, sizeof(double) * r0, src + h->P.SS * s, sizeof(double) * h->RM, sizeof(double) * r0, n, hipMemcpyDeviceToHost));
    return TTX_OK;
}

// ---- tensor trains from outside the sweep: host cores and the reference's stream file ----------------
extern "C" int ttx_get_modes(const ttx_engine *h, int32_t *d, int32_t *n)
{
    if (!h || !d) return fail(TTX_EINVAL, "ttx_get_modes: null argument");
    *d = h->d;
    if (n) for (int k = 1; k <= h->d; k++) n[k - 1] = h->n1[k];
    return TTX_OK;
}

extern "C" int ttx_from_tt(ttx_engine **out, int32_t d, const int32_t *n, const int32_t *r, const double *cores, int32_t device)
{
    if (!out || !n || !r || !cores) return fail(TTX_EINVAL, "ttx_from_tt: null argument");
    *out = nullptr;
    if (d < 2) return fail(TTX_EINVAL, "ttx_from_tt: at least two cores are needed (got %d)", d);
    int rmax = 1;
    for (int k = 0; k <= d; k++) { if (r[k] < 1) return fail(TTX_EINVAL, "ttx_from_tt: rank r(%d)=%d", k, r[k]); rmax = std::max(rmax, (int)r[k]); }
    // ort/svd may pass through r(k) = min(r(k-1)*n(k), ...) <= the incoming ranks, so max(r) is enough storage
    ttx_config c{};
    c.d = d; c.n = n; c.fun_id = 0; c.accuracy = -1.0; c.maxrank = rmax; c.pivoting = 0; c.nproc = 1; c.device = device; c.world_size = 1;
    ttx_engine *h = nullptr;
    int rc = create_impl(&h, &c, true);
    if (rc) return rc;
    GroupState *g0 = (GroupState *)calloc(1, sizeof(GroupState));     // only the bond range is read by the quad kernels
    g0->first = 1; g0->last = d - 1; g0->gglobal = 0;
    hipError_t e = hipMemcpy(h->P.gs, g0, offsetof(GroupState, S), hipMemcpyHostToDevice);
    free(g0);
    if (e != hipSuccess) { ttx_destroy(h); return fail(TTX_EHIP, "ttx_from_tt: %s", hipGetErrorString(e)); }
    h->rfinal.assign(r, r + d + 1);
    size_t off = 0;
    for (int k = 1; k <= d; k++) {
        const int r0 = r[k - 1], r1 = r[k], nk = n[k - 1];
        double *dst = h->P.arg + (size_t)(k - 1) * h->P.CS;
        for (int s = 0; s < r1; s++) {        // compact host -> padded device slabs (inverse of ttx_get_core)
            e = hipMemcpy2D(dst + h->P.SS * s, sizeof(double) * h->RM, cores + off + (size_t)r0 * nk * s, sizeof(double) * r0, sizeof(double) * r0, nk, hipMemcpyHostToDevice);
            if (e != hipSuccess) { ttx_destroy(h); return fail(TTX_EHIP, "ttx_from_tt: %s", hipGetErrorString(e)); }
        }
        off += (size_t)r0 * nk * r1;
    }
    std::vector<int32_t> rr((size_t)(d + 2), 1);
    for (int p = 0; p <= d; p++) rr[p] = r[p];
    e = hipMemcpy(h->P.r, rr.data(), sizeof(int32_t) * rr.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { ttx_destroy(h); return fail(TTX_EHIP, "ttx_from_tt: %s", hipGetErrorString(e)); }
    h->ran = true;
    *out = h;
    return TTX_OK;
}

// lib/ttio.f90:10-17 `tthead`: 'TT      ', ver(2)=(1,0), inf(4)=(tt_size,0,0,0), comment*64, i(8) with i(1:2)=(l,m); 128 bytes
namespace {
struct TTFileHead { char txt[8]; int32_t ver[2]; int32_t inf[4]; char comment[64]; int32_t i[8]; };
static_assert(sizeof(TTFileHead) == 128, "stream header is 128 bytes");
}
extern "C" int ttx_write(const ttx_engine *h, const char *path)
{
    if (!h || !path || !h->ran) return fail(TTX_ESTATE, "dtt_write: no tensor train to write");
    if (h->W > 1) {     // collective: every process takes part in the replica, rank 0 writes the file
        ttx_engine *hh = const_cast<ttx_engine *>(h);
        return with_replica(hh, [&](ttx_engine *e) { return hh->wrank == 0 ? ttx_write(e, path) : TTX_OK; });
    }
    const int d = h->d;
    size_t sz = 0;
    for (int k = 1; k <= d; k++) sz += (size_t)h->rfinal[k - 1] * h->n1[k] * h->rfinal[k];
    if (sz == 0) return fail(TTX_EINVAL, "dtt_write: tt structure has invalid size: 0");      // lib/ttio.f90:60-61
    std::vector<double> x(sz);
    size_t off = 0;
    for (int k = 1; k <= d; k++) { int rc = ttx_get_core(h, k, x.data() + off); if (rc) return rc; off += (size_t)ttx_core_size(h, k); }
    FILE *f = fopen(path, "wb");
    if (!f) return fail(TTX_EINVAL, "dtt_write: error opening file: %s", path);                // :85-88
    TTFileHead hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.txt, "TT      ", 8);
    hd.ver[0] = 1; hd.ver[1] = 0; hd.inf[0] = 2048;
    hd.i[0] = 1; hd.i[1] = d;
    const int32_t lm[2] = {1, d};
    bool ok = fwrite(&hd, sizeof hd, 1, f) == 1 && fwrite(lm, sizeof lm, 1, f) == 1;          // :75-76
    ok = ok && fwrite(&h->n1[1], sizeof(int32_t), d, f) == (size_t)d && fwrite(h->rfinal.data(), sizeof(int32_t), d + 1, f) == (size_t)d + 1;   // :77
    ok = ok && fwrite(x.data(), sizeof(double), sz, f) == sz;                                 // :78
    ok = (fclose(f) == 0) && ok;
    if (!ok) return fail(TTX_EINVAL, "dtt_write: error writing file: %s", path);
    return TTX_OK;
}

// ---- HDF5 layout of lib/utils.f90:8-57 (save_dtt_to_hdf5): group "TT", datasets "modes" (m ints), "ranks" (m+1 ints),
//      "core_k" (k = 0..m-1) with the Fortran shape (r(k-1), n(k), r(k)) -- i.e. the C dataspace (r(k), n(k), r(k-1)) over the
//      column-major bytes.  libhdf5 is an optional run-time dependency, resolved with dlopen like librccl.
namespace {
typedef int64_t hid_t_; typedef int herr_t_; typedef unsigned long long hsize_t_;
struct Hdf5Api {
    void *lib = nullptr;
    herr_t_ (*open)() = nullptr;
    hid_t_ (*Fcreate)(const char *, unsigned, hid_t_, hid_t_) = nullptr;
    hid_t_ (*Fopen)(const char *, unsigned, hid_t_) = nullptr;
    herr_t_ (*Fclose)(hid_t_) = nullptr;
    hid_t_ (*Gcreate2)(hid_t_, const char *, hid_t_, hid_t_, hid_t_) = nullptr;
    herr_t_ (*Gclose)(hid_t_) = nullptr;
    hid_t_ (*Screate_simple)(int, const hsize_t_ *, const hsize_t_ *) = nullptr;
    herr_t_ (*Sclose)(hid_t_) = nullptr;
    hid_t_ (*Dcreate2)(hid_t_, const char *, hid_t_, hid_t_, hid_t_, hid_t_, hid_t_) = nullptr;
    hid_t_ (*Dopen2)(hid_t_, const char *, hid_t_) = nullptr;
    hid_t_ (*Dget_space)(hid_t_) = nullptr;
    int (*Sget_simple_extent_dims)(hid_t_, hsize_t_ *, hsize_t_ *) = nullptr;
    herr_t_ (*Dwrite)(hid_t_, hid_t_, hid_t_, hid_t_, hid_t_, const void *) = nullptr;
    herr_t_ (*Dread)(hid_t_, hid_t_, hid_t_, hid_t_, hid_t_, void *) = nullptr;
    herr_t_ (*Dclose)(hid_t_) = nullptr;
    herr_t_ (*Eset_auto2)(hid_t_, void *, void *) = nullptr;
    hid_t_ t_int = -1, t_double = -1;
};
Hdf5Api g_h5;
int hdf5_load()
{
    if (g_h5.lib) return TTX_OK;
    void *L = nullptr;
    for (const char *nm : {"libhdf5.so", "libhdf5.so.103", "/opt/conda/lib/libhdf5.so", "libhdf5_serial.so"}) if ((L = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!L) return fail(TTX_EINVAL, "save_dtt_to_hdf5: libhdf5.so not found (%s)", dlerror());
#define H5_(f, name) *(void **)(&g_h5.f) = dlsym(L, name); if (!g_h5.f) return fail(TTX_EINVAL, "libhdf5.so lacks %s", name);
    H5_(open, "H5open") H5_(Fcreate, "H5Fcreate") H5_(Fopen, "H5Fopen") H5_(Fclose, "H5Fclose") H5_(Gcreate2, "H5Gcreate2") H5_(Gclose, "H5Gclose")
    H5_(Screate_simple, "H5Screate_simple") H5_(Sclose, "H5Sclose") H5_(Dcreate2, "H5Dcreate2") H5_(Dopen2, "H5Dopen2") H5_(Dget_space, "H5Dget_space")
    H5_(Sget_simple_extent_dims, "H5Sget_simple_extent_dims") H5_(Dwrite, "H5Dwrite") H5_(Dread, "H5Dread") H5_(Dclose, "H5Dclose") H5_(Eset_auto2, "H5Eset_auto2")
#undef H5_
    if (g_h5.open() < 0) return fail(TTX_EINVAL, "H5open failed");
    hid_t_ *ti = (hid_t_ *)dlsym(L, "H5T_NATIVE_INT_g"), *td = (hid_t_ *)dlsym(L, "H5T_NATIVE_DOUBLE_g");
    if (!ti || !td) return fail(TTX_EINVAL, "libhdf5.so lacks the native type ids");
    g_h5.t_int = *ti; g_h5.t_double = *td;
    g_h5.Eset_auto2(0, nullptr, nullptr);                      // errors are reported through return codes here
    g_h5.lib = L;
    return TTX_OK;
}
}
extern "C" int ttx_write_hdf5(const ttx_engine *h, const char *path)
{
    if (!h || !path || !h->ran) return fail(TTX_ESTATE, "save_dtt_to_hdf5: no tensor train to write");
    if (h->W > 1) return fail(TTX_EINVAL, "save_dtt_to_hdf5: single-process engines only");
    int rc = hdf5_load();
    if (rc) return rc;
    const int d = h->d;
    const hid_t_ f = g_h5.Fcreate(path, 2u /* H5F_ACC_TRUNC */, 0, 0);
    if (f < 0) return fail(TTX_EINVAL, "save_dtt_to_hdf5: cannot create %s", path);
    const hid_t_ grp = g_h5.Gcreate2(f, "TT", 0, 0, 0);
    bool ok = grp >= 0;
    auto put = [&](const char *name, int rank, const hsize_t_ *dims, hid_t_ type, const void *buf) {
        const hid_t_ sp = g_h5.Screate_simple(rank, dims, nullptr);
        const hid_t_ ds = (sp >= 0) ? g_h5.Dcreate2(grp, name, type, sp, 0, 0, 0) : -1;
        if (ds < 0 || g_h5.Dwrite(ds, type, 0, 0, 0, buf) < 0) ok = false;
        if (ds >= 0) g_h5.Dclose(ds);
        if (sp >= 0) g_h5.Sclose(sp);
    };
    if (ok) {
        hsize_t_ d1 = (hsize_t_)d;
        put("modes", 1, &d1, g_h5.t_int, &h->n1[1]);                                 // utils.f90:25-30
        d1 = (hsize_t_)d + 1;
        put("ranks", 1, &d1, g_h5.t_int, h->rfinal.data());                          // :32-37
        std::vector<double> x;
        for (int k = 1; k <= d && ok; k++) {                                         // :40-51
            x.resize((size_t)ttx_core_size(h, k));
            if ((rc = ttx_get_core(h, k, x.data()))) { ok = false; break; }
            const hsize_t_ d3[3] = {(hsize_t_)h->rfinal[k], (hsize_t_)h->n1[k], (hsize_t_)h->rfinal[k - 1]};   // Fortran (r0, n, r1) reversed
            char nm[32]; snprintf(nm, sizeof nm, "core_%d", k - 1);
            put(nm, 3, d3, g_h5.t_double, x.data());
        }
    }
    if (grp >= 0) g_h5.Gclose(grp);
    g_h5.Fclose(f);
    if (!ok) return rc ? rc : fail(TTX_EINVAL, "save_dtt_to_hdf5: error writing %s", path);
    return TTX_OK;
}
extern "C" int ttx_read_hdf5(ttx_engine **out, const char *path, int32_t device)
{
    if (!out || !path) return fail(TTX_EINVAL, "ttx_read_hdf5: null argument");
    *out = nullptr;
    int rc = hdf5_load();
    if (rc) return rc;
    const hid_t_ f = g_h5.Fopen(path, 0u /* H5F_ACC_RDONLY */, 0);
    if (f < 0) return fail(TTX_EINVAL, "ttx_read_hdf5: cannot open %s", path);
    auto dims_of = [&](const char *name, int want, hsize_t_ *dims) -> hid_t_ {
        const hid_t_ ds = g_h5.Dopen2(f, name, 0);
        if (ds < 0) return -1;
        const hid_t_ sp = g_h5.Dget_space(ds);
        const int nd = (sp >= 0) ? g_h5.Sget_simple_extent_dims(sp, dims, nullptr) : -1;
        if (sp >= 0) g_h5.Sclose(sp);
        if (nd != want) { g_h5.Dclose(ds); return -1; }
        return ds;
    };
    hsize_t_ dm[3];
    std::vector<int32_t> n, r;
    std::vector<double> cores;
    bool ok = true;
    hid_t_ ds = dims_of("/TT/modes", 1, dm);
    if (ds < 0) ok = false;
    else { n.resize(dm[0]); ok = g_h5.Dread(ds, g_h5.t_int, 0, 0, 0, n.data()) >= 0; g_h5.Dclose(ds); }
    if (ok) { ds = dims_of("/TT/ranks", 1, dm); if (ds < 0 || dm[0] != n.size() + 1) ok = false; if (ds >= 0) { r.resize(dm[0]); ok = ok && g_h5.Dread(ds, g_h5.t_int, 0, 0, 0, r.data()) >= 0; g_h5.Dclose(ds); } }
    for (size_t k = 0; ok && k < n.size(); k++) {
        char nm[40]; snprintf(nm, sizeof nm, "/TT/core_%zu", k);
        ds = dims_of(nm, 3, dm);
        if (ds < 0 || (int)dm[0] != r[k + 1] || (int)dm[1] != n[k] || (int)dm[2] != r[k]) { ok = false; if (ds >= 0) g_h5.Dclose(ds); break; }
        const size_t off = cores.size(), sz = (size_t)r[k] * n[k] * r[k + 1];
        cores.resize(off + sz);
        ok = g_h5.Dread(ds, g_h5.t_double, 0, 0, 0, cores.data() + off) >= 0;
        g_h5.Dclose(ds);
    }
    g_h5.Fclose(f);
    if (!ok) return fail(TTX_EINVAL, "ttx_read_hdf5: %s does not hold a tensor train in the layout of lib/utils.f90", path);
    return ttx_from_tt(out, (int32_t)n.size(), n.data(), r.data(), cores.data(), device);
}

extern "C" int ttx_read(ttx_engine **out, const char *path, int32_t device)
{
    if (!out || !path) return fail(TTX_EINVAL, "dtt_read: null argument");
    *out = nullptr;
    FILE *f = fopen(path, "rb");
    if (!f) return fail(TTX_EINVAL, "dtt_read: file not exist: %s", path);                     // lib/ttio.f90:210-214
    TTFileHead hd;
    int32_t lm[2];
    auto bad = [&](const char *what) { fclose(f); return fail(TTX_EINVAL, "dtt_read: %s: %s", what, path); };
    if (fread(&hd, sizeof hd, 1, f) != 1) return bad("error reading header");                // :276-279
    if (hd.txt[0] != 'T' || hd.txt[1] != 'T') return bad("not TT header in file");            // :236-240
    if (hd.ver[0] != 1) return bad("not correct version of TT file");                         // :241-245
    if (fread(lm, sizeof lm, 1, f) != 1) return bad("error reading lm");
    const int l = lm[0], m = lm[1];
    if (l < 1 || m < l || m > 2048) return bad("read strange l,m");                           // :249-251, tt_size
    const int d = m - l + 1;
    std::vector<int32_t> n(d), r(d + 1);
    if (fread(n.data(), sizeof(int32_t), d, f) != (size_t)d || fread(r.data(), sizeof(int32_t), d + 1, f) != (size_t)d + 1) return bad("error reading nr");
    size_t sz = 0;
    for (int k = 0; k < d; k++) {
        if (n[k] < 1 || r[k] < 1 || r[k + 1] < 1 || n[k] > 32000 || r[k] > 128 || r[k + 1] > 128) return bad("tt structure has invalid size");
        sz += (size_t)r[k] * n[k] * r[k + 1];
    }
    std::vector<double> x(sz);
    if (fread(x.data(), sizeof(double), sz, f) != sz) return bad("error reading cores");
    fclose(f);
    // the device engine numbers cores 1..d; a file with l > 1 keeps its shape but loses the offset (every driver has l = 1)
    return ttx_from_tt(out, d, n.data(), r.data(), x.data(), device);
}

extern "C" int ttx_quad(ttx_engine *h, const double *w, double *val)
{
    if (!h || !val || !h->ran) return fail(TTX_ESTATE, "ttx_quad: run first");
    HIPCHECK(hipSetDevice(h->cfg.device));
    double *dw = nullptr;
    std::vector<double> wp;
    if (w) {
        wp.assign((size_t)(h->d + 1) * h->NM, 0.0);
        size_t off = 0;
        for (int k = 1; k <= h->d; k++) { for (int j = 0; j < h->n1[k]; j++) wp[(size_t)k * h->NM + j] = w[off + j]; off += h->n1[k]; }
        HIPCHECK(hipMalloc((void **)&dw, sizeof(double) * wp.size()));
        HIPCHECK(hipMemcpy(dw, wp.data(), sizeof(double) * wp.size(), hipMemcpyHostToDevice));
    }
    int rc = launch_quad(h, 1, dw);
    if (!rc) rc = readback(h);
    if (dw) (void)hipFree(dw);
    if (rc) return rc;
    HIPCHECK(hipGetLastError());
    *val = h->h_sum[SUM_VAL];
    return TTX_OK;
}

template <int FUN>
static int accchk_impl(ttx_engine *h, int nlot, double *einf, double *efro, double *ainf, double *afro, int32_t *pivot)
{
    DevProb &P = h->P;
    const int d = h->d;
    std::vector<int> owner(d + 2, 0);
    for (int k = 1; k <= d; k++) owner[k] = owner_of_core(h, k);
    // the stream position where dtt_dmrgg left the run-time generator (rank 0 of the reference = group 0)
    GroupState g0s;
    HIPCHECK(hipMemcpy(&g0s, P.gs, sizeof(GroupState), hipMemcpyDeviceToHost));
    int *downer, *dind; double *dout;
    HIPCHECK(hipMalloc((void **)&downer, sizeof(int) * (d + 2))); HIPCHECK(hipMalloc((void **)&dind, sizeof(int) * (size_t)nlot * d));
    HIPCHECK(hipMalloc((void **)&dout, sizeof(double) * 4 * (size_t)nlot));
    HIPCHECK(hipMemcpy(downer, owner.data(), sizeof(int) * (d + 2), hipMemcpyHostToDevice));
    size_t lds = sizeof(double) * (h->par.size() + 2 * h->RM + 4) + sizeof(int) * (d + 4);
    if (FUN == FUN_HOST) {
        // chunks of at most G*HS samples: index pass, the user's function on the host, value pass
        const int chunk = (int)std::min<size_t>((size_t)h->G * h->HS, 65535);
        for (int il0 = 0; il0 < nlot; il0 += chunk) {
            const int cnt = std::min(chunk, nlot - il0);
            DevProb Q = P;
            Q.hostpass = 1;
            hipLaunchKernelGGL(k_accchk<FUN>, dim3(cnt), dim3(64), lds, h->stream, Q, (unsigned long long)g0s.rngpos, nlot, (const int *)downer, dout, dind, il0);
            if (int rc_ = host_eval(h)) return rc_;
            Q.hostpass = 2;
            hipLaunchKernelGGL(k_accchk<FUN>, dim3(cnt), dim3(64), lds, h->stream, Q, (unsigned long long)g0s.rngpos, nlot, (const int *)downer, dout, dind, il0);
        }
    } else
    hipLaunchKernelGGL(k_accchk<FUN>, dim3(nlot), dim3(64), lds, h->stream, P, (unsigned long long)g0s.rngpos, nlot, (const int *)downer, dout, dind, 0);
    std::vector<double> o(4 * (size_t)nlot);
    std::vector<int> ind((size_t)nlot * d);
    HIPCHECK(hipMemcpyAsync(o.data(), dout, sizeof(double) * o.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipMemcpyAsync(ind.data(), dind, sizeof(int) * ind.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipGetLastError());
    // the reference's sequential bookkeeping over the samples (:1126-1133): strict '<' keeps the first worst sample
    double e1 = 0.0, e2 = 0.0, a1 = 0.0, a2 = 0.0; int worst = -1;
    for (int il = 0; il < nlot; il++) {
        if (e1 < o[4 * (size_t)il]) { e1 = o[4 * (size_t)il]; worst = il; }
        e2 = e2 + o[4 * (size_t)il + 1];
        a1 = std::max(a1, o[4 * (size_t)il + 2]);
        a2 = a2 + o[4 * (size_t)il + 3];
    }
    *einf = e1; *efro = std::sqrt(e2); *ainf = a1; *afro = std::sqrt(a2);
    if (pivot && worst >= 0) for (int i = 0; i < d; i++) pivot[i] = ind[(size_t)worst * d + i];
    (void)hipFree(downer); (void)hipFree(dind); (void)hipFree(dout);
    return TTX_OK;
}

extern "C" int ttx_accchk(ttx_engine *h, int32_t nlot, double *einf, double *efro, double *ainf, double *afro, int32_t *pivot)
{
    if (!h || !einf || !efro || !ainf || !afro || !h->ran) return fail(TTX_ESTATE, "ttx_accchk: run first");
    if (h->W > 1)       // every rank needs all cores: the check runs on a replica of the job's train (identical result on every process)
        return with_replica(h, [&](ttx_engine *e) { return ttx_accchk(e, nlot, einf, efro, ainf, afro, pivot); });
    if (nlot < 1) return fail(TTX_EINVAL, "dtt_accchk: nlot must be positive");
    if (h->cfg.fun_id == 0) return fail(TTX_ESTATE, "dtt_accchk: this engine holds a loaded tensor train and has no integrand");
    HIPCHECK(hipSetDevice(h->cfg.device));
    switch (h->cfg.fun_id) {
        case TTX_FUN_ISING: return accchk_impl<FUN_ISING>(h, nlot, einf, efro, ainf, afro, pivot);
        case TTX_FUN_STDNORM: return accchk_impl<FUN_STDNORM>(h, nlot, einf, efro, ainf, afro, pivot);
        case TTX_FUN_HOST:
            if (!h->hfun) return fail(TTX_ESTATE, "dtt_accchk: call ttx_set_integrand_host first");
            return accchk_impl<FUN_HOST>(h, nlot, einf, efro, ainf, afro, pivot);
        default: return accchk_impl<FUN_MVN>(h, nlot, einf, efro, ainf, afro, pivot);
    }
}

// ---- tt_lib utilities (ort / svd / norm / dot) -------------------------------------------------------------
static int tt_prepare(ttx_engine *h, const char *who)
{
    if (!h || !h->ran) return fail(TTX_ESTATE, "%s: run dtt_dmrgg first", who);
    if (h->W > 1) return fail(TTX_EINVAL, "%s: single-process engines only", who);
    HIPCHECK(hipSetDevice(h->cfg.device));
    if (!h->Wa) {
        const size_t CS = h->P.CS, RM = h->RM;
        int rc;
        if ((rc = dev_alloc(h, &h->Wa, CS)) || (rc = dev_alloc(h, &h->Wb, CS)) || (rc = dev_alloc(h, &h->Wc, CS)) || (rc = dev_alloc(h, &h->Wd, CS))) return rc;
        if ((rc = dev_alloc(h, &h->Sm, 8 * RM * RM + 4 * RM + 16)) || (rc = dev_alloc(h, &h->Si, 2 * RM + 16))) return rc;
    }
    return TTX_OK;
}
static inline dim3 g1(size_t n) { return dim3((unsigned)std::min<size_t>((n + 255) / 256, 4096)); }
// threads of the one-workgroup factorisation kernels (a multiple of 64, at most 1024): the kernels are chains of short phases
// separated by workgroup barriers, whose cost grows with the number of waves -- TTX_QR_THREADS / TTX_JAC_THREADS override
static int tt_threads(const char *env, int dflt)
{
    int v = dflt;
    if (const char *e = getenv(env)) v = atoi(e);
    v = std::max(64, std::min(1024, v)) & ~63;
    return v;
}
static int gemm(ttx_engine *h, int M, int N, int K, const double *A, int lda, const double *B, int ldb, double *C, int ldc)
{
    hipLaunchKernelGGL(k_gemm_mfma, dim3((N + 63) / 64, (M + 15) / 16), dim3(256), 0, h->stream, M, N, K, A, lda, B, ldb, C, ldc);
    return TTX_OK;
}
// ---- QR of an unfolding ------------------------------------------------------------------------------------------------
// k_qr_own<MR, NC> (matrix in registers, one barrier per reflector, ttx_ttops.h): NC = columns per wave, MR = rows per lane
struct QrOwnShape { int nt, mr, nc, rows_cap; };
static QrOwnShape qr_own_shape(int n)
{
    QrOwnShape s{0, 0, 0, 0};
    if (n < 1 || n > 128 || (getenv("TTX_QR_OWN") && atoi(getenv("TTX_QR_OWN")) == 0)) return s;
    s.nt = tt_threads("TTX_QR_THREADS", 1024);
    const int nw = s.nt / 64;
    int nc = 1; while (nc * nw < n) nc *= 2;
    if (nc > 8) return QrOwnShape{0, 0, 0, 0};
    s.nc = nc;
    s.mr = nc <= 2 ? 8 : 4;                                                  // rows per lane the instantiations hold without spilling
    const size_t budget = 150 * 1024 / sizeof(double);
    const long lds_rows = (long)((budget - (size_t)n - 4) / (size_t)n);
    s.rows_cap = (int)std::min<long>(64L * s.mr, lds_rows);
    return s;
}
// one launch over P panels of rbs rows (P = 1, rbs = rows: the whole matrix); false: shape not covered
static bool qr_own_launch(ttx_engine *h, const QrOwnShape &s, int rows, int n, int rbs, int P, const double *M, int ldm, double *Q, int ldq,
                          double *R, int ldr, int rstep, double *tau, int *rc_out)
{
    int mr = 1; while (mr * 64 < rbs) mr *= 2;
    if (mr < 2) mr = 2;
    if (mr > s.mr) return false;
    const size_t lds = sizeof(double) * qr_own_lds_doubles(rbs, n);
    const void *fn = nullptr;
#define QRO(MRv, NCv) if (mr == MRv && s.nc == NCv) fn = reinterpret_cast<const void *>(k_qr_own<MRv, NCv>);
    QRO(2, 1) QRO(4, 1) QRO(8, 1) QRO(2, 2) QRO(4, 2) QRO(8, 2) QRO(2, 4) QRO(4, 4) QRO(2, 8) QRO(4, 8)
#undef QRO
    if (!fn) return false;
    {   // the dynamic-LDS limit set so far per instantiation (engines may be driven from several host threads)
        static std::map<const void *, size_t> lds_set;
        static std::mutex lds_mu;
        std::lock_guard<std::mutex> lk(lds_mu);
        if ((*rc_out = ensure_lds(fn, lds, lds_set[fn]))) return true;
    }
    void *args[] = {&rows, &n, &rbs, &M, &ldm, &Q, &ldq, &R, &ldr, &rstep, &tau};
    hipError_t e = hipLaunchKernel(fn, dim3(P), dim3(s.nt), args, lds, h->stream);
    if (e != hipSuccess) *rc_out = fail(TTX_EHIP, "k_qr_own: %s", hipGetErrorString(e));
    return true;
}
// Tall-skinny QR of A (m x n, m >> n) over several workgroups (ttx_ttops.h): levels of panel factorisations side by side, then
// the explicit Q from the top level down.  Scratch: Wb (Q panels of level 0), Wc (stacked triangles of the levels / their
// accumulated Q), Wd (Q panels of the levels >= 1) -- all free while a qr() is running.  false: shape not eligible.
static bool qr_tsqr(ttx_engine *h, int m, int n, double *A, double *R, double *tau, int *rc_out)
{
    *rc_out = TTX_OK;
    if (A != h->Wa || n > 96 || m < 4 * n || (getenv("TTX_TSQR") && atoi(getenv("TTX_TSQR")) == 0)) return false;
    const QrOwnShape own = qr_own_shape(n);
    const bool use_own = own.nt && own.rows_cap >= 2 * n;
    const size_t budget = 150 * 1024 / sizeof(double);
    // rows of a panel: the register kernel's time per reflector grows with the rows per lane, so its panels are short (4 n rows,
    // at least 256 -- measured optimum for n = 32); the LDS kernel takes what fits with its reflector
    int RB = use_own ? std::min(own.rows_cap, std::max(256, 4 * n)) : (int)((budget - 2 * n - 2) / (size_t)(n + 1));
    if (const char *e = getenv("TTX_QR_PANEL")) if (use_own && atoi(e) >= 2 * n) RB = std::min(own.rows_cap, atoi(e));
    if (RB < 2 * n) return false;
    struct Lvl { int rows, P, rbs; double *M, *Q; };
    std::vector<Lvl> lv;
    double *Sbuf = h->Wc, *Tbuf = h->Wd;
    int rows = m; double *M = A;
    auto top_fits = [&](int rws) { return use_own ? rws <= own.rows_cap : (size_t)rws * n + rws + 2 * n + 4 <= budget; };
    while (!top_fits(rows)) {                                                   // until one workgroup takes the rest
        const int P = (rows + RB - 1) / RB, rbs = (rows + P - 1) / P;
        if (rows - (P - 1) * rbs < n) return false;                             // a last panel shorter than n: keep the one-workgroup path
        Lvl L{rows, P, rbs, M, lv.empty() ? h->Wb : Tbuf};
        if (!lv.empty()) Tbuf += (size_t)rows * n;
        lv.push_back(L);
        M = Sbuf; Sbuf += (size_t)P * n * n; rows = P * n;
    }
    if (lv.empty()) return false;
    static size_t a_qp = 0, a_q1 = 0;
    for (size_t l = 0; l < lv.size(); l++) {
        const Lvl &L = lv[l];
        double *Rst = (l + 1 < lv.size()) ? lv[l + 1].M : M;                    // the next level's matrix (P n x n)
        if (use_own && qr_own_launch(h, own, L.rows, n, L.rbs, L.P, L.M, L.rows, L.Q, L.rows, Rst, L.P * n, n, nullptr, rc_out)) {
            if (*rc_out) return true;
            continue;
        }
        const size_t lds = sizeof(double) * qr_panel_lds_doubles(L.rbs, n);
        if ((*rc_out = ensure_lds(reinterpret_cast<const void *>(k_qr_panel), lds, a_qp))) return true;
        hipLaunchKernelGGL(k_qr_panel, dim3(L.P), dim3(tt_threads("TTX_QR_THREADS", 1024)), lds, h->stream, L.rows, n, L.rbs, L.M, L.Q, Rst, L.P * n);
    }
    // top: one workgroup, in place: M -> Q_top (rows x n), R (n x n)
    if (!(use_own && qr_own_launch(h, own, rows, n, rows, 1, M, rows, M, rows, R, std::min(rows, n), 0, tau, rc_out))) {
        const size_t lds_all = sizeof(double) * ((size_t)rows + 2 * n + 4 + (size_t)rows * n);
        if ((*rc_out = ensure_lds(reinterpret_cast<const void *>(k_qr<true>), lds_all, a_q1))) return true;
        hipLaunchKernelGGL(k_qr<true>, dim3(1), dim3(tt_threads("TTX_QRTOP_THREADS", 1024)), lds_all, h->stream, rows, n, M, R, tau);
    }
    if (*rc_out) return true;
    // explicit Q, top down: Qacc(level l) = blockdiag(Q_p) * Qacc(level l+1); level l's own matrix buffer takes the result
    const double *Qup = M; int ldup = rows;
    for (int l = (int)lv.size() - 1; l >= 0; l--) {
        const Lvl &L = lv[l];
        double *out = L.M;                                                      // level 0: A itself
        hipLaunchKernelGGL(k_gemm_mfma_panels, dim3((n + 63) / 64, (L.rbs + 15) / 16, L.P), dim3(256), 0, h->stream,
                           L.rows, n, L.rbs, (const double *)L.Q, L.rows, Qup, ldup, out, L.rows);
        Qup = out; ldup = L.rows;
    }
    return true;
}
static int qr(ttx_engine *h, int m, int n, double *A, double *R, double *tau)
{
    { int rc = TTX_OK; if (qr_tsqr(h, m, n, A, R, tau, &rc)) return rc; }
    {   // small unfoldings: one workgroup, matrix in registers
        const QrOwnShape own = qr_own_shape(n);
        int rc = TTX_OK;
        if (own.nt && m <= own.rows_cap && qr_own_launch(h, own, m, n, m, 1, A, m, A, m, R, std::min(m, n), 0, tau, &rc)) return rc;
    }
    const size_t lds = sizeof(double) * ((size_t)m + n + 4);
    if (lds > 150 * 1024) return fail(TTX_EINVAL, "dtt_ort: unfolding with %d rows does not fit the LDS-staged reflector", m);
    // small unfoldings are factored entirely inside LDS; larger ones stream the panel from L2 with threads mapped to rows
    const size_t lds_all = lds + sizeof(double) * ((size_t)m * n + n);
    static size_t a_q0 = 0, a_q1 = 0;
    if (lds_all <= 150 * 1024) {
        if (int rc = ensure_lds(reinterpret_cast<const void *>(k_qr<true>), lds_all, a_q1)) return rc;
        hipLaunchKernelGGL(k_qr<true>, dim3(1), dim3(tt_threads("TTX_QR_THREADS", 1024)), lds_all, h->stream, m, n, A, R, tau);
    } else {
        if (int rc = ensure_lds(reinterpret_cast<const void *>(k_qr<false>), lds, a_q0)) return rc;
        hipLaunchKernelGGL(k_qr<false>, dim3(1), dim3(1024), lds, h->stream, m, n, A, R, tau);
    }
    return TTX_OK;
}
static int jacobi(ttx_engine *h, int p, int q, double *X, double *V, double *sv, int *perm, int *info, double tol, int rmax)
{
    const size_t lds = sizeof(double) * ((size_t)p + q) * q;
    const int in_lds = lds <= 140 * 1024;
    static size_t a_j = 0;
    if (in_lds) { if (int rc = ensure_lds(reinterpret_cast<const void *>(k_jacobi_svd), lds, a_j)) return rc; }
    hipLaunchKernelGGL(k_jacobi_svd, dim3(1), dim3(tt_threads("TTX_JAC_THREADS", 256)), in_lds ? lds : 0, h->stream, p, q, X, V, sv, perm, info, 1, tol, rmax, in_lds);
    return TTX_OK;
}
static int sumsq(ttx_engine *h, size_t n, const double *x, double *out_host)
{
    double *d = h->Sm + 8 * (size_t)h->RM * h->RM + 4 * h->RM;      // scratch scalar
    hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(1024), 0, h->stream, n, x, d);
    HIPCHECK(hipMemcpyAsync(out_host, d, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return TTX_OK;
}

static int ort_impl(ttx_engine *h)
{
    const int d = h->d, RM = h->RM; const size_t SS = h->P.SS;
    std::vector<int32_t> &r = h->rfinal;
    double *Rm = h->Sm, *tau = h->Sm + 8 * (size_t)RM * RM;
    double *acc = h->Sm + 8 * (size_t)RM * RM + 4 * RM + 2;             // device scalars: sum of log norms, last 1/norm
    int rc;
    HIPCHECK(hipMemsetAsync(acc, 0, 2 * sizeof(double), h->stream));
    // the whole left-to-right pass is enqueued without a host round trip: the new ranks min(r0*n, r1) are known on the
    // host, the norm equalisation (log of each R's norm, final rescaling) stays on the device
    for (int k = 1; k <= d - 1; k++) {                                  // lib/tt.f90:149-181
        const int r0 = r[k - 1], n = h->n1[k], r1 = r[k], mm = r0 * n, nn = r1, mn = std::min(mm, nn), kk = h->n1[k + 1] * r[k + 1];
        hipLaunchKernelGGL(k_pack_core, g1((size_t)mm * nn), dim3(256), 0, h->stream, core_dev(h, k), h->Wa, r0, n, r1, RM, SS, 0);
        if ((rc = qr(h, mm, nn, h->Wa, Rm, tau))) return rc;
        hipLaunchKernelGGL(k_norm_log, dim3(1), dim3(1024), 0, h->stream, (size_t)mn * nn, Rm, acc, 1);
        hipLaunchKernelGGL(k_unpack_core, g1((size_t)mm * mn), dim3(256), 0, h->stream, core_dev(h, k), h->Wa, r0, n, mn, RM, SS, 0, 1.0);
        hipLaunchKernelGGL(k_pack_core, g1((size_t)nn * kk), dim3(256), 0, h->stream, core_dev(h, k + 1), h->Wb, nn, h->n1[k + 1], r[k + 1], RM, SS, 0);
        gemm(h, mn, kk, nn, Rm, mn, h->Wb, nn, h->Wc, mn);             // R pushed into the next core (:175), fp64 MFMA
        r[k] = mn;
        hipLaunchKernelGGL(k_unpack_core, g1((size_t)mn * kk), dim3(256), 0, h->stream, core_dev(h, k + 1), h->Wc, mn, h->n1[k + 1], r[k + 1], RM, SS, 0, 1.0);
    }
    const size_t last = (size_t)r[d - 1] * h->n1[d] * r[d];
    hipLaunchKernelGGL(k_pack_core, g1(last), dim3(256), 0, h->stream, core_dev(h, d), h->Wa, r[d - 1], h->n1[d], r[d], RM, SS, 0);
    hipLaunchKernelGGL(k_norm_log, dim3(1), dim3(1024), 0, h->stream, last, h->Wa, acc, 0);     // :184-188
    for (int k = 1; k <= d; k++)                                        // :190-194
        hipLaunchKernelGGL(k_scal_core_acc, g1((size_t)r[k - 1] * h->n1[k] * r[k]), dim3(256), 0, h->stream, core_dev(h, k), r[k - 1], h->n1[k], r[k], RM, SS,
                           acc, d, (k == d) ? 1 : 0);
    push_ranks(h);
    HIPCHECK(hipGetLastError());
    return TTX_OK;
}

// rank, Jacobi sweeps and singular values of the core just decomposed: written by the device into pinned memory, polled here
static int svd_fetch(ttx_engine *h, const double *sv, const int *info, int q, int *inf2, double *svh)
{
    if (!h->h_svd) HIPCHECK(hipHostMalloc((void **)&h->h_svd, sizeof(double) * ((size_t)h->RM + 8)));
    if (getenv("TTX_SVD_POLL") && atoi(getenv("TTX_SVD_POLL")) == 0) {
        HIPCHECK(hipMemcpyAsync(inf2, info, sizeof(int) * 2, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipMemcpyAsync(svh, sv, sizeof(double) * q, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        return TTX_OK;
    }
    h->svd_seq += 1.0;
    hipLaunchKernelGGL(k_svd_report, dim3(1), dim3(64), 0, h->stream, sv, info, q, (volatile double *)h->h_svd, h->svd_seq);
    volatile double *hv = h->h_svd;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (hv[0] != h->svd_seq) {
        if ((++spins & 0xfff) == 0) {
            if (hipStreamQuery(h->stream) == hipSuccess && hv[0] != h->svd_seq) { HIPCHECK(hipStreamSynchronize(h->stream)); break; }   // (kernel failed to launch: do not spin for ever)
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) return fail(TTX_EHIP, "dtt_svd: no report from the device within 30 s");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    inf2[0] = (int)hv[1]; inf2[1] = (int)hv[2];
    for (int j = 0; j < q; j++) svh[j] = hv[3 + j];
    return TTX_OK;
}
static int svd_impl(ttx_engine *h, double tol, int rmax)
{
    const int d = h->d, RM = h->RM; const size_t SS = h->P.SS;
    if (d <= 1) return TTX_OK;
    int rc = ort_impl(h);
    if (rc) return rc;
    std::vector<int32_t> &r = h->rfinal;
    double *Rm = h->Sm, *Rt = Rm + (size_t)RM * RM, *Vb = Rt + (size_t)RM * RM, *US = Vb + (size_t)RM * RM, *Vs = US + (size_t)RM * RM;
    double *tau = h->Sm + 8 * (size_t)RM * RM, *sv = tau + RM;
    int *perm = h->Si, *info = h->Si + RM;
    double lognrm = 0.0, s2;
    std::vector<double> svh(RM);
    for (int k = d; k >= 2; k--) {                                      // lib/tt.f90:329-356
        const int mm = r[k - 1], n = h->n1[k], nn = n * r[k], kk = r[k - 2] * h->n1[k - 1];
        if (mm > nn) {
            // tall unfolding (only possible for trains that do not come from a cross): A (mm x nn) = Q R, R = Ub S Vb^T
            //   =>  A = (Q Ub S) Vb^T: Q Ub S goes into the previous core, Vb^T is the new core
            hipLaunchKernelGGL(k_pack_core, g1((size_t)mm * nn), dim3(256), 0, h->stream, core_dev(h, k), h->Wa, mm, n, r[k], RM, SS, 0);
            if ((rc = qr(h, mm, nn, h->Wa, Rm, tau))) return rc;        // Wa -> Q (mm x nn), Rm = R (nn x nn)
            if ((rc = jacobi(h, nn, nn, Rm, Vb, sv, perm, info, tol, rmax))) return rc;
            int inf2[2];
            if ((rc = svd_fetch(h, sv, info, nn, inf2, svh.data()))) return rc;
            const int rr = inf2[0];
            s2 = 0.0; for (int j = 0; j < rr; j++) s2 += svh[j] * svh[j];
            const double nrm = std::sqrt(s2);
            if (nrm != 0.0) lognrm += std::log(nrm);
            hipLaunchKernelGGL(k_take_cols, g1((size_t)nn * rr), dim3(256), 0, h->stream, nn, rr, Rm, nn, perm, sv, nrm != 0.0 ? 1.0 / nrm : 1.0, US);
            gemm(h, mm, rr, nn, h->Wa, mm, US, nn, h->Wd, mm);          // Q Ub S  (mm x rr)
            hipLaunchKernelGGL(k_pack_core, g1((size_t)kk * mm), dim3(256), 0, h->stream, core_dev(h, k - 1), h->Wb, r[k - 2], h->n1[k - 1], mm, RM, SS, 0);
            gemm(h, kk, rr, mm, h->Wb, kk, h->Wd, mm, h->Wc, kk);
            hipLaunchKernelGGL(k_unpack_core, g1((size_t)kk * rr), dim3(256), 0, h->stream, core_dev(h, k - 1), h->Wc, r[k - 2], h->n1[k - 1], rr, RM, SS, 0, 1.0);
            hipLaunchKernelGGL(k_take_cols, g1((size_t)nn * rr), dim3(256), 0, h->stream, nn, rr, Vb, nn, perm, (const double *)nullptr, 1.0, Vs);
            hipLaunchKernelGGL(k_unpack_core, g1((size_t)rr * nn), dim3(256), 0, h->stream, core_dev(h, k), Vs, rr, n, r[k], RM, SS, 1, 1.0);
            r[k - 1] = rr;
            continue;
        }
        // A (mm x nn) = R1^T Q1^T with A^T = Q1 R1 ; R1^T = Ub S Vb^T  =>  A = Ub S (Q1 Vb)^T
        hipLaunchKernelGGL(k_pack_core, g1((size_t)mm * nn), dim3(256), 0, h->stream, core_dev(h, k), h->Wa, mm, n, r[k], RM, SS, 1);
        if ((rc = qr(h, nn, mm, h->Wa, Rm, tau))) return rc;            // Wa -> Q1 (nn x mm), Rm = R1 (mm x mm)
        hipLaunchKernelGGL(k_transpose, g1((size_t)mm * mm), dim3(256), 0, h->stream, mm, mm, Rm, Rt);
        if ((rc = jacobi(h, mm, mm, Rt, Vb, sv, perm, info, tol, rmax))) return rc;
        int inf[2];
        if ((rc = svd_fetch(h, sv, info, mm, inf, svh.data()))) return rc;
        const int rr = inf[0];
        if (getenv("TTX_JAC_TRACE")) fprintf(stderr, "svd core %d: %d x %d, %d Jacobi sweeps, rank %d\n", k, mm, mm, inf[1], rr);
        s2 = 0.0; for (int j = 0; j < rr; j++) s2 += svh[j] * svh[j];
        const double nrm = std::sqrt(s2);
        if (nrm != 0.0) lognrm += std::log(nrm);
        hipLaunchKernelGGL(k_take_cols, g1((size_t)mm * rr), dim3(256), 0, h->stream, mm, rr, Rt, mm, perm, sv, nrm != 0.0 ? 1.0 / nrm : 1.0, US);
        hipLaunchKernelGGL(k_pack_core, g1((size_t)kk * mm), dim3(256), 0, h->stream, core_dev(h, k - 1), h->Wb, r[k - 2], h->n1[k - 1], mm, RM, SS, 0);
        gemm(h, kk, rr, mm, h->Wb, kk, US, mm, h->Wc, kk);              // U S pushed into the previous core (:344)
        hipLaunchKernelGGL(k_unpack_core, g1((size_t)kk * rr), dim3(256), 0, h->stream, core_dev(h, k - 1), h->Wc, r[k - 2], h->n1[k - 1], rr, RM, SS, 0, 1.0);
        hipLaunchKernelGGL(k_take_cols, g1((size_t)mm * rr), dim3(256), 0, h->stream, mm, rr, Vb, mm, perm, (const double *)nullptr, 1.0, Vs);
        gemm(h, nn, rr, mm, h->Wa, nn, Vs, mm, h->Wd, nn);              // Y = Q1 Vb(:, kept)  (nn x rr)
        hipLaunchKernelGGL(k_unpack_core, g1((size_t)rr * nn), dim3(256), 0, h->stream, core_dev(h, k), h->Wd, rr, n, r[k], RM, SS, 1, 1.0);
        r[k - 1] = rr;
    }
    const size_t first = (size_t)r[0] * h->n1[1] * r[1];
    hipLaunchKernelGGL(k_pack_core, g1(first), dim3(256), 0, h->stream, core_dev(h, 1), h->Wa, r[0], h->n1[1], r[1], RM, SS, 0);
    if ((rc = sumsq(h, first, h->Wa, &s2))) return rc;
    double nf = std::sqrt(s2), firstscale = 1.0;
    if (nf != 0.0) { firstscale = 1.0 / nf; lognrm += std::log(nf); }
    lognrm /= d;
    const double nrm = std::exp(lognrm);
    for (int k = 1; k <= d; k++)
        hipLaunchKernelGGL(k_scal_core, g1((size_t)r[k - 1] * h->n1[k] * r[k]), dim3(256), 0, h->stream, core_dev(h, k), r[k - 1], h->n1[k], r[k], RM, SS,
                           (k == 1) ? nrm * firstscale : nrm);
    push_ranks(h);
    HIPCHECK(hipGetLastError());
    return TTX_OK;
}

// (ort / svd change the train in place: on a multi-process engine that would mean redistributing cores whose ranks have changed;
//  take a replica with ttx_replicate and work on that)
extern "C" int ttx_ort(ttx_engine *h) { int rc = tt_prepare(h, "dtt_ort"); return rc ? rc : ort_impl(h); }
extern "C" int ttx_svd(ttx_engine *h, double tol, int32_t rmax) { int rc = tt_prepare(h, "dtt_svd"); return rc ? rc : svd_impl(h, tol, rmax); }

extern "C" int ttx_norm(ttx_engine *h, double tol, double *val)
{
    if (h && h->ran && h->W > 1) return with_replica(h, [&](ttx_engine *e) { return ttx_norm(e, tol, val); });
    int rc = tt_prepare(h, "dtt_norm");
    if (rc) return rc;
    if (!val) return fail(TTX_EINVAL, "dtt_norm: null result");
    // the reference works on a copy (tmp = arg, lib/tt.f90:1082): back the cores and ranks up, restore afterwards
    const size_t tot = (size_t)h->G * h->NC * h->P.CS;
    if (!h->bak) { if ((rc = dev_alloc(h, &h->bak, tot))) return rc; }
    HIPCHECK(hipMemcpyAsync(h->bak, h->P.arg, sizeof(double) * tot, hipMemcpyDeviceToDevice, h->stream));
    std::vector<int32_t> rsave = h->rfinal;
    const int d = h->d;
    double s2 = 0.0;
    if (tol >= 0.0) {
        rc = svd_impl(h, tol, 0);
        if (!rc) { const size_t sz = (size_t)h->rfinal[0] * h->n1[1] * h->rfinal[1];
                   hipLaunchKernelGGL(k_pack_core, g1(sz), dim3(256), 0, h->stream, core_dev(h, 1), h->Wa, h->rfinal[0], h->n1[1], h->rfinal[1], h->RM, h->P.SS, 0);
                   rc = sumsq(h, sz, h->Wa, &s2); }
    } else {
        rc = ort_impl(h);
        if (!rc) { const size_t sz = (size_t)h->rfinal[d - 1] * h->n1[d] * h->rfinal[d];
                   hipLaunchKernelGGL(k_pack_core, g1(sz), dim3(256), 0, h->stream, core_dev(h, d), h->Wa, h->rfinal[d - 1], h->n1[d], h->rfinal[d], h->RM, h->P.SS, 0);
                   rc = sumsq(h, sz, h->Wa, &s2); }
    }
    HIPCHECK(hipMemcpyAsync(h->P.arg, h->bak, sizeof(double) * tot, hipMemcpyDeviceToDevice, h->stream));
    h->rfinal = rsave;
    push_ranks(h);
    if (rc) return rc;
    *val = std::pow(std::sqrt(s2), d);                                  // :1089
    return TTX_OK;
}

extern "C" int ttx_dot(ttx_engine *x, ttx_engine *y, double *val)
{
    if (x && y && x->ran && y->ran && (x->W > 1 || y->W > 1)) {      // replicas of whichever train is spread over processes
        if (x->W > 1) return with_replica(x, [&](ttx_engine *ex) { return ttx_dot(ex, y, val); });
        return with_replica(y, [&](ttx_engine *ey) { return ttx_dot(x, ey, val); });
    }
    int rc = tt_prepare(x, "dtt_dot");
    if (rc || (rc = tt_prepare(y, "dtt_dot"))) return rc;
    if (!val) return fail(TTX_EINVAL, "dtt_dot: null result");
    if (x->d != y->d) return fail(TTX_EINVAL, "dtt_dot: dimensions not match");          // lib/tt.f90:1162
    for (int k = 1; k <= x->d; k++) if (x->n1[k] != y->n1[k]) return fail(TTX_EINVAL, "dtt_dot: sizes not match");
    if (x->cfg.device != y->cfg.device) return fail(TTX_EINVAL, "dtt_dot: both tensor trains must live on the same GPU");
    const int d = x->d;
    double *phi = x->Sm, *phi2 = x->Sm + (size_t)x->RM * x->RM;        // needs rx*ry <= RM_x^2: checked below
    const double one = 1.0;
    HIPCHECK(hipMemcpyAsync(phi, &one, sizeof(double), hipMemcpyHostToDevice, x->stream));
    HIPCHECK(hipStreamSynchronize(y->stream));
    for (int i = 1; i <= d; i++) {
        const int rx0 = x->rfinal[i - 1], rx1 = x->rfinal[i], ry0 = y->rfinal[i - 1], ry1 = y->rfinal[i], n = x->n1[i];
        // x's scratch holds: core i of y packed (ry0*n*ry1 -> Wb), phi*core (rx0*n*ry1 -> Wc), the r x r interface matrices (Sm)
        if ((size_t)rx1 * ry1 > (size_t)x->RM * x->RM || (size_t)rx0 * ry0 > (size_t)x->RM * x->RM || (size_t)rx0 * n * ry1 > x->P.CS ||
            (size_t)ry0 * n * ry1 > x->P.CS)
            return fail(TTX_EINVAL, "dtt_dot: ranks of y exceed the work space of x (call dot_product(y, x) or raise maxrank of x)");
        hipLaunchKernelGGL(k_pack_core, g1((size_t)ry0 * n * ry1), dim3(256), 0, x->stream, core_dev(y, i), x->Wb, ry0, n, ry1, y->RM, y->P.SS, 0);
        gemm(x, rx0, n * ry1, ry0, phi, rx0, x->Wb, ry0, x->Wc, rx0);                       // :1169
        hipLaunchKernelGGL(k_pack_core, g1((size_t)rx0 * n * rx1), dim3(256), 0, x->stream, core_dev(x, i), x->Wa, rx0, n, rx1, x->RM, x->P.SS, 0);
        hipLaunchKernelGGL(k_transpose, g1((size_t)rx0 * n * rx1), dim3(256), 0, x->stream, rx0 * n, rx1, x->Wa, x->Wd);
        gemm(x, rx1, ry1, rx0 * n, x->Wd, rx1, x->Wc, rx0 * n, phi2, rx1);                  // :1170
        std::swap(phi, phi2);
    }
    HIPCHECK(hipMemcpyAsync(val, phi, sizeof(double), hipMemcpyDeviceToHost, x->stream));
    HIPCHECK(hipStreamSynchronize(x->stream));
    HIPCHECK(hipGetLastError());
    return TTX_OK;
}

// ztt_quad of a MULTI-PROCESS job (lib/dmrgg.f90:1418-1523 is collective over mybonds): every process chains the matrices of the
// cores it holds, the partial products travel by a SUM all-reduce into zero-padded slots, every process folds them in rank order
static int zquad_multi(ttx_engine *h, int32_t nf, const double *w, double *out)
{
    if (!h->ran) return fail(TTX_ESTATE, "ztt_quad: run dtt_dmrgg first");
    HIPCHECK(hipSetDevice(h->cfg.device));
    const int d = h->d, RM = h->RM, W = h->W, nproc = h->cfg.nproc;
    const size_t lds = sizeof(double) * 4 * (size_t)RM * RM;
    if (lds > 160 * 1024) return fail(TTX_EINVAL, "ztt_quad: maxrank %d needs %zu bytes of LDS for the chain (limit 160 KB, maxrank <= 71)", RM, lds);
    size_t sumn = 0;
    for (int k = 1; k <= d; k++) sumn += h->n1[k];
    auto lo_of = [&](int wr) { return h->own[(int)((long long)nproc * wr / W)]; };
    auto hi_of = [&](int wr) { const int ge = (int)((long long)nproc * (wr + 1) / W); return ge == nproc ? d : h->own[ge] - 1; };
    const int plo = lo_of(h->wrank), phi = hi_of(h->wrank);
    std::vector<const double *> cp(d + 2, nullptr);
    for (int k = plo; k <= phi; k++) cp[k] = core_dev(h, k);
    std::vector<int> rr(h->rfinal.begin(), h->rfinal.end()), dims(2 * W);
    for (int wr = 0; wr < W; wr++) { dims[2 * wr] = rr[lo_of(wr) - 1]; dims[2 * wr + 1] = rr[hi_of(wr)]; }
    struct Tmp { std::vector<void *> p; ~Tmp() { for (void *q : p) (void)hipFree(q); } } tmp;
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(q, bytes); if (e == hipSuccess) tmp.p.push_back(*q); return e; };
    double *dw, *dtq, *dout, *dpart; const double **dcp; int *dr, *ddims;
    const size_t npart = (size_t)nf * W * 2 * RM * RM;
    HIPCHECK(dalloc((void **)&dw, sizeof(double) * 2 * sumn * nf)); HIPCHECK(dalloc((void **)&dtq, sizeof(double) * (size_t)nf * (d + 1) * 2 * RM * RM));
    HIPCHECK(dalloc((void **)&dout, sizeof(double) * 2 * nf)); HIPCHECK(dalloc((void **)&dcp, sizeof(double *) * (d + 2))); HIPCHECK(dalloc((void **)&dr, sizeof(int) * (d + 1)));
    HIPCHECK(dalloc((void **)&dpart, sizeof(double) * npart)); HIPCHECK(dalloc((void **)&ddims, sizeof(int) * 2 * W));
    HIPCHECK(hipMemcpy(dw, w, sizeof(double) * 2 * sumn * nf, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dcp, cp.data(), sizeof(double *) * (d + 2), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dr, rr.data(), sizeof(int) * (d + 1), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(ddims, dims.data(), sizeof(int) * 2 * W, hipMemcpyHostToDevice));
    HIPCHECK(hipMemsetAsync(dpart, 0, sizeof(double) * npart, h->stream));
    hipLaunchKernelGGL(k_zquad_build, dim3(d, nf), dim3(256), 0, h->stream, d, RM, h->NM, h->P.SS, h->P.n, (const int *)dr, (const double *const *)dcp, (const double *)dw, 2 * sumn, dtq);
    static size_t a_zs = 0, a_zf = 0;
    if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_zquad_chain_seg), lds, a_zs)) return rc_;
    if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_zquad_fold), lds, a_zf)) return rc_;
    hipLaunchKernelGGL(k_zquad_chain_seg, dim3(nf), dim3(256), lds, h->stream, d, RM, (const int *)dr, (const double *)dtq, plo, phi, h->wrank, W, dpart);
    if (int rc_ = allreduce_big(h, dpart, npart)) return rc_;
    hipLaunchKernelGGL(k_zquad_fold, dim3(nf), dim3(256), lds, h->stream, RM, W, (const int *)ddims, (const double *)dpart, dout);
    HIPCHECK(hipMemcpyAsync(out, dout, sizeof(double) * 2 * nf, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipGetLastError());
    if (h->cb_error) return fail(TTX_EHIP, "host transport: a sendrecv / allreduce callback failed");
    return TTX_OK;
}

extern "C" int ttx_zquad(ttx_engine *h, int32_t nf, const double *w, double *out)
{
    if (h && h->W > 1) { if (nf < 1 || !w || !out) return fail(TTX_EINVAL, "ztt_quad: bad argument"); return zquad_multi(h, nf, w, out); }
    int rc = tt_prepare(h, "ztt_quad");
    if (rc) return rc;
    if (nf < 1 || !w || !out) return fail(TTX_EINVAL, "ztt_quad: bad argument");
    const int d = h->d, RM = h->RM;
    size_t sumn = 0;
    for (int k = 1; k <= d; k++) sumn += h->n1[k];
    std::vector<const double *> cp(d + 2, nullptr);
    for (int k = 1; k <= d; k++) cp[k] = core_dev(h, k);
    std::vector<int> rr(h->rfinal.begin(), h->rfinal.end());
    // the chain kernel keeps two complex r x r matrices of the LARGEST OCCURRING rank in LDS
    int rmax = 1;
    for (int k = 0; k <= d; k++) rmax = std::max(rmax, rr[k]);
    const size_t lds = sizeof(double) * 4 * (size_t)RM * RM;
    if (lds > 160 * 1024) return fail(TTX_EINVAL, "ztt_quad: maxrank %d needs %zu bytes of LDS for the chain (limit 160 KB, maxrank <= 71)", RM, lds);
    // the five temporaries are released on every path out of this function
    struct Tmp { std::vector<void *> p; ~Tmp() { for (void *q : p) (void)hipFree(q); } } tmp;
    auto dalloc = [&](void **q, size_t bytes) -> hipError_t { hipError_t e = hipMalloc(q, bytes); if (e == hipSuccess) tmp.p.push_back(*q); return e; };
    double *dw, *dtq, *dout; const double **dcp; int *dr;
    HIPCHECK(dalloc((void **)&dw, sizeof(double) * 2 * sumn * nf)); HIPCHECK(dalloc((void **)&dtq, sizeof(double) * (size_t)nf * (d + 1) * 2 * RM * RM));
    HIPCHECK(dalloc((void **)&dout, sizeof(double) * 2 * nf)); HIPCHECK(dalloc((void **)&dcp, sizeof(double *) * (d + 2))); HIPCHECK(dalloc((void **)&dr, sizeof(int) * (d + 1)));
    HIPCHECK(hipMemcpy(dw, w, sizeof(double) * 2 * sumn * nf, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dcp, cp.data(), sizeof(double *) * (d + 2), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dr, rr.data(), sizeof(int) * (d + 1), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_zquad_build, dim3(d, nf), dim3(256), 0, h->stream, d, RM, h->NM, h->P.SS, h->P.n, (const int *)dr, (const double *const *)dcp, (const double *)dw, 2 * sumn, dtq);
    static size_t a_zq = 0;
    if (int rc_ = ensure_lds(reinterpret_cast<const void *>(k_zquad_chain), lds, a_zq)) return rc_;
    hipLaunchKernelGGL(k_zquad_chain, dim3(nf), dim3(256), lds, h->stream, d, RM, (const int *)dr, (const double *)dtq, dout);
    HIPCHECK(hipMemcpyAsync(out, dout, sizeof(double) * 2 * nf, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipGetLastError());
    return TTX_OK;
}

extern "C" int ttx_ijk(ttx_engine *h, const int32_t *ind, double *val)
{
    int rc = tt_prepare(h, "dtt_ijk");
    if (rc) return rc;
    if (!ind || !val) return fail(TTX_EINVAL, "dtt_ijk: null argument");
    const int d = h->d;
    for (int k = 1; k <= d; k++) if (ind[k - 1] <= 0 || ind[k - 1] > h->n1[k]) { *val = -3.0; return TTX_OK; }   // lib/tt.f90:639
    // x = U_d(:, ind_d, 1); for i = d-1..1: x = U_i(:, ind_i, :) x -- matrix-vector steps through the GEMM kernel
    double *xv = h->Sm, *zv = h->Sm + h->RM;
    hipLaunchKernelGGL(k_pack_core, g1((size_t)h->rfinal[d - 1] * h->n1[d] * h->rfinal[d]), dim3(256), 0, h->stream, core_dev(h, d), h->Wa, h->rfinal[d - 1], h->n1[d], h->rfinal[d], h->RM, h->P.SS, 0);
    HIPCHECK(hipMemcpyAsync(xv, h->Wa + (size_t)h->rfinal[d - 1] * (ind[d - 1] - 1), sizeof(double) * h->rfinal[d - 1], hipMemcpyDeviceToDevice, h->stream));
    for (int i = d - 1; i >= 1; i--) {
        const int q0 = h->rfinal[i - 1], q1 = h->rfinal[i], n = h->n1[i];
        hipLaunchKernelGGL(k_pack_core, g1((size_t)q0 * n * q1), dim3(256), 0, h->stream, core_dev(h, i), h->Wa, q0, n, q1, h->RM, h->P.SS, 0);
        gemm(h, q0, 1, q1, h->Wa + (size_t)q0 * (ind[i - 1] - 1), q0 * n, xv, q1, zv, q0);
        std::swap(xv, zv);
    }
    HIPCHECK(hipMemcpyAsync(val, xv, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    HIPCHECK(hipGetLastError());
    return TTX_OK;
}

extern "C" int ttx_arith(const ttx_engine *h) { return h ? h->P.arith : -1; }
extern "C" int ttx_sweep_path(const ttx_engine *h) { return !h ? -1 : h->cluster ? 2 : h->fused ? 1 : 0; }
extern "C" int64_t ttx_resid_halfsteps(const ttx_engine *h) { return h ? h->n_resid : 0; }
extern "C" int ttx_cluster_fallbacks(const ttx_engine *h) { return h ? h->cluster_fallbacks : 0; }
extern "C" int ttx_det_fallbacks(const ttx_engine *h) { return h ? h->det_fallbacks + h->de5_fallbacks : 0; }
extern "C" int ttx_fun_id(const ttx_engine *h) { return h ? h->cfg.fun_id : -1; }
extern "C" int ttx_set_profile(ttx_engine *h, int on) { if (!h) return fail(TTX_EINVAL, "null"); h->profile = on != 0; return TTX_OK; }
extern "C" int ttx_kernel_stats(const ttx_engine *h, int64_t launches[TTX_K_NKINDS], double ms[TTX_K_NKINDS], double bytes[TTX_K_NKINDS])
{
    if (!h) return fail(TTX_EINVAL, "null");
    for (int k = 0; k < TTX_K_NKINDS; k++) { launches[k] = h->k_launches[k]; ms[k] = h->k_ms[k]; bytes[k] = h->k_bytes[k]; }
    return TTX_OK;
}

// ---- kernel-level test entry points ------------------------------------------------------------------
extern "C" int ttx_k_residual_argmax(int32_t device, int32_t m, int32_t r, const double *a, const double *F, const double *x,
                                     double *b_out, int32_t *imax, double *bmax)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    if (r > 256 || m < 1) return fail(TTX_EINVAL, "ttx_k_residual_argmax: bad sizes");
    double *da, *dF, *dx, *db; Partial *dp;
    int nb = (m + TTX_BLK - 1) / TTX_BLK;
    HIPCHECK(hipMalloc((void **)&da, sizeof(double) * m)); HIPCHECK(hipMalloc((void **)&dF, sizeof(double) * (size_t)m * std::max(r, 1)));
    HIPCHECK(hipMalloc((void **)&dx, sizeof(double) * std::max(r, 1))); HIPCHECK(hipMalloc((void **)&db, sizeof(double) * m));
    HIPCHECK(hipMalloc((void **)&dp, sizeof(Partial) * nb));
    HIPCHECK(hipMemcpy(da, a, sizeof(double) * m, hipMemcpyHostToDevice));
    if (r > 0) { HIPCHECK(hipMemcpy(dF, F, sizeof(double) * (size_t)m * r, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dx, x, sizeof(double) * r, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_resid_argmax, dim3(nb), dim3(TTX_BLK), 0, 0, m, r, (size_t)m, da, dF, dx, db, dp);
    HIPCHECK(hipDeviceSynchronize());
    std::vector<Partial> parts(nb);
    HIPCHECK(hipMemcpy(parts.data(), dp, sizeof(Partial) * nb, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(b_out, db, sizeof(double) * m, hipMemcpyDeviceToHost));
    double ba = -1; int bi = INT_MAX; double bv = 0;
    for (auto &p : parts) if (p.absmax > ba || (p.absmax == ba && p.idx < bi)) { ba = p.absmax; bi = p.idx; bv = p.val; }
    *imax = bi; *bmax = bv;
    (void)hipFree(da); (void)hipFree(dF); (void)hipFree(dx); (void)hipFree(db); (void)hipFree(dp);
    return TTX_OK;
}

extern "C" int ttx_k_residual_bench(int32_t device, int64_t m, int32_t r, int32_t iters, double *avg_ms, double *bytes)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    if (m < 1 || m > 2000000000LL || r < 1 || r > 256 || iters < 1) return fail(TTX_EINVAL, "ttx_k_residual_bench: bad sizes");
    double *da, *dF, *dx, *db; Partial *dp;
    const int nb = 256 * 8;                                 // 8 blocks per CU, grid-stride
    HIPCHECK(hipMalloc((void **)&da, sizeof(double) * m)); HIPCHECK(hipMalloc((void **)&dF, sizeof(double) * (size_t)m * r));
    HIPCHECK(hipMalloc((void **)&dx, sizeof(double) * r)); HIPCHECK(hipMalloc((void **)&db, sizeof(double) * m));
    HIPCHECK(hipMalloc((void **)&dp, sizeof(Partial) * nb));
    hipLaunchKernelGGL(k_fill_synth, dim3(2048), dim3(256), 0, 0, da, (size_t)m, 1ull);
    hipLaunchKernelGGL(k_fill_synth, dim3(2048), dim3(256), 0, 0, dF, (size_t)m * r, 2ull);
    hipLaunchKernelGGL(k_fill_synth, dim3(1), dim3(256), 0, 0, dx, (size_t)r, 3ull);
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_resid_argmax_stream, dim3(nb), dim3(TTX_BLK), 0, 0, (long long)m, r, (size_t)m, da, dF, dx, db, dp);
    HIPCHECK(hipEventRecord(e0, 0));
    for (int w = 0; w < iters; w++) hipLaunchKernelGGL(k_resid_argmax_stream, dim3(nb), dim3(TTX_BLK), 0, 0, (long long)m, r, (size_t)m, da, dF, dx, db, dp);
    HIPCHECK(hipEventRecord(e1, 0));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = ms / iters;
    *bytes = 8.0 * ((double)m * r + r + 2.0 * m);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(da); (void)hipFree(dF); (void)hipFree(dx); (void)hipFree(db); (void)hipFree(dp);
    return TTX_OK;
}

// ---- latency probe: the unit costs of the dependent chains that bound the sweep kernels at BASELINE sizes ----------
// one wave, one chain each, timed with the 100 MHz wall clock (s_memrealtime): out[0] ns per dependent fp64 multiply,
// out[1] ns per dependent fp64 add after multiply pair (the running sums of the Ising C integrand), out[2] ns per
// dependent L2 round trip (pointer chase with L1-bypassing loads, 64 KB ring = L2-resident), out[3] ns per dependent
// LDS read, out[4] ns per fp64 IEEE division in a dependent chain
__global__ __launch_bounds__(64) void k_latency_probe(const unsigned *ring, double *out, double seed)
{
    __shared__ unsigned lds_ring[1024];
    const int lane = threadIdx.x;
    const int N = 4096;
    for (int x = lane; x < 1024; x += 64) lds_ring[x] = (unsigned)((x * 37 + 11) & 1023);
    __syncthreads();
    double x = seed, y = 1.0 + 1e-9 * lane;
    long long t0 = wall_clock64();
#pragma unroll 16
    for (int k = 0; k < N; k++) x = x * y;
    long long t1 = wall_clock64();
    double s = seed, pk = 1.0;
#pragma unroll 16
    for (int k = 0; k < N; k++) { pk = pk * y; s = s + pk; }
    long long t2 = wall_clock64();
    unsigned p = (unsigned)lane;
    for (int k = 0; k < 1024; k++) { unsigned v; asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(ring + p) : "memory"); p = v; }
    long long t3 = wall_clock64();
    unsigned q = (unsigned)lane;
#pragma unroll 8
    for (int k = 0; k < N; k++) q = lds_ring[q];
    long long t4 = wall_clock64();
    double z = seed + 3.0;
#pragma unroll 4
    for (int k = 0; k < 1024; k++) z = (z - 1.0) / (z + 1.0) + 3.0;
    long long t5 = wall_clock64();
    if (lane == 0) {
        out[0] = 10.0 * (double)(t1 - t0) / N; out[1] = 10.0 * (double)(t2 - t1) / N; out[2] = 10.0 * (double)(t3 - t2) / 1024;
        out[3] = 10.0 * (double)(t4 - t3) / N; out[4] = 10.0 * (double)(t5 - t4) / 1024;
        out[5] = x + s + (double)p + (double)q + z;      // keep the chains alive
    }
}
// development probe: ns per element of the LDS-broadcast folds (one wave per block, `nblk` blocks), rows of `len` doubles
__global__ __launch_bounds__(64) void k_fold_probe(int len, int reps, double *out)
{
    __shared__ double row[1024];
    const int lane = threadIdx.x;
    for (int x = lane; x < 1024; x += 64) row[x] = 1.0 + 1e-9 * x;
    __syncthreads();
    double a = 1.0, s = 0.0;
    long long t0 = wall_clock64();
    for (int r = 0; r < reps; r++) a = lds_chain(a, row + (r & 7), len);
    long long t1 = wall_clock64();
    for (int r = 0; r < reps; r++) s = lds_sum_chain(s, row + (r & 7), len);
    long long t2 = wall_clock64();
    double u = 0.7 + 1e-9 * lane, b = 1.0;
    for (int r = 0; r < reps; r++) de_run<true>(b, u, 0.999, row, len);
    long long t3 = wall_clock64();
    // issue throughput of independent instructions on one wave: 8 streams each
    double f[8], g8[8]; float h8[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { f[q] = 1.0 + 1e-9 * (lane + q); g8[q] = 1.5 + 1e-3 * (lane + q); h8[q] = 1.5f + 1e-3f * (lane + q); }
    long long t4 = wall_clock64();
    for (int r = 0; r < 512; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) f[q] = __builtin_fma(f[q], 1.0000001, 1e-9);
    }
    long long t5 = wall_clock64();
    for (int r = 0; r < 512; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) g8[q] = __builtin_amdgcn_rcp(g8[q]);
    }
    long long t6 = wall_clock64();
    for (int r = 0; r < 512; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) h8[q] = __builtin_amdgcn_rcpf(h8[q]);
    }
    long long t7 = wall_clock64();
    double n8[8], d8[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { n8[q] = -0.3 - 1e-3 * q; d8[q] = 1.7 + 1e-3 * (lane + q); }
    for (int r = 0; r < 512; r++) {
#pragma unroll
        for (int q = 0; q < 8; q++) n8[q] = fdiv_unit(n8[q], d8[q]);
    }
    long long t8 = wall_clock64();
    if (lane == 0 && blockIdx.x == 0) {
        out[0] = 10.0 * (double)(t1 - t0) / ((double)reps * len); out[1] = 10.0 * (double)(t2 - t1) / ((double)reps * len);
        out[2] = 10.0 * (double)(t3 - t2) / ((double)reps * (len + 1));
        out[3] = 10.0 * (double)(t5 - t4) / 4096.0; out[4] = 10.0 * (double)(t6 - t5) / 4096.0; out[5] = 10.0 * (double)(t7 - t6) / 4096.0;
        out[6] = 10.0 * (double)(t8 - t7) / 4096.0;
        double acc = a + s + b;
#pragma unroll
        for (int q = 0; q < 8; q++) acc += f[q] + g8[q] + (double)h8[q] + n8[q];
        out[7] = acc;
    }
}
extern "C" int ttx_k_fold_probe(int32_t device, int32_t nblk, int32_t len, double out[7])
{
    HIPCHECK(hipSetDevice(device));
    double *d; HIPCHECK(hipMalloc((void **)&d, 128));
    double o[8];
    hipLaunchKernelGGL(k_fold_probe, dim3(nblk), dim3(64), 0, 0, len, 200, d);
    hipLaunchKernelGGL(k_fold_probe, dim3(nblk), dim3(64), 0, 0, len, 200, d);
    HIPCHECK(hipMemcpy(o, d, 64, hipMemcpyDeviceToHost));
    for (int k = 0; k < 7; k++) out[k] = o[k];
    (void)hipFree(d);
    return TTX_OK;
}

extern "C" int ttx_k_latency_probe(int32_t device, double out[5])
{
    if (!out) return fail(TTX_EINVAL, "ttx_k_latency_probe: null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    const int R = 16384;                                       // 64 KB ring of indices, a single cycle with stride 4099
    std::vector<unsigned> ring(R);
    for (int i = 0; i < R; i++) ring[i] = (unsigned)((i + 4099) % R);
    unsigned *dr; double *dout;
    HIPCHECK(hipMalloc((void **)&dr, sizeof(unsigned) * R)); HIPCHECK(hipMalloc((void **)&dout, sizeof(double) * 8));
    HIPCHECK(hipMemcpy(dr, ring.data(), sizeof(unsigned) * R, hipMemcpyHostToDevice));
    double best[5] = {1e30, 1e30, 1e30, 1e30, 1e30}, o[8];
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(k_latency_probe, dim3(1), dim3(64), 0, 0, (const unsigned *)dr, dout, 1.0 + 1e-12 * rep);
        HIPCHECK(hipMemcpy(o, dout, sizeof(double) * 6, hipMemcpyDeviceToHost));
        for (int k = 0; k < 5; k++) best[k] = std::min(best[k], o[k]);
    }
    for (int k = 0; k < 5; k++) out[k] = best[k];
    (void)hipFree(dr); (void)hipFree(dout);
    return TTX_OK;
}

static int k_eval_impl(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
                       const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out, int arith);
extern "C" int ttx_k_eval(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
                          const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out)
{ return k_eval_impl(device, fun_id, d, n, par, npar, aux, naux, npts, ind, out, 0); }
extern "C" int ttx_k_eval_arith(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
                                const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out, int32_t arith)
{ return k_eval_impl(device, fun_id, d, n, par, npar, aux, naux, npts, ind, out, arith); }
static int k_eval_impl(int32_t device, int32_t fun_id, int32_t d, const int32_t *n, const double *par, int32_t npar,
                       const double *aux, int32_t naux, int64_t npts, const int32_t *ind, double *out, int arith)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    DevProb P{};
    std::vector<int> n1(d + 2, 1);
    for (int k = 1; k <= d; k++) n1[k] = n[k - 1];
    int *dn, *dind; double *dpar, *daux = nullptr, *dout;
    HIPCHECK(hipMalloc((void **)&dn, sizeof(int) * (d + 2))); HIPCHECK(hipMalloc((void **)&dpar, sizeof(double) * npar));
    HIPCHECK(hipMalloc((void **)&dind, sizeof(int) * npts * d)); HIPCHECK(hipMalloc((void **)&dout, sizeof(double) * npts));
    HIPCHECK(hipMemcpy(dn, n1.data(), sizeof(int) * (d + 2), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dpar, par, sizeof(double) * npar, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dind, ind, sizeof(int) * npts * d, hipMemcpyHostToDevice));
    if (aux && naux > 0) { HIPCHECK(hipMalloc((void **)&daux, sizeof(double) * naux)); HIPCHECK(hipMemcpy(daux, aux, sizeof(double) * naux, hipMemcpyHostToDevice)); }
    P.d = d; P.n = dn; P.par = dpar; P.aux = daux; P.npar = npar; P.fun_id = fun_id;
    P.ising_id = (fun_id == TTX_FUN_ISING) ? (int)par[2 * n[0]] : 0;
    P.arith = (arith == TTX_ARITH_FAST && fun_id == TTX_FUN_ISING && P.ising_id != 1) ? 1 : 0;     // the one-thread evaluator f_ising_fast (nodes must lie in [0,1])
    P.mvn_norm = (fun_id == TTX_FUN_MVN) ? std::sqrt(powi(2.0 * 3.141592653589793, d) * aux[d + (size_t)d * d]) : 1.0;
    dim3 grid((unsigned)((npts + 255) / 256));
    size_t lds = sizeof(double) * (npar + 2);
    if (fun_id == TTX_FUN_ISING) hipLaunchKernelGGL(k_eval_list<FUN_ISING>, grid, dim3(256), lds, 0, P, (long long)npts, dind, dout);
    else if (fun_id == TTX_FUN_STDNORM) hipLaunchKernelGGL(k_eval_list<FUN_STDNORM>, grid, dim3(256), lds, 0, P, (long long)npts, dind, dout);
    else hipLaunchKernelGGL(k_eval_list<FUN_MVN>, grid, dim3(256), lds, 0, P, (long long)npts, dind, dout);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, dout, sizeof(double) * npts, hipMemcpyDeviceToHost));
    (void)hipFree(dn); (void)hipFree(dpar); (void)hipFree(dind); (void)hipFree(dout); if (daux) (void)hipFree(daux);
    return TTX_OK;
}

__global__ void k_exp_list(long long n, const double *x, double *out)
{
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) out[t] = ttx_exp(x[t]);
}
extern "C" int ttx_k_exp(int32_t device, int64_t n, const double *x, double *out)
{
    if (!x || !out || n < 1) return fail(TTX_EINVAL, "ttx_k_exp: bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    double *dx, *dy;
    HIPCHECK(hipMalloc((void **)&dx, sizeof(double) * n)); HIPCHECK(hipMalloc((void **)&dy, sizeof(double) * n));
    HIPCHECK(hipMemcpy(dx, x, sizeof(double) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_exp_list, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (long long)n, dx, dy);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(out, dy, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dy);
    return TTX_OK;
}
extern "C" int ttx_exp_host(int64_t n, const double *x, double *out)
{
    if (!x || !out || n < 0) return fail(TTX_EINVAL, "ttx_exp_host: bad argument");
    for (int64_t i = 0; i < n; i++) out[i] = ttx_exp(x[i]);
    return TTX_OK;
}

extern "C" int ttx_k_xcc_map(int32_t device, int32_t nblocks, int32_t *out)
{
    if (!out || nblocks < 1) return fail(TTX_EINVAL, "ttx_k_xcc_map: bad argument");
    HIPCHECK(hipSetDevice(device));
    int *d = nullptr;
    HIPCHECK(hipMalloc((void **)&d, sizeof(int) * nblocks));
    hipLaunchKernelGGL(k_xcc_probe, dim3(nblocks), dim3(256), 0, 0, d);
    HIPCHECK(hipMemcpy(out, d, sizeof(int) * nblocks, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    return TTX_OK;
}
extern "C" int ttx_k_lottery(int32_t device, int32_t npnt, int32_t m, int32_t n, int32_t nz, const int32_t *zcol,
                             const int32_t *zrow, uint64_t rngpos, int32_t *points)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(TTX_ENODEV, "no HIP device");
    HIPCHECK(hipSetDevice(device));
    int *dzc, *dzr, *dp;
    HIPCHECK(hipMalloc((void **)&dzc, sizeof(int) * (nz + 1))); HIPCHECK(hipMalloc((void **)&dzr, sizeof(int) * (nz + 1)));
    HIPCHECK(hipMalloc((void **)&dp, sizeof(int) * 2 * npnt));
    if (nz > 0) { HIPCHECK(hipMemcpy(dzc, zcol, sizeof(int) * nz, hipMemcpyHostToDevice)); HIPCHECK(hipMemcpy(dzr, zrow, sizeof(int) * nz, hipMemcpyHostToDevice)); }
    hipLaunchKernelGGL(k_lottery_only, dim3(1), dim3(256), 0, 0, npnt, m, n, nz, dzc, dzr, (unsigned long long)rngpos, dp);
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(points, dp, sizeof(int) * 2 * npnt, hipMemcpyDeviceToHost));
    (void)hipFree(dzc); (void)hipFree(dzr); (void)hipFree(dp);
    return TTX_OK;
}
