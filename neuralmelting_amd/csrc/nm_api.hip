// nm_api.hip — C-ABI (include/nm.h) over the gfx950 kernels.  The context owns all device memory, one HIP
// stream and the HIP events used for measurement; nothing here falls back to a CPU path: without a usable
// HIP device nm_create fails with NM_ERR_HIP.
#include "../../include/nm.h"
#include "nm_kernels.h"
#include "nm_distr.h"
#include "nm_format.h"
#include "nm_parse.h"
#include "nm_lattice.h"
#include "../../include/nm_distr.h"
#include "../../include/nm_parse.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace nm;

namespace {

#ifndef NM_SMALL_BLOCK
#define NM_SMALL_BLOCK 512 // 8 waves: 256 VGPRs per lane, no spills (1024 threads cap at 128 and spilled ~200)
#define NM_SMALL_TPA 2
#endif
// cluster variants of the small kernel: Q workgroups per replica, threads-per-atom scaled so that all 512 threads work
// (a cluster stores the list rows of its own atoms only, and keeps the list twice: a rejected move goes back to the one it started from)
typedef Cfg<NM_SMALL_BLOCK, 2 * NM_SMALL_TPA, 256, 192, unsigned char, true, true, 0, 128, true, true> CfgSmallQ2;
#if NM_SMALL_BLOCK == 1024
typedef Cfg<NM_SMALL_BLOCK, 4 * NM_SMALL_TPA, 256, 256, unsigned char, true, true, 0, 64, true, true> CfgSmallQ4;
#else
typedef Cfg<NM_SMALL_BLOCK, 4 * NM_SMALL_TPA, 256, 192, unsigned char, true, true, 0, 64, true, true> CfgSmallQ4;
#endif
#if NM_SMALL_BLOCK == 1024 // (experiment: 16 waves per workgroup; 32 threads per row would not fit a DPP row)
typedef Cfg<NM_SMALL_BLOCK, 4 * NM_SMALL_TPA, 256, 256, unsigned char, true, true, 0, 64, true, true> CfgSmallQ8;
#else
typedef Cfg<NM_SMALL_BLOCK, 8 * NM_SMALL_TPA, 256, 256, unsigned char, true, true, 0, 32, true, true> CfgSmallQ8; // grids of <= 32 replicas
#endif
typedef Cfg<NM_SMALL_BLOCK, NM_SMALL_TPA, 256, 192, unsigned char, true, true, 0, 256, true, true> CfgSmall;     // N <= 256: everything incl. the byte lists in LDS
// element Al: Sutton-Chen EAM, 4^3 cells only (BASELINE config 4); 200 neighbour slots (134 within rc+skin in the crystal)
typedef Cfg<NM_SMALL_BLOCK, NM_SMALL_TPA, 256, 256, unsigned char, true, true, 1> CfgSmallSC;
typedef Cfg<NM_SMALL_BLOCK, 2 * NM_SMALL_TPA, 256, 256, unsigned char, true, true, 1, 128, true, true> CfgSmallSCQ2; // (own rows, two lists)
typedef Cfg<NM_SMALL_BLOCK, 4 * NM_SMALL_TPA, 256, 256, unsigned char, true, true, 1, 64, true, true> CfgSmallSCQ4;
typedef Cfg<512, 1, 864, 192, unsigned short, false, true> CfgMid;     // N <= 864: list in HBM/L2, saved copies in LDS (here: at 2 workgroups per replica)
// the same at ONE workgroup per replica (more replicas than CUs: the reference's run.sh setting): every pair lies inside the workgroup and
// is listed once (Cfg::HALF, nm_kernels.h)
typedef Cfg<512, 1, 864, 192, unsigned short, false, true, 0, 864, true, false, true> CfgMidH;
// cluster variants: own atoms 216 / 108
typedef Cfg<512, 2, 864, 192, unsigned short, false, true> CfgMidQ4;
// 8 workgroups per replica: 108 own atoms, whose list rows (16-bit, 41 KB at 192 slots) fit in LDS once the saved velocities moved to the spill.
// 192 list slots per atom since round 4 (160 before): at the skin of 0.55 that these sizes now run with, the dense crystals of the P* = 7.5-8 rows
// reach 147 entries (scripts/probe_maxrow6.py)
typedef Cfg<512, 4, 864, 192, unsigned short, true, true, 0, 108, false> CfgMidQ8;
// N <= 2048: saved copies spill to HBM as well.  224 list slots: the list lives in HBM, so slots are cheap, and at skin 0.6 the dense
// crystals (P* = 8: 140 neighbours inside 3.1, the next shell of 36 just beyond) came within ~10 % of the 160 there were
typedef Cfg<512, 1, 2048, 224, unsigned short, false, false> CfgLarge;
// the same at ONE workgroup per replica (more replicas than CUs): half lists (Cfg::HALF), as CfgMidH.  Measured at 256 replicas of 2048 atoms: 254 k
// against 231 k sweeps/s sustained (+10 %)
typedef Cfg<512, 1, 2048, 224, unsigned short, false, false, 0, 2048, false, false, true> CfgLargeH;

thread_local std::string g_create_error;

struct EvPair { hipEvent_t a, b; bool used; uint32_t launch_id; };

} // namespace

struct nm_ctx {
    nm_config cfg;
    int N, nslots, slot0, kind; // kind: 0 small, 1 mid, 2 large
    int cus;                    // workgroups per replica
    bool whole_rows;            // the slot range is made of whole pressure rows (nm_exchange can run on the device)
    double *d_xbuf;
    uint32_t launch_id;
    size_t lds_bytes, aux_doubles;
    double lat, mass, kB, mvv2e, ftm2v, nktv2p, skin, rc;
    int pot; // 0 lj/cut, 1 Sutton-Chen EAM
    uint32_t step;
    hipStream_t stream;
    // device
    double *d_x, *d_v, *d_box, *d_steps, *d_therm, *d_count, *d_et, *d_pf, *d_tq, *d_stats;
    float *d_ratio;
    int *d_slot2buf, *d_status, *d_status_acc, *d_halt, *d_rerun, *d_nswaps, *d_order;
    unsigned long long *d_last_ticks;
    bool use_order; // one workgroup per replica and more replicas than CUs: launch the slowest slots first (nm_order_kernel)
    unsigned int *d_census; // residency census of cluster launches (nm_kernels.h): the grid's counter, then one per cluster
    unsigned int census_base = 0, census_cbase = 0; // what those counters stand at (they only grow; reset_census zeroes both)
    unsigned int *d_rowsync = nullptr; // nm_run_cycles: per local row an arrival counter and a release word, then the abort word (KParams::rowbar, rowgo, cyc_abort)
    bool over;              // the grid holds twice the clusters the chip does at once (pick_q)
    // Calls queued on the stream since the host last looked at the outcome (settle): if a block of them stopped because its
    // cluster grid was not resident or a hand-over timed out, nothing after it has run (KParams::halt) and the same calls are
    // issued again with fewer workgroups per replica.
    struct Op { int kind, arg, trace; uint32_t step, launch_id; };
    std::vector<Op> journal;
    int heals = 0;              // blocks re-issued at a lower Q so far
    int cluster_launches = 0;   // (NM_INJECT_CENSUS counts these)
    std::string note;           // what nm_create's residency probe and later re-issues had to give up (nm_create_note)
    double *d_tape, *d_xtape, *d_trace, *d_xcrit, *d_evalU, *d_evalW, *d_evalF, *d_aux;
    int *d_tape_off;
    void *d_nbr;
    unsigned long long *d_prof; // diagnostic build only (NM_PROF)
    unsigned long long *d_tline; // experiment build only
    // output snapshots (nm_snapshot / nm_snapshot_fetch): two slots, each a device copy and a pinned host copy of everything a recorded
    // cycle writes (x, box, therm, steps, count, ratio, slot2buf); the D2H runs on a side stream so that the main stream never waits
    hipStream_t side = nullptr;
    struct Snap { double *d = nullptr, *h = nullptr; hipEvent_t taken = nullptr, landed = nullptr; bool pending = false; } snap[2];
    size_t snap_doubles = 0;
    int snap_head = 0, snap_count = 0; // oldest pending slot, number pending
    double *h_stage = nullptr;   // pinned host staging area (nm_set_state / nm_get_state)
    size_t stage_cap = 0;
    size_t trace_cap;
    int trace_on, trace_mod;
    int xtape_n;
    std::vector<double> h_et, h_pf, h_tq;
    std::vector<float> h_P, h_T; // the context's own copy of the grids (cfg.P / cfg.T point here after nm_create)
    std::vector<EvPair> ev;
    int ev_next;
    int launches;
    double total_ms;
    std::string err;
};

namespace {

int fail(nm_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail((c), NM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

template <class T>
hipError_t dalloc(T **p, size_t n) { return hipMalloc((void **)p, (n ? n : 1) * sizeof(T)); }

// test-only environment hooks (NM_ASSUME_CUS, NM_INJECT_OVERFLOW, NM_INJECT_CENSUS) are honoured only under NM_TESTING=1, so that a
// stray variable cannot change Q or stop a production run
bool testing()
{
    const char *e = std::getenv("NM_TESTING");
    return e && std::atoi(e) != 0;
}

void fill_params(const nm_ctx *c, KParams &p)
{
    std::memset(&p, 0, sizeof p);
    p.N = c->N; p.nslots = c->nslots; p.slot0 = c->slot0;
    p.nstps = c->cfg.nstps; p.bulk = c->cfg.bulk; p.iter_revert = c->cfg.iter_revert;
    p.seed = c->cfg.seed; p.step = c->step;
    p.ppos = c->cfg.ppos; p.pvol = c->cfg.pvol; p.lat = c->lat; p.mass = c->mass;
    p.kB = c->kB; p.mvv2e = c->mvv2e; p.ftm2v = c->ftm2v; p.nktv2p = c->nktv2p;
    p.rc = c->rc; p.skin = c->skin;
    p.sc_eps = 0.033147; p.sc_a2 = 4.05 * 4.05; p.sc_c = 16.399; // Sutton & Chen, Phil. Mag. Lett. 61 (1990) 139: Al
    p.x = c->d_x; p.v = c->d_v; p.box = c->d_box; p.steps = c->d_steps; p.therm = c->d_therm;
    p.count = c->d_count; p.ratio = c->d_ratio; p.slot2buf = c->d_slot2buf;
    p.et = c->d_et; p.pf = c->d_pf; p.tq = c->d_tq;
    p.status = c->d_status; p.stats = c->d_stats;
    p.tape = c->d_tape; p.tape_off = c->d_tape_off;
    p.nbr_g = c->d_nbr; p.aux_g = c->d_aux;
    p.prof = c->d_prof;
    p.cus = c->cus; p.xbuf = c->d_xbuf; p.launch_id = c->launch_id;
    p.census = c->cus > 1 ? c->d_census : nullptr; p.over = (c->cus > 1 && c->over) ? 1 : 0;
    p.census_base = c->census_base; p.census_cbase = c->census_cbase;
    p.plain_granules = 1;
    if (const char *e = std::getenv("NM_PLAIN_GRANULES")) p.plain_granules = std::atoi(e);
    p.dbg = 0;
    p.tline = c->d_tline;
    if (const char *e = std::getenv("NM_DBG")) p.dbg = std::atoi(e);
    p.inj_rebuild = -1; p.inj_q = 0;
    const char *e = testing() ? std::getenv("NM_INJECT_OVERFLOW") : nullptr; // tests of the error path only
    if (e) {
        int n = -1, q = 0;
        if (std::sscanf(e, "%d,%d", &n, &q) >= 1) { p.inj_rebuild = n; p.inj_q = q; }
    }
    p.status_acc = c->d_status_acc; p.halt = c->d_halt; p.rerun_mask = nullptr; p.inj_census = 0;
    p.order = (c->use_order && (c->cus == 1 || c->over)) ? c->d_order : nullptr; p.last_ticks = c->d_last_ticks;
}

// workgroups of a launch: 8 Q ceil(nslots / 8), the block kernel's cluster mapping (nm_kernels.h); nslots x Q when 8 divides nslots or Q = 1
unsigned int nm_grid(int nslots, int q) { return q == 1 ? (unsigned int)nslots : (unsigned int)(8 * q * ((nslots + 7) / 8)); }
size_t census_words(int nslots) { return 1 + (size_t)8 * ((nslots + 7) / 8); } // the grid's counter + one per cluster

template <class C>
hipError_t launch_block(const nm_ctx *c, const KParams &p)
{
    if (p.order) { // longest first, by what each slot's previous block took
        hipLaunchKernelGGL(nm_order_kernel, dim3((c->nslots + 255) / 256), dim3(256), 0, c->stream, c->nslots, c->d_last_ticks, c->d_order);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(nm_block_kernel<C>, dim3(nm_grid(c->nslots, c->cus)), dim3(C::BLOCK), C::LDS_BYTES, c->stream, p);
    return hipGetLastError();
}

// The census counters only grow (KParams::census_base): zero them and the bases.  At creation, around the residency probes, after a
// census that was given up (its abort bit would fail every later launch) and long before they could wrap.
hipError_t reset_census(nm_ctx *c)
{
    c->census_base = c->census_cbase = 0;
    return hipMemsetAsync(c->d_census, 0, sizeof(unsigned int) * census_words(c->nslots), c->stream);
}

template <class C>
hipError_t launch_probe(const nm_ctx *c, KParams p)
{
    hipError_t e = hipFuncSetAttribute((const void *)nm_probe_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(c->d_census, 0, sizeof(unsigned int) * census_words(c->nslots), c->stream);
    if (e != hipSuccess) return e;
    p.census_base = p.census_cbase = 0; // (the probe starts from zeroed counters; pick_q zeroes them again behind it)
    hipLaunchKernelGGL(nm_probe_kernel<C>, dim3(nm_grid(c->nslots, c->cus)), dim3(C::BLOCK), C::LDS_BYTES, c->stream, p);
    return hipGetLastError();
}

// workgroups of the block kernel one CU admits (LDS, registers), for the configuration launch_kind would pick at q per replica
template <class C>
int blocks_per_cu()
{
    int n = 0;
    if (hipFuncSetAttribute((const void *)nm_block_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, nm_block_kernel<C>, C::BLOCK, C::LDS_BYTES) != hipSuccess) return 0;
    return n;
}

template <class C>
hipError_t launch_cycles(const nm_ctx *c, const KParams &p)
{
    hipError_t e = hipFuncSetAttribute((const void *)nm_cycles_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nm_cycles_kernel<C>, dim3(nm_grid(c->nslots, c->cus)), dim3(C::BLOCK), C::LDS_BYTES, c->stream, p);
    return hipGetLastError();
}

// nm_cycles_kernel is instantiated for the 4^3 clusters (LJ and Al) and the 6^3 cluster of eight.  nm_run_cycles uses it where it measured faster
// than the loop of single launches: the 4^3 clusters of 2 and 4 workgroups (64-128 replicas: +2.4 to +2.9 % LJ, +0.5 % Al).  Clusters of 8 run 32 replicas
// or fewer, whose rows have little spread to hide, and there the block compiled inside the loop over cycles (~1 % slower) costs more than the rows gain
// (-0.8 % at 4^3, -2.9 % at 6^3): the loop of single launches, unless NM_FUSED_CYCLES=all.  NM_FUSED_CYCLES=0: never.  DESIGN.md §7.4 (6).
bool cycles_kind_built(const nm_ctx *c) { return (c->kind == 0 && c->cus >= 2) || (c->kind == 1 && c->cus == 8); }
bool cycles_kind_ok(const nm_ctx *c)
{
    if (!cycles_kind_built(c)) return false;
    const char *e = std::getenv("NM_FUSED_CYCLES");
    if (e && !std::strcmp(e, "all")) return true;
    if (e && !std::strcmp(e, "0")) return false;
    return c->kind == 0 && (c->cus == 2 || c->cus == 4);
}

hipError_t launch_cycles_kind(const nm_ctx *c, const KParams &p)
{
    if (c->kind == 0) {
        if (c->pot == 1) return c->cus == 4 ? launch_cycles<CfgSmallSCQ4>(c, p) : launch_cycles<CfgSmallSCQ2>(c, p);
        return c->cus == 8 ? launch_cycles<CfgSmallQ8>(c, p) : c->cus == 4 ? launch_cycles<CfgSmallQ4>(c, p) : launch_cycles<CfgSmallQ2>(c, p);
    }
    return launch_cycles<CfgMidQ8>(c, p);
}

hipError_t launch_kind(const nm_ctx *c, const KParams &p)
{
    switch (c->kind) {
    case 0:
        if (c->pot == 1) return c->cus == 4 ? launch_block<CfgSmallSCQ4>(c, p) : c->cus == 2 ? launch_block<CfgSmallSCQ2>(c, p) : launch_block<CfgSmallSC>(c, p);
        return c->cus == 8 ? launch_block<CfgSmallQ8>(c, p) : c->cus == 4 ? launch_block<CfgSmallQ4>(c, p) : c->cus == 2 ? launch_block<CfgSmallQ2>(c, p) : launch_block<CfgSmall>(c, p);
    case 1: return c->cus == 8 ? launch_block<CfgMidQ8>(c, p) : c->cus == 4 ? launch_block<CfgMidQ4>(c, p) : c->cus == 2 ? launch_block<CfgMid>(c, p) : launch_block<CfgMidH>(c, p);
    default: return c->cus == 1 ? launch_block<CfgLargeH>(c, p) : launch_block<CfgLarge>(c, p);
    }
}

int blocks_per_cu_kind(int kind, int pot, int q)
{
    switch (kind) {
    case 0:
        if (pot == 1) return q == 4 ? blocks_per_cu<CfgSmallSCQ4>() : q == 2 ? blocks_per_cu<CfgSmallSCQ2>() : blocks_per_cu<CfgSmallSC>();
        return q == 8 ? blocks_per_cu<CfgSmallQ8>() : q == 4 ? blocks_per_cu<CfgSmallQ4>() : q == 2 ? blocks_per_cu<CfgSmallQ2>() : blocks_per_cu<CfgSmall>();
    case 1: return q == 8 ? blocks_per_cu<CfgMidQ8>() : q == 4 ? blocks_per_cu<CfgMidQ4>() : q == 2 ? blocks_per_cu<CfgMid>() : blocks_per_cu<CfgMidH>();
    default: return q == 1 ? blocks_per_cu<CfgLargeH>() : blocks_per_cu<CfgLarge>();
    }
}

hipError_t probe_kind(const nm_ctx *c, const KParams &p)
{
    switch (c->kind) {
    case 0:
        if (c->pot == 1) return c->cus == 4 ? launch_probe<CfgSmallSCQ4>(c, p) : c->cus == 2 ? launch_probe<CfgSmallSCQ2>(c, p) : launch_probe<CfgSmallSC>(c, p);
        return c->cus == 8 ? launch_probe<CfgSmallQ8>(c, p) : c->cus == 4 ? launch_probe<CfgSmallQ4>(c, p) : c->cus == 2 ? launch_probe<CfgSmallQ2>(c, p) : launch_probe<CfgSmall>(c, p);
    case 1: return c->cus == 8 ? launch_probe<CfgMidQ8>(c, p) : c->cus == 4 ? launch_probe<CfgMidQ4>(c, p) : c->cus == 2 ? launch_probe<CfgMid>(c, p) : launch_probe<CfgMidH>(c, p);
    default: return c->cus == 1 ? launch_probe<CfgLargeH>(c, p) : launch_probe<CfgLarge>(c, p);
    }
}

// drain one event pair into the timing accumulators
void harvest(nm_ctx *c, EvPair &e)
{
    if (!e.used) return;
    hipEventSynchronize(e.b);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { c->total_ms += ms; c->launches += 1; }
    e.used = false;
}

std::string status_text(const nm_ctx *c, int k, int bits)
{
    char buf[640];
    std::snprintf(buf, sizeof buf,
                  "replica slot %d (global %d) left the supported regime:%s%s%s%s%s%s", k, c->slot0 + k,
                  (bits & ST_LIST_OVERFLOW) ? " neighbour list overflow;" : "",
                  (bits & ST_BOX_TOO_SMALL) ? " box edge < 2*rc (minimum image invalid);" : "",
                  (bits & ST_TAPE_EXHAUSTED) ? " rng tape exhausted;" : "",
                  (bits & ST_NONFINITE) ? " non-finite energy;" : "",
                  (bits & ST_SYNC_TIMEOUT) ? " cluster hand-off timed out (workgroups not co-resident?);" : "",
                  (bits & ST_NOT_RESIDENT) ? " the launch's workgroups were not resident together (CUs taken by another process, stream or a CU "
                                             "mask): nothing was changed;" : "");
    return buf;
}

enum : int { OP_BLOCK = 0, OP_ADAPT = 1, OP_EXCHANGE = 2, OP_MD = 3, OP_CYCLES = 4 }; // OP_CYCLES: arg = MOD, trace = number of cycles

// ---- the queued calls (nm_run_block, nm_run_md, nm_adapt, nm_exchange) as stream operations; issued by the API call and, after a
// block had to be given up, again by settle()
int check_status(nm_ctx *c);

// a cluster launch has been queued: where its census leaves the counters
void census_advance(nm_ctx *c)
{
    if (c->cus <= 1) return;
    if (c->over) c->census_cbase += (unsigned int)c->cus; else c->census_base += nm_grid(c->nslots, c->cus);
}
void fill_census(const nm_ctx *c, KParams &p) { p.census_base = c->census_base; p.census_cbase = c->census_cbase; }

int issue_block(nm_ctx *c, int kind, int arg, uint32_t step, int trace, const int *d_mask, bool timed)
{
    if (c->journal.size() > 4096 && !d_mask) { // a caller that never looks: bound the journal
        const int rc = check_status(c);
        if (rc) return rc;
    }
    ++c->launch_id;
    KParams p;
    const uint32_t keep = c->step;
    c->step = step;
    fill_params(c, p);
    c->step = keep;
    p.rerun_mask = d_mask;
    if (kind == OP_MD) { p.mod = 1; p.md_mode = 1; p.nstps = arg; p.tape = nullptr; }
    else {
        p.mod = arg;
        if (trace) {
            const size_t need = (size_t)c->nslots * arg * NM_TRACE_COLS;
            if (need > c->trace_cap) {
                if (c->d_trace) HIPCHK(c, hipFree(c->d_trace));
                c->d_trace = nullptr;
                HIPCHK(c, dalloc(&c->d_trace, need));
                c->trace_cap = need;
            }
            if (!d_mask) HIPCHK(c, hipMemsetAsync(c->d_trace, 0, need * sizeof(double), c->stream));
            p.trace = c->d_trace;
            c->trace_mod = arg;
        }
    }
    if (testing())
        if (const char *e = std::getenv("NM_INJECT_CENSUS")) // "n": the context's n-th cluster launch fails its residency census
            if (c->cus > 1 && std::atoi(e) == c->cluster_launches) p.inj_census = 1;
    if (c->cus > 1) ++c->cluster_launches;
    // status[] holds the bits of the LAST launch (status_acc[] everything since the host last looked); a launch that finds the halt
    // word armed does nothing, and the memset in front of it must not wipe the failed block's bits: they are in status_acc[]
    // (status[]: cleared by each slot's writer inside the kernel; census counters: monotonic — no memsets in front of a launch)
    if (c->cus > 1 && (c->census_base > 0x3F000000u || c->census_cbase > 0x3F000000u)) { HIPCHK(c, reset_census(c)); fill_census(c, p); }
    if (timed) {
        EvPair &e = c->ev[c->ev_next];
        harvest(c, e);
        HIPCHK(c, hipEventRecord(e.a, c->stream));
        HIPCHK(c, launch_kind(c, p));
        HIPCHK(c, hipEventRecord(e.b, c->stream));
        e.used = true; e.launch_id = c->launch_id;
        c->ev_next = (c->ev_next + 1) % (int)c->ev.size();
    } else HIPCHK(c, launch_kind(c, p));
    census_advance(c);
    c->journal.push_back({ kind, arg, trace, step, c->launch_id });
    return NM_OK;
}

int issue_adapt(nm_ctx *c)
{
    hipLaunchKernelGGL(nm_adapt_kernel, dim3((c->nslots + 63) / 64), dim3(64), 0, c->stream, c->nslots, c->d_slot2buf,
                       c->d_steps, c->d_count, c->d_ratio, c->d_halt);
    HIPCHK(c, hipGetLastError());
    c->journal.push_back({ OP_ADAPT, 0, 0, 0u, 0u });
    return NM_OK;
}

int issue_exchange(nm_ctx *c, uint32_t step)
{
    hipLaunchKernelGGL(nm_exchange_kernel, dim3(1), dim3(64), 0, c->stream, c->cfg.nrows, c->cfg.nt,
                       c->cfg.row0, c->cfg.seed, step, c->d_slot2buf, c->d_therm, c->d_et, c->d_pf,
                       c->xtape_n ? c->d_xtape : nullptr, c->d_xcrit, c->d_nswaps, c->d_halt);
    HIPCHK(c, hipGetLastError());
    c->journal.push_back({ OP_EXCHANGE, 0, 0, step, 0u });
    return NM_OK;
}

// nm_run_cycles as stream operations: ONE launch of nm_cycles_kernel where the configuration has one and the whole grid is resident and checked by the
// census (clusters), else the same cycles as single blocks, adapts and exchanges
int issue_cycles(nm_ctx *c, int ncycles, int mod, uint32_t step, bool timed)
{
    const bool fused = c->whole_rows && c->cus > 1 && !c->over && cycles_kind_ok(c) && !c->d_tape && !c->xtape_n && !c->trace_on &&
                       !(c->use_order && (c->cus == 1 || c->over));
    if (!fused) {
        for (int k = 0; k < ncycles; ++k) {
            int rc = issue_block(c, OP_BLOCK, mod, step + (uint32_t)k, 0, nullptr, timed);
            if (!rc) rc = issue_adapt(c);
            if (!rc) rc = issue_exchange(c, step + (uint32_t)k);
            if (rc) return rc;
        }
        return NM_OK;
    }
    // a launch holds at most 64 cycles (what the rows gain by not waiting is there after ~10; a kernel that runs for minutes serves nobody)
    constexpr int MAX_PER_LAUNCH = 64;
    if (ncycles > MAX_PER_LAUNCH) {
        for (int k = 0; k < ncycles; k += MAX_PER_LAUNCH) {
            const int rc = issue_cycles(c, std::min(MAX_PER_LAUNCH, ncycles - k), mod, step + (uint32_t)k, timed);
            if (rc) return rc;
        }
        return NM_OK;
    }
    const int nrows = c->cfg.nrows;
    if (!c->d_rowsync) HIPCHK(c, dalloc(&c->d_rowsync, (size_t)2 * nrows + 2));
    HIPCHK(c, hipMemsetAsync(c->d_rowsync, 0, sizeof(unsigned int) * ((size_t)2 * nrows + 2), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_nswaps, 0, sizeof(int), c->stream));
    const uint32_t first_id = c->launch_id + 1;
    c->launch_id += (uint32_t)ncycles;   // one id per cycle: the hand-over granules of successive blocks must differ
    KParams p;
    const uint32_t keep = c->step;
    c->step = step;
    fill_params(c, p);
    c->step = keep;
    p.launch_id = first_id;
    p.mod = mod; p.ncycles = ncycles; p.nt = c->cfg.nt; p.row0 = c->cfg.row0;
    p.rowbar = c->d_rowsync; p.rowgo = c->d_rowsync + nrows; p.cyc_abort = (int *)(c->d_rowsync + 2 * nrows); p.nswaps = c->d_nswaps;
    p.order = nullptr; p.tape = nullptr; p.trace = nullptr;
    if (testing())
        if (const char *e = std::getenv("NM_INJECT_CENSUS"))
            if (std::atoi(e) == c->cluster_launches) p.inj_census = 1;
    ++c->cluster_launches;
    if (c->census_base > 0x3F000000u || c->census_cbase > 0x3F000000u) { HIPCHK(c, reset_census(c)); fill_census(c, p); }
    if (timed) {
        EvPair &e = c->ev[c->ev_next];
        harvest(c, e);
        HIPCHK(c, hipEventRecord(e.a, c->stream));
        HIPCHK(c, launch_cycles_kind(c, p));
        HIPCHK(c, hipEventRecord(e.b, c->stream));
        e.used = true; e.launch_id = first_id;
        c->ev_next = (c->ev_next + 1) % (int)c->ev.size();
    } else HIPCHK(c, launch_cycles_kind(c, p));
    census_advance(c);
    c->journal.push_back({ OP_CYCLES, mod, ncycles, step, first_id });
    return NM_OK;
}

// Workgroups per replica.  A cluster spins on its peers, so every workgroup of the grid must be resident at once.  The candidates
// are the Q <= qmax for which the occupancy query admits the whole grid (workgroups per CU x CUs); the first one whose grid
// actually gathers in a residency census (nm_probe_kernel: the block kernel's launch shape, census only, nm_kernels.h) is taken,
// so a masked or busy CU lowers Q instead of stalling every block.  Leaves c->cus set; 1 when no cluster gathers.
int pick_q(nm_ctx *c, int qmax, std::string &note)
{
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->cfg.device));
    int cu = prop.multiProcessorCount;
    if (testing())
        if (const char *e = std::getenv("NM_ASSUME_CUS")) { const int v = std::atoi(e); if (v > 0) cu = v; } // tests of the fallback
    const int maxq = c->kind == 0 ? (c->pot == 0 ? 8 : 4) : c->kind == 1 ? 8 : 4; // own-atom ranges the instantiated thread mappings cover
    c->cus = 1; c->over = false;
    // The large cells (N > 864) at 4 workgroups per replica when that makes a grid of (nearly) TWICE the chip: the clusters run in
    // two rounds, longest block first, each with its own census.  Their blocks differ by more than 2x across an equilibrated PxT grid
    // (31-73 ms at 8^3, two workgroups each), so a resident grid of two workgroups per replica lasts as long as its slowest member while
    // a third of the CUs have nothing left to do; four workgroups make a block ~1.6x shorter, and two rounds of them, dealt out longest first,
    // end within a few ms of one another.  Measured on C5's per-GPU share (128 x 2048 atoms): equilibrated chains 221 k sweeps/s against
    // 211 k for the resident grid (+4.5 %: 74 ms per launch where perfect packing of the same blocks would give 67), but 276 k against
    // 324 k while all replicas are still alike (nothing to even out, and 4 workgroups per replica cost 1.27x the CU time of 2).  So it
    // is opt-in: NM_OVERSUBSCRIBE=1 (and a hundred cycles on, 203 k against 219 k: the gain belongs to the cycles in which the replicas differ most).  (It also leans on workgroups being dispatched in index order within an XCD, which HIP does not
    // promise; a cluster whose members do not gather fails its census and the block is re-issued on the resident grid, settle().)
    bool try_over = false;
    if (const char *e = std::getenv("NM_OVERSUBSCRIBE")) try_over = c->kind == 2 && qmax >= 4 && std::atoi(e) != 0;
    for (int pass = try_over ? 0 : 1; pass < 2; ++pass)
    for (int qq : { 8, 4, 2 }) {
        if (qq > maxq || qq > qmax) continue;
        const int per_cu = blocks_per_cu_kind(c->kind, c->pot, qq);
        const long grid = (long)nm_grid(c->nslots, qq), room = (long)per_cu * cu;
        const bool over = pass == 0;
        if (over) { if (qq != 4 || grid > 2 * room || 4 * grid < 7 * room) continue; } // 1.75 .. 2 chips' worth of workgroups
        else if (grid > room) continue;
        c->cus = qq; c->over = over; // probe: does the grid of this Q gather?
        KParams p;
        fill_params(c, p);
        p.status_acc = nullptr; p.halt = nullptr;
        HIPCHK(c, hipMemsetAsync(c->d_status, 0, sizeof(int) * c->nslots, c->stream));
        HIPCHK(c, probe_kind(c, p));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        std::vector<int> st((size_t)c->nslots);
        HIPCHK(c, hipMemcpy(st.data(), c->d_status, sizeof(int) * c->nslots, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int v : st) ok = ok && !(v & ST_NOT_RESIDENT);
        HIPCHK(c, hipMemset(c->d_status, 0, sizeof(int) * c->nslots));
        HIPCHK(c, reset_census(c)); // (the probe's arrivals, or its abort bit, must not meet the first block)
        if (ok) return NM_OK;
        char buf[200];
        std::snprintf(buf, sizeof buf, "%d workgroups per replica (%d in all) did not gather on this device; falling back. ", qq, (int)nm_grid(c->nslots, qq));
        note += buf;
        c->cus = 1; c->over = false;
    }
    return NM_OK;
}

// the buffers whose size depends on the workgroups per replica: the per-workgroup spill and the hand-over granules
int alloc_cluster_buffers(nm_ctx *c)
{
    const size_t ns = c->nslots;
    size_t aux_doubles, lds_bytes = c->lds_bytes;
    if (c->kind == 0) aux_doubles = CfgSmall::AUX_DOUBLES;
    else if (c->kind == 1) { aux_doubles = c->cus == 8 ? CfgMidQ8::AUX_DOUBLES : CfgMid::AUX_DOUBLES; lds_bytes = c->cus == 8 ? CfgMidQ8::LDS_BYTES : CfgMid::LDS_BYTES; }
    else aux_doubles = CfgLarge::AUX_DOUBLES;
    // the new buffers first, swapped in only when both exist: a failure leaves the context with the buffers (and the workgroups per
    // replica) it can still run with
    double *aux = nullptr, *xbuf = nullptr;
    const size_t xbd = c->kind == 0 ? CfgSmall::XBUF_DOUBLES : c->kind == 1 ? CfgMid::XBUF_DOUBLES : CfgLarge::XBUF_DOUBLES;
    hipError_t e = hipSuccess;
    if (aux_doubles) e = dalloc(&aux, ns * c->cus * aux_doubles);
    if (e == hipSuccess && c->cus > 1) {
        e = dalloc(&xbuf, ns * 2 * xbd);
        if (e == hipSuccess) e = hipMemset(xbuf, 0, ns * 2 * xbd * sizeof(double));
    }
    if (e != hipSuccess) {
        if (aux) hipFree(aux);
        if (xbuf) hipFree(xbuf);
        if (!c->d_xbuf) { c->cus = 1; c->over = false; } // nothing to hand over with: one workgroup per replica (its spill area, if any, is a subset of what is there)
        return fail(c, NM_ERR_HIP, std::string("alloc_cluster_buffers: ") + hipGetErrorString(e));
    }
    if (c->d_aux) hipFree(c->d_aux);
    if (c->d_xbuf) hipFree(c->d_xbuf);
    c->d_aux = aux; c->d_xbuf = xbuf; c->aux_doubles = aux_doubles; c->lds_bytes = lds_bytes;
    return NM_OK;
}

// The outcome of everything queued so far; the stream has been synchronised by the caller.  A block that stopped because its
// cluster grid was not resident (NM_ST_NOT_RESIDENT: nothing was touched) or a hand-over timed out (NM_ST_SYNC_TIMEOUT: the
// replicas concerned are as they were at the block's start) armed the halt word, so nothing queued behind it has run either
// (the reference's analogue is Dask retrying a failed gen_sample task, remcmc:921-922): the context drops to the next lower
// number of workgroups per replica that gathers and issues the same calls again — the failed block for the replicas that did
// not complete it, everything behind it as it was.  Any other status is the caller's to deal with: NM_ERR_STATE once, with
// the reason; the replicas' state is that of the failed block's start and the context stays usable.
int settle(nm_ctx *c)
{
    for (int round = 0; round < 4; ++round) {
        int halt = 0;
        HIPCHK(c, hipMemcpy(&halt, c->d_halt, sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> st((size_t)c->nslots);
        HIPCHK(c, hipMemcpy(st.data(), c->d_status_acc, sizeof(int) * c->nslots, hipMemcpyDeviceToHost));
        int any = 0;
        for (int v : st) any |= v;
        if (!halt && !any) { c->journal.clear(); c->err.clear(); return NM_OK; } // (nm_last_error is empty after a call that returned NM_OK)
        const int healable = ST_NOT_RESIDENT | ST_SYNC_TIMEOUT;
        size_t at = c->journal.size();
        bool mid_cycles = false; // the halt lies inside a launch of several cycles, behind its first: rows are at different cycles, nothing to re-issue
        for (size_t k = 0; k < c->journal.size(); ++k) {
            const nm_ctx::Op &o = c->journal[k];
            if ((o.kind == OP_BLOCK || o.kind == OP_MD) && (int)o.launch_id == halt) { at = k; break; }
            if (o.kind == OP_CYCLES && (uint32_t)halt >= o.launch_id && (uint32_t)halt < o.launch_id + (uint32_t)o.trace) {
                at = k; mid_cycles = (uint32_t)halt != o.launch_id || (any & ST_SYNC_TIMEOUT); break;
            }
        }
        if (halt && !(any & ~healable) && c->cus > 1 && at < c->journal.size() && !mid_cycles) {
            // ---- re-issue at a lower Q
            const int q_old = c->cus;
            std::string why;
            int rc = pick_q(c, q_old / 2, why);
            if (rc) return rc;
            HIPCHK(c, reset_census(c)); // (the failed launch's census may have left its abort bit)
            if ((rc = alloc_cluster_buffers(c))) return rc;
            std::vector<int> mask((size_t)c->nslots);
            for (int k = 0; k < c->nslots; ++k) mask[k] = st[k] ? 1 : 0;
            HIPCHK(c, hipMemcpy(c->d_rerun, mask.data(), sizeof(int) * c->nslots, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemset(c->d_status_acc, 0, sizeof(int) * c->nslots));
            HIPCHK(c, hipMemset(c->d_halt, 0, sizeof(int)));
            char buf[256];
            std::snprintf(buf, sizeof buf, "block of step %u stopped at %d workgroups per replica (%s); re-issued at %d. %s",
                          c->journal[at].step, q_old, (any & ST_NOT_RESIDENT) ? "grid not resident" : "hand-over timed out", c->cus, why.c_str());
            c->note += buf;
            ++c->heals;
            // timing: the launch that halted did nothing for the replicas that are re-issued, and the blocks queued behind it found the
            // halt word armed and left at once — their event pairs would count launches that did no work.  Drop them (first wait for them:
            // an event pair is recycled) and time the re-issued blocks instead, so that nm_timing_get keeps meaning "the launches that did
            // the work".  nm_heal_count tells a caller that a region contained a re-issue.
            for (auto &e : c->ev)
                if (e.used && e.launch_id >= c->journal[at].launch_id) { hipEventSynchronize(e.b); e.used = false; }
            std::vector<nm_ctx::Op> todo(c->journal.begin() + at, c->journal.end());
            c->journal.clear();
            for (size_t k = 0; k < todo.size(); ++k) {
                const nm_ctx::Op &o = todo[k];
                if (o.kind == OP_BLOCK || o.kind == OP_MD) rc = issue_block(c, o.kind, o.arg, o.step, o.trace, k == 0 ? c->d_rerun : nullptr, o.kind == OP_BLOCK);
                else if (o.kind == OP_CYCLES) rc = issue_cycles(c, o.trace, o.arg, o.step, true); // (its census failed: nothing had run)
                else if (o.kind == OP_ADAPT) rc = issue_adapt(c);
                else rc = issue_exchange(c, o.step);
                if (rc) return rc;
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            continue; // look at the outcome of the re-issue
        }
        // ---- not curable here: report once, leave the context usable
        c->journal.clear();
        // launches queued behind the one that stopped found the halt word armed and left BEFORE their residency census, while the host had already
        // advanced the census base for them: start the counters afresh, or the next launch's census comes up short and is taken for a grid that
        // is not resident (re-issued at fewer workgroups per replica for no reason)
        if (halt) HIPCHK(c, reset_census(c));
        HIPCHK(c, hipMemcpy(c->d_status, st.data(), sizeof(int) * c->nslots, hipMemcpyHostToDevice)); // nm_get_status: what stopped which slot
        HIPCHK(c, hipMemset(c->d_status_acc, 0, sizeof(int) * c->nslots));
        HIPCHK(c, hipMemset(c->d_halt, 0, sizeof(int)));
        for (int k = 0; k < c->nslots; ++k)
            if (st[k]) return fail(c, NM_ERR_STATE, status_text(c, k, st[k]));
        return fail(c, NM_ERR_STATE, "a block stopped on an error that left no status bits");
    }
    return fail(c, NM_ERR_STATE, "a block could not be completed at any number of workgroups per replica");
}

int check_status(nm_ctx *c)
{
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return settle(c);
}

// everything a context owns; safe on a half-built one (nm_create's failure paths: `new nm_ctx()` zero-initialises the pointers)
void free_ctx(nm_ctx *c)
{
    if (c->stream) hipStreamSynchronize(c->stream);
    for (auto &e : c->ev) { if (e.a) hipEventDestroy(e.a); if (e.b) hipEventDestroy(e.b); }
    void *ptrs[] = { c->d_x, c->d_v, c->d_box, c->d_steps, c->d_therm, c->d_count, c->d_ratio, c->d_et, c->d_pf, c->d_tq,
                     c->d_stats, c->d_slot2buf, c->d_status, c->d_nswaps, c->d_evalU, c->d_evalW, c->d_evalF, c->d_xcrit,
                     c->d_xtape, c->d_tape, c->d_tape_off, c->d_trace, c->d_nbr, c->d_aux, c->d_prof, c->d_tline, c->d_xbuf, c->d_census,
                     c->d_status_acc, c->d_halt, c->d_rerun, c->d_order, c->d_last_ticks, c->d_rowsync };
    for (void *q : ptrs) if (q) hipFree(q);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->side) hipStreamSynchronize(c->side);
    for (auto &sn : c->snap) {
        if (sn.d) hipFree(sn.d);
        if (sn.h) hipHostFree(sn.h);
        if (sn.taken) hipEventDestroy(sn.taken);
        if (sn.landed) hipEventDestroy(sn.landed);
    }
    if (c->side) hipStreamDestroy(c->side);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

} // namespace

extern "C" {

const char *nm_last_error(const nm_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char *nm_create_note(const nm_ctx *ctx) { return ctx ? ctx->note.c_str() : ""; }
int nm_nslots(const nm_ctx *ctx) { return ctx ? ctx->nslots : NM_ERR_ARG; }
int nm_natoms(const nm_ctx *ctx) { return ctx ? ctx->N : NM_ERR_ARG; }
int nm_cus_per_replica(const nm_ctx *ctx) { return ctx ? ctx->cus : NM_ERR_ARG; }
int nm_heal_count(const nm_ctx *ctx) { return ctx ? ctx->heals : NM_ERR_ARG; }

int nm_create(const nm_config *cfg, nm_ctx **out)
{
    if (!cfg || !out) return fail(nullptr, NM_ERR_ARG, "nm_create: null argument");
    *out = nullptr;
    if (cfg->size != (int32_t)sizeof(nm_config)) return fail(nullptr, NM_ERR_ARG, "nm_create: nm_config size mismatch (ABI)");
    if (cfg->natoms < 2 || cfg->natoms > 2048) return fail(nullptr, NM_ERR_ARG, "nm_create: natoms must be in [2, 2048]");
    const bool by_slots = cfg->nslots > 0;
    if (cfg->np < 1 || cfg->nt < 1) return fail(nullptr, NM_ERR_ARG, "nm_create: bad grid");
    if (by_slots ? (cfg->slot0 < 0 || cfg->slot0 + cfg->nslots > cfg->np * cfg->nt)
                 : (cfg->nrows < 1 || cfg->row0 < 0 || cfg->row0 + cfg->nrows > cfg->np))
        return fail(nullptr, NM_ERR_ARG, "nm_create: bad row / slot range");
    if (!cfg->P || !cfg->T) return fail(nullptr, NM_ERR_ARG, "nm_create: P and T grids are required");
    if (cfg->nstps < 1 || cfg->ppos < 0 || cfg->pvol < 0 || cfg->ppos + cfg->pvol > 1.0)
        return fail(nullptr, NM_ERR_ARG, "nm_create: bad move parameters");
    if (cfg->element != NM_EL_LJ && cfg->element != NM_EL_AL)
        return fail(nullptr, NM_ERR_UNSUPPORTED, "nm_create: elements LJ and Al have device force kernels in this build");
    if (cfg->element == NM_EL_AL && cfg->natoms > 256)
        return fail(nullptr, NM_ERR_UNSUPPORTED, "nm_create: element Al (EAM) is built for up to 256 atoms (4^3 cells)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, NM_ERR_HIP, "nm_create: no HIP device available (this engine has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, NM_ERR_ARG, "nm_create: device ordinal out of range");

    nm_ctx *c = new nm_ctx();
    c->cfg = *cfg;
    c->h_P.assign(cfg->P, cfg->P + cfg->np); c->h_T.assign(cfg->T, cfg->T + cfg->nt);
    c->cfg.P = c->h_P.data(); c->cfg.T = c->h_T.data();
    c->N = cfg->natoms;
    c->nslots = by_slots ? cfg->nslots : cfg->nrows * cfg->nt;
    c->slot0 = by_slots ? cfg->slot0 : cfg->row0 * cfg->nt;
    c->whole_rows = (c->slot0 % cfg->nt == 0) && (c->nslots % cfg->nt == 0);
    c->cfg.row0 = c->slot0 / cfg->nt;                       // meaningful only for whole rows
    c->cfg.nrows = c->whole_rows ? c->nslots / cfg->nt : 0;
    c->step = 0;
    c->trace_on = 0; c->trace_mod = 0; c->trace_cap = 0; c->xtape_n = 0;
    c->ev_next = 0; c->launches = 0; c->total_ms = 0.0;
    // material tables, remcmc:873-893
    // Verlet-list skin: not observable in results, only in the rebuild rate.  (Round 1 took 0.3, LAMMPS's lj default, as best at 256 atoms —
    // measured in the first cycles after the lattice start, when HMC steps are still short and rebuilds rare;
    // the O(N^2) rebuild of the larger cells favours fewer rebuilds (0.45: 6^3 -10 %, 8^3 -45 % per move)
    // (8^3, equilibrated chains, where a rebuild costs ~10 evaluations: 0.6 gives 107 ms per launch of C5's share against 121 at 0.45,
    // for +7 % while the chains still reject everything; 0.75 overflows the 160 list slots of the dense crystals)
    c->skin = cfg->natoms <= 512 ? 0.4 : cfg->natoms <= 1024 ? 0.55 : 0.6; // (256 atoms, equilibrated: 7.1 / 6.8 / 6.8 ms per launch at 0.3 / 0.4 / 0.5)
    // (round 4, sustained, same box each: 500 atoms at one workgroup per replica 1.385 / 1.411 / 1.392 / 1.390 M sweeps/s at 0.35 / 0.40 / 0.45 / 0.50;
    //  864 atoms: C3 share (Q = 8) 352 / 359 / 368 / 372 / 370 k at 0.40 / 0.45 / 0.50 / 0.55 / 0.60, 256 replicas at Q = 1 615 / 617 / 632 / 642 k at
    //  0.40 / 0.45 / 0.50 / 0.55: the O(N^2) rebuild of the larger cells wants fewer rebuilds; 0.45 for both until then)
    if (const char *e = std::getenv("NM_SKIN")) { const double v = std::atof(e); if (v > 0.0 && v < 1.0) c->skin = v; }
    c->lat = 1.122; c->mass = 1.0; c->kB = 1.0; c->mvv2e = 1.0; c->ftm2v = 1.0; c->nktv2p = 1.0;
    c->rc = 2.5; c->pot = 0;
    if (cfg->element == NM_EL_AL) { // units metal (LAMMPS update.cpp constants), remcmc:880,886
        c->lat = 4.046; c->mass = 29.982; c->kB = 8.617343e-5; c->mvv2e = 1.0364269e-4; c->ftm2v = 1.0 / 1.0364269e-4;
        c->nktv2p = 1.6021765e6; c->rc = 7.5; c->skin = 0.6; c->pot = 1; // (skin: 0.8 until the rebuild's scan and append got cheaper in round 3;
        // equilibrated C4 on one box: 685 / 690 / 706 k sweeps/s at 0.8 / 0.7 / 0.6 A, 2.64 / 2.73 / 2.95 rebuilds per sweep)
        if (const char *e = std::getenv("NM_SKIN_AL")) { const double v = std::atof(e); if (v > 0.0 && v < 3.0) c->skin = v; }
    }

    // init_constant (remcmc:114-132) in float64 on the float32-rounded grid values (NumPy-1.x promotion)
    c->h_et.resize(c->nslots); c->h_pf.resize(c->nslots); c->h_tq.resize(c->nslots);
    for (int k = 0; k < c->nslots; ++k) {
        const int i = (c->slot0 + k) / cfg->nt, j = (c->slot0 + k) % cfg->nt;
        const double Pi = (double)cfg->P[i], Tj = (double)cfg->T[j];
        if (cfg->element == NM_EL_AL) { // remcmc:124-127
            const double kb = 8.61733e-5;
            c->h_et[k] = kb * Tj;
            c->h_pf[k] = 1e-30 * (1e5 * Pi) / (1.60218e-19 * kb * Tj);
        } else {                        // remcmc:128-131
            const double kb = 1.0;
            c->h_et[k] = kb * Tj;
            c->h_pf[k] = Pi / (kb * Tj);
        }
        c->h_tq[k] = Tj;
    }

    size_t nbr_elems;
    if (c->N <= CfgSmall::NMAX) { c->kind = 0; c->lds_bytes = c->pot == 1 ? CfgSmallSC::LDS_BYTES : CfgSmall::LDS_BYTES; /* Q8 variant: set at launch */ c->aux_doubles = CfgSmall::AUX_DOUBLES; nbr_elems = CfgSmall::NBR_G_ELEMS; }
    else if (c->N <= CfgMid::NMAX) { c->kind = 1; c->lds_bytes = CfgMid::LDS_BYTES; c->aux_doubles = CfgMid::AUX_DOUBLES; nbr_elems = CfgMid::NBR_G_ELEMS; }
    else { c->kind = 2; c->lds_bytes = CfgLarge::LDS_BYTES; c->aux_doubles = CfgLarge::AUX_DOUBLES; nbr_elems = CfgLarge::NBR_G_ELEMS; }

#define CHK(call)                                                                                     \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            fail(nullptr, NM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));             \
            free_ctx(c);                                                                              \
            return NM_ERR_HIP;                                                                        \
        }                                                                                             \
    } while (0)
    CHK(hipSetDevice(cfg->device));
    CHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->cus = 1; c->d_xbuf = nullptr; c->d_aux = nullptr; c->launch_id = 0; c->d_census = nullptr; c->over = false;
    c->d_status_acc = nullptr; c->d_halt = nullptr; c->d_rerun = nullptr; c->d_order = nullptr; c->d_last_ticks = nullptr; c->use_order = false;
    std::string note;
    {
        int want = 8;
        if (const char *e = std::getenv("NM_CUS_PER_REPLICA")) want = std::atoi(e);
        CHK(dalloc(&c->d_census, census_words(c->nslots)));
        CHK(dalloc(&c->d_status, (size_t)c->nslots));
        CHK(hipMemset(c->d_status, 0, sizeof(int) * c->nslots));
        CHK(dalloc(&c->d_status_acc, (size_t)c->nslots));
        CHK(hipMemset(c->d_status_acc, 0, sizeof(int) * c->nslots));
        CHK(dalloc(&c->d_rerun, (size_t)c->nslots));
        CHK(dalloc(&c->d_order, (size_t)c->nslots));
        CHK(dalloc(&c->d_last_ticks, (size_t)c->nslots));
        CHK(hipMemset(c->d_last_ticks, 0, sizeof(unsigned long long) * c->nslots)); // (first launch: all ties, i.e. index order)
        CHK(dalloc(&c->d_halt, 1));
        CHK(hipMemset(c->d_halt, 0, sizeof(int)));
        CHK(dalloc(&c->d_slot2buf, (size_t)c->nslots));
        {
            std::vector<int> ident((size_t)c->nslots);
            for (int k = 0; k < c->nslots; ++k) ident[k] = k;
            CHK(hipMemcpy(c->d_slot2buf, ident.data(), sizeof(int) * c->nslots, hipMemcpyHostToDevice));
        }
        if (pick_q(c, want, note) != NM_OK) { g_create_error = c->err; free_ctx(c); return NM_ERR_HIP; } // (workgroups per replica: see pick_q)
        if (!note.empty()) note = "nm_create: " + note;
        if (testing())
            if (const char *e = std::getenv("NM_TEST_CENSUS_BASE")) { // tests: start the monotonic census counters just below the value at
                // which issue_block zeroes them (a wrap is otherwise 16 million launches away)
                const unsigned int b = (unsigned int)std::strtoul(e, nullptr, 0);
                std::vector<unsigned int> w(census_words(c->nslots), b);
                CHK(hipMemcpy(c->d_census, w.data(), sizeof(unsigned int) * w.size(), hipMemcpyHostToDevice));
                c->census_base = c->census_cbase = b;
            }
        {
            hipDeviceProp_t prop;
            CHK(hipGetDeviceProperties(&prop, cfg->device));
            c->use_order = c->over || c->nslots > prop.multiProcessorCount; // (more workgroups / clusters than the chip holds at once: longest first)
            if (const char *e = std::getenv("NM_LAUNCH_ORDER")) c->use_order = std::atoi(e) != 0;
        }
    }
    const size_t ns = c->nslots, n3 = (size_t)3 * c->N;
    CHK(dalloc(&c->d_x, ns * n3)); CHK(dalloc(&c->d_v, ns * n3));
    CHK(dalloc(&c->d_box, ns)); CHK(dalloc(&c->d_steps, ns * 3)); CHK(dalloc(&c->d_therm, ns * 5));
    CHK(dalloc(&c->d_count, ns * 6)); CHK(dalloc(&c->d_ratio, ns * 3));
    CHK(dalloc(&c->d_et, ns)); CHK(dalloc(&c->d_pf, ns)); CHK(dalloc(&c->d_tq, ns));
    CHK(dalloc(&c->d_stats, ns * NM_STATS_COLS));
    CHK(dalloc(&c->d_nswaps, 1)); // (d_slot2buf, d_status: allocated for the census probe above)
    CHK(dalloc(&c->d_evalU, ns)); CHK(dalloc(&c->d_evalW, ns)); CHK(dalloc(&c->d_evalF, ns * n3));
    const int npairs = c->cfg.nrows * cfg->nt * (cfg->nt - 1) / 2;
    CHK(dalloc(&c->d_xcrit, (size_t)npairs)); CHK(dalloc(&c->d_xtape, (size_t)npairs));
    c->d_prof = nullptr;
    c->d_tline = nullptr;
#ifdef NM_EXPERIMENT
    CHK(dalloc(&c->d_tline, (size_t)8 * 8 * 512 * 8)); CHK(hipMemset(c->d_tline, 0, (size_t)8 * 8 * 512 * 8 * sizeof(unsigned long long)));
#endif
#ifdef NM_PROF
    CHK(dalloc(&c->d_prof, ns * 16)); CHK(hipMemset(c->d_prof, 0, ns * 16 * sizeof(unsigned long long)));
#endif
    c->d_tape = nullptr; c->d_tape_off = nullptr; c->d_trace = nullptr; c->d_nbr = nullptr;
    if (nbr_elems) CHK(hipMalloc(&c->d_nbr, ns * 2 * nbr_elems * sizeof(unsigned short))); // two lists per slot (Cfg::LIST2)
    if (alloc_cluster_buffers(c) != NM_OK) { g_create_error = c->err; free_ctx(c); return NM_ERR_HIP; }
    CHK(hipMemset(c->d_x, 0, ns * n3 * sizeof(double))); CHK(hipMemset(c->d_v, 0, ns * n3 * sizeof(double)));
    CHK(hipMemset(c->d_box, 0, ns * sizeof(double))); CHK(hipMemset(c->d_steps, 0, ns * 3 * sizeof(double)));
    CHK(hipMemset(c->d_therm, 0, ns * 5 * sizeof(double))); CHK(hipMemset(c->d_count, 0, ns * 6 * sizeof(double)));
    CHK(hipMemset(c->d_ratio, 0, ns * 3 * sizeof(float))); CHK(hipMemset(c->d_stats, 0, ns * NM_STATS_COLS * sizeof(double)));
    CHK(hipMemset(c->d_nswaps, 0, sizeof(int)));
    CHK(hipMemcpy(c->d_et, c->h_et.data(), ns * sizeof(double), hipMemcpyHostToDevice));
    CHK(hipMemcpy(c->d_pf, c->h_pf.data(), ns * sizeof(double), hipMemcpyHostToDevice));
    CHK(hipMemcpy(c->d_tq, c->h_tq.data(), ns * sizeof(double), hipMemcpyHostToDevice));
    // dynamic LDS above 64 KiB has to be requested per kernel
    if (c->kind == 0) {
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmall>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmall::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallQ2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallQ2::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallQ4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallQ4::LDS_BYTES));
        static_assert(CfgSmall::LDS_BYTES <= 160 * 1024 && CfgSmallQ2::LDS_BYTES <= 160 * 1024, "two byte lists per workgroup must fit the CU's LDS");
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallQ8::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallSC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallSC::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallSCQ2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallSCQ2::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgSmallSCQ4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgSmallSCQ4::LDS_BYTES));
        static_assert(CfgSmallSCQ2::LDS_BYTES <= 160 * 1024 && CfgSmallSC::LDS_BYTES <= 160 * 1024, "");
    }
    else if (c->kind == 1) {
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgMid>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgMid::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgMidH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgMidH::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgMidQ4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgMidQ4::LDS_BYTES));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgMidQ8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgMidQ8::LDS_BYTES));
        static_assert(CfgMidQ8::LDS_BYTES <= 160 * 1024, "the 6^3 cluster configuration must fit the CU's LDS");
    }
    else {
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgLarge>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->lds_bytes));
        CHK(hipFuncSetAttribute((const void *)nm_block_kernel<CfgLargeH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CfgLargeH::LDS_BYTES));
    }
    c->ev.assign(32, EvPair{ nullptr, nullptr, false, 0u });
    for (auto &e : c->ev) { CHK(hipEventCreate(&e.a)); CHK(hipEventCreate(&e.b)); }
#undef CHK
    c->err.clear();
    c->note = note; // nm_create_note(ctx): what the residency probe had to give up, if anything
    *out = c;
    return NM_OK;
}

int nm_destroy(nm_ctx *c)
{
    if (!c) return NM_ERR_ARG;
    hipSetDevice(c->cfg.device);
    free_ctx(c);
    return NM_OK;
}

int nm_get_const(const nm_ctx *c, double *et, double *pf)
{
    if (!c) return NM_ERR_ARG;
    if (et) std::memcpy(et, c->h_et.data(), sizeof(double) * c->nslots);
    if (pf) std::memcpy(pf, c->h_pf.data(), sizeof(double) * c->nslots);
    return NM_OK;
}

int nm_set_step(nm_ctx *c, uint32_t step)
{
    if (!c) return NM_ERR_ARG;
    c->step = step;
    return NM_OK;
}

static int slot_map(nm_ctx *c, std::vector<int> &m)
{
    m.resize(c->nslots);
    const int rc = check_status(c); // everything queued has run (or has been re-issued) before state is read or replaced
    if (rc) return rc;
    HIPCHK(c, hipMemcpy(m.data(), c->d_slot2buf, sizeof(int) * c->nslots, hipMemcpyDeviceToHost));
    return NM_OK;
}

// pinned staging area of the context (grown on demand): state moves between the caller's arrays and HBM as whole device
// arrays, one asynchronous copy each on the context's stream and one wait, instead of up to four small copies per replica
static int stage_reserve(nm_ctx *c, size_t doubles)
{
    if (doubles <= c->stage_cap) return NM_OK;
    if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
    c->h_stage = nullptr;
    c->stage_cap = 0;
    HIPCHK(c, hipHostMalloc((void **)&c->h_stage, doubles * sizeof(double), hipHostMallocDefault));
    c->stage_cap = doubles;
    return NM_OK;
}

int nm_set_state(nm_ctx *c, int k0, int nk, const double *x, const double *v, const double *box, const double *dxdvdt)
{
    if (!c || k0 < 0 || nk < 0 || k0 + nk > c->nslots) return fail(c, NM_ERR_ARG, "nm_set_state: slot range");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    const size_t n3 = (size_t)3 * c->N, ns = (size_t)c->nslots;
    const bool all = nk == c->nslots;
    if (!all && nk <= 8) { // a few replicas (the split-row exchange re-seats the ones that swapped): copy just those
        for (int q = 0; q < nk; ++q) {
            const size_t bq = (size_t)m[k0 + q];
            if (x) HIPCHK(c, hipMemcpyAsync(c->d_x + bq * n3, x + q * n3, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            if (v) HIPCHK(c, hipMemcpyAsync(c->d_v + bq * n3, v + q * n3, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
            if (box) {
                const double vol = std::pow(box[q], 3.0);
                HIPCHK(c, hipMemcpyAsync(c->d_box + bq, box + q, sizeof(double), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->d_therm + 5 * bq + 4, &vol, sizeof(double), hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream)); // (vol lives on this stack frame)
            }
            if (dxdvdt) HIPCHK(c, hipMemcpyAsync(c->d_steps + 3 * bq, dxdvdt + 3 * q, 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return NM_OK;
    }
    // a partial range is merged into the device arrays: read them back first (set-up path, not timed)
    if ((rc = stage_reserve(c, 2 * ns * n3 + 9 * ns))) return rc;
    double *hx = c->h_stage, *hv = hx + ns * n3, *hb = hv + ns * n3, *hs = hb + ns, *ht = hs + 3 * ns;
    if (!all) {
        if (x) HIPCHK(c, hipMemcpyAsync(hx, c->d_x, ns * n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (v) HIPCHK(c, hipMemcpyAsync(hv, c->d_v, ns * n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (box) HIPCHK(c, hipMemcpyAsync(hb, c->d_box, ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (dxdvdt) HIPCHK(c, hipMemcpyAsync(hs, c->d_steps, 3 * ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    if (box) HIPCHK(c, hipMemcpyAsync(ht, c->d_therm, 5 * ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < nk; ++q) {
        const size_t bq = (size_t)m[k0 + q];
        if (x) std::memcpy(hx + bq * n3, x + q * n3, n3 * sizeof(double));
        if (v) std::memcpy(hv + bq * n3, v + q * n3, n3 * sizeof(double));
        if (box) { hb[bq] = box[q]; ht[5 * bq + 4] = std::pow(box[q], 3.0); }
        if (dxdvdt) std::memcpy(hs + 3 * bq, dxdvdt + 3 * q, 3 * sizeof(double));
    }
    if (x) HIPCHK(c, hipMemcpyAsync(c->d_x, hx, ns * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (v) HIPCHK(c, hipMemcpyAsync(c->d_v, hv, ns * n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (box) {
        HIPCHK(c, hipMemcpyAsync(c->d_box, hb, ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_therm, ht, 5 * ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    if (dxdvdt) HIPCHK(c, hipMemcpyAsync(c->d_steps, hs, 3 * ns * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the staging area is reused by the next call
    return NM_OK;
}

int nm_lattice_state(int element, int sz, int np, int nt, const float *P, uint32_t seed, int gslot, double dx, int interpolate, double *x,
                     double *box)
{
    if ((element != NM_EL_LJ && element != NM_EL_AL) || sz < 1 || np < 1 || nt < 1 || !P || gslot < 0 || gslot >= np * nt || !x || !box)
        return fail(nullptr, NM_ERR_ARG, "nm_lattice_state: bad argument");
    std::vector<double> frac;
    lat::fcc_fractional(sz, frac);
    const double b = lat::relax_box(element, sz, (double)P[gslot / nt]);
    lat::init_state(element, sz, b, seed, gslot, gslot % nt, nt, dx, interpolate, frac, x, box);
    return NM_OK;
}

int nm_init_lattice(nm_ctx *c, double dx, double dv, int interpolate)
{
    if (!c) return NM_ERR_ARG;
    int sz = 1;
    while (4 * sz * sz * sz < c->N) ++sz;
    if (4 * sz * sz * sz != c->N) return fail(c, NM_ERR_ARG, "nm_init_lattice: natoms is not 4*sz^3 (fcc)");
    const int ns = c->nslots, nt = c->cfg.nt;
    const size_t n3 = (size_t)3 * c->N;
    std::vector<double> frac, x((size_t)ns * n3), v((size_t)ns * n3, 0.0), box((size_t)ns), d((size_t)3 * ns);
    lat::fcc_fractional(sz, frac);
    double brow = 0.0;
    int row = -1;
    for (int k = 0; k < ns; ++k) {
        const int g = c->slot0 + k, i = g / nt, j = g % nt;
        if (i != row) { brow = lat::relax_box(c->cfg.element, sz, (double)c->cfg.P[i]); row = i; }
        lat::init_state(c->cfg.element, sz, brow, c->cfg.seed, g, j, nt, dx, interpolate, frac, x.data() + (size_t)k * n3, &box[k]);
        d[3 * k] = dx; d[3 * k + 1] = dv; d[3 * k + 2] = 0.00390625; // TIMESTEP, remcmc:891-893 (lj and metal)
    }
    return nm_set_state(c, 0, ns, x.data(), v.data(), box.data(), d.data());
}

int nm_set_thermo(nm_ctx *c, int k0, int nk, const double *th)
{
    if (!c || !th || k0 < 0 || nk < 0 || k0 + nk > c->nslots) return fail(c, NM_ERR_ARG, "nm_set_thermo: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    for (int q = 0; q < nk; ++q)
        HIPCHK(c, hipMemcpy(c->d_therm + 5 * (size_t)m[k0 + q], th + 5 * q, 5 * sizeof(double), hipMemcpyHostToDevice));
    return NM_OK;
}

int nm_get_state(nm_ctx *c, int k0, int nk, double *x, double *v, double *box, double *dxdvdt)
{
    if (!c || k0 < 0 || nk < 0 || k0 + nk > c->nslots) return fail(c, NM_ERR_ARG, "nm_get_state: slot range");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    const size_t n3 = (size_t)3 * c->N, ns = (size_t)c->nslots;
    if (nk <= 8 && nk < c->nslots) { // a few replicas: copy just those
        for (int q = 0; q < nk; ++q) {
            const size_t bq = (size_t)m[k0 + q];
            if (x) HIPCHK(c, hipMemcpyAsync(x + q * n3, c->d_x + bq * n3, n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (v) HIPCHK(c, hipMemcpyAsync(v + q * n3, c->d_v + bq * n3, n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (box) HIPCHK(c, hipMemcpyAsync(box + q, c->d_box + bq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            if (dxdvdt) HIPCHK(c, hipMemcpyAsync(dxdvdt + 3 * q, c->d_steps + 3 * bq, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return NM_OK;
    }
    if ((rc = stage_reserve(c, 2 * ns * n3 + 9 * ns))) return rc;
    double *hx = c->h_stage, *hv = hx + ns * n3, *hb = hv + ns * n3, *hs = hb + ns;
    if (x) HIPCHK(c, hipMemcpyAsync(hx, c->d_x, ns * n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (v) HIPCHK(c, hipMemcpyAsync(hv, c->d_v, ns * n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (box) HIPCHK(c, hipMemcpyAsync(hb, c->d_box, ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (dxdvdt) HIPCHK(c, hipMemcpyAsync(hs, c->d_steps, 3 * ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int q = 0; q < nk; ++q) {
        const size_t bq = (size_t)m[k0 + q];
        if (x) std::memcpy(x + q * n3, hx + bq * n3, n3 * sizeof(double));
        if (v) std::memcpy(v + q * n3, hv + bq * n3, n3 * sizeof(double));
        if (box) box[q] = hb[bq];
        if (dxdvdt) std::memcpy(dxdvdt + 3 * q, hs + 3 * bq, 3 * sizeof(double));
    }
    return NM_OK; // (slot_map has looked at the outcome of everything queued)
}

// The replicas named in slots[] only (the split-row exchange re-seats the ones that swapped, neuralmelting_amd/exchange.py): ONE look at
// the queue's outcome, the listed buffers copied back to back on the context's stream, ONE wait.  th[nk][5] = temp, pe, ke, virial, vol
// (nm_set_thermo's columns); small items go through the pinned staging area so that nothing on this stack frame is read after return.
int nm_get_slots(nm_ctx *c, int nk, const int *slots, double *x, double *v, double *box, double *dxdvdt, double *th)
{
    if (!c || nk < 0 || (nk && !slots)) return fail(c, NM_ERR_ARG, "nm_get_slots: bad argument");
    for (int q = 0; q < nk; ++q) if (slots[q] < 0 || slots[q] >= c->nslots) return fail(c, NM_ERR_ARG, "nm_get_slots: slot out of range");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    const size_t n3 = (size_t)3 * c->N;
    for (int q = 0; q < nk; ++q) {
        const size_t bq = (size_t)m[slots[q]];
        if (x) HIPCHK(c, hipMemcpyAsync(x + q * n3, c->d_x + bq * n3, n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (v) HIPCHK(c, hipMemcpyAsync(v + q * n3, c->d_v + bq * n3, n3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (box) HIPCHK(c, hipMemcpyAsync(box + q, c->d_box + bq, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (dxdvdt) HIPCHK(c, hipMemcpyAsync(dxdvdt + 3 * q, c->d_steps + 3 * bq, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (th) HIPCHK(c, hipMemcpyAsync(th + 5 * q, c->d_therm + 5 * bq, 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NM_OK;
}

int nm_set_slots(nm_ctx *c, int nk, const int *slots, const double *x, const double *v, const double *box, const double *dxdvdt,
                 const double *th)
{
    if (!c || nk < 0 || (nk && !slots)) return fail(c, NM_ERR_ARG, "nm_set_slots: bad argument");
    for (int q = 0; q < nk; ++q) if (slots[q] < 0 || slots[q] >= c->nslots) return fail(c, NM_ERR_ARG, "nm_set_slots: slot out of range");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    const size_t n3 = (size_t)3 * c->N;
    if ((rc = stage_reserve(c, (size_t)nk + 1))) return rc;
    double *vol = c->h_stage; // the volumes that go with the boxes (therm column 4), pinned
    for (int q = 0; q < nk; ++q) {
        const size_t bq = (size_t)m[slots[q]];
        if (x) HIPCHK(c, hipMemcpyAsync(c->d_x + bq * n3, x + q * n3, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (v) HIPCHK(c, hipMemcpyAsync(c->d_v + bq * n3, v + q * n3, n3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (box) {
            vol[q] = std::pow(box[q], 3.0);
            HIPCHK(c, hipMemcpyAsync(c->d_box + bq, box + q, sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_therm + 5 * bq + 4, vol + q, sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        if (dxdvdt) HIPCHK(c, hipMemcpyAsync(c->d_steps + 3 * bq, dxdvdt + 3 * q, 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (th) HIPCHK(c, hipMemcpyAsync(c->d_therm + 5 * bq, th + 5 * q, 5 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // the caller's arrays (pageable: staged by the runtime) and the staging area are free again
    return NM_OK;
}

// ---- output snapshots: write_outputs (remcmc:259-286) without stopping the stream.  nm_snapshot, queued right behind nm_run_block, copies what
// a recorded cycle writes — positions, box, the 17 thermo columns' sources — device to device on the context's stream (behind the block, in front of
// nm_adapt), and a side stream brings the copy to pinned host memory while the main stream goes on with nm_adapt, nm_exchange and the next block.
// nm_snapshot_fetch hands the OLDEST pending snapshot out; it waits for that snapshot's copy only, never for the main stream.  Two slots: a driver
// fetches cycle s - 1 after it has queued cycle s.  (nm_get_thermo + nm_get_state, the synchronous way, stop the GPU between two blocks for the
// copies and the host's turn-around: 7 % of a recorded C2 run.)  Neither call looks at the queue's outcome: a block that stopped on an error is
// reported by the next synchronising call as always, and its snapshot holds the state the block started from.
static size_t snap_layout(const nm_ctx *c, size_t off[8])
{
    const size_t ns = c->nslots, n3 = (size_t)3 * c->N;
    size_t o = 0;
    off[0] = o; o += ns * n3;        // x
    off[1] = o; o += ns;             // box
    off[2] = o; o += 5 * ns;         // therm
    off[3] = o; o += 3 * ns;         // steps
    off[4] = o; o += 6 * ns;         // count
    off[5] = o; o += (3 * ns + 1) / 2; // ratio (float)
    off[6] = o; o += (ns + 1) / 2;   // slot2buf (int)
    return o;
}

int nm_snapshot(nm_ctx *c)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    if (c->snap_count == 2) return fail(c, NM_ERR_STATE, "nm_snapshot: two snapshots are pending; fetch one first (nm_snapshot_fetch)");
    size_t off[8];
    const size_t nd = snap_layout(c, off);
    if (!c->side) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
        c->snap_doubles = nd;
        for (auto &sn : c->snap) {
            HIPCHK(c, dalloc(&sn.d, nd));
            HIPCHK(c, hipHostMalloc((void **)&sn.h, nd * sizeof(double), hipHostMallocDefault));
            HIPCHK(c, hipEventCreateWithFlags(&sn.taken, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&sn.landed, hipEventDisableTiming));
        }
    }
    nm_ctx::Snap &sn = c->snap[(c->snap_head + c->snap_count) & 1];
    const size_t ns = c->nslots, n3 = (size_t)3 * c->N;
    SnapArgs a;
    a.x = c->d_x; a.box = c->d_box; a.therm = c->d_therm; a.steps = c->d_steps; a.count = c->d_count; a.ratio = c->d_ratio; a.slot2buf = c->d_slot2buf;
    a.dst = sn.d;
    const size_t cnt[7] = { ns * n3, ns, 5 * ns, 3 * ns, 6 * ns, 3 * ns, ns };
    for (int q = 0; q < 7; ++q) { a.off[q] = off[q]; a.n[q] = cnt[q]; }
    const unsigned int blocks = (unsigned int)std::min<size_t>((ns * n3 + 1023) / 1024, 64); // (a handful of CUs for a few microseconds, between two blocks)
    hipLaunchKernelGGL(nm_snapshot_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, c->stream, a);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(sn.taken, c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->side, sn.taken, 0));
    HIPCHK(c, hipMemcpyAsync(sn.h, sn.d, nd * sizeof(double), hipMemcpyDeviceToHost, c->side));
    HIPCHK(c, hipEventRecord(sn.landed, c->side));
    sn.pending = true;
    ++c->snap_count;
    return NM_OK;
}

int nm_snapshot_fetch(nm_ctx *c, double *rows, double *x, double *box)
{
    if (!c) return NM_ERR_ARG;
    if (c->snap_count == 0) return fail(c, NM_ERR_STATE, "nm_snapshot_fetch: no snapshot is pending");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    nm_ctx::Snap &sn = c->snap[c->snap_head];
    HIPCHK(c, hipEventSynchronize(sn.landed));
    size_t off[8];
    snap_layout(c, off);
    const size_t ns = c->nslots, n3 = (size_t)3 * c->N;
    const double *hx = sn.h + off[0], *hb = sn.h + off[1], *th = sn.h + off[2], *st = sn.h + off[3], *cn = sn.h + off[4];
    const float *ra = (const float *)(sn.h + off[5]);
    const int *m = (const int *)(sn.h + off[6]);
    for (size_t k = 0; k < ns; ++k) {
        const size_t b = (size_t)m[k];
        if (rows) {
            double *r = rows + k * NM_THERMO_COLS;
            for (int q = 0; q < 5; ++q) r[q] = th[5 * b + q];
            for (int q = 0; q < 3; ++q) r[5 + q] = st[3 * b + q];
            for (int q = 0; q < 6; ++q) r[8 + q] = cn[6 * k + q];
            for (int q = 0; q < 3; ++q) r[14 + q] = (double)ra[3 * k + q];
        }
        if (x) std::memcpy(x + k * n3, hx + b * n3, n3 * sizeof(double));
        if (box) box[k] = hb[b];
    }
    sn.pending = false;
    c->snap_head ^= 1;
    --c->snap_count;
    return NM_OK;
}

int nm_run_block(nm_ctx *c, int mod)
{
    if (!c || mod < 0) return fail(c, NM_ERR_ARG, "nm_run_block: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return issue_block(c, OP_BLOCK, mod, c->step, c->trace_on, nullptr, true);
}

int nm_run_md(nm_ctx *c, int nsteps)
{
    if (!c || nsteps < 1) return fail(c, NM_ERR_ARG, "nm_run_md: bad argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return issue_block(c, OP_MD, nsteps, c->step, 0, nullptr, false);
}

int nm_run_cycles(nm_ctx *c, int ncycles, int mod)
{
    if (!c || ncycles < 1 || mod < 0) return fail(c, NM_ERR_ARG, "nm_run_cycles: bad argument");
    if (!c->whole_rows) return fail(c, NM_ERR_UNSUPPORTED, "nm_run_cycles: this context holds a partial pressure row (the exchange spans contexts)");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return issue_cycles(c, ncycles, mod, c->step, true);
}

int nm_get_thermo(nm_ctx *c, double *rows)
{
    if (!c || !rows) return fail(c, NM_ERR_ARG, "nm_get_thermo: null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::vector<int> m;
    int rc = slot_map(c, m);
    if (rc) return rc;
    const size_t ns = c->nslots;
    std::vector<double> th(ns * 5), st(ns * 3), cn(ns * 6);
    std::vector<float> ra(ns * 3);
    HIPCHK(c, hipMemcpy(th.data(), c->d_therm, th.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(st.data(), c->d_steps, st.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(cn.data(), c->d_count, cn.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(ra.data(), c->d_ratio, ra.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < ns; ++k) {
        const int b = m[k];
        double *r = rows + k * NM_THERMO_COLS;
        r[0] = th[5 * b]; r[1] = th[5 * b + 1]; r[2] = th[5 * b + 2]; r[3] = th[5 * b + 3]; r[4] = th[5 * b + 4];
        r[5] = st[3 * b]; r[6] = st[3 * b + 1]; r[7] = st[3 * b + 2];
        for (int q = 0; q < 6; ++q) r[8 + q] = cn[6 * k + q];
        for (int q = 0; q < 3; ++q) r[14 + q] = (double)ra[3 * k + q];
    }
    return NM_OK; // (slot_map has looked at the outcome of everything queued)
}

int nm_adapt(nm_ctx *c)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return issue_adapt(c);
}

int nm_exchange(nm_ctx *c, int *nswaps)
{
    if (!c) return NM_ERR_ARG;
    if (!c->whole_rows)
        return fail(c, NM_ERR_UNSUPPORTED, "nm_exchange: this context holds a partial pressure row; the sweep spans contexts (host exchange over RCCL)");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc = issue_exchange(c, c->step);
    if (rc) return rc;
    if (nswaps) {
        if ((rc = check_status(c))) return rc;
        HIPCHK(c, hipMemcpy(nswaps, c->d_nswaps, sizeof(int), hipMemcpyDeviceToHost));
    }
    return NM_OK;
}

int nm_synchronize(nm_ctx *c)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return check_status(c);
}

int nm_get_status(nm_ctx *c, int *status)
{
    if (!c || !status) return fail(c, NM_ERR_ARG, "nm_get_status: null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    (void)check_status(c); // re-issues a halted queue; an error that cannot be cured leaves its bits in d_status (settle), which is what this call is for
    HIPCHK(c, hipMemcpy(status, c->d_status, sizeof(int) * c->nslots, hipMemcpyDeviceToHost));
    return NM_OK;
}

int nm_timing_reset(nm_ctx *c)
{
    if (!c) return NM_ERR_ARG;
    for (auto &e : c->ev) harvest(c, e);
    c->launches = 0; c->total_ms = 0.0;
    return NM_OK;
}

int nm_timing_get(nm_ctx *c, int *launches, double *total_ms)
{
    if (!c) return NM_ERR_ARG;
    for (auto &e : c->ev) harvest(c, e);
    if (launches) *launches = c->launches;
    if (total_ms) *total_ms = c->total_ms;
    return NM_OK;
}

int nm_stats_get(nm_ctx *c, double *stats, int reset)
{
    if (!c || !stats) return fail(c, NM_ERR_ARG, "nm_stats_get: null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    const size_t n = (size_t)c->nslots * NM_STATS_COLS;
    HIPCHK(c, hipMemcpy(stats, c->d_stats, n * sizeof(double), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(c, hipMemset(c->d_stats, 0, n * sizeof(double)));
    return NM_OK;
}

#ifdef NM_EXPERIMENT
int nm_tline_get(nm_ctx *c, unsigned long long *out)
{
    if (!c || !out) return NM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_tline, (size_t)8 * 8 * 512 * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return NM_OK;
}
#endif
#ifdef NM_PROF
// diagnostic build only: out-of-range indices into the global spill / list arrays since the last call (counted and redirected by
// the kernel, nm_kernels.h NM_CHECK_INDEX); scripts/check_bounds.py runs the large-cell parity cases under it
int nm_prof_oob(nm_ctx *c, unsigned int *count)
{
    if (!c || !count) return NM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyFromSymbol(count, HIP_SYMBOL(nm::nm_oob_count), sizeof(unsigned int)));
    const unsigned int zero = 0;
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(nm::nm_oob_count), &zero, sizeof(unsigned int)));
    return NM_OK;
}
// diagnostic build only: rows of freshly built neighbour lists that lacked an atom inside rc + skin (exact fp64 check after every
// rebuild, Replica::rebuild) since the last call
int nm_prof_list_miss(nm_ctx *c, unsigned int *count)
{
    if (!c || !count) return NM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyFromSymbol(count, HIP_SYMBOL(nm::nm_list_miss), sizeof(unsigned int)));
    const unsigned int zero = 0;
    HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(nm::nm_list_miss), &zero, sizeof(unsigned int)));
    return NM_OK;
}
int nm_prof_miss_info(nm_ctx *c, double *info16)
{
    if (!c || !info16) return NM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpyFromSymbol(info16, HIP_SYMBOL(nm::nm_miss_info), 16 * sizeof(double)));
    return NM_OK;
}
// diagnostic build only: cycle sums per section and slot, [nslots][16]
int nm_prof_get(nm_ctx *c, unsigned long long *out, int reset)
{
    if (!c || !out) return NM_ERR_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, c->d_prof, (size_t)c->nslots * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    if (reset) HIPCHK(c, hipMemset(c->d_prof, 0, (size_t)c->nslots * 16 * sizeof(unsigned long long)));
    return NM_OK;
}
#endif

int nm_eval(nm_ctx *c, double *U, double *W, double *f)
{
    if (!c || !U || !W) return fail(c, NM_ERR_ARG, "nm_eval: null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int rc0 = check_status(c); // (whatever is queued first: an evaluation is not re-issued)
    if (rc0) return rc0;
    ++c->launch_id;
    KParams p;
    fill_params(c, p);
    p.eval_only = 1; p.tape = nullptr;
    p.evalU = c->d_evalU; p.evalW = c->d_evalW; p.evalF = f ? c->d_evalF : nullptr;
    HIPCHK(c, launch_kind(c, p));
    census_advance(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t ns = c->nslots;
    HIPCHK(c, hipMemcpy(U, c->d_evalU, ns * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(W, c->d_evalW, ns * sizeof(double), hipMemcpyDeviceToHost));
    if (f) HIPCHK(c, hipMemcpy(f, c->d_evalF, ns * 3 * c->N * sizeof(double), hipMemcpyDeviceToHost));
    return check_status(c);
}

int nm_set_rng_tape(nm_ctx *c, const double *tape, const int *offsets)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    if (c->d_tape) { HIPCHK(c, hipFree(c->d_tape)); c->d_tape = nullptr; }
    if (c->d_tape_off) { HIPCHK(c, hipFree(c->d_tape_off)); c->d_tape_off = nullptr; }
    if (!tape || !offsets) return NM_OK;
    const int total = offsets[c->nslots];
    if (total < 0) return fail(c, NM_ERR_ARG, "nm_set_rng_tape: bad offsets");
    HIPCHK(c, dalloc(&c->d_tape, (size_t)total));
    HIPCHK(c, dalloc(&c->d_tape_off, (size_t)c->nslots + 1));
    HIPCHK(c, hipMemcpy(c->d_tape, tape, sizeof(double) * total, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_tape_off, offsets, sizeof(int) * (c->nslots + 1), hipMemcpyHostToDevice));
    return NM_OK;
}

int nm_set_exchange_tape(nm_ctx *c, const double *tape, int n)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    const int npairs = c->cfg.nrows * c->cfg.nt * (c->cfg.nt - 1) / 2;
    if (!tape) { c->xtape_n = 0; return NM_OK; }
    if (n != npairs) return fail(c, NM_ERR_ARG, "nm_set_exchange_tape: need one uniform per pair of the sweep");
    HIPCHK(c, hipMemcpy(c->d_xtape, tape, sizeof(double) * n, hipMemcpyHostToDevice));
    c->xtape_n = n;
    return NM_OK;
}

int nm_set_trace(nm_ctx *c, int enable)
{
    if (!c) return NM_ERR_ARG;
    c->trace_on = enable ? 1 : 0;
    return NM_OK;
}

int nm_get_trace(nm_ctx *c, double *trace, int mod)
{
    if (!c || !trace) return fail(c, NM_ERR_ARG, "nm_get_trace: null argument");
    if (!c->d_trace || mod != c->trace_mod) return fail(c, NM_ERR_ARG, "nm_get_trace: no trace of that length recorded");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    HIPCHK(c, hipMemcpy(trace, c->d_trace, (size_t)c->nslots * mod * NM_TRACE_COLS * sizeof(double), hipMemcpyDeviceToHost));
    return NM_OK;
}

int nm_set_counters(nm_ctx *c, const double *count, const float *ratio)
{
    if (!c) return NM_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    if (count) HIPCHK(c, hipMemcpy(c->d_count, count, sizeof(double) * 6 * c->nslots, hipMemcpyHostToDevice));
    if (ratio) HIPCHK(c, hipMemcpy(c->d_ratio, ratio, sizeof(float) * 3 * c->nslots, hipMemcpyHostToDevice));
    return NM_OK;
}

int nm_get_perm(nm_ctx *c, int *perm)
{
    if (!c || !perm) return fail(c, NM_ERR_ARG, "nm_get_perm: null argument");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    HIPCHK(c, hipMemcpy(perm, c->d_slot2buf, sizeof(int) * c->nslots, hipMemcpyDeviceToHost));
    return NM_OK;
}

int nm_get_exchange_crit(nm_ctx *c, double *crit, int n)
{
    if (!c || !crit) return fail(c, NM_ERR_ARG, "nm_get_exchange_crit: null argument");
    const int npairs = c->cfg.nrows * c->cfg.nt * (c->cfg.nt - 1) / 2;
    if (n != npairs) return fail(c, NM_ERR_ARG, "nm_get_exchange_crit: wrong pair count");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    { const int rc_ = check_status(c); if (rc_) return rc_; } // (a halted queue is re-issued, or its error reported, before anything is read or replaced)
    HIPCHK(c, hipMemcpy(crit, c->d_xcrit, sizeof(double) * n, hipMemcpyDeviceToHost));
    return NM_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// structural histograms (include/nm_distr.h; lammps_distr.py:123-171)
namespace {
thread_local std::string g_distr_error;
int dfail(int code, const std::string &m) { g_distr_error = m; return code; }
} // namespace

extern "C" {

const char *nm_distr_last_error(void) { return g_distr_error.c_str(); }

int nm_distr_histograms(int device, int ns, int natoms, const float *pos, const float *box, int sbins, const double *r_edges,
                        int cbins, const double *rv_edges, float *rdf, float *cdf)
{
#define DCHK(call)                                                                                     \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) return dfail(NM_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
    if (ns < 0 || natoms < 1 || !pos || !box) return dfail(NM_ERR_ARG, "nm_distr_histograms: bad argument");
    if (rdf && (!r_edges || sbins < 2 || sbins > DISTR_MAXS)) return dfail(NM_ERR_ARG, "nm_distr_histograms: bad spherical bins");
    if (cdf && (!rv_edges || cbins < 1 || cbins > DISTR_MAXC)) return dfail(NM_ERR_ARG, "nm_distr_histograms: bad cartesian bins");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return dfail(NM_ERR_HIP, "nm_distr_histograms: no HIP device available");
    if (device < 0 || device >= ndev) return dfail(NM_ERR_ARG, "nm_distr_histograms: device ordinal out of range");
    if (ns == 0) return NM_OK;
    DCHK(hipSetDevice(device));
    const int sb = rdf ? sbins : 0, cb = cdf ? cbins : 0;
    const size_t nc = (size_t)cb * cb * cb;
    const int npad = (natoms + 63) & ~63;
    const size_t lds = (((size_t)3 * (natoms + npad) * sizeof(float) + 7) & ~(size_t)7) + (size_t)(sb + cb + 1) * sizeof(double)
                     + ((size_t)sb + nc) * sizeof(unsigned int);
    if (lds > 160 * 1024) return dfail(NM_ERR_ARG, "nm_distr_histograms: working set exceeds LDS");
    if (natoms >= 4096) return dfail(NM_ERR_ARG, "nm_distr_histograms: natoms^2 must stay below 2^24 (float32 counts, as in the reference)");
    DCHK(hipFuncSetAttribute((const void *)nm_distr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int chunk = 4096; // samples per launch: bounds device memory (pos 12 N B + counts) for long trajectories
    float *d_pos = nullptr, *d_box = nullptr;
    double *d_re = nullptr, *d_ve = nullptr;
    float *d_r = nullptr, *d_c = nullptr;
    const int cs = ns < chunk ? ns : chunk;
    DCHK(hipMalloc((void **)&d_pos, (size_t)cs * natoms * 3 * sizeof(float)));
    DCHK(hipMalloc((void **)&d_box, (size_t)cs * sizeof(float)));
    if (rdf) { DCHK(hipMalloc((void **)&d_re, sbins * sizeof(double))); DCHK(hipMalloc((void **)&d_r, (size_t)cs * sbins * sizeof(float)));
               DCHK(hipMemcpy(d_re, r_edges, sbins * sizeof(double), hipMemcpyHostToDevice)); }
    if (cdf) { DCHK(hipMalloc((void **)&d_ve, (cbins + 1) * sizeof(double))); DCHK(hipMalloc((void **)&d_c, (size_t)cs * nc * sizeof(float)));
               DCHK(hipMemcpy(d_ve, rv_edges, (cbins + 1) * sizeof(double), hipMemcpyHostToDevice)); }
    for (int s0 = 0; s0 < ns; s0 += cs) {
        const int n = (ns - s0) < cs ? (ns - s0) : cs;
        DCHK(hipMemcpy(d_pos, pos + (size_t)s0 * natoms * 3, (size_t)n * natoms * 3 * sizeof(float), hipMemcpyHostToDevice));
        DCHK(hipMemcpy(d_box, box + s0, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
        if (rdf) DCHK(hipMemset(d_r, 0, (size_t)n * sbins * sizeof(float)));
        if (cdf) DCHK(hipMemset(d_c, 0, (size_t)n * nc * sizeof(float)));
        hipLaunchKernelGGL(nm_distr_kernel, dim3(n * 27), dim3(DISTR_BLOCK), lds, 0, natoms, d_pos, d_box, sb, d_re, cb, d_ve, d_r, d_c);
        DCHK(hipGetLastError());
        DCHK(hipDeviceSynchronize());
        if (rdf) DCHK(hipMemcpy(rdf + (size_t)s0 * sbins, d_r, (size_t)n * sbins * sizeof(float), hipMemcpyDeviceToHost));
        if (cdf) DCHK(hipMemcpy(cdf + (size_t)s0 * nc, d_c, (size_t)n * nc * sizeof(float), hipMemcpyDeviceToHost));
    }
    hipFree(d_pos); hipFree(d_box); hipFree(d_re); hipFree(d_ve); hipFree(d_r); hipFree(d_c);
#undef DCHK
    return NM_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// output formatting (include/nm.h; write_outputs, remcmc:235-286): plain host code
extern "C" {

int nm_format_thrm(const double *row17, char *out, int cap)
{
    if (!row17 || !out || cap < 17 * 14 + 2) return NM_ERR_ARG;
    return format_thrm(row17, out);
}

int nm_format_traj(int natoms, double box, const double *x, char *out, int cap)
{
    if (natoms < 0 || !x || !out || cap < 32 + 42 * natoms) return NM_ERR_ARG;
    return format_traj(natoms, box, x, out);
}

int nm_append_outputs(int nk, int natoms, const char *const *thrm_paths, const char *const *traj_paths, const double *rows,
                      const double *x, const double *box, int nthreads)
{
    if (nk < 0 || natoms < 0 || !thrm_paths || !traj_paths || !rows || !x || !box) return NM_ERR_ARG;
    if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > nk) nthreads = nk > 0 ? nk : 1;
    std::vector<int> err((size_t)nthreads, 0);
    auto work = [&](int t) {
        std::vector<char> buf((size_t)64 + 42 * (size_t)natoms + 17 * 14);
        for (int k = t; k < nk; k += nthreads) {
            int n = format_thrm(rows + 17 * (size_t)k, buf.data());
            FILE *f = std::fopen(thrm_paths[k], "a");
            if (!f || std::fwrite(buf.data(), 1, (size_t)n, f) != (size_t)n) err[t] = 1;
            if (f) std::fclose(f);
            n = format_traj(natoms, box[k], x + 3 * (size_t)natoms * k, buf.data());
            f = std::fopen(traj_paths[k], "a");
            if (!f || std::fwrite(buf.data(), 1, (size_t)n, f) != (size_t)n) err[t] = 1;
            if (f) std::fclose(f);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &t : th) t.join();
    for (int e : err) if (e) return NM_ERR_ARG;
    return NM_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// .thrm / .traj reader (include/nm_parse.h; lammps_parse.py:48-49, 88-96)
// ------------------------------------------------------------------------------------------------------------------
namespace {
thread_local std::string g_parse_error;
}
extern "C" {
const char *nm_parse_last_error(void) { return g_parse_error.c_str(); }

int nm_parse_thrm(const char *path, float *rows, long cap_rows, long *nrows, int nthreads)
{
    if (!path || (rows && cap_rows < 0)) { g_parse_error = "nm_parse_thrm: bad argument"; return NM_ERR_ARG; }
    try {
        return nm::parse_thrm(path, rows, cap_rows, nrows, nthreads, g_parse_error) == 0 ? NM_OK : NM_ERR_ARG;
    } catch (const std::exception &e) { g_parse_error = e.what(); return NM_ERR_ARG; }
}

int nm_parse_traj(const char *path, uint16_t *natoms, float *box, float *pos, long cap_frames, long cap_posrows, long *nframes,
                  long *nposrows, int nthreads)
{
    if (!path) { g_parse_error = "nm_parse_traj: bad argument"; return NM_ERR_ARG; }
    try {
        return nm::parse_traj(path, natoms, box, pos, cap_frames, cap_posrows, nframes, nposrows, nthreads, g_parse_error) == 0
                   ? NM_OK : NM_ERR_ARG;
    } catch (const std::exception &e) { g_parse_error = e.what(); return NM_ERR_ARG; }
}
}
