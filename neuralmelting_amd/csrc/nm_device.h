// nm_device.h — device-side building blocks of the MI355X (gfx950) NPT-HMC engine.
//
// One workgroup owns one replica for a whole block of MOD moves.  Coordinates, velocities, forces,
// the saved copies for Metropolis rejection and (for N <= 256) the byte-indexed Verlet list live
// in LDS for the entire block; HBM is touched once at block start and once at block end.
// Wave width is 64 throughout (CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nm {

// RNG stream ids: counter = (index, stream, tag, step), key = (seed, global slot).  DESIGN.md §RNG.
enum : uint32_t { S_ROLL = 0, S_ACC = 1, S_VOL = 2, S_DISP_XY = 3, S_DISP_Z = 4, S_VEL_A = 5, S_VEL_B = 6,
                  S_EXCH = 7, S_ITER_XY = 8, S_ITER_Z = 9, S_ITER_ACC = 10 };

// status bits per slot
enum : int { ST_LIST_OVERFLOW = 1, ST_BOX_TOO_SMALL = 2, ST_TAPE_EXHAUSTED = 4, ST_NONFINITE = 8, ST_SYNC_TIMEOUT = 16, ST_NOT_RESIDENT = 32 };

struct KParams {
    int N, nslots, slot0;          // atoms, local replicas, global index of local slot 0
    int mod, nstps, bulk, iter_revert;
    int eval_only;                 // nm_eval: evaluate the loaded states and leave
    int md_mode;                   // nm_run_md: velocities at T, then nstps velocity-Verlet steps, no Metropolis test
    uint32_t seed, step;
    double ppos, pvol, lat, mass;
    double kB, mvv2e, ftm2v, nktv2p;
    double rc, skin;
    double sc_eps, sc_a2, sc_c;    // Sutton-Chen EAM (element Al): E = eps [ 1/2 sum (a/r)^7 - c sum sqrt(rho) ], rho = sum (a/r)^6
    // per buffer
    double *x, *v, *box, *steps, *therm;
    // per slot
    double *count;
    float *ratio;
    const int *slot2buf;
    const double *et, *pf, *tq;
    int *status;
    double *stats;
    const double *tape;
    const int *tape_off;
    double *trace;                 // [slot][mod][4] or null
    double *evalU, *evalW, *evalF; // nm_eval outputs
    void *nbr_g;                   // global neighbour lists (N > 256): [slot][maxnb*N] uint16
    double *aux_g;                 // global spill of the saved copies (large N): [slot][AUX_DOUBLES(N)]
    unsigned long long *prof;      // diagnostic build only
    int cus;                       // workgroups (CUs) cooperating on one replica
    int dbg;                       // NM_DBG: timing experiments only (skips work, results are wrong)
    int inj_rebuild, inj_q;        // fault injection for the tests of the error path (NM_INJECT_OVERFLOW=n,q): the n-th list rebuild of
                                   // a block reports an overflow in workgroup q of every cluster; -1 = off
    unsigned long long *tline;     // experiment build only: 100 MHz timestamps of slot 0's evaluations [q][wave][eval][8]
    double *xbuf;                  // [slot][2][XBUF_DOUBLES]: force slices + partial sums exchanged inside a cluster
    uint32_t launch_id;            // distinguishes the granules of successive launches
    int plain_granules;            // 1: clusters found to sit on one XCD hand over through that XCD's L2 (plain stores); 0: always write-through
    unsigned int *census;          // cluster launches: arrival counter of the grid's workgroups, then one per cluster.  The counters only ever
                                   // grow: a launch is complete at census_base + its own count (no memset in front of every launch); the
                                   // host zeroes them, and the bases, after a census that was given up (abort bit) and before they could wrap
    unsigned int census_base, census_cbase; // what the grid's counter / every cluster's counter stood at before this launch
    int *status_acc;               // per slot: status bits of every block since the host last looked (status[] holds the last launch's)
    int *halt;                     // launch id of the first block that stopped on an error; later launches do nothing until the host
                                   // has dealt with it (nm_api.hip settle): a failed block never has successors running on its state
    const int *rerun_mask;         // re-issue of a block after a hand-over timeout: only the slots marked here run; null = all
    int inj_census;                // fault injection (NM_TESTING=1, NM_INJECT_CENSUS): this launch's residency census fails
    int over;                      // the grid holds more clusters than the chip at once: every cluster takes its own census (census[1 + cluster])
    const int *order;              // one workgroup per replica and more replicas than the chip holds at once: workgroup b runs slot order[b],
                                   // slowest first (by the time each slot's previous block took); null = identity
    unsigned long long *last_ticks; // per slot: duration of its last block (100 MHz ticks), what nm_order_kernel sorts by
    // nm_run_cycles (nm_cycles_kernel): several cycles of block / adapt / exchange in ONE launch; the replicas of a pressure row meet at the end of
    // every block, the rows never wait for one another
    int ncycles, nt, row0;         // cycles of this launch; temperatures per row; global index of the context's first row
    unsigned int *rowbar, *rowgo;  // per local row: workgroups arrived (monotonic within the launch), cycles released by the row's leader
    int *cyc_abort;                // set by whoever leaves the launch early: nobody waits for a row that will not complete
    int *nswaps;                   // accepted swaps of the launch (atomic)
};

// status bits reach the host through status[] (last launch) and status_acc[] (everything since the host last looked); a block
// that stopped (stops) arms the halt word with its launch id
__device__ __forceinline__ void report_status(const KParams &p, int slot, int bits, bool stops)
{
    if (!bits) return;
    atomicOr(&p.status[slot], bits);
    if (p.status_acc) atomicOr(&p.status_acc[slot], bits);
    if (stops && p.halt) atomicCAS(p.halt, 0, (int)p.launch_id);
}
// true when an earlier launch stopped on an error the host has not dealt with yet
__device__ __forceinline__ bool halted(const KParams &p)
{
    if (!p.halt) return false;
    const int h = __hip_atomic_load(p.halt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return h != 0 && h != (int)p.launch_id;
}

// ------------------------------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t (&o)[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo)
{
    const unsigned long long w = ((unsigned long long)hi << 32) | (unsigned long long)lo;
    return (double)(w >> 11) * (1.0 / 9007199254740992.0);
}

// value of float('%f' % x): the reference hands box edge, timestep, temperature and displacement
// amplitude to LAMMPS through '%f' strings (remcmc:466,483,571,604,607)
__device__ __forceinline__ double q6(double x)
{
    // the product and the difference must be individually rounded: hipcc's default fp-contract=fast would fuse
    // p - n into fma(x, 1e6, -n) and lose the tie test (0.03125*1.122 -> 35062.5 is such a tie)
#pragma clang fp contract(off)
    const double p = x * 1.0e6;
    double n = rint(p);
    if (fabs(p - n) == 0.5) {
        const double e = __builtin_fma(x, 1.0e6, -p); // exact residual of the product
        if (e > 0.0) n = floor(p) + 1.0;
        else if (e < 0.0) n = floor(p);
    }
    return n / 1.0e6;
}

// A value every lane holds identically, moved to scalar registers: tells the compiler it is wave-uniform so it
// stops occupying a VGPR pair (the block-wide sums and everything derived from them are such values).
__device__ __forceinline__ double uniform(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------------------------------ reductions
// Workgroup-wide OR of two flags with one barrier (bit 0 = a, bit 1 = b)
template <int NW, int NVMAX>
__device__ __forceinline__ int block_any2(bool a, bool b, double *red, int &parity)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *r = (int *)(red + parity * (NW * NVMAX));
    const unsigned long long ma = __ballot(a), mb = __ballot(b);
    if (lane == 0) r[wv] = ((ma != 0ull) ? 1 : 0) | ((mb != 0ull) ? 2 : 0);
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) any |= r[w];
    parity ^= 1;
    return __builtin_amdgcn_readfirstlane(any);
}

// Workgroup-wide OR of three flags with one barrier (bits 0, 1, 2)
template <int NW, int NVMAX>
__device__ __forceinline__ int block_any3(bool a, bool b, bool c, double *red, int &parity)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *r = (int *)(red + parity * (NW * NVMAX));
    const unsigned long long ma = __ballot(a), mb = __ballot(b), mc = __ballot(c);
    if (lane == 0) r[wv] = ((ma != 0ull) ? 1 : 0) | ((mb != 0ull) ? 2 : 0) | ((mc != 0ull) ? 4 : 0);
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) any |= r[w];
    parity ^= 1;
    return __builtin_amdgcn_readfirstlane(any);
}

// Workgroup-wide OR with one barrier: a ballot per wave, one LDS word per wave (HIP's __syncthreads_or funnels all
// 512 threads through LDS atomics: ~4k cycles per call in this kernel).  Same two-halves trick as block_sum.
template <int NW, int NVMAX>
__device__ __forceinline__ bool block_any(bool flag, double *red, int &parity)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int *r = (int *)(red + parity * (NW * NVMAX));
    const unsigned long long m = __ballot(flag);
    if (lane == 0) r[wv] = (m != 0ull) ? 1 : 0;
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) any |= r[w];
    parity ^= 1;
    return __builtin_amdgcn_readfirstlane(any) != 0;
}

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Butterfly over the 64 lanes: every lane ends with the bit-identical total (each step adds the partner's partial, and the
// partner adds this lane's: a + b == b + a).  The four steps inside a 16-lane row are DPP moves (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror: no LDS round trip), only the two steps across rows go through ds_bpermute.
__device__ __forceinline__ double wave_sum(double v)
{
    v += dpp_mov<0xB1>(v);  // quad_perm:[1,0,3,2]
    v += dpp_mov<0x4E>(v);  // quad_perm:[2,3,0,1]
    v += dpp_mov<0x141>(v); // row_half_mirror: lane i <-> 7-i of its 8 lanes (the other quad's sum)
    v += dpp_mov<0x140>(v); // row_mirror: lane i <-> 15-i of its row (the other half row's sum)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// Sum NV values over the workgroup; every thread returns the same bits (fixed order: butterfly inside a
// wave, then waves 0..NW-1).  `red` has 2*NW*NVMAX doubles and is used in two alternating halves so that
// one barrier per reduction suffices.
// nact: waves 0 .. nact-1 may hold non-zero addends (a sum over N < BLOCK items dealt out by thread index); the others skip the
// butterflies — sixteen of them in velocity_create, on SIMDs they share with the waves that do have work — and are left out of
// the second stage.  Adding their zeros or not gives the same bits.
template <int NV, int NW, int NVMAX>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int &parity, int nact = NW)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    double *r = red + parity * (NW * NVMAX);
    if (wv < nact) {
        if constexpr (NV == 16) {
            // Sixteen sums at once (the moments of `velocity create`): instead of sixteen butterflies of six steps, ONE in which a lane
            // hands half of what it still carries to its partner and keeps the other half — 8 + 4 + 2 + 1 additions and exchanges, then
            // two steps across the 16-lane rows with the one value that is left: ~115 instructions instead of ~290.  Lane l ends
            // with the wave's total of value 8 b0 + 4 b1 + 2 b2 + b3 (b = the bits of l); the order of the additions is fixed, the same
            // bits in every lane that holds the same value.
            {
                const bool up = (lane & 1) != 0;
#pragma unroll
                for (int q = 0; q < 8; ++q) { const double keep = up ? v[q + 8] : v[q], give = up ? v[q] : v[q + 8]; v[q] = keep + dpp_mov<0xB1>(give); } // lane ^ 1
            }
            {
                const bool up = (lane & 2) != 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const double keep = up ? v[q + 4] : v[q], give = up ? v[q] : v[q + 4]; v[q] = keep + dpp_mov<0x4E>(give); } // lane ^ 2
            }
            {
                const bool up = (lane & 4) != 0;
#pragma unroll
                for (int q = 0; q < 2; ++q) { const double keep = up ? v[q + 2] : v[q], give = up ? v[q] : v[q + 2]; v[q] = keep + __shfl_xor(give, 4, 64); }
            }
            {
                const bool up = (lane & 8) != 0;
                const double keep = up ? v[1] : v[0], give = up ? v[0] : v[1];
                v[0] = keep + __shfl_xor(give, 8, 64);
            }
            v[0] += __shfl_xor(v[0], 16, 64);
            v[0] += __shfl_xor(v[0], 32, 64);
            if (lane < 16) r[wv * NVMAX + ((lane & 1) * 8 + ((lane >> 1) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 3) & 1))] = v[0];
        } else {
#pragma unroll
            for (int q = 0; q < NV; ++q) {
                const double s = wave_sum(v[q]);
                if (lane == 0) r[wv * NVMAX + q] = s;
            }
        }
    }
    __syncthreads();
    // second stage: lane q < NV of every wave adds the partial sums of value q (NW reads instead of NW*NV per thread), then
    // the totals are handed to all lanes through v_readlane: scalar registers, the same bits everywhere
    double s = 0.0;
    if (lane < NV)
        for (int w = 0; w < nact; ++w) s += r[w * NVMAX + lane];
    const long long sb = __double_as_longlong(s);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const uint32_t lo = __builtin_amdgcn_readlane((int)sb, q), hi = __builtin_amdgcn_readlane((int)(sb >> 32), q);
        v[q] = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
    parity ^= 1;
}

} // namespace nm
