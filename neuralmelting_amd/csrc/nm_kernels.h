// nm_kernels.h — the persistent per-replica block kernel (gen_sample, remcmc:665-691) and the small
// adapt / exchange kernels (remcmc:726-745, 776-803) for gfx950.
//
// Mapping to the reference (every LAMMPS command the reference issues is re-implemented here):
//   Replica::wrap + eval        "run 0"             remcmc:469,484,496,527,573,588,608,633
//   Replica::eval               pair_style lj/cut 2.5 (or the Al EAM) energy/force/virial over a Verlet list
//   nm_block_kernel, PH_BULK    bulk_position_mc    remcmc:477-502  (displace_atoms all random ...)
//   Replica::iter_pmc           iter_position_mc    remcmc:505-549  (reference mode: iter_pmc_all, the N trials of a move at once)
//   nm_block_kernel, PH_VMC     volume_mc           remcmc:552-595  (change_box ... + scaled scatter)
//   nm_block_kernel, PH_HMC_*   hamiltonian_mc      remcmc:598-640  (Replica::hmc_velocities = velocity create/zero; fix nve run NSTPS)
//   end of nm_block_kernel      lammps_extract      remcmc:377-391 + ratios remcmc:685-688
//   Replica::advance_and_share, exchange_sums (no counterpart)  hand-over between the workgroups that share one replica
#pragma once
#include "nm_device.h"
#include "nm_math.h"

namespace nm {

#ifndef NM_AB
#define NM_AB 0 // A/B experiments: a variant library is built with -DNM_AB=k (scripts/ab_multi.sh), the shipped one with 0
#endif
#ifndef NM_POLL_SLEEP
#define NM_POLL_SLEEP 1 // s_sleep units (64 clocks each) between two polls of a hand-over granule (0: none)
#endif
#ifndef NM_PAIR_W
#define NM_PAIR_W 2 // listed neighbours a thread works on at once (pair_vec)
#endif
#ifndef NM_HALF_LIST
#define NM_HALF_LIST 1 // one thread per row (lists in HBM/L2): a pair of two atoms of the same workgroup is listed ONCE (Cfg::HALF); 0 = full lists
#endif
#ifndef NM_SPREAD
#define NM_SPREAD 1 // pair loop over LDS lists: the row's epilogue on three lanes, its operands prefetched (0: one lane, as in rounds 1-3)
#endif
// pair loop over LDS lists: list entry at which the two waves of a SIMD swap priorities (see pair_loop), by threads per row — a thread
// holds (row length) / TPA entries, and the best swap point sits at ~0.6 of them.  Equilibrated, same box (round 4): TPA 8 (C2):
// 1.415 / 1.420 / 1.409 / 1.403 M sweeps/s at 8 / 10 / 12 / 14 (12 in rounds 2-3); TPA 4 (C3 share): 360 k at 20 vs 354 k at 10 or 12,
// 354 k at 26.  A swap point proportional to the row's own length — a run-time compare in every trip instead of one the compiler
// folds — measured 3 % slower.
#ifndef NM_PRIO_SW16
#define NM_PRIO_SW16 6
#endif
#ifndef NM_PRIO_SW8
#define NM_PRIO_SW8 10
#endif
#ifndef NM_PRIO_SW4
#define NM_PRIO_SW4 20
#endif
#ifndef NM_PRIO_SW2
#define NM_PRIO_SW2 40
#endif

constexpr int NVMAX = 16; // widest block reduction (the 16 raw moments of hmc_velocities)

// Diagnostic build only (-DNM_PROF, never the shipped library): shader-clock stamps per section, summed by lane 0
// of each workgroup into KParams::prof[slot][NM_PROF_SLOTS].
// NM_DBG(bit): timing experiments that skip work (results are wrong); compiled in only for the diagnostic build
#if defined(NM_PROF) || defined(NM_EXPERIMENT)
#define NM_DBG(bit) (p.dbg & (bit))
#else
#define NM_DBG(bit) false
#endif
// NM_EXPERIMENT build: TLINE(k) records the constant-rate clock (100 MHz, chip-wide) at point k of an evaluation, per wave, for
// the workgroups of slot 0 only: lets the phases of one evaluation be laid on a common time axis across the cluster.
#ifdef NM_EXPERIMENT
#define TL_EVALS 512
#define TLINE(k) do { if (tl && tl_n < TL_EVALS && (tid & 63) == 0) tl[((size_t)(q * NW + (tid >> 6)) * TL_EVALS + tl_n) * 8 + (k)] = wall_clock64(); } while (0)
// the same for the hand-over that follows evaluation tl_n - 1
#define TLINE_PREV(k) do { if (tl && tl_n > 0 && tl_n <= TL_EVALS && (tid & 63) == 0) tl[((size_t)(q * NW + (tid >> 6)) * TL_EVALS + tl_n - 1) * 8 + (k)] = wall_clock64(); } while (0)
#ifdef NM_TL_REBUILD // a second experiment build: slots 3 .. 7 of an evaluation hold stamps from INSIDE Replica::rebuild instead (scripts/probe_rebuild.py)
#define RTL(k) do { if (tl && tl_n < TL_EVALS && (tid & 63) == 0) tl[((size_t)(q * NW + (tid >> 6)) * TL_EVALS + tl_n) * 8 + (k)] = wall_clock64(); } while (0)
#undef TLINE
#undef TLINE_PREV
#define TLINE(k) do { if ((k) < 3 && tl && tl_n < TL_EVALS && (tid & 63) == 0) tl[((size_t)(q * NW + (tid >> 6)) * TL_EVALS + tl_n) * 8 + (k)] = wall_clock64(); } while (0)
#define TLINE_PREV(k) do { } while (0)
#else
#define RTL(k) do { } while (0)
#endif
#else
#define TLINE(k) do { } while (0)
#define TLINE_PREV(k) do { } while (0)
#define RTL(k) do { } while (0)
#endif
#ifdef NM_PROF
#define NM_PROF_SLOTS 16
#define PROF_DECL unsigned long long prof_acc[NM_PROF_SLOTS] = {}; unsigned long long prof_t0 = 0;
#define PROF_BEGIN() do { __builtin_amdgcn_sched_barrier(0); prof_t0 = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PROF_END(k) do { __builtin_amdgcn_sched_barrier(0); prof_acc[k] += __builtin_readcyclecounter() - prof_t0; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PROF_DECL
#define PROF_BEGIN() do { } while (0)
#define PROF_END(k) do { } while (0)
#endif

// The dynamic LDS block of the workgroup.  Arrays inside it are addressed as CONSTANT offsets from this symbol through the empty
// proxy types below: the compiler then knows the address space (ds_read/ds_write with immediate offsets).  Plain `double *`
// members pointing into LDS were kept by the compiler as generic 64-bit pointers in scratch memory and dereferenced with
// flat loads — a scratch round trip in front of every elementwise phase.
extern __shared__ __align__(16) unsigned char nm_lds[];

template <typename T, size_t OFF>
struct LdsArr {
    __device__ __forceinline__ T *ptr() const { return (T *)(nm_lds + OFF); }
    template <typename I> __device__ __forceinline__ T &operator[](I i) const { return ptr()[i]; }
    __device__ __forceinline__ operator T *() const { return ptr(); }
};
// the same interface over a pointer into global memory (configurations whose saved copies / lists do not fit in LDS)
#ifdef NM_PROF
// Diagnostic build: every index into a global spill / list array is checked against the array's extent; a violation is counted
// (nm_oob_count, read by nm_prof_oob) and redirected to element 0, so the diagnostic run itself cannot fault.
__device__ unsigned int nm_oob_count;
__device__ unsigned int nm_list_miss; // rows of a freshly built list that lack an atom inside rc + skin (Replica::rebuild's self-check)
__device__ double nm_miss_info[16];    // details of the first such row
#define NM_BOUND(arr, count) (arr).n = (size_t)(count)
#define NM_CHECK_INDEX(i, count) ((size_t)(i) < (size_t)(count) ? (size_t)(i) : (atomicAdd(&nm_oob_count, 1u), (size_t)0))
#else
#define NM_BOUND(arr, count) do { } while (0)
#define NM_CHECK_INDEX(i, count) (i)
#endif
template <typename T>
struct GlobArr {
    T *g = nullptr;
#ifdef NM_PROF
    size_t n = 0;
    template <typename I> __device__ __forceinline__ T &operator[](I i) const { return g[NM_CHECK_INDEX(i, n)]; }
#else
    template <typename I> __device__ __forceinline__ T &operator[](I i) const { return g[i]; }
#endif
    __device__ __forceinline__ T *ptr() const { return g; }
    __device__ __forceinline__ operator T *() const { return g; }
};
template <bool LDS, typename T, size_t OFF> struct ArrSel { using type = LdsArr<T, OFF>; };
template <typename T, size_t OFF> struct ArrSel<false, T, OFF> { using type = GlobArr<T>; };

// NMAX (array stride) and MAXNB (neighbour slots per atom) are compile-time so that every LDS array sits at a
// constant offset: no address registers are needed for them (with run-time strides the eighteen array bases were
// spilled to scratch and reloaded inside the pair loop).
template <int BLOCK_, int TPA_, int NMAX_, int MAXNB_, typename IdxT_, bool LIST_LDS_, bool SAVE_LDS_, int POT_ = 0, int NLIST_ = NMAX_,
          bool SAVEV_LDS_ = SAVE_LDS_, bool LDS_LIST2_ = false, bool HALF_ = false>
struct Cfg {
    // LDS_LIST2: an LDS list is kept twice as well (see LIST2 below): the second copy, the reference positions and the row lengths
    // that belong to the list a move started from live in LDS too.  Affordable once a cluster configuration stores only the rows
    // of its own atoms (NLIST = NMAX / Q).
    static constexpr bool LDS_LIST2 = LDS_LIST2_;
    static_assert(!LDS_LIST2_ || (LIST_LDS_ && SAVE_LDS_), "a second LDS list needs the first one and the saved copies in LDS");
    // NLIST: rows of an LDS list.  A workgroup only builds and reads the rows of its own atoms; a cluster configuration whose own
    // range is a fixed fraction of NMAX stores just those (row = i - a0).  SAVEV_LDS: the saved velocities (touched twice per
    // HMC move) may live in the global spill while the saved positions and the list reference stay in LDS — together these two
    // make room for the list of the 6^3 system at 8 workgroups per replica.
    static constexpr int NLIST = NLIST_;
    static constexpr bool SAVEV_LDS = SAVEV_LDS_;
    static_assert(SAVE_LDS_ || !SAVEV_LDS_, "saved velocities in LDS only together with the saved positions");
    static constexpr int POT = POT_; // 0 = lj/cut 2.5, 1 = Sutton-Chen EAM (two-pass: densities, then forces)
    // HALF (round 4): the configurations whose lists live in HBM / L2 with one thread per row evaluate a pair of two atoms of the SAME
    // workgroup once instead of twice.  The row that holds the pair — chosen by a checkerboard rule, (j > i) == ((i + j) even), so that
    // every row keeps about half of its own-range neighbours whatever its index — adds the force to its own sum and the opposite force
    // to the partner's entry of the force array.  That addition is an LDS atomic on a 64-bit INTEGER (fixed point, 2^-36 of the force
    // unit = 1.5e-11, range 3.3e4: Replica::fixed_add): integer sums do not depend on the order in which the waves' atomics land, so
    // results stay bit-reproducible (an fp64 atomic add would not be).  Pairs with an atom of another workgroup stay in both rows.
    // Halves the list bytes streamed per evaluation and the pair evaluations of the own range (all of them at one workgroup per replica).
    // Measured (round 4, same box, sustained rate, half against full lists): 500 atoms at ONE workgroup per replica (the run.sh setting)
    // +8.6 %; 864 atoms at two per replica -4 %; 2048 atoms at two per replica -9 %.  Where it loses: the pair loop of the larger cells is
    // bound by the LDS pipe already (64 unrelated atoms per gather instruction), three atomics per entry make 2.4 times the LDS work of an
    // entry, only the own range's pairs go away (75 % of the entries at two workgroups per replica), and a cluster loses the epilogue that
    // integrates and publishes a row's atom as soon as its force is known.  At ONE workgroup per replica it pays at 2048 atoms too (256 replicas:
    // +10 %).  So HALF_ is set for the one-workgroup configurations of the lists in HBM only (CfgMidH, CfgLargeH, nm_api.hip); -DNM_HALF_LIST=2 builds every one-thread-per-row configuration with half lists for the A/B.
    static constexpr bool HALF = !LIST_LDS_ && TPA_ == 1 && POT_ == 0 && (NM_HALF_LIST == 2 || (NM_HALF_LIST == 1 && HALF_));
    static constexpr int BLOCK = BLOCK_, TPA = TPA_, NW = BLOCK_ / 64, G = BLOCK_ / TPA_, NMAX = NMAX_, MAXNB = MAXNB_;
    static constexpr bool LIST_LDS = LIST_LDS_, SAVE_LDS = SAVE_LDS_;
    using IdxT = IdxT_;
    static constexpr size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }
    static constexpr size_t A3 = (size_t)3 * NMAX * sizeof(double);
    static constexpr size_t OFF_POS = 0, OFF_VEL = A3, OFF_FRC = 2 * A3;
    static constexpr size_t OFF_SAV = 3 * A3, OFF_SAVV = 4 * A3, OFF_X0 = (SAVEV_LDS ? 5 : 4) * A3; // only when SAVE_LDS (/ SAVEV_LDS)
    // saved forces: a rejected move then also gets its forces back, and a trajectory that follows it starts without
    // re-evaluating the restored configuration.  In LDS for the small systems (6 KB), in the global spill otherwise.
    static constexpr bool SAVEF_LDS = SAVE_LDS && NMAX <= 256;
    static constexpr size_t OFF_SAVF = 6 * A3;
    static_assert(!SAVEF_LDS || SAVEV_LDS, "");
    static constexpr size_t OFF_RED = (3 + (SAVE_LDS ? 2 : 0) + (SAVEV_LDS ? 1 : 0) + (SAVEF_LDS ? 1 : 0)) * A3;
    // per-wave copies of block-uniform scalars that are touched once per move (counters, the move's saved energies, the slot's
    // constants): kept in LDS instead of ~40 scalar registers that the hot loops would otherwise spill and reload
    static constexpr int UST_PER_WAVE = 32;
    static constexpr size_t OFF_UST = OFF_RED + (size_t)2 * NW * NVMAX * sizeof(double);
    static constexpr size_t OFF_CNT = OFF_UST + (size_t)NW * UST_PER_WAVE * sizeof(double);
    // image flags (int16) and the iter-PMC wrap counts (int8): in LDS with the saved copies, else in the global spill
    static constexpr size_t OFF_IMG = OFF_CNT + pad8((size_t)NMAX * sizeof(unsigned short));
    static constexpr size_t OFF_WN = OFF_IMG + pad8((size_t)3 * NMAX * sizeof(short));
    static constexpr size_t OFF_NBR = SAVE_LDS ? OFF_WN + pad8((size_t)3 * NMAX) : OFF_IMG;
    static constexpr size_t LIST_BYTES = LIST_LDS ? pad8((size_t)MAXNB * NLIST * sizeof(IdxT)) : 0; // one LDS list
    static constexpr size_t OFF_RHO = OFF_NBR + (LDS_LIST2 ? 2 : 1) * LIST_BYTES; // EAM densities
    static constexpr size_t OFF_X0S = OFF_RHO + (POT ? (size_t)NMAX * sizeof(double) : 0); // LDS_LIST2: reference positions of the saved list
    static constexpr size_t OFF_CNTS = OFF_X0S + (LDS_LIST2 ? A3 : 0);                     //            and its row lengths
    // PREFETCH: the gaussians of the NEXT move's `velocity create` are drawn while the closing exchange of the current move's
    // energy sums is in flight (Replica::prefetch_gaussians) and wait here, 3 NMAX doubles
    // (measured 1.4 % SLOWER on C2 and C4, round 3: the 400 instructions per thread delay the poll more than the flight time they fill,
    //  and the kernel spills 75 instead of 52 VGPRs.  Kept behind NM_AB == 6 for the record; off in the shipped build.)
    static constexpr bool PREFETCH = (NM_AB == 6) && NMAX_ <= 256;
    static constexpr size_t OFF_GAUSS = OFF_CNTS + (LDS_LIST2 ? pad8((size_t)NMAX * sizeof(unsigned short)) : 0);
    static constexpr size_t LDS_NATURAL = OFF_GAUSS + (PREFETCH ? A3 : 0);
    // One workgroup per CU, by construction: a cluster's census and the Q selection count on it (nm_probe_kernel asserts it), and
    // two workgroups of different replicas on one CU measured slower.  A configuration that would fit twice
    // into the CU's 160 KB asks for a little more than half of them.
#if NM_AB == 4 // experiment: 256-thread workgroups (-DNM_SMALL_BLOCK=256 -DNM_SMALL_TPA=1), 8 per replica, two (of different replicas) per CU
    static constexpr size_t LDS_BYTES = (BLOCK_ == 256 || LDS_NATURAL > (size_t)82 * 1024) ? LDS_NATURAL : (size_t)82 * 1024;
#else
    static constexpr size_t LDS_BYTES = LDS_NATURAL > (size_t)82 * 1024 ? LDS_NATURAL : (size_t)82 * 1024;
#endif
    // per-slot global spill when the saved copies do not fit in LDS: sav, savv, x0 (9 NMAX doubles) + images + wrap counts
    static constexpr size_t AUX_SAVES = SAVE_LDS ? (SAVEV_LDS ? 0 : (size_t)3 * NMAX) : (size_t)9 * NMAX + ((size_t)3 * NMAX * 3 + 7) / 8;
    // lists outside LDS are kept TWICE per slot (LIST2): a trial that rebuilt and is then rejected goes back to the list it started
    // from instead of rebuilding again (Replica::save / rebuild / restore); with it the reference positions and the row lengths
    // of that list (3 NMAX doubles + NMAX 16-bit counts per workgroup)
    static constexpr bool LIST2 = !LIST_LDS_ || LDS_LIST2_;
    static constexpr size_t AUX_LIST2 = !LIST_LDS_ ? (size_t)3 * NMAX + ((size_t)NMAX + 3) / 4 : 0;
    static constexpr size_t AUX_DOUBLES = AUX_SAVES + (SAVEF_LDS ? 0 : (size_t)3 * NMAX) + AUX_LIST2; // ... + the saved forces + LIST2
    static constexpr size_t NBR_G_ELEMS = LIST_LDS ? 0 : (size_t)MAXNB * NMAX; // per-slot global list
    // Lists that live in HBM/L2 are stored in chunks of CH consecutive neighbours of one atom ([chunk][atom][CH]) so that one
    // 8-byte load brings four indices: the dependent L2 round trip per neighbour was the cost there.  LDS lists stay [slot][atom].
    static constexpr int CH = LIST_LDS ? 1 : 4;
    static_assert(MAXNB % 4 == 0, "MAXNB must be a multiple of the chunk size");
    // LDS byte lists are stored so that the TPA threads of an atom each fetch EIGHT of their neighbours with one conflict-free
    // ds_read_b64: neighbour slot r of atom i belongs to thread sub = r % TPA as its k-th neighbour (k = r / TPA) and sits at byte
    // (((k / 8) * NMAX + i) * TPA + sub) * 8 + k % 8.  (One ds_read_u8 per neighbour was an 8-way bank conflict: the 8 threads of
    // an atom read rows 256 B apart; with the three position gathers it made the pair loop LDS-bound, scripts/ubench_pair.hip.)
    // 16-bit indices (N > 256) use the same layout with four entries per 8-byte word.
    static constexpr int PW = 8 / (int)sizeof(IdxT); // list entries per 8-byte word
    static constexpr int LOG2PW = sizeof(IdxT) == 1 ? 3 : 2;
    static_assert(!LIST_LDS || ((sizeof(IdxT) == 1 || sizeof(IdxT) == 2) && MAXNB % (PW * TPA) == 0), "LDS lists: MAXNB a multiple of PW*TPA");
    static constexpr int QMAX = 8;                                               // most workgroups per replica
    static constexpr size_t XBUF_GRANULES = (size_t)4 * NMAX + 4 * QMAX;         // forces by component, EAM densities, per-workgroup partials
    static constexpr size_t XG_PART = (size_t)3 * NMAX, XG_RHO = (size_t)3 * NMAX + 4 * QMAX; // granule indices
    static constexpr size_t XBUF_DOUBLES = 2 * XBUF_GRANULES;                    // one exchange buffer (there are two per slot)
};

// velocity all create / zero linear / zero angular (see Replica::hmc_velocities' comment below for the algebra).  A function of its
// own, NOT inlined: it runs once per HMC move, and inlined its registers (sixteen running moments, Philox, log / sin / cos) were
// live-range neighbours of everything the trajectory loop keeps, which the allocator then spilled into that loop.  All state it
// touches is in LDS at constant offsets; scalars come and go by value.  Makes ONE block reduction (the caller flips its buffer
// parity) and returns sum m |v|^2 of the velocities it leaves.
// the gaussians of `velocity all create ... dist gaussian`, divided by sqrt(m): 2N work items over all threads — item w < N draws
// (vx, vy) of atom w, item N + w draws vz — written to three arrays of NMAX doubles at byte offset `off` of the LDS block (the
// velocities, or the prefetch area).  A function of its own (Philox, log, sin, cos: ~400 instructions), called from two places.
template <class C>
__device__ __attribute__((noinline)) void gaussian_fill(uint32_t tag, int N, int gslot, double mass, uint32_t seed, uint32_t step, int off)
{
    constexpr int BLOCK = C::BLOCK;
    const int tid = threadIdx.x;
    N = __builtin_amdgcn_readfirstlane(N); gslot = __builtin_amdgcn_readfirstlane(gslot); off = __builtin_amdgcn_readfirstlane(off);
    tag = __builtin_amdgcn_readfirstlane(tag); seed = __builtin_amdgcn_readfirstlane(seed); step = __builtin_amdgcn_readfirstlane(step);
    mass = uniform(mass);
    double *const gx = (double *)(nm_lds + off), *const gy = gx + C::NMAX, *const gz = gy + C::NMAX;
    const double twopi = 6.283185307179586476925286766559;
    const double fac = 1.0 / sqrt(mass);
    for (int w = tid; w < 2 * N; w += BLOCK) {
        const int part = w >= N ? 1 : 0, i = w - part * N;
        uint32_t o[4];
        philox4x32_10((uint32_t)i, part ? S_VEL_B : S_VEL_A, tag, step, seed, (uint32_t)gslot, o);
        const double u1 = u01(o[0], o[1]), u2 = u01(o[2], o[3]);
        const double r = sqrt(-2.0 * log_pos(1.0 - u1)); // (nm_math.h: the same functions of the same arguments, within 1 ulp like the library's)
        double sn, cs;
        sincos_2pi(twopi * u2, sn, cs);
        if (part) gz[i] = r * cs * fac;
        else { gx[i] = r * cs * fac; gy[i] = r * sn * fac; }
    }
}

template <class C>
__device__ __attribute__((noinline)) double velocity_create(double t, uint32_t tag, double L, int N, int gslot, int parity, double mass,
                                                         double mvv2e, double kB, uint32_t seed, uint32_t step, short *im_g, int prefetched)
{
    constexpr int BLOCK = C::BLOCK, NW = C::NW;
    constexpr size_t A1 = (size_t)C::NMAX * sizeof(double);
    const int tid = threadIdx.x;
    LdsArr<double, C::OFF_POS> px; LdsArr<double, C::OFF_POS + A1> py; LdsArr<double, C::OFF_POS + 2 * A1> pz;
    LdsArr<double, C::OFF_VEL> vx; LdsArr<double, C::OFF_VEL + A1> vy; LdsArr<double, C::OFF_VEL + 2 * A1> vz;
    LdsArr<double, C::OFF_RED> red;
    typename ArrSel<C::SAVE_LDS, short, C::OFF_IMG>::type im; // LAMMPS image flags: LDS, or the per-workgroup global spill
    if constexpr (!C::SAVE_LDS) {
        const unsigned long long b = (unsigned long long)(uintptr_t)im_g;
        im.g = (short *)(uintptr_t)(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b)); // (the builtin returns int:
        // without the uint32_t cast a low word with bit 31 set sign-extends over the high word — a wild pointer)
        NM_BOUND(im, 3 * C::NMAX);
    }
    N = __builtin_amdgcn_readfirstlane(N); gslot = __builtin_amdgcn_readfirstlane(gslot); parity = __builtin_amdgcn_readfirstlane(parity);
    tag = __builtin_amdgcn_readfirstlane(tag); seed = __builtin_amdgcn_readfirstlane(seed); step = __builtin_amdgcn_readfirstlane(step);
    t = uniform(t); L = uniform(L); mass = uniform(mass); mvv2e = uniform(mvv2e); kB = uniform(kB);
    const double m = mass, mt = m * N;
    prefetched = __builtin_amdgcn_readfirstlane(prefetched);
    if (C::PREFETCH && prefetched) { // drawn while the previous move's energy sums were crossing the cluster (prefetch_gaussians)
        const double *const g = (const double *)(nm_lds + C::OFF_GAUSS);
        for (int i = tid; i < N; i += BLOCK) { vx[i] = g[i]; vy[i] = g[C::NMAX + i]; vz[i] = g[2 * C::NMAX + i]; }
    } else gaussian_fill<C>(tag, N, gslot, mass, seed, step, (int)C::OFF_VEL);
    __syncthreads();
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = 0.0;
    for (int i = tid; i < N; i += BLOCK) {
        const double ax = vx[i], ay = vy[i], az = vz[i];
        const double X = px[i] + im[3 * i] * L, Y = py[i] + im[3 * i + 1] * L, Z = pz[i] + im[3 * i + 2] * L;
        a[0] += m * ax; a[1] += m * ay; a[2] += m * az;
        a[3] += m * X; a[4] += m * Y; a[5] += m * Z;
        a[6] += m * (ax * ax + ay * ay + az * az);
        a[7] += m * (Y * az - Z * ay);
        a[8] += m * (Z * ax - X * az);
        a[9] += m * (X * ay - Y * ax);
        a[10] += m * (Y * Y + Z * Z);
        a[11] += m * (X * X + Z * Z);
        a[12] += m * (X * X + Y * Y);
        a[13] -= m * X * Y;
        a[14] -= m * Y * Z;
        a[15] -= m * X * Z;
    }
#if NM_AB == 1
    block_sum<16, NW, NVMAX>(a, red, parity);
#else
    block_sum<16, NW, NVMAX>(a, red, parity, min(NW, (N + 63) >> 6)); // (atoms are dealt out by thread index: N <= 256 leaves half the waves empty)
#endif
    const double imt = 1.0 / mt; // (one division instead of six, and one for the inertia tensor below instead of three: every thread does all of this)
    const double c0 = a[0] * imt, c1 = a[1] * imt, c2 = a[2] * imt;   // COM velocity
    const double cx = a[3] * imt, cy = a[4] * imt, cz = a[5] * imt;   // centre of mass (unwrapped)
    const double dof = 3.0 * N - 3.0;
    const double s2 = a[6] - mt * (c0 * c0 + c1 * c1 + c2 * c2);
    const double tcur = s2 * mvv2e / (dof * kB);
    const double sc = sqrt(t / tcur);
    const double L0_ = sc * (a[7] - mt * (cy * c2 - cz * c1));
    const double L1_ = sc * (a[8] - mt * (cz * c0 - cx * c2));
    const double L2_ = sc * (a[9] - mt * (cx * c1 - cy * c0));
    const double I00 = a[10] - mt * (cy * cy + cz * cz), I11 = a[11] - mt * (cx * cx + cz * cz), I22 = a[12] - mt * (cx * cx + cy * cy);
    const double I01 = a[13] + mt * cx * cy, I12 = a[14] + mt * cy * cz, I02 = a[15] + mt * cx * cz;
    const double det = I00 * I11 * I22 + I01 * I12 * I02 + I02 * I01 * I12 - I00 * I12 * I12 - I01 * I01 * I22 - I02 * I11 * I02;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0;
    if (det > 0.0) {
        const double i00 = I11 * I22 - I12 * I12, i01 = -(I01 * I22 - I02 * I12), i02 = I01 * I12 - I02 * I11;
        const double i10 = -(I01 * I22 - I12 * I02), i11 = I00 * I22 - I02 * I02, i12 = -(I00 * I12 - I02 * I01);
        const double i20 = I01 * I12 - I11 * I02, i21 = -(I00 * I12 - I01 * I02), i22 = I00 * I11 - I01 * I01;
        const double idet = 1.0 / det;
        w0 = (i00 * L0_ + i01 * L1_ + i02 * L2_) * idet;
        w1 = (i10 * L0_ + i11 * L1_ + i12 * L2_) * idet;
        w2 = (i20 * L0_ + i21 * L1_ + i22 * L2_) * idet;
    }
    for (int i = tid; i < N; i += BLOCK) {
        const double dx = px[i] + im[3 * i] * L - cx, dy = py[i] + im[3 * i + 1] * L - cy, dz = pz[i] + im[3 * i + 2] * L - cz;
        vx[i] = (vx[i] - c0) * sc - (w1 * dz - w2 * dy);
        vy[i] = (vy[i] - c1) * sc - (w2 * dx - w0 * dz);
        vz[i] = (vz[i] - c2) * sc - (w0 * dy - w1 * dx);
    }
    // sum m |v'|^2 of what was just written, from the same moments: with u = sc (v - c), r = X - Xc and I omega = L,
    // sum m |u - omega x r|^2 = sc^2 sum m |v - c|^2 - 2 omega.L + omega.I.omega = sc^2 s2 - omega.L   (the caller's H0 needs no pass of its own)
    return sc * sc * s2 - (w0 * L0_ + w1 * L1_ + w2 * L2_);
}

// Loop over the atoms this workgroup owns (a0 <= i < a1): atom a0 + t on thread t, i.e. on the workgroup's OLDEST waves, which leave the
// pair loop first (walking them as "atom i on thread i mod BLOCK", the mapping of the other elementwise phases, measured 1.5 % slower at
// 4^3: the first kick, the publication of the new positions and the fetching of the peers' then start on whatever wave owns the range).
// Two mappings need ordering wherever an own atom changes hands between them; the three places are marked "(mapping)": round 2 had
// none of them ordered — seen once, in round 3, as a stale velocity in a block's closing kinetic-energy sum.
#define NM_FOR_OWN(i) for (int i = a0 + tid; i < a1; i += BLOCK)

template <class C>
struct Replica {
    using IdxT = typename C::IdxT;
    static constexpr int BLOCK = C::BLOCK, TPA = C::TPA, NW = C::NW, G = C::G, NMAX = C::NMAX, MAXNB = C::MAXNB;

    const KParams &p;
    const int tid, N, gslot;
    // cluster: Q workgroups (one per CU) run the same replica redundantly and split only the pair work by atom range
    const int Q, q, a0, a1;
    double *xb;
    int gen = 0;
    unsigned long long *tl = nullptr; // experiment build
    int tl_n = 0;
    static constexpr size_t A1 = (size_t)C::NMAX * sizeof(double);
    LdsArr<double, C::OFF_POS> px; LdsArr<double, C::OFF_POS + A1> py; LdsArr<double, C::OFF_POS + 2 * A1> pz;
    LdsArr<double, C::OFF_VEL> vx; LdsArr<double, C::OFF_VEL + A1> vy; LdsArr<double, C::OFF_VEL + 2 * A1> vz;
    LdsArr<double, C::OFF_FRC> fx; LdsArr<double, C::OFF_FRC + A1> fy; LdsArr<double, C::OFF_FRC + 2 * A1> fz;
    // saved positions / velocities, list reference positions, image flags, wrap counts: LDS or the per-workgroup global spill
    typename ArrSel<C::SAVE_LDS, double, C::OFF_SAV>::type sx; typename ArrSel<C::SAVE_LDS, double, C::OFF_SAV + A1>::type sy;
    typename ArrSel<C::SAVE_LDS, double, C::OFF_SAV + 2 * A1>::type sz;
    typename ArrSel<C::SAVEV_LDS, double, C::OFF_SAVV>::type svx; typename ArrSel<C::SAVEV_LDS, double, C::OFF_SAVV + A1>::type svy;
    typename ArrSel<C::SAVEV_LDS, double, C::OFF_SAVV + 2 * A1>::type svz;
    typename ArrSel<C::SAVEF_LDS, double, C::OFF_SAVF>::type sfx; typename ArrSel<C::SAVEF_LDS, double, C::OFF_SAVF + A1>::type sfy;
    typename ArrSel<C::SAVEF_LDS, double, C::OFF_SAVF + 2 * A1>::type sfz; // saved forces of the own atoms
    typename ArrSel<C::SAVE_LDS, double, C::OFF_X0>::type x0; typename ArrSel<C::SAVE_LDS, double, C::OFF_X0 + A1>::type y0;
    typename ArrSel<C::SAVE_LDS, double, C::OFF_X0 + 2 * A1>::type z0;
    typename ArrSel<C::SAVE_LDS, short, C::OFF_IMG>::type im;      // LAMMPS image flags
    typename ArrSel<C::SAVE_LDS, signed char, C::OFF_WN>::type wn; // wrap counts of the coordinates gathered at the start of an iter-PMC move
    LdsArr<unsigned short, C::OFF_CNT> cnt;
    typename ArrSel<C::LIST_LDS, IdxT, C::OFF_NBR>::type nbr;
    LdsArr<double, C::OFF_RED> red;
    LdsArr<double, C::OFF_RHO> rho; // EAM: densities, then 1/sqrt(density)
    int parity = 0;
    // block-uniform scalars
    double L = 0.0, L0 = 0.0, U = 0.0, W = 0.0;
    // derived from (L, L0), refreshed by box_consts() only when one of them changed (three fp64 divisions otherwise sit on the
    // critical path between two pair loops of an HMC trajectory): 1/L, L/L0 and the squared list-validity bound
    double bc_L = -1.0, bc_L0 = -1.0, bc_invL = 0.0, bc_sc = 0.0, bc_thr2 = 0.0;
    bool bc_bad = true;
    double psum[4] = { 0.0, 0.0, 0.0, 0.0 }; // partial (then cluster-wide) sums of the last energy evaluation: 2U, 2W, 2 pairs; sum m v.v of the
                                             // own atoms when the evaluation made the last half kick of a trajectory
    // slot k of this wave's copy of the rarely-touched uniform scalars (every lane of the wave reads / writes the same value)
    __device__ __forceinline__ double &ust(int k) const { return ((double *)(nm_lds + C::OFF_UST))[(tid >> 6) * C::UST_PER_WAVE + k]; }
    // block-uniform flags in ONE scalar register (three separate bools cost lane masks and spilled scalars in the hot loops)
    enum : int { F_LIST_OK = 1, F_FRESH = 2, F_SAVED_FRESH = 4, F_LIST_SAVED = 8, F_REBUILT = 16 };
    int flags = 0;
    __device__ __forceinline__ bool fresh() const { return (flags & F_FRESH) != 0; }
    __device__ __forceinline__ void set_fresh(bool b) { flags = b ? (flags | F_FRESH) : (flags & ~F_FRESH); }
    int status = 0;
    // LIST2: the other list of the slot, and what belongs to the list a move started from
    typename ArrSel<C::LDS_LIST2, double, C::OFF_X0S>::type x0s; typename ArrSel<C::LDS_LIST2, double, C::OFF_X0S + A1>::type y0s;
    typename ArrSel<C::LDS_LIST2, double, C::OFF_X0S + 2 * A1>::type z0s;
    typename ArrSel<C::LDS_LIST2, unsigned short, C::OFF_CNTS>::type cnts;
    double L0s = 0.0;
    int list_cur = 0;
    // the list in force: LDS lists kept twice sit one behind the other
    __device__ __forceinline__ IdxT *nbr_cur() const
    {
        if constexpr (C::LDS_LIST2) return (IdxT *)(nm_lds + C::OFF_NBR + (size_t)list_cur * C::LIST_BYTES);
        else return nbr.ptr();
    }
    bool same_xcd = false; // all workgroups of the cluster run on one XCD (read from the hardware, not assumed from blockIdx)
    const double *tape = nullptr;
    int tpos = 0, tlen = 0;
    double st_evals = 0.0, st_rebuilds = 0.0, st_eevals = 0.0, st_pairs = 0.0;
    int st_maxc = 0; // longest list row this thread has built (stats column 8)
    PROF_DECL

    __device__ Replica(const KParams &p_, int slot, int q_)
        : p(p_), tid(threadIdx.x), N(p_.N), gslot(p_.slot0 + slot), Q(p_.cus), q(q_), a0((p_.N * q_) / p_.cus),
          a1((p_.N * (q_ + 1)) / p_.cus)
    {
        xb = p.xbuf ? p.xbuf + (size_t)slot * 2 * C::XBUF_DOUBLES : nullptr;
#ifdef NM_EXPERIMENT
        if (slot == 0) tl = p.tline;
#endif
        if constexpr (!C::SAVE_LDS) {
            double *a = p.aux_g + ((size_t)slot * p.cus + q_) * C::AUX_DOUBLES; // one spill area per workgroup
            sx.g = a; sy.g = a + NMAX; sz.g = a + 2 * (size_t)NMAX;
            svx.g = a + 3 * (size_t)NMAX; svy.g = a + 4 * (size_t)NMAX; svz.g = a + 5 * (size_t)NMAX;
            x0.g = a + 6 * (size_t)NMAX; y0.g = a + 7 * (size_t)NMAX; z0.g = a + 8 * (size_t)NMAX;
            im.g = (short *)(a + 9 * (size_t)NMAX);
            wn.g = (signed char *)(im.g + 3 * (size_t)NMAX);
            NM_BOUND(sx, NMAX); NM_BOUND(sy, NMAX); NM_BOUND(sz, NMAX); NM_BOUND(svx, NMAX); NM_BOUND(svy, NMAX); NM_BOUND(svz, NMAX);
            NM_BOUND(x0, NMAX); NM_BOUND(y0, NMAX); NM_BOUND(z0, NMAX); NM_BOUND(im, 3 * NMAX); NM_BOUND(wn, 3 * NMAX);
        }
        else if constexpr (!C::SAVEV_LDS) {
            double *a = p.aux_g + ((size_t)slot * p.cus + q_) * C::AUX_DOUBLES;
            svx.g = a; svy.g = a + NMAX; svz.g = a + 2 * (size_t)NMAX;
            NM_BOUND(svx, NMAX); NM_BOUND(svy, NMAX); NM_BOUND(svz, NMAX);
        }
        if constexpr (!C::SAVEF_LDS) {
            double *a = p.aux_g + ((size_t)slot * p.cus + q_) * C::AUX_DOUBLES + C::AUX_SAVES;
            sfx.g = a; sfy.g = a + NMAX; sfz.g = a + 2 * (size_t)NMAX;
            NM_BOUND(sfx, NMAX); NM_BOUND(sfy, NMAX); NM_BOUND(sfz, NMAX);
        }
        if constexpr (!C::LIST_LDS) {
            nbr.g = (IdxT *)p.nbr_g + (size_t)slot * 2 * C::NBR_G_ELEMS; // two lists per slot, list_cur = 0
            double *a = p.aux_g + ((size_t)slot * p.cus + q_) * C::AUX_DOUBLES + (C::AUX_DOUBLES - C::AUX_LIST2);
            x0s.g = a; y0s.g = a + NMAX; z0s.g = a + 2 * (size_t)NMAX;
            cnts.g = (unsigned short *)(a + 3 * (size_t)NMAX);
            NM_BOUND(x0s, NMAX); NM_BOUND(y0s, NMAX); NM_BOUND(z0s, NMAX); NM_BOUND(cnts, NMAX); NM_BOUND(nbr, C::NBR_G_ELEMS);
        }
        if (p.tape) { tape = p.tape + p.tape_off[slot]; tlen = p.tape_off[slot + 1] - p.tape_off[slot]; }
    }

    // ------------------------------------------------------------------ random draws
    __device__ __forceinline__ double draw_scalar(uint32_t stream, uint32_t m, uint32_t index)
    {
        if (tape) { // test-only: uniforms recorded from the reference's np.random stream
            double v = 2.0;
            if (tpos < tlen) v = tape[tpos]; else status |= ST_TAPE_EXHAUSTED;
            ++tpos;
            return v;
        }
        uint32_t o[4];
        philox4x32_10(index, stream, m, p.step, p.seed, (uint32_t)gslot, o);
        return u01(o[0], o[1]);
    }
    __device__ __forceinline__ uint32_t draw_tag(uint32_t m)
    {
        if (tape) { // np.random.randint(1, 2**16), remcmc:482,603
            uint32_t v = 0;
            if (tpos < tlen) v = (uint32_t)(tape[tpos] * 65536.0); else status |= ST_TAPE_EXHAUSTED; // randint/65536
            ++tpos;
            return v;
        }
        return m;
    }
    // remcmc:487-500: metcrit = exp(-c); +inf -> reject without drawing; else accept iff u <= min(1, metcrit)
    __device__ __forceinline__ bool metropolis(double c, uint32_t stream, uint32_t m, uint32_t index)
    {
        const double metcrit = exp(-c);
        if (isinf(metcrit)) return false;
        const double u = draw_scalar(stream, m, index);
        const double mm = (metcrit != metcrit) ? metcrit : (metcrit < 1.0 ? metcrit : 1.0);
        return u <= mm;
    }

    // ------------------------------------------------------------------ HBM <-> LDS
    __device__ void load(int buf)
    {
        const double *gx = p.x + (size_t)buf * 3 * N, *gv = p.v + (size_t)buf * 3 * N;
        for (int a = tid; a < 3 * N; a += BLOCK) { // coalesced interleaved read, de-interleave into SoA
            const int i = a / 3, c = a - 3 * i;
            (c == 0 ? px : c == 1 ? py : pz)[i] = gx[a];
            (c == 0 ? vx : c == 1 ? vy : vz)[i] = gv[a];
            im[a] = 0; // a new LAMMPS instance per block starts with zero image flags (remcmc:462-463)
        }
        __syncthreads();
    }
    // every workgroup of a cluster holds all positions; velocities are current only for the atoms it integrates
    __device__ void store(int buf)
    {
        __syncthreads();
        double *gx = p.x + (size_t)buf * 3 * N, *gv = p.v + (size_t)buf * 3 * N;
        if (q == 0)
            for (int a = tid; a < 3 * N; a += BLOCK) {
                const int i = a / 3, c = a - 3 * i;
                gx[a] = (c == 0 ? px : c == 1 ? py : pz)[i];
            }
        for (int a = 3 * a0 + tid; a < 3 * a1; a += BLOCK) {
            const int i = a / 3, c = a - 3 * i;
            gv[a] = (c == 0 ? vx : c == 1 ? vy : vz)[i];
        }
    }

    // ------------------------------------------------------------------ per-atom elementwise phases
    // Ownership: atom i belongs to thread i % BLOCK in every elementwise phase, so these need no barrier
    // among themselves; eval() opens with one.
    __device__ __forceinline__ int wrap1(double &x) // domain->remap of one coordinate; returns the image increment
    {
        int d = 0;
        if (x < 0.0 || x >= L) {
            const double nb = floor(x / L);
            x -= nb * L;
            d = (int)nb;
            if (x >= L) { x -= L; d += 1; }
            if (x < 0.0) x = 0.0;
        }
        return d;
    }
    __device__ void wrap()
    {
        for (int i = tid; i < N; i += BLOCK) {
            im[3 * i] = (short)(im[3 * i] + wrap1(px[i]));
            im[3 * i + 1] = (short)(im[3 * i + 1] + wrap1(py[i]));
            im[3 * i + 2] = (short)(im[3 * i + 2] + wrap1(pz[i]));
        }
    }
    __device__ void save(bool with_v)
    {
        if (with_v && Q > 1) {
            // (mapping) a Hamiltonian move: the first kick and drift that follow overwrite x, v of the own atoms on thread i - a0, so
            // that thread saves them, and the caller's barrier after wrap() has made them current for it; the other atoms as usual
            for (int i = tid; i < N; i += BLOCK)
                if (i < a0 || i >= a1) { sx[i] = px[i]; sy[i] = py[i]; sz[i] = pz[i]; svx[i] = vx[i]; svy[i] = vy[i]; svz[i] = vz[i]; }
            NM_FOR_OWN(i) { sx[i] = px[i]; sy[i] = py[i]; sz[i] = pz[i]; svx[i] = vx[i]; svy[i] = vy[i]; svz[i] = vz[i]; }
        } else
        for (int i = tid; i < N; i += BLOCK) {
            sx[i] = px[i]; sy[i] = py[i]; sz[i] = pz[i];
            if (with_v) { svx[i] = vx[i]; svy[i] = vy[i]; svz[i] = vz[i]; }
        }
        // LIST2: the list in force now is the one to come back to if this move rebuilds and is then rejected
        flags &= ~(F_LIST_SAVED | F_REBUILT);
        if (C::LIST2 && (flags & F_LIST_OK)) flags |= F_LIST_SAVED;
        // the forces of the own atoms, if they belong to these positions
        flags = fresh() ? (flags | F_SAVED_FRESH) : (flags & ~F_SAVED_FRESH);
        if (fresh())
            NM_FOR_OWN(i) { sfx[i] = fx[i]; sfy[i] = fy[i]; sfz[i] = fz[i]; }
    }
    // The reference answers a rejection with scatter_atoms + `run 0`, i.e. it re-evaluates the old configuration.  The result
    // is what was there before the move (the caller puts U, W back); where the forces were saved too they come back as well and
    // the configuration counts as evaluated: a trajectory that follows starts from them.
    __device__ void restore(bool with_v)
    {
        for (int i = tid; i < N; i += BLOCK) {
            px[i] = sx[i]; py[i] = sy[i]; pz[i] = sz[i]; // scatter_atoms: x only, LAMMPS image flags keep what the remaps did
            if (with_v) { vx[i] = svx[i]; vy[i] = svy[i]; vz[i] = svz[i]; }
        }
        set_fresh(false);
        if (flags & F_SAVED_FRESH) {
            NM_FOR_OWN(i) { fx[i] = sfx[i]; fy[i] = sfy[i]; fz[i] = sfz[i]; }
            set_fresh(true);
        }
        if constexpr (C::LIST2) {
            // the move rebuilt the list on the way and is rejected: the positions are back where the previous list was built
            // for, and that list is still there (the rebuild went to the slot's other buffer).  Its validity is tested by the
            // next evaluation like that of any list.
            if ((flags & F_LIST_SAVED) && (flags & F_REBUILT)) {
                list_cur ^= 1;
                if constexpr (!C::LIST_LDS) nbr.g = (IdxT *)p.nbr_g + ((size_t)(gslot - p.slot0) * 2 + list_cur) * C::NBR_G_ELEMS;
                for (int i = tid; i < N; i += BLOCK) { x0[i] = x0s[i]; y0[i] = y0s[i]; z0[i] = z0s[i]; }
                NM_FOR_OWN(i) cnt[i] = cnts[i];
                L0 = L0s;
                flags |= F_LIST_OK;
            }
            flags &= ~(F_LIST_SAVED | F_REBUILT);
        }
    }
    __device__ double sum_mv2()
    {
        double s[1] = { 0.0 };
        for (int i = tid; i < N; i += BLOCK) s[0] += p.mass * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
        block_sum<1, NW, NVMAX>(s, red, parity);
        return s[0];
    }

    // sum over the atoms this workgroup integrates (its velocities of the other atoms go stale during a trajectory)
    __device__ double own_mv2()
    {
        double s[1] = { 0.0 };
        NM_FOR_OWN(i) s[0] += p.mass * (vx[i] * vx[i] + vy[i] * vy[i] + vz[i] * vz[i]);
        block_sum<1, NW, NVMAX>(s, red, parity);
        return s[0];
    }

    // ------------------------------------------------------------------ Verlet list
    // Wave-cooperative build: one wave per atom i, 64 candidate j per step, ballot compaction keeps the
    // list sorted by j, so the list (and every sum over it) is independent of scheduling.
    // row of atom i in an LDS list
    __device__ __forceinline__ int lrow(int i) const { return C::NLIST == NMAX ? i : i - a0; }
    // element index of neighbour slot r of atom i in the list
    __device__ __forceinline__ size_t nbr_at(int r, int i) const
    {
        if constexpr (C::LIST_LDS) { const int sub = r % TPA, k = r / TPA; return ((((size_t)(k >> C::LOG2PW) * C::NLIST + lrow(i)) * TPA + sub) << C::LOG2PW) + (k & (C::PW - 1)); }
        else return ((size_t)(r / C::CH) * NMAX + i) * C::CH + (r % C::CH);
    }

    // One candidate test of the list rebuild on 16-bit fixed-point coordinates (units of L / 65536): xy holds x | y << 16, z the
    // third coordinate.  v_pk_sub_i16 wraps modulo 2^16, which IS the minimum image; three v_mad_i32_i16 (the second on the high
    // halves) give the squared separation (at most 3 x 2^30: no wrap as an unsigned 32-bit number); the compare's carry is shifted
    // into the mask by v_addc (m + m + carry): seven VALU instructions per test (scripts/ubench_int16.hip: 36 cycles per wave, 48
    // with two v_dot2c_i32_i16 and the wait states a dot product's result needs before another instruction may read it — which
    // hipcc does not insert for a reader inside an asm statement, so the dot form is not used).  The result of the b-th test of a
    // run ends up in bit n - 1 - b (the callers reverse).
    typedef short v2s16 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ unsigned int test16(unsigned int m, unsigned int xyi, unsigned int zi, unsigned int xyj, unsigned int zj,
                                                          unsigned int t2)
    {
        const unsigned int d0 = __builtin_bit_cast(unsigned int, (v2s16)(__builtin_bit_cast(v2s16, xyi) - __builtin_bit_cast(v2s16, xyj)));
        const unsigned int d1 = __builtin_bit_cast(unsigned int, (v2s16)(__builtin_bit_cast(v2s16, zi) - __builtin_bit_cast(v2s16, zj)));
        unsigned int r;
        asm("v_mad_i32_i16 %1, %2, %2, 0\n\t"
            "v_mad_i32_i16 %1, %2, %2, %1 op_sel:[1,1,0,0]\n\t"
            "v_mad_i32_i16 %1, %3, %3, %1\n\t"
            "v_cmp_gt_u32 vcc, %4, %1\n\t"
            "v_addc_co_u32 %0, vcc, %0, %0, vcc"
            : "+v"(m), "=&v"(r) : "v"(d0), "v"(d1), "v"(t2) : "vcc");
        return m;
    }

    // rows ibase + tid (and ibase + BLOCK + tid) of a list that lives in HBM/L2: 64 candidates per block.  Each lane fetches ONE of
    // them (a coalesced read), and the block is then walked with v_readlane: a candidate's coordinates reach all lanes as scalar
    // operands, no LDS access and no load latency per candidate.  The tests are branch-free, one bit each: appending inside this
    // loop would put a divergent branch behind every test (some lane of the wave is in range of nearly every candidate; measured 5x
    // the arithmetic).  A thread packs four indices into the 8-byte chunk the pair loop reads and stores it whole.
    template <int NR>
    __device__ __forceinline__ void rebuild_rows_hbm(int ibase, const unsigned long long *cf, unsigned int t2, int &ovf)
    {
        const int lane = tid & 63;
        unsigned long long *g = (unsigned long long *)nbr.ptr();
        bool active[NR];
        int i[NR], c[NR];
        unsigned int xyi[NR], zi[NR], lo[NR], hi[NR]; // lo, hi: the chunk being filled, four 16-bit indices
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            active[r] = ibase + r * BLOCK + tid < a1;
            i[r] = active[r] ? ibase + r * BLOCK + tid : a1 - 1;
            const unsigned long long ci = cf[i[r]];
            xyi[r] = (unsigned int)ci; zi[r] = (unsigned int)(ci >> 32);
            c[r] = 0; lo[r] = hi[r] = 0u;
        }
        for (int j0 = 0; j0 < N; j0 += 64) {
            const int jl = j0 + lane < N ? j0 + lane : N - 1;
            const unsigned long long cj = cf[jl];
            const int cxy = (int)(unsigned int)cj, cz = (int)(unsigned int)(cj >> 32);
            unsigned int mm[NR][2];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                unsigned int m[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) m[r] = 0u;
#pragma unroll 1
                for (int b0 = 0; b0 < 32; b0 += 8) // eight candidates in flight; unrolled further the kernel spills more than it gains
#pragma unroll
                    for (int b = b0; b < b0 + 8; ++b) {
                        const int ln = 32 * half + b;
                        const unsigned int sxy = (unsigned int)__builtin_amdgcn_readlane(cxy, ln), sz_ = (unsigned int)__builtin_amdgcn_readlane(cz, ln);
#pragma unroll
                        for (int r = 0; r < NR; ++r) m[r] = test16(m[r], xyi[r], zi[r], sxy, sz_, t2);
                    }
                const int valid = N - j0 - 32 * half; // candidates of this half that exist
                const unsigned int vm = valid >= 32 ? 0xFFFFFFFFu : valid > 0 ? (1u << valid) - 1u : 0u;
#pragma unroll
                for (int r = 0; r < NR; ++r) mm[r][half] = __brev(m[r]) & vm;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if constexpr (C::HALF) {
                    // candidates of the own range: keep j iff (j > i) == ((i + j) even); the atom itself falls out with them.  below(n): bits 0 .. n-1
                    auto below = [](int n) -> unsigned int { return n <= 0 ? 0u : n >= 32 ? 0xFFFFFFFFu : (1u << n) - 1u; };
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int jb = j0 + 32 * half;
                        const unsigned int own = below(a1 - jb) & ~below(a0 - jb);
                        const int d = i[r] - jb; // bit of atom i itself (anywhere: below 0 = every candidate lies above it)
                        const unsigned int above = ~below(d + 1), under = below(d);
                        const unsigned int even = ((i[r] + jb) & 1) ? 0xAAAAAAAAu : 0x55555555u; // bits b with i + jb + b even
                        mm[r][half] &= ~own | (above & even) | (under & ~even);
                    }
                } else {
                if ((unsigned int)(i[r] - j0) < 32u) mm[r][0] &= ~(1u << (i[r] - j0)); // not the atom itself
                else if ((unsigned int)(i[r] - j0) < 64u) mm[r][1] &= ~(1u << (i[r] - j0 - 32));
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    unsigned int m = mm[r][half];
                    while (m) { // a few bits per thread and block
                        const unsigned int j = (unsigned int)(j0 + 32 * half + __builtin_ctz(m));
                        m &= m - 1u;
                        if (c[r] < MAXNB) {
                            const unsigned int v = (j << 3) << (16 * (c[r] & 1)); // the entry is the byte offset 8 j (pair_vec<.., BYTES>)
                            if (c[r] & 2) hi[r] |= v; else lo[r] |= v;
                            if ((c[r] & 3) == 3) {
                                if (active[r]) g[NM_CHECK_INDEX((size_t)(c[r] >> 2) * NMAX + i[r], C::NBR_G_ELEMS / 4)] = ((unsigned long long)hi[r] << 32) | lo[r];
                                lo[r] = hi[r] = 0u;
                            }
                        }
                        ++c[r];
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (active[r] && c[r] < MAXNB && (c[r] & 3)) // the last, partial chunk
                g[NM_CHECK_INDEX((size_t)(c[r] >> 2) * NMAX + i[r], C::NBR_G_ELEMS / 4)] = ((unsigned long long)hi[r] << 32) | lo[r];
            st_maxc = max(st_maxc, c[r]);
            if (c[r] > MAXNB) { ovf = 1; c[r] = MAXNB; }
            if (active[r]) cnt[i[r]] = (unsigned short)c[r];
        }
    }

    // Verlet-list rebuild.
    // * The candidate test runs on a 16-bit fixed-point copy of the positions, u = round(65536 x / L) mod 2^16 per coordinate (round 3;
    //   fp32 in round 2, fp64 in round 1): VALU issue bounds the build, and in this form a test is seven instructions (test16) against
    //   fourteen to eighteen in fp32.  The copy lives in the force array, which is dead here: every caller of rebuild() is about to
    //   run the pair loop that rewrites it.  Each coordinate is off by at most half a unit, each component of a separation by one,
    //   the separation itself by at most sqrt(3) units: the test radius is rc + skin + 1.8 units (1.7e-4 at L = 6.1, 3.4e-4 at
    //   L = 12.2), so the list is a superset of the exact one: the extra entries lie beyond rc + skin and are masked by the pair
    //   loop's exact fp64 cutoff test, contributing an exact zero in the same place of the sum.  (The list itself is not observable
    //   in any result.)  tests/test_list_fixed_point.py restates the arithmetic in numpy and checks the superset property.
    // * Lists in LDS (N <= 256, and the 6^3 system at 8 workgroups per replica): TPA threads per row (see below).
    // * Lists in HBM/L2 (one thread per atom in the pair loop): one THREAD per row.  All lanes of a wave test the same candidate
    //   j at the same time: each lane fetches one of 64 candidates (a coalesced read) and the block is walked with v_readlane, so a
    //   candidate reaches all lanes as two scalar operands, there is no cross-lane step at all, and the loop over j unrolls into
    //   independent tests; a thread packs four indices into the 8-byte chunk the pair loop reads and stores it whole (lanes =
    //   consecutive atoms = consecutive words of the [chunk][atom] layout).  The wave-per-row form of round 1 took ~0.45 ms per
    //   rebuild of a 2048-atom replica — bound by its dependent chain index read -> gathers -> ballot, 32 times per row — which
    //   was ~70 % of the 8^3 kernel once the chains have equilibrated (one to two rebuilds per move when HMC accepts); a (y, z)
    //   column binning with a 5 x 5 stencil in front of it gained only 1.3x and is gone again.
    __device__ void rebuild()
    {
        [[maybe_unused]] const int lane = tid & 63;
        // one 8-byte word per atom: {x | y << 16, z}; LDS lists: skewed copy, element j at j + j / 32
        unsigned long long *cf = (unsigned long long *)(nm_lds + C::OFF_FRC);
        static_assert((size_t)(NMAX + NMAX / 32 + 64 + 2) * sizeof(unsigned long long) <= (size_t)3 * NMAX * sizeof(double), "");
        const double sc16 = 65536.0 / L, invLd = 1.0 / L;
        for (int i = tid; i < N; i += BLOCK) {
            const int is = C::LIST_LDS ? i + (i >> 5) : i;
            // (v_fract_f64 first: a trajectory that is about to be rejected for an astronomic energy — the reference's never-undone
            //  iterative trials produce overlapping atoms — may carry coordinates far beyond the integer range.  Beyond 2^40 box edges, or
            //  not a number at all, an atom gets a made-up place of its own: such a state is rejected whatever its list says (its kinetic
            //  energy is not finite either), but NaNs must not all land on the origin and overflow one another's rows — the fp32 test of
            //  round 2 simply never matched them.)
            const double sx = px[i] * invLd, sy = py[i] * invLd, sz_ = pz[i] * invLd;
            unsigned int ux = (unsigned int)__double2int_rn(__builtin_amdgcn_fract(sx) * 65536.0) & 0xFFFFu,
                         uy = (unsigned int)__double2int_rn(__builtin_amdgcn_fract(sy) * 65536.0) & 0xFFFFu,
                         uz = (unsigned int)__double2int_rn(__builtin_amdgcn_fract(sz_) * 65536.0) & 0xFFFFu;
            if (!(fabs(sx) < 1.0e12 && fabs(sy) < 1.0e12 && fabs(sz_) < 1.0e12)) {
                ux = ((unsigned int)i * 0x9E37u) & 0xFFFFu; uy = ((unsigned int)i * 0x85EBu + 0x1234u) & 0xFFFFu; uz = ((unsigned int)i * 0xC2B3u + 0x5678u) & 0xFFFFu;
            }
            cf[is] = (unsigned long long)(ux | (uy << 16)) | ((unsigned long long)uz << 32);
        }
        set_fresh(false); // the forces are gone
        if constexpr (C::LIST2) {
            if ((flags & F_LIST_SAVED) && !(flags & F_REBUILT)) { // first rebuild since save(): keep the list the move started from
                for (int i = tid; i < N; i += BLOCK) { x0s[i] = x0[i]; y0s[i] = y0[i]; z0s[i] = z0[i]; }
                NM_FOR_OWN(i) cnts[i] = cnt[i];
                L0s = L0;
                list_cur ^= 1;
                if constexpr (!C::LIST_LDS) nbr.g = (IdxT *)p.nbr_g + ((size_t)(gslot - p.slot0) * 2 + list_cur) * C::NBR_G_ELEMS;
                flags |= F_REBUILT;
            }
        }
        const double rt = (p.rc + p.skin) * sc16 + 1.8;                      // test radius in units of L / 65536 (< 2^15: L >= 2 rc)
        const unsigned int t2 = (unsigned int)__double2uint_rd(rt * rt) + 1u; // accepted: squared separation < t2
        int ovf = 0;
        RTL(3); // conversion and the copy of the saved list's reference issued
        __syncthreads(); // the fixed-point copy is complete
        RTL(4);
        if constexpr (C::LIST_LDS) {
            // TPA threads per row (the pair loop's grouping): thread (row, sub) tests the contiguous block of candidates
            // j = sub * CH + k, k < CH = ceil(N / TPA) — 32 per round, branch-free, one bit each.  An exclusive scan of the hit
            // counts over the TPA threads of the row gives every thread its place, and it appends its hits there: with one round
            // (N <= 32 TPA: 256 atoms at 4 or 8 workgroups) the row comes out sorted by j, with more it is ordered by (round, j) — any
            // order that is a function of the positions alone serves, it only fixes the summation order of the pair loop; what matters
            // is that consecutive entries are mostly consecutive atoms, which keeps the pair loop's gathers off each other's banks
            // (a (round, sub)-strided assignment, j = sub + TPA k, measured 5 % slower in the pair loop for that reason).  All 512
            // threads work; the wave-per-row form this replaces kept one wave on a row through 256 candidates with a ballot + mbcnt +
            // scattered byte stores per 64: ~5 us per rebuild of a 256-atom replica at 4 workgroups, as much as an HMC step, 2.4
            // times per move once the chains have equilibrated.
            // The copy is read as cf[j + j / 32]: the TPA blocks start 32 apart, and without the skew the candidates a wave
            // instruction touches (one per sub) would all sit in one LDS bank.
            const int g = tid / TPA, sub = tid - g * TPA;
            const int CH = (N + TPA - 1) / TPA;
            IdxT *const nbr_list = nbr_cur();
            for (int i0 = a0; i0 < a1; i0 += G) { // uniform trip count: the scans below need every lane
                const bool active = i0 + g < a1;
                const int i = active ? i0 + g : a1 - 1;
                const unsigned long long ci = cf[i + (i >> 5)];
                const unsigned int xyi = (unsigned int)ci, zi = (unsigned int)(ci >> 32);
                int base = 0; // entries of the row placed by earlier rounds
                for (int k0 = 0; k0 < CH; k0 += 32) {
                    unsigned int m = 0u;
                    const int jb = sub * CH + k0;
                    const int ntest = (min(32, CH - k0) + 7) & ~7; // (16 candidates per thread at 8 workgroups per replica: two groups of eight)
#pragma unroll 1
                    for (int b0 = 0; b0 < ntest; b0 += 8) { // eight candidates in flight: all eight reads first, then the tests (written as
                        // one loop the compiler put `s_waitcnt lgkmcnt(0)` behind every single read: 32 LDS latencies in a row per thread,
                        // half of a rebuild.  Round 4 tried the NEXT group's reads ahead of the current group's tests (sixteen more
                        // registers): -0.8 % on C2, dropped)
                        unsigned long long cj[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int j = jb + b0 + u; // (beyond the block or N: whatever lies there, masked below)
                            cj[u] = cf[j + (j >> 5)];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) m = test16(m, xyi, zi, (unsigned int)cj[u], (unsigned int)(cj[u] >> 32), t2);
                    }
                    RTL(5); // (last round of tests done)
                    m = __brev(m) >> (32 - ntest); // test b sat in bit ntest - 1 - b
                    const int valid = min(32, min(CH - k0, N - jb)); // candidates of this round that exist
                    m &= valid >= 32 ? 0xFFFFFFFFu : valid > 0 ? (1u << valid) - 1u : 0u;
                    if ((unsigned int)(i - jb) < 32u) m &= ~(1u << (i - jb)); // not the atom itself
                    int incl = __popc(m);
                    const int mine_n = incl;
                    // inclusive scan over the row's TPA threads: DPP row shifts (the group sits inside one 16-lane row; a lane whose
                    // source would lie in the neighbouring group does not add), no LDS round trips
                    if constexpr (TPA >= 2) { const int t = __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true); if (sub >= 1) incl += t; } // row_shr:1
                    if constexpr (TPA >= 4) { const int t = __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true); if (sub >= 2) incl += t; } // row_shr:2
                    if constexpr (TPA >= 8) { const int t = __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true); if (sub >= 4) incl += t; } // row_shr:4
                    if constexpr (TPA >= 16) { const int t = __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true); if (sub >= 8) incl += t; } // row_shr:8
                    unsigned int r = (unsigned int)(base + incl - mine_n);
                    base += __shfl(incl, TPA - 1, TPA);
                    // the hits go to consecutive slots of the row.  Unsigned slot arithmetic, no test inside the loop: a slot beyond the
                    // row's capacity lands on its last one (an overflow stops the block anyway), and the threads of a surplus group
                    // (rows < groups) write what the row's own group writes
                    constexpr unsigned int LT = TPA == 1 ? 0 : TPA == 2 ? 1 : TPA == 4 ? 2 : TPA == 8 ? 3 : 4; // log2 TPA
                    const unsigned int rowc = ((unsigned int)lrow(i) * TPA) << C::LOG2PW;
                    while (m) {
                        const unsigned int j = (unsigned int)jb + (unsigned int)__builtin_ctz(m);
                        m &= m - 1u;
                        const unsigned int rr = min(r, (unsigned int)MAXNB - 1u), k = rr >> LT;
                        nbr_list[(k >> C::LOG2PW) * ((unsigned int)C::NLIST * TPA * C::PW) + rowc + (((rr & (TPA - 1)) << C::LOG2PW) | (k & (C::PW - 1)))] = (IdxT)j;
                        ++r;
                    }
                }
                st_maxc = max(st_maxc, base);
                if (base > MAXNB) { ovf = 1; base = MAXNB; }
                if (active && sub == 0) cnt[i] = (unsigned short)base;
                RTL(6); // (scan and appends done)
            }
        } else {
            static_assert(C::CH == 4 && sizeof(IdxT) == 2 && MAXNB % 4 == 0, "four 16-bit indices per 8-byte chunk");
            static_assert((size_t)NMAX * 8 <= 65536, "the entries hold 8 x the atom index");
            // one thread per row, two rows at a time where a thread has two (rows i and i + BLOCK): the candidate's two v_readlane
            // serve both tests.  Uniform trip count: every lane of a wave must take part in the candidate loads (v_readlane reads
            // lanes whatever their exec bit).
            for (int i0 = a0; i0 < a1; i0 += 2 * BLOCK) {
                if (i0 + BLOCK + (tid & ~63) < a1) rebuild_rows_hbm<2>(i0, cf, t2, ovf); // (wave-uniform)
                else rebuild_rows_hbm<1>(i0, cf, t2, ovf);
            }
        }
#ifdef NM_PROF
        if (p.dbg & 8) { // diagnostic build, NM_DBG=8 (scripts/check_bounds.py; off for the section timings of the same build): every own
            // row must hold ALL atoms whose exact (fp64, minimum-image) separation is below rc + skin.  The entries of a row are distinct
            // atoms, so it does iff as many of its entries lie inside that radius as atoms do.
            __syncthreads();
            const double rl2 = (p.rc + p.skin) * (p.rc + p.skin), iL = 1.0 / L;
            NM_FOR_OWN(i) {
                int n_exact = 0, n_list = 0;
                auto inside = [&](int j) {
                    if (!(fabs(px[j]) < 1.0e6 * L && fabs(py[j]) < 1.0e6 * L && fabs(pz[j]) < 1.0e6 * L)) return false; // (see below)
                    double dx = px[i] - px[j], dy = py[i] - py[j], dz = pz[i] - pz[j];
                    dx -= L * rint(dx * iL); dy -= L * rint(dy * iL); dz -= L * rint(dz * iL);
                    return dx * dx + dy * dy + dz * dz < rl2;
                };
                auto listed_here = [&](int j) { // HALF: the row that holds a pair of two own atoms
                    if (!C::HALF || j < a0 || j >= a1) return true;
                    return (j > i) == (((i + j) & 1) == 0);
                };
                for (int j = 0; j < N; ++j) n_exact += (j != i && listed_here(j) && inside(j)) ? 1 : 0;
                const int c = cnt[i];
                for (int r = 0; r < c; ++r) {
                    int j;
                    if constexpr (C::LIST_LDS) j = (int)nbr_cur()[nbr_at(r, i)];
                    else j = (int)((((const unsigned long long *)nbr.ptr())[(size_t)(r >> 2) * NMAX + i] >> (16 * (r & 3))) & 0xFFFFull) >> 3;
                    n_list += inside(j) ? 1 : 0;
                }
                if (n_exact != n_list && c < MAXNB && fabs(px[i]) < 1.0e6 * L && fabs(py[i]) < 1.0e6 * L && fabs(pz[i]) < 1.0e6 * L) {
                    // (exploded coordinates — see above — are beyond the exact check's own arithmetic)
                    if (atomicAdd(&nm_list_miss, 1u) == 0u) { // the first one: which atom is missing, and how far away it is
                        int jm = -1;
                        for (int j = 0; j < N && jm < 0; ++j) {
                            if (j == i || !listed_here(j) || !inside(j)) continue;
                            bool listed = false;
                            for (int r = 0; r < c; ++r) {
                                int jj;
                                if constexpr (C::LIST_LDS) jj = (int)nbr_cur()[nbr_at(r, i)];
                                else jj = (int)((((const unsigned long long *)nbr.ptr())[(size_t)(r >> 2) * NMAX + i] >> (16 * (r & 3))) & 0xFFFFull) >> 3;
                                listed = listed || jj == j;
                            }
                            if (!listed) jm = j;
                        }
                        double *o = nm_miss_info;
                        o[0] = gslot; o[1] = i; o[2] = jm; o[3] = n_exact; o[4] = n_list; o[5] = c; o[6] = L; o[7] = p.rc + p.skin;
                        if (jm >= 0) {
                            o[8] = px[i]; o[9] = py[i]; o[10] = pz[i]; o[11] = px[jm]; o[12] = py[jm]; o[13] = pz[jm];
                            const unsigned long long ci = cf[C::LIST_LDS ? i + (i >> 5) : i], cj = cf[C::LIST_LDS ? jm + (jm >> 5) : jm];
                            o[14] = (double)ci; o[15] = (double)cj;
                        }
                    }
                }
            }
            __syncthreads();
        }
#endif
        for (int i = tid; i < N; i += BLOCK) { x0[i] = px[i]; y0[i] = py[i]; z0[i] = pz[i]; }
        L0 = L;
        flags |= F_LIST_OK;
        if (p.inj_rebuild >= 0 && (int)st_rebuilds == p.inj_rebuild && q == p.inj_q % Q) ovf = 1; // fault injection (tests)
        st_rebuilds += 1.0;
        if (block_any<NW, NVMAX>(ovf != 0, red, parity)) status |= ST_LIST_OVERFLOW;
        RTL(7);
    }

    // sum over the TPA consecutive lanes of an atom, total in the lane with sub == 0 (the other lanes hold partial garbage).
    // DPP row shifts (lane l reads lane l+n inside its 16-lane row) instead of ds_bpermute shuffles: no LDS round trips.  Same
    // summation tree as an xor butterfly (n = TPA/2, ..., 2, 1), hence the same bits.
    template <int CTRL>
    __device__ __forceinline__ double dpp_shl(double v)
    {
        const long long b = __double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
        return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    __device__ __forceinline__ double group_sum(double v)
    {
        static_assert(TPA <= 16, "an atom's threads must sit inside one 16-lane DPP row");
        if constexpr (TPA >= 16) v += dpp_shl<0x108>(v); // row_shl:8
        if constexpr (TPA >= 8) v += dpp_shl<0x104>(v);  // row_shl:4
        if constexpr (TPA >= 4) v += dpp_shl<0x102>(v);  // row_shl:2
        if constexpr (TPA >= 2) v += dpp_shl<0x101>(v);  // row_shl:1
        return v;
    }

    // 1/r2 by v_rcp_f64 + two Newton steps (about 1 ulp) instead of the 12-instruction IEEE division sequence
    __device__ __forceinline__ double recip(double a)
    {
        double y = __builtin_amdgcn_rcp(a);
        double e = __builtin_fma(-a, y, 1.0);
        y = __builtin_fma(y, e, y);
        e = __builtin_fma(-a, y, 1.0);
        return __builtin_fma(y, e, y);
    }

    // W listed neighbours at once, written stage by stage so that W independent dependency chains interleave (fp64 results
    // are not available to the next instruction of the same chain for several cycles).  Coordinates of the neighbours are
    // already in registers (xj, yj, zj); a masked entry holds atom i itself (r2 = 0: finite, multiplied away).
    template <bool WANT_E, int W>
    __device__ __forceinline__ void pair_pre(const double *xj, const double *yj, const double *zj, const bool *ok, double xi, double yi,
                                             double zi, double invL, double rc2, double &ax, double &ay, double &az, double &e, double &w,
                                             double &np)
    {
        // minimum image: with c_i = x_i / L + 1/2 (the caller's xi, yi, zi; once per row) the separation is
        // (fract(c_i - x_j / L) - 1/2) L: fma, v_fract_f64, fma per component instead of subtraction, multiplication, v_rndne, fma
        double dx[W], dy[W], dz[W], r2[W], y[W], t[W], fp[W];
        const double mhL = -0.5 * L;
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_fma(-xj[q], invL, xi); dy[q] = __builtin_fma(-yj[q], invL, yi); dz[q] = __builtin_fma(-zj[q], invL, zi); }
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_amdgcn_fract(dx[q]); dy[q] = __builtin_amdgcn_fract(dy[q]); dz[q] = __builtin_amdgcn_fract(dz[q]); }
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_fma(dx[q], L, mhL); dy[q] = __builtin_fma(dy[q], L, mhL); dz[q] = __builtin_fma(dz[q], L, mhL); }
        // A listed neighbour beyond the cutoff, or a padded entry, is switched off by zeroing 1 / r^2 AFTER the reciprocal: every
        // term below is a polynomial in it.  (A padded entry may be the atom itself: r2 = 0, reciprocal inf, Newton step NaN,
        // all replaced by the select.)  Three instructions per neighbour fewer than guarding r2 in front and multiplying a 0 / 1
        // mask into the force.
        bool in[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            r2[q] = dx[q] * dx[q] + dy[q] * dy[q] + dz[q] * dz[q];
            in[q] = ok[q] && r2[q] < rc2;
        }
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = __builtin_amdgcn_rcp(r2[q]);
        // v_rcp_f64 is good to 2^-24.4 (scripts/ubench_rcp.hip): one Newton step gives 2e-15 relative (20 ulp), two give the
        // correctly rounded quotient.  Energy evaluations take two, the force-only evaluations inside an HMC trajectory one.
#pragma unroll
        for (int q = 0; q < W; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
        if (WANT_E) {
#pragma unroll
            for (int q = 0; q < W; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
            for (int q = 0; q < W; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
        } // y = 1/r2
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = in[q] ? y[q] : 0.0;
#pragma unroll
        for (int q = 0; q < W; ++q) t[q] = y[q] * y[q] * y[q];              // 1/r6
#pragma unroll
        for (int q = 0; q < W; ++q) {
            // in units of 24 (force, virial) and 4 (energy): 2 t - 1 and t - 1 take inline constants, whereas 48 t - 24 as an fma
            // into a register preloaded with -24 cost two v_mov per neighbour; pair_loop scales the row's sums once at the end
            fp[q] = t[q] * __builtin_fma(2.0, t[q], -1.0) * y[q];
            if (WANT_E) { e += t[q] * (t[q] - 1.0); np += in[q] ? 1.0 : 0.0; }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
            ax += dx[q] * fp[q]; ay += dy[q] * fp[q]; az += dz[q] * fp[q];
            if (WANT_E) w += r2[q] * fp[q];
        }
    }
    // the same with the gathers from LDS in front (lists that live in HBM/L2: their index loads are what is prefetched there)
    // (BYTES: j holds byte offsets into the coordinate arrays, 8 x the atom index — what the lists in HBM store, so that the
    //  gather's address needs no shift)
    template <bool WANT_E, int W, bool BYTES = false>
    __device__ __forceinline__ void pair_vec(const int (&j)[W], const bool (&ok)[W], double xi, double yi, double zi, double invL,
                                             double rc2, double &ax, double &ay, double &az, double &e, double &w, double &np)
    {
        double xj[W], yj[W], zj[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            if constexpr (BYTES) {
                xj[q] = *(const double *)((const char *)px.ptr() + j[q]); yj[q] = *(const double *)((const char *)py.ptr() + j[q]);
                zj[q] = *(const double *)((const char *)pz.ptr() + j[q]);
            } else { xj[q] = px[j[q]]; yj[q] = py[j[q]]; zj[q] = pz[j[q]]; }
        }
        pair_pre<WANT_E, W>(xj, yj, zj, ok, xi, yi, zi, invL, rc2, ax, ay, az, e, w, np);
    }

    // HALF: the force array of the own atoms as 64-bit fixed point while a pair loop runs.  fma(t, scale, 1.5 * 2^52) leaves
    // round(t * scale) in the low mantissa bits (|t * scale| < 2^51); the difference of the bit patterns is that integer.
    static constexpr double FIX_SCALE = 68719476736.0 /* 2^36 */, FIX_MAGIC = 6755399441055744.0 /* 1.5 * 2^52 */;
    __device__ __forceinline__ void fixed_add(unsigned int byte_off, double t, double scale) const
    {
        const double y = __builtin_fma(t, scale, FIX_MAGIC);
        const long long n = __double_as_longlong(y) - __double_as_longlong(FIX_MAGIC);
        (void)__hip_atomic_fetch_add((long long *)__builtin_assume_aligned(nm_lds + byte_off, 8), n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __device__ __forceinline__ void half_begin() // own entries of the force array = 0 (as integers and as doubles); ends with a barrier
    {
        NM_FOR_OWN(i) { fx[i] = 0.0; fy[i] = 0.0; fz[i] = 0.0; }
        __syncthreads();
    }
    // behind the pair loop: a barrier (every row's and every partner's addend has landed; nobody reads positions any more), then the
    // own atoms' entries become doubles again, on the thread that integrates them (NM_FOR_OWN, as every later reader of f[] of the own
    // atoms does).  final_kick: the energy evaluation that ends a trajectory makes the last half kick here and sums the kinetic energy.
    __device__ __forceinline__ void half_end(bool final_kick, double dtfm, double &kacc)
    {
        __syncthreads();
        NM_FOR_OWN(i) {
            const double gx = fixed_get(fx[i]), gy = fixed_get(fy[i]), gz = fixed_get(fz[i]);
            fx[i] = gx; fy[i] = gy; fz[i] = gz;
            if (final_kick) {
                const double ux = __builtin_fma(dtfm, gx, vx[i]), uy = __builtin_fma(dtfm, gy, vy[i]), uz = __builtin_fma(dtfm, gz, vz[i]);
                vx[i] = ux; vy[i] = uy; vz[i] = uz;
                kacc += p.mass * (ux * ux + uy * uy + uz * uz);
            }
        }
    }
    __device__ __forceinline__ double fixed_get(const double &slot) const
    {
        return (double)__double_as_longlong(slot) * (1.0 / FIX_SCALE);
    }
    // W listed neighbours of row i at once (lists in HBM/L2, j = 8 x the atom index); entries that are atoms of this workgroup
    // (own8 = 8 a0, ownn8 = 8 (a1 - a0)) stand for the pair in BOTH directions: the opposite force goes to the partner, the pair's energy,
    // virial and count are taken twice (the caller halves the cluster-wide sums as it does for a full list)
    // (the neighbours' coordinates are handed in: the caller gathers those of the NEXT trip before it calls this for the current one,
    //  so that the gathers stand in front of this trip's atomics in the LDS queue, which returns in order — with the gathers behind
    //  them every trip waited for six atomics to drain before its coordinates arrived: C5 share -11 %, run.sh setting +5 %; round 4)
    template <bool WANT_E, int W>
    __device__ __forceinline__ void pair_vec_half(const int (&j)[W], const bool (&ok)[W], const double (&xj)[W], const double (&yj)[W],
                                                  const double (&zj)[W], double xi, double yi, double zi, double invL,
                                                  double rc2, double &ax, double &ay, double &az, double &e, double &w, double &np,
                                                  unsigned int own8, unsigned int ownn8)
    {
        double dx[W], dy[W], dz[W], r2[W], y[W], t[W], fp[W];
        const double mhL = -0.5 * L;
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_fma(-xj[q], invL, xi); dy[q] = __builtin_fma(-yj[q], invL, yi); dz[q] = __builtin_fma(-zj[q], invL, zi); }
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_amdgcn_fract(dx[q]); dy[q] = __builtin_amdgcn_fract(dy[q]); dz[q] = __builtin_amdgcn_fract(dz[q]); }
#pragma unroll
        for (int q = 0; q < W; ++q) { dx[q] = __builtin_fma(dx[q], L, mhL); dy[q] = __builtin_fma(dy[q], L, mhL); dz[q] = __builtin_fma(dz[q], L, mhL); }
        bool in[W], own[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            r2[q] = dx[q] * dx[q] + dy[q] * dy[q] + dz[q] * dz[q];
            in[q] = ok[q] && r2[q] < rc2;
            own[q] = ((unsigned int)j[q] - own8) < ownn8;
        }
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = __builtin_amdgcn_rcp(r2[q]);
#pragma unroll
        for (int q = 0; q < W; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
        if (WANT_E) {
#pragma unroll
            for (int q = 0; q < W; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
            for (int q = 0; q < W; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
        }
#pragma unroll
        for (int q = 0; q < W; ++q) y[q] = in[q] ? y[q] : 0.0;
#pragma unroll
        for (int q = 0; q < W; ++q) t[q] = y[q] * y[q] * y[q];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            fp[q] = t[q] * __builtin_fma(2.0, t[q], -1.0) * y[q]; // (pair_pre's units: 24 for force and virial, 4 for the energy)
            if (WANT_E) { const double wg = own[q] ? 2.0 : 1.0; e += wg * (t[q] * (t[q] - 1.0)); np += in[q] ? wg : 0.0; w += wg * (r2[q] * fp[q]); }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const double tx = dx[q] * fp[q], ty = dy[q] * fp[q], tz = dz[q] * fp[q];
            ax += tx; ay += ty; az += tz;
            // the partner's share, -24 t (the row's own sum is scaled by 24 at its end).  UNCONDITIONAL: an entry beyond the cutoff adds an
            // exact zero (fp = 0), an atom of another workgroup gets a zero scale (its entry of the force array is not read by anybody).
            // Guarded by `if (own && in)` the compiler gave every neighbour a basic block of its own and the two neighbours of a trip no
            // longer overlapped: the half list then ran 8-14 % SLOWER than the full one (run.sh setting, C5 share; round 4).
            const double sc = own[q] ? -24.0 * FIX_SCALE : 0.0;
            fixed_add((unsigned int)C::OFF_FRC + (unsigned int)j[q], tx, sc);
            fixed_add((unsigned int)(C::OFF_FRC + A1) + (unsigned int)j[q], ty, sc);
            fixed_add((unsigned int)(C::OFF_FRC + 2 * A1) + (unsigned int)j[q], tz, sc);
        }
    }

    // The two waves that share a SIMD (w and w + NW/2) do not share it evenly: the older one wins every arbitration it is
    // ready for, finishes its rows ~1 us before the other, and the younger one then runs alone below the SIMD's fp64 issue
    // rate.  So in the loops over LDS lists the younger wave holds the higher priority for its first NM_PRIO_SW list entries and
    // the older one after that: both stay in the loop to the end (measured +2.9 % on the 4^3 cluster; either wave favoured
    // throughout: no gain).
    static constexpr int PRIO_SW = TPA >= 16 ? NM_PRIO_SW16 : TPA >= 8 ? NM_PRIO_SW8 : TPA >= 4 ? NM_PRIO_SW4 : NM_PRIO_SW2;
    __device__ __forceinline__ bool young() const { return (tid >> 6) >= NW / 2; }
    __device__ __forceinline__ void prio_begin() const { if constexpr (C::LIST_LDS) { if (young()) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); } }
    __device__ __forceinline__ void prio_swap() const { if constexpr (C::LIST_LDS) { if (young()) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); } }
    __device__ __forceinline__ void prio_end() const { if constexpr (C::LIST_LDS) __builtin_amdgcn_s_setprio(0); }

    // fuse (force-only evaluations inside an HMC trajectory): the lane that holds atom i's force integrates it on the spot —
    // both half kicks around this evaluation and the drift, same arithmetic as advance_and_share(two) — publishes the NEW
    // position to the cluster and parks it in f[i] (the force has no other reader); positions themselves stay untouched until
    // the whole workgroup is through its pair loop.  The peers thus get the positions as early as they used to get forces.
    template <bool WANT_E>
    __device__ __forceinline__ void pair_loop(double invL, double &eacc, double &wacc, double &nacc, double &kacc, bool fuse, double dtfm, double h)
    {
        double *xg = xb ? xb + (size_t)(gen & 1) * C::XBUF_DOUBLES : nullptr;
        const int g = tid / TPA, sub = tid - g * TPA;
        const double rc2 = p.rc * p.rc;
        // SPREAD (round 4): the row's epilogue — integrate, publish, store — is done by three lanes, one component each, instead of
        // lane 0 doing all three in a row (TPA >= 4; the byte / 16-bit LDS lists)
        constexpr bool SPREAD = (NM_SPREAD != 0) && TPA >= 4 && !C::HALF;
        for (int i0 = a0; i0 < a1; i0 += G) { // uniform trip count keeps the shuffles below convergent
            const int i = i0 + g;
            double ax = 0.0, ay = 0.0, az = 0.0, e = 0.0, w = 0.0, np = 0.0;
            [[maybe_unused]] double vpre = 0.0, ppre = 0.0;
            prio_begin();
            if (i < a1) {
                // (the row's first list word is asked for FIRST: the first gathers wait for it and for nothing else)
                [[maybe_unused]] unsigned long long wn_first = 0ull;
                if constexpr (C::LIST_LDS) wn_first = ((const unsigned long long *)nbr_cur())[(size_t)lrow(i) * TPA + sub];
                const int c = cnt[i];
                const double xi = __builtin_fma(px[i], invL, 0.5), yi = __builtin_fma(py[i], invL, 0.5), zi = __builtin_fma(pz[i], invL, 0.5); // (pair_pre)
                if constexpr (SPREAD && !C::LIST_LDS) { if (fuse && sub < 3) { vpre = vx.ptr()[sub * NMAX + i]; ppre = px.ptr()[sub * NMAX + i]; } }
                if constexpr (C::LIST_LDS) {
                    constexpr int W = NM_PAIR_W, PW = C::PW, BITS = 8 * (int)sizeof(IdxT);
                    static_assert(PW % W == 0, "");
                    const unsigned long long *nb64 = (const unsigned long long *)nbr_cur();
                    const int mine = (c - sub + TPA - 1) / TPA; // neighbours of atom i that this thread handles: slots sub, sub+TPA, ...
                    constexpr int KLAST = MAXNB / TPA - PW;     // first entry of a thread's last list word
                    unsigned long long wn = wn_first;
                    if constexpr (SPREAD) {
                        // lanes 0, 1, 2 of the row will integrate the x, y, z component (the epilogue below): what they need of the atom is
                        // asked for NOW — behind the first list word, which the first gathers wait for — and arrives under the neighbours'
                        // arithmetic.  Read in the epilogue, two dependent LDS round trips stood behind every pair loop while the LDS pipe
                        // was busy with the other waves' gathers.
                        if (fuse && sub < 3) { vpre = vx.ptr()[sub * NMAX + i]; ppre = px.ptr()[sub * NMAX + i]; }
                    }
                    TLINE(1); // (experiment build: the row's prologue is issued)
                    for (int k0 = 0; k0 < mine; k0 += PW) {     // one conflict-free 8-byte read = PW of them
                        const unsigned long long wd = wn;
                        // the NEXT word is asked for now and looked at PW entries later (unconditionally: clamped to the row's last word) —
                        // read where it is needed, its LDS latency stood in front of every PW-th pair evaluation
                        wn = nb64[((size_t)(min(k0 + PW, KLAST) >> C::LOG2PW) * C::NLIST + lrow(i)) * TPA + sub];
#pragma unroll
                        for (int e0 = 0; e0 < PW; e0 += W) {
                            if (k0 + e0 == PRIO_SW) prio_swap();
                            if (k0 + e0 < mine) {
                                int jj[W];
                                bool ok[W];
#pragma unroll
                                for (int q = 0; q < W; ++q) {
                                    ok[q] = (k0 + e0 + q) < mine;
                                    jj[q] = ok[q] ? (int)((wd >> (BITS * (e0 + q))) & ((1ull << BITS) - 1ull)) : i; // a masked lane looks at itself: finite, ignored
                                }
                                pair_vec<WANT_E, W>(jj, ok, xi, yi, zi, invL, rc2, ax, ay, az, e, w, np);
                            }
                        }
                    }
                } else {
                    static_assert(sizeof(IdxT) == 2 && C::CH == 4, "lists outside LDS: chunks of four 16-bit indices");
                    // The list lives in HBM/L2: one 8-byte load = four neighbours, ~1 us away.  The loads of the NEXT four
                    // chunks are issued before the arithmetic on the current four, so the latency is paid once per 16 neighbours
                    // and hidden behind ~16 pair evaluations (issued one chunk at a time it dominated the 6^3 / 8^3 kernels).
                    constexpr int PF = 4, W = 2;
                    const unsigned long long *nb64 = (const unsigned long long *)nbr.ptr();
                    const int nch = (c + C::CH - 1) / C::CH;              // chunks of atom i
                    const int mych = (nch - sub + TPA - 1) / TPA;         // chunks sub, sub+TPA, ... belong to this thread
                    unsigned long long cur[PF], nxt[PF];
#pragma unroll
                    for (int q = 0; q < PF; ++q) cur[q] = q < mych ? nb64[NM_CHECK_INDEX((size_t)(sub + q * TPA) * NMAX + i, C::NBR_G_ELEMS / 4)] : 0ull;
                    if constexpr (C::HALF) {
                        // software pipeline over the trips of two neighbours: indices and coordinates of the next trip are fetched before
                        // the current one is evaluated (an entry that does not exist reads atom 0 and is masked)
                        static_assert(TPA == 1 && W == 2 && C::CH == 4, "");
                        int jn[W]; double xn[W], yn[W], zn[W];
                        auto fetch = [&](unsigned long long word, int e0) {
#pragma unroll
                            for (int r = 0; r < W; ++r) {
                                jn[r] = (int)((word >> (16 * (e0 + r))) & 0xFFFFull); // 8 x the atom index
                                xn[r] = *(const double *)((const char *)px.ptr() + jn[r]); yn[r] = *(const double *)((const char *)py.ptr() + jn[r]);
                                zn[r] = *(const double *)((const char *)pz.ptr() + jn[r]);
                            }
                        };
                        fetch(cur[0], 0);
                        const unsigned int own8 = 8u * (unsigned int)a0, ownn8 = 8u * (unsigned int)(a1 - a0);
                        for (int k0 = 0; k0 < mych; k0 += PF) {
#pragma unroll
                            for (int q = 0; q < PF; ++q) nxt[q] = (k0 + PF + q) < mych ? nb64[NM_CHECK_INDEX((size_t)(k0 + PF + q) * NMAX + i, C::NBR_G_ELEMS / 4)] : 0ull;
#pragma unroll
                            for (int q = 0; q < PF; ++q) {
                                if (k0 + q < mych) {
                                    const int first = (k0 + q) * C::CH;
#pragma unroll
                                    for (int e0 = 0; e0 < C::CH; e0 += W) {
                                        int jj[W]; bool ok[W]; double xc[W], yc[W], zc[W];
#pragma unroll
                                        for (int r = 0; r < W; ++r) { jj[r] = jn[r]; xc[r] = xn[r]; yc[r] = yn[r]; zc[r] = zn[r]; ok[r] = (first + e0 + r) < c; }
                                        if (e0 + W < C::CH) fetch(cur[q], e0 + W);
                                        else if (q + 1 < PF) fetch(cur[q + 1], 0);
                                        else fetch(nxt[0], 0);
                                        pair_vec_half<WANT_E, W>(jj, ok, xc, yc, zc, xi, yi, zi, invL, rc2, ax, ay, az, e, w, np, own8, ownn8);
                                    }
                                }
                            }
#pragma unroll
                            for (int q = 0; q < PF; ++q) cur[q] = nxt[q];
                        }
                    } else
                    for (int k0 = 0; k0 < mych; k0 += PF) {
#pragma unroll
                        for (int q = 0; q < PF; ++q) nxt[q] = (k0 + PF + q) < mych ? nb64[NM_CHECK_INDEX((size_t)(sub + (k0 + PF + q) * TPA) * NMAX + i, C::NBR_G_ELEMS / 4)] : 0ull;
#pragma unroll
                        for (int q = 0; q < PF; ++q) {
                            if (k0 + q < mych) {
                                const int first = (sub + (k0 + q) * TPA) * C::CH; // list slot of this chunk's first entry
#pragma unroll
                                for (int e0 = 0; e0 < C::CH; e0 += W) {
                                    int jj[W];
                                    bool ok[W];
#pragma unroll
                                    for (int r = 0; r < W; ++r) {
                                        ok[r] = (first + e0 + r) < c; // the tail entries of a row's last chunk are zero (rebuild): atom 0, ignored
                                        jj[r] = (int)((cur[q] >> (16 * (e0 + r))) & 0xFFFFull); // 8 x the atom index
                                    }
                                    pair_vec<WANT_E, W, true>(jj, ok, xi, yi, zi, invL, rc2, ax, ay, az, e, w, np);
                                }
                            }
                        }
#pragma unroll
                        for (int q = 0; q < PF; ++q) cur[q] = nxt[q];
                    }
                }
            }
            prio_end();
            TLINE(5); // (experiment build: the row's neighbours are through; what follows is the epilogue)
            ax *= 24.0; ay *= 24.0; az *= 24.0; // (pair_pre's units)
            if (WANT_E) { e *= 4.0; w *= 24.0; }
            ax = group_sum(ax); ay = group_sum(ay); az = group_sum(az);
            if (WANT_E) { e = group_sum(e); w = group_sum(w); np = group_sum(np); }
            if constexpr (C::HALF) {
                // the row's own sum joins what the partners' rows add to this atom (fixed point, like theirs: every addend of an entry is
                // an integer, the total does not depend on the order); integration waits for the barrier behind the loop (half_end)
                if (i < a1) {
                    eacc += e; wacc += w; nacc += np;
                    fixed_add((unsigned int)C::OFF_FRC + 8u * (unsigned int)i, ax, FIX_SCALE);
                    fixed_add((unsigned int)(C::OFF_FRC + A1) + 8u * (unsigned int)i, ay, FIX_SCALE);
                    fixed_add((unsigned int)(C::OFF_FRC + 2 * A1) + 8u * (unsigned int)i, az, FIX_SCALE);
                }
                continue;
            }
            if constexpr (SPREAD) {
                // the three sums sit in lane 0 of the row (group_sum): y moves to lane 1, z to lane 2 (row_shr), and each of the three
                // lanes finishes its component with the same arithmetic as the one-lane epilogue below (bit-identical forces, velocities
                // and positions; the kinetic energy is summed per component instead of per atom)
                const double sy = dpp_mov<0x111>(ay), sz = dpp_mov<0x112>(az); // (moved by EVERY lane, outside any branch: a DPP read
                // of a lane that is masked off returns zero)
                const double ac = sub == 0 ? ax : sub == 1 ? sy : sz;
                if (i < a1 && sub == 0) { eacc += e; wacc += w; nacc += np; }
                if (i < a1 && sub < 3) {
                    const int ci = sub * NMAX + i; // x, y, z follow one another NMAX doubles apart in every array
                    if (WANT_E && fuse) { // the energy evaluation that ends a trajectory: final_integrate on the spot, kinetic energy along
                        const double u = __builtin_fma(dtfm, ac, vpre);
                        vx.ptr()[ci] = u;
                        kacc += p.mass * (u * u);
                        fx.ptr()[ci] = ac;
                    } else if (!WANT_E && fuse) {
                        double u = __builtin_fma(dtfm, ac, vpre);
                        u = __builtin_fma(dtfm, ac, u);
                        vx.ptr()[ci] = u;
                        const double pn = __builtin_fma(h, u, ppre);
                        if (Q > 1) put_granule(xg + 2 * (size_t)ci, pn, my_magic());
                        fx.ptr()[ci] = pn;
                    } else fx.ptr()[ci] = ac;
                }
                continue;
            }
            if (i < a1 && sub == 0) {
                eacc += e; wacc += w; nacc += np;
                if (WANT_E && fuse) { // the energy evaluation that ends a trajectory: final_integrate on the spot, kinetic energy along
                    const double ux = __builtin_fma(dtfm, ax, vx[i]), uy = __builtin_fma(dtfm, ay, vy[i]), uz = __builtin_fma(dtfm, az, vz[i]);
                    vx[i] = ux; vy[i] = uy; vz[i] = uz;
                    kacc += p.mass * (ux * ux + uy * uy + uz * uz);
                }
                if (!WANT_E && fuse) {
                    double ux = __builtin_fma(dtfm, ax, vx[i]), uy = __builtin_fma(dtfm, ay, vy[i]), uz = __builtin_fma(dtfm, az, vz[i]);
                    ux = __builtin_fma(dtfm, ax, ux); uy = __builtin_fma(dtfm, ay, uy); uz = __builtin_fma(dtfm, az, uz);
                    vx[i] = ux; vy[i] = uy; vz[i] = uz;
                    ax = __builtin_fma(h, ux, px[i]); ay = __builtin_fma(h, uy, py[i]); az = __builtin_fma(h, uz, pz[i]);
                    if (Q > 1) {
                        const unsigned long long mgp = my_magic();
                        put_granule(xg + 2 * (size_t)i, ax, mgp);
                        put_granule(xg + 2 * (size_t)(NMAX + i), ay, mgp);
                        put_granule(xg + 2 * (size_t)(2 * NMAX + i), az, mgp);
                    }
                }
                fx[i] = ax; fy[i] = ay; fz[i] = az;
            }
        }
    }

    // ------------------------------------------------------------------ Sutton-Chen EAM (element Al; the build's own choice
    // for BASELINE config 4: the reference's MEAM parameter files are not part of its tree)
    // E = eps [ 1/2 sum_ij (a/r)^7 - c sum_i sqrt(rho_i) ],  rho_i = sum_j (a/r)^6,  r < rc.
    // Two passes over the same full list: densities of the own atoms, (cluster: exchange them,) then forces with
    // fp = eps [ 7 (a/r)^7 - 6 (c/2)(1/sqrt(rho_i) + 1/sqrt(rho_j)) (a/r)^6 ] / r^2.
    template <bool WANT_E>
    __device__ __forceinline__ void pair_loop_sc(double invL, double &eacc, double &wacc, double &nacc, double &kacc, bool fuse, double dtfm, double h);

    // ------------------------------------------------------------------ cluster hand-off (Q workgroups per replica)
    // Data-tagged granules (MI355X guide, hand-off price list "handoff-1to1"): every exchanged double travels as ONE
    // 16-byte write-through (sc1) store {bits, bits ^ magic}, magic = f(launch, generation).  A reader polls the granule
    // itself with sc1 loads until the two words agree with the magic of the generation it is in: no flag, no counter, no
    // fence, one memory round trip.  A torn or stale granule fails the check (probability 2^-64 otherwise) and is re-read.
    // Every exchange (positions, partial sums, EAM densities) is all-to-all and takes the next generation; two buffers
    // alternate; a workgroup can only run one exchange ahead of the slowest one (it needs everybody's data of generation g
    // to finish g), so buffer g&1 is never overwritten while someone still reads it.  Spins are bounded: a cluster that is
    // not co-resident reports ST_SYNC_TIMEOUT instead of hanging.
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    // a workgroup whose own list overflowed publishes under magic ^ POISON: the peers accept the granule and learn the status
    // from it (the only status bit that is not identical in all workgroups of a cluster), so no status words are exchanged
    static constexpr unsigned long long POISON = 0x5555555555555554ull;
    __device__ __forceinline__ unsigned long long magic() const
    {
        return ((unsigned long long)p.launch_id << 32 | (unsigned long long)(uint32_t)(gen + 1)) * 0x9E3779B97F4A7C15ull | 1ull;
    }
    __device__ __forceinline__ void put_granule(double *g, double v, unsigned long long mg)
    {
        u64x2 w;
        w.x = (unsigned long long)__double_as_longlong(v);
        w.y = w.x ^ mg;
        // hipcc adds no wait states for an asm statement: a VMEM store of more than 8 bytes needs one before its data
        // registers may be overwritten (gfx9 hazard), hence the s_nop inside the string
        // one XCD (same_xcd, established at block start): a plain store leaves the line in that XCD's L2, where the peers' sc1
        // loads (L1 bypassed) find it ~0.5 us sooner than after the write-through that sc1 stores force (they drop the line from
        // L2, so the reader goes out to the fabric).  Across XCDs only the write-through form is visible at all.
        if (same_xcd) asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(w) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(w) : "memory");
    }
    // up to three granules per call, issued back to back and waited for once; `poisoned` is set when a granule carries the
    // publisher's overflow mark
    template <int K>
    __device__ __forceinline__ bool get_granules(double *const (&g)[K], unsigned long long mg, double (&out)[K], int &timeout, int &poisoned)
    {
        unsigned long long t0 = 0; // 100 MHz clock, read only once polls have failed
        int spins = 0;
        for (;;) {
            u64x2 w[K];
            static_assert(K == 1 || K == 2 || K == 3 || K == 4, "granule reads come in ones to fours");
            if constexpr (K == 2)
                asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]) : "v"(g[0]), "v"(g[1]) : "memory");
            else
            if constexpr (K == 4)
                asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
                             "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]), "=&v"(w[3]) : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]) : "memory");
            else if constexpr (K == 3)
                asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\t"
                             "global_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(w[0]), "=&v"(w[1]), "=&v"(w[2]) : "v"(g[0]), "v"(g[1]), "v"(g[2]) : "memory");
            else
                asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w[0]) : "v"(g[0]) : "memory");
            bool ok = true;
            int po = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const unsigned long long x = w[k].x ^ w[k].y;
                ok = ok && (x == mg || x == (mg ^ POISON));
                po |= (x == (mg ^ POISON)) ? 1 : 0;
            }
            if (ok) {
#pragma unroll
                for (int k = 0; k < K; ++k) out[k] = __longlong_as_double((long long)w[k].x);
                poisoned |= po;
                return true;
            }
#if NM_POLL_SLEEP > 0
            __builtin_amdgcn_s_sleep(NM_POLL_SLEEP); // a failed poll is not repeated at once: fewer loads in the way of the peers' stores
#endif
            if ((++spins & 63) == 0) {
                const unsigned long long now = wall_clock64();
                if (t0 == 0) t0 = now | 1ull;
                else if (now - t0 > 200000000ull) { timeout = 1; return false; } // 2 s
            }
        }
    }

    // What crosses the cluster (Q > 1): every workgroup holds all positions but computes forces for, and integrates, only its
    // own atoms a0..a1.  So an HMC step exchanges the NEW POSITIONS of the own atoms (advance_and_share), an energy evaluation
    // exchanges per-workgroup partial sums (exchange_sums), the EAM adds its densities (pair_loop_sc); forces never travel.
    // Every exchange is all-to-all and takes the next generation number; a workgroup whose list overflowed marks whatever it
    // publishes next with POISON, and whoever reads it takes over the status bit, so the cluster leaves the block together.
    __device__ __forceinline__ unsigned long long my_magic() const { return magic() ^ ((status & ST_LIST_OVERFLOW) ? POISON : 0ull); }

    // cluster-wide sums of K per-workgroup partial sums (identical in all threads of the workgroup on entry): lane r of every
    // wave fetches workgroup r's K granules in one round trip, then the lanes are added in workgroup order, so every thread
    // of every workgroup ends with the identical bits.  Ends with a barrier.
    template <int K>
    __device__ void exchange_sums(double (&s)[K])
    {
        static_assert(K == 4, "one instantiation: callers pad with zeros");
        if (Q == 1) return;
        double *xg = xb + (size_t)(gen & 1) * C::XBUF_DOUBLES;
        const unsigned long long mg = magic();
        if (tid < K) {
            double mine = s[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mine = (tid == k) ? s[k] : mine;
            put_granule(xg + 2 * (size_t)(C::XG_PART + 4 * q + tid), mine, my_magic());
        }
        int timeout = 0, poisoned = 0;
        const int lane = tid & 63;
        double v[K];
#pragma unroll
        for (int k = 0; k < K; ++k) v[k] = 0.0;
        if (lane < Q) {
            double *gs[K];
#pragma unroll
            for (int k = 0; k < K; ++k) gs[k] = xg + 2 * (C::XG_PART + 4 * lane + k);
            get_granules<K>(gs, mg, v, timeout, poisoned);
        }
        double t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            t[k] = 0.0;
            for (int r = 0; r < Q; ++r) t[k] += __shfl(v[k], r, 64);
        }
        ++gen;
        TLINE(5);
        const int fl = block_any2<NW, NVMAX>(timeout != 0, poisoned != 0, red, parity);
        if (fl & 1) status |= ST_SYNC_TIMEOUT;
        if (fl & 2) status |= ST_LIST_OVERFLOW; // a peer's list overflowed
#pragma unroll
        for (int k = 0; k < K; ++k) s[k] = uniform(t[k]);
    }

    // the same in two halves, for work that can be done while the granules are in flight
    __device__ __forceinline__ void exchange_put(const double (&s)[4])
    {
        double *xg = xb + (size_t)(gen & 1) * C::XBUF_DOUBLES;
        if (tid < 4) {
            double mine = s[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) mine = (tid == k) ? s[k] : mine;
            put_granule(xg + 2 * (size_t)(C::XG_PART + 4 * q + tid), mine, my_magic());
        }
    }
    __device__ __forceinline__ void exchange_get(double (&s)[4])
    {
        double *xg = xb + (size_t)(gen & 1) * C::XBUF_DOUBLES;
        const unsigned long long mg = magic();
        int timeout = 0, poisoned = 0;
        const int lane = tid & 63;
        double v[4] = { 0.0, 0.0, 0.0, 0.0 };
        if (lane < Q) {
            double *gs[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) gs[k] = xg + 2 * (C::XG_PART + 4 * lane + k);
            get_granules<4>(gs, mg, v, timeout, poisoned);
        }
        double t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = 0.0;
            for (int r = 0; r < Q; ++r) t[k] += __shfl(v[k], r, 64);
        }
        ++gen;
        const int fl = block_any2<NW, NVMAX>(timeout != 0, poisoned != 0, red, parity);
        if (fl & 1) status |= ST_SYNC_TIMEOUT;
        if (fl & 2) status |= ST_LIST_OVERFLOW; // a peer's list overflowed
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = uniform(t[k]);
    }

    // a workgroup that leaves the block on its own finding (list overflow) marks everything its peers could be waiting for
    // next — position granules of its atoms and its partial sums — so that they leave too instead of timing out
    __device__ void publish_poison()
    {
        double *xg = xb + (size_t)(gen & 1) * C::XBUF_DOUBLES;
        const unsigned long long mgp = magic() ^ POISON;
        NM_FOR_OWN(i) {
            put_granule(xg + 2 * (size_t)i, 0.0, mgp);
            put_granule(xg + 2 * (size_t)(NMAX + i), 0.0, mgp);
            put_granule(xg + 2 * (size_t)(2 * NMAX + i), 0.0, mgp);
        }
        if (tid < 4) put_granule(xg + 2 * (size_t)(C::XG_PART + 4 * q + tid), 0.0, mgp);
    }

    // the list-validity test of one atom at (x, y, z)
    struct ListCheck { double sc, thr2, invL; int bad; };
    __device__ __forceinline__ void box_consts()
    {
        if (L == bc_L && L0 == bc_L0) return;
        bc_L = L; bc_L0 = L0;
        bc_invL = 1.0 / L;
        // the list built at (x0, L0) still covers every pair within rc of the affinely rescaled reference if
        // max_i |x_i - (L/L0) x0_i| <= ((L/L0)(rc+skin) - rc)/2
        bc_sc = L / L0;
        const double thr = 0.5 * (bc_sc * (p.rc + p.skin) - p.rc);
        bc_thr2 = thr * thr;
        bc_bad = !(thr > 0.0);
    }
    __device__ __forceinline__ void check_begin(ListCheck &c)
    {
        box_consts();
        c.sc = bc_sc; c.thr2 = bc_thr2; c.invL = bc_invL; c.bad = bc_bad ? 1 : 0;
    }
    __device__ __forceinline__ void check_atom(ListCheck &c, int i, double x, double y, double z) const
    {
        check_atom(c, x, y, z, x0[i], y0[i], z0[i]);
    }
    // (the reference position handed in: a caller that waits for something else first reads it before that wait)
    __device__ __forceinline__ void check_atom(ListCheck &c, double x, double y, double z, double rx, double ry, double rz) const
    {
        double dx = x - c.sc * rx, dy = y - c.sc * ry, dz = z - c.sc * rz;
        dx -= L * rint(dx * c.invL); dy -= L * rint(dy * c.invL); dz -= L * rint(dz * c.invL);
        if (dx * dx + dy * dy + dz * dz > c.thr2) c.bad = 1;
    }

    // One integrator step of the own atoms — v += dtfm f (fix nve initial_integrate), or twice that when the final_integrate of
    // the previous step is folded in (same arithmetic as two separate half kicks), then x += h v — and the hand-over of their
    // new positions.  The own atoms' threads publish three granules each; the threads of the following waves fetch the
    // other workgroups' atoms (one thread per atom, three granules in flight, ONE memory round trip) and write them to LDS.
    // Every position is tested against the list-validity bound by the thread that writes it, so the one block-wide OR at the
    // end is the barrier that publishes the positions AND the rebuild decision: the evaluation that follows starts at once.
    // Requires that no thread of the workgroup still reads positions (the caller's evaluation ended with a barrier).
    // mode 1: one half kick + drift; 3: the pair loop did it already (eval_force): f[] holds the new positions of the own atoms
    // and they are published; 2 (HALF configurations, whose forces are complete only behind the pair loop's barrier): the two half kicks
    // around the evaluation that just ended — final_integrate of its step, initial_integrate of the next — and the drift.
    __device__ bool advance_and_share(int mode, double dtfm, double h)
    {
        int timeout = 0, poisoned = 0;
        double *xg = xb ? xb + (size_t)(gen & 1) * C::XBUF_DOUBLES : nullptr;
        const unsigned long long mg = magic(), mgp = my_magic();
        // the own atoms first and nothing in their way: the peers are waiting for these granules
        NM_FOR_OWN(i) {
            const double gx = fx[i], gy = fy[i], gz = fz[i];
            if (mode == 3) { px[i] = gx; py[i] = gy; pz[i] = gz; continue; }
            double ux = __builtin_fma(dtfm, gx, vx[i]), uy = __builtin_fma(dtfm, gy, vy[i]), uz = __builtin_fma(dtfm, gz, vz[i]);
            if (mode == 2) { ux = __builtin_fma(dtfm, gx, ux); uy = __builtin_fma(dtfm, gy, uy); uz = __builtin_fma(dtfm, gz, uz); } // (as the fused epilogue)
            vx[i] = ux; vy[i] = uy; vz[i] = uz;
            const double nx = __builtin_fma(h, ux, px[i]), ny = __builtin_fma(h, uy, py[i]), nz = __builtin_fma(h, uz, pz[i]);
            px[i] = nx; py[i] = ny; pz[i] = nz;
            if (Q > 1) {
                put_granule(xg + 2 * (size_t)i, nx, mgp);
                put_granule(xg + 2 * (size_t)(NMAX + i), ny, mgp);
                put_granule(xg + 2 * (size_t)(2 * NMAX + i), nz, mgp);
            }
        }
        ListCheck c;
        check_begin(c);
        NM_FOR_OWN(i) check_atom(c, i, px[i], py[i], pz[i]);
        if (Q > 1) {
            const int nown = a1 - a0, nother = N - nown;
            const int shift = ((nown + 63) & ~63) % BLOCK; // the fetching starts on the waves after the ones that integrate
            for (int o = (tid + BLOCK - shift) % BLOCK; o < nother; o += BLOCK) {
                const int i = o < a0 ? o : o + nown;
                double *const g3[3] = { xg + 2 * (size_t)i, xg + 2 * (size_t)(NMAX + i), xg + 2 * (size_t)(2 * NMAX + i) };
                double x3[3];
                const double rx = x0[i], ry = y0[i], rz = z0[i]; // (read before the poll: at 8^3 they come from the global spill, a memory latency that
                // now lies under the granules' round trip instead of behind it)
                if (get_granules<3>(g3, mg, x3, timeout, poisoned)) {
                    px[i] = x3[0]; py[i] = x3[1]; pz[i] = x3[2];
                    check_atom(c, x3[0], x3[1], x3[2], rx, ry, rz);
                }
            }
            ++gen;
        }
        set_fresh(false);
        TLINE_PREV(6);
        const int fl = block_any3<NW, NVMAX>(timeout != 0, poisoned != 0, c.bad != 0, red, parity);
        TLINE_PREV(7);
        if (fl & 1) status |= ST_SYNC_TIMEOUT;
        if (fl & 2) status |= ST_LIST_OVERFLOW;
        return (fl & 4) != 0;
    }

    // ------------------------------------------------------------------ lj/cut 2.5 energy, forces, virial
    // pair_lj_cut: r2inv, r6inv, fpair = r6inv*(48 r6inv - 24)*r2inv, evdwl = r6inv*(4 r6inv - 4), no shift/tail.
    // Full (both-direction) list: thread group (i, sub) owns f_i, no scatter, no atomics, fixed summation order.
    // have_need: the caller already holds the rebuild decision for the current positions (advance_and_share) and the barrier
    // that goes with it.  An energy evaluation leaves this workgroup's partial sums in psum[]; finish_sums() makes U, W of them
    // (across the cluster, together with one more partial sum of the caller: the kinetic energy at the end of a trajectory).
    __device__ void eval(bool have_need = false, bool pre_need = false, bool final_kick = false, double dtfm = 0.0)
    {
        if (status & (ST_SYNC_TIMEOUT | ST_LIST_OVERFLOW)) return; // learnt from the last hand-over: the cluster is leaving
        if (!(L >= 2.0 * p.rc)) { status |= ST_BOX_TOO_SMALL; __syncthreads(); return; } // minimum-image limit
        bool need = !(flags & F_LIST_OK);
        TLINE(0);
        PROF_BEGIN();
        // The validity check reads only a thread's own atoms (written by itself) and x0 (settled since the last rebuild), so
        // it needs no barrier before it; its own block-wide OR is the barrier that publishes the new positions to everybody.
        if (have_need && !need) need = pre_need;
        else if (need || NM_DBG(4)) __syncthreads();
        else {
            ListCheck c;
            check_begin(c);
            for (int i = tid; i < N; i += BLOCK) check_atom(c, i, px[i], py[i], pz[i]);
            need = block_any<NW, NVMAX>(c.bad != 0, red, parity);
        }
        PROF_END(1);
        TLINE(1);
        PROF_BEGIN();
        if (need) rebuild();
        PROF_END(2);
        TLINE(2);
        box_consts();
        const double invL = bc_invL;

        double eacc = 0.0, wacc = 0.0, nacc = 0.0, kacc = 0.0;
        PROF_BEGIN();
        if constexpr (C::HALF) half_begin();
        if (NM_DBG(16)) { }
        else if constexpr (C::POT == 1) pair_loop_sc<true>(invL, eacc, wacc, nacc, kacc, final_kick, dtfm, 0.0);
        else pair_loop<true>(invL, eacc, wacc, nacc, kacc, final_kick, dtfm, 0.0);
        if constexpr (C::HALF) half_end(final_kick, dtfm, kacc);
        PROF_END(3);
        TLINE(3);
        PROF_BEGIN();
        st_evals += 1.0;
        double s[4] = { eacc, wacc, nacc, kacc };
        block_sum<4, NW, NVMAX>(s, red, parity); // over this workgroup's atoms
        PROF_END(4);
        TLINE(4);
        psum[0] = s[0]; psum[1] = s[1]; psum[2] = s[2]; psum[3] = s[3];
        ++tl_n;
        set_fresh(true);
    }

    // The force-only evaluation inside an HMC trajectory (fix nve between two `run` boundaries): the rebuild decision for the
    // current positions came with the hand-over (advance_and_share) together with the barrier that published them.  The pair
    // loop integrates the own atoms on the spot (both half kicks around this evaluation, the drift) and publishes their new
    // positions; it ends with the barrier after which positions may be overwritten.  ONE call site (the trajectory loop of the
    // block kernel), so this and eval() hold one pair loop each.
    __device__ __forceinline__ void eval_force(bool need, double dtfm, double h)
    {
        TLINE(0);
        TLINE(1);
        PROF_BEGIN();
        if (need || !(flags & F_LIST_OK)) rebuild();
        PROF_END(2);
        TLINE(2);
        box_consts();
        const double invL = bc_invL;
        double eacc = 0.0, wacc = 0.0, nacc = 0.0, kacc = 0.0;
        PROF_BEGIN();
        if constexpr (C::HALF) half_begin();
        if (NM_DBG(16)) { }
        else if constexpr (C::POT == 1) pair_loop_sc<false>(invL, eacc, wacc, nacc, kacc, true, dtfm, h);
        else pair_loop<false>(invL, eacc, wacc, nacc, kacc, true, dtfm, h);
        PROF_END(3);
        TLINE(3);
        PROF_BEGIN();
        st_evals += 1.0;
        if constexpr (C::HALF) half_end(false, 0.0, kacc); // its barrier is the one below; the caller integrates (advance_and_share mode 2)
        else
        __syncthreads(); // nobody reads positions any more: they may be advanced
        PROF_END(4);
        TLINE(4);
        ++tl_n;
        set_fresh(true);
    }

    __device__ void take_sums()
    {
        U = 0.5 * psum[0]; W = 0.5 * psum[1];
        st_eevals += 1.0; st_pairs += 0.5 * psum[2];
        if (!(U == U) || isinf(U)) status |= ST_NONFINITE;
    }
    // completes an energy evaluation; `extra` is one more per-workgroup partial sum that rides along.  next_m >= 0: the move that
    // follows is move next_m of the block; if it is going to be a Hamiltonian move its gaussians are drawn between the publication of
    // this workgroup's partial sums and the poll for the peers' — the ~1 us the sums need to cross the cluster otherwise passes idle.
    // Philox is counter-based, so the values are those velocity_create would draw: results do not change by a bit.
    int gauss_for = -1; // move index whose gaussians wait in the prefetch area
    __device__ double finish_sums(double extra, int next_m = -1)
    {
        double s[4] = { psum[0], psum[1], psum[2], extra };
#if NM_AB == 5
        next_m = -1;
#endif
        if (C::PREFETCH && Q > 1 && next_m >= 0 && next_m < p.mod && tape == nullptr && !p.md_mode) {
            exchange_put(s);
            uint32_t o[4];
            philox4x32_10(0u, S_ROLL, (uint32_t)next_m, p.step, p.seed, (uint32_t)gslot, o);
            const double roll = u01(o[0], o[1]); // (draw_scalar(S_ROLL, next_m, 0): block-uniform)
            if (roll > p.ppos + p.pvol) {
                gaussian_fill<C>((uint32_t)next_m, N, gslot, p.mass, p.seed, p.step, (int)C::OFF_GAUSS);
                gauss_for = next_m; // (visible to every thread after exchange_get's barrier)
            }
            exchange_get(s);
        } else exchange_sums<4>(s);
        psum[0] = s[0]; psum[1] = s[1]; psum[2] = s[2];
        take_sums();
        return s[3];
    }

    // ------------------------------------------------------------------ velocity commands (remcmc:604-606)
    // velocity all create t seed dist gaussian   (gaussians/sqrt(m) per atom id, COM momentum removed, rescaled to exactly
    //                                             t with dof = 3N-3: LAMMPS velocity.cpp create(), defaults mom yes rot no)
    // velocity all zero linear                   (again removes the COM momentum, now round-off only)
    // velocity all zero angular                  (omega = I^-1 L about the centre of mass of the unwrapped coordinates)
    // LAMMPS makes four passes with three global sums.  Here: one pass for the gaussians (2N work items over all threads:
    // item w < N draws (vx, vy) of atom w, item N + w draws vz), ONE block reduction of the 16 raw moments
    //   sum m v (3), sum m X (3), sum m v.v, sum m X x v (3), sum m (second moments of X) (6),
    // from which the quantities of the later passes follow algebraically:
    //   sum m |v - c|^2            = sum m v.v - M |c|^2                       (c = COM velocity, M = total mass)
    //   L about the COM, scaled    = sc (sum m X x v - M Xc x c)               (Xc = centre of mass)
    //   I about the COM            = raw second moments - M (|Xc|^2 1 - Xc Xc^T)   (parallel axis)
    // and one pass that writes v = sc (v - c) - omega x (X - Xc).  The second "zero linear" would subtract the round-off of
    // sum m v' / M (~1e-17 relative); it is left out.  Differences to the four-pass arithmetic are ~1e-15 relative.
    __device__ __forceinline__ double hmc_velocities(double t, uint32_t tag, int prefetched)
    {
        short *img = nullptr;
        if constexpr (!C::SAVE_LDS) img = im.g;
        const double mv2 = velocity_create<C>(t, tag, L, N, gslot, parity, p.mass, p.mvv2e, p.kB, p.seed, p.step, img, prefetched);
        parity ^= 1;
        return uniform(mv2); // (the caller's barrier after wrap() orders these velocities for the threads that take the own atoms next)
    }

    // ------------------------------------------------------------------ the moves
    // single-particle energy difference against all other atoms (identical to the reference's difference of two
    // full-system energies up to summation order)
    __device__ void delta_single(int k, double ox, double oy, double oz, double nx, double ny, double nz, double &dE, double &dW)
    {
        box_consts();
        const double invL = bc_invL, rc2 = p.rc * p.rc;
        double s[2] = { 0.0, 0.0 };
        // One work item per atom j, carrying BOTH pairs (new position, j) and (old position, j): atom j's coordinates are then
        // read by one thread only, the thread that owns j in every elementwise phase (j mod BLOCK) and the only one that ever
        // writes them in iter_pmc, so a trial needs no barrier besides the one inside its block reduction.  (With the old and
        // the new pair as separate items the old-position item of atom k's own thread lived in another wave whenever
        // N mod BLOCK != 0 and read px[k] while the accepted position of the previous trial was being written.)  The two
        // pair evaluations are independent dependency chains and interleave in the pipeline.
        for (int j = tid; j < N; j += BLOCK) {
            const double xj = px[j], yj = py[j], zj = pz[j];
            double ax = nx - xj, ay = ny - yj, az = nz - zj, bx = ox - xj, by = oy - yj, bz = oz - zj;
            ax -= L * rint(ax * invL); ay -= L * rint(ay * invL); az -= L * rint(az * invL);
            bx -= L * rint(bx * invL); by -= L * rint(by * invL); bz -= L * rint(bz * invL);
            const double ra = ax * ax + ay * ay + az * az, rb = bx * bx + by * by + bz * bz;
            const bool ina = (j != k) && ra < rc2, inb = (j != k) && rb < rc2;
            const double ia = recip(ina ? ra : 1.0), ib = recip(inb ? rb : 1.0);
            const double a6 = ia * ia * ia, b6 = ib * ib * ib;
            const double sa = ina ? 1.0 : 0.0, sb = inb ? -1.0 : 0.0;
            s[0] += sa * (a6 * (4.0 * a6 - 4.0));
            s[0] += sb * (b6 * (4.0 * b6 - 4.0));
            s[1] += sa * (a6 * (48.0 * a6 - 24.0));
            s[1] += sb * (b6 * (48.0 * b6 - 24.0));
        }
        block_sum<2, NW, NVMAX>(s, red, parity);
        dE = s[0]; dW = s[1];
    }

    // The same for the Sutton-Chen EAM: moving atom k changes its pair terms AND the density of every neighbour, old or new:
    //   dE = eps [ sum_j ((a/r'_kj)^7 - (a/r_kj)^7) - c ( sum_j (sqrt(rho_j + drho_j) - sqrt(rho_j)) + sqrt(rho'_k) - sqrt(rho_k) ) ],
    //   drho_j = (a/r'_kj)^6 - (a/r_kj)^6,  rho'_k = sum_j (a/r'_kj)^6.
    // rho[] holds the densities of the current configuration (iter_densities at move start, kept up to date by the trial loop); thread
    // j keeps its drho_j until the decision is known.  One work item per atom (N <= BLOCK), one block reduction.
    __device__ void delta_single_sc(int k, double ox, double oy, double oz, double nx, double ny, double nz, double &dE, double &drho_mine,
                                    double &rho_k_new)
    {
        static_assert(C::POT != 1 || NMAX <= BLOCK, "one thread per atom");
        box_consts();
        const double invL = bc_invL, rc2 = p.rc * p.rc, a2 = p.sc_a2;
        double s[3] = { 0.0, 0.0, 0.0 };
        drho_mine = 0.0;
        const int j = tid;
        if (j < N && j != k) {
            const double xj = px[j], yj = py[j], zj = pz[j];
            double ax = nx - xj, ay = ny - yj, az = nz - zj, bx = ox - xj, by = oy - yj, bz = oz - zj;
            ax -= L * rint(ax * invL); ay -= L * rint(ay * invL); az -= L * rint(az * invL);
            bx -= L * rint(bx * invL); by -= L * rint(by * invL); bz -= L * rint(bz * invL);
            const double ra = ax * ax + ay * ay + az * az, rb = bx * bx + by * by + bz * bz;
            const bool ina = ra < rc2, inb = rb < rc2;
            const double qa = a2 * recip(ina ? ra : 1.0), qb = a2 * recip(inb ? rb : 1.0); // (a/r)^2
            const double ga = ina ? qa * qa * qa : 0.0, gb = inb ? qb * qb * qb : 0.0;     // (a/r)^6
            s[0] = ga * sqrt(qa) - gb * sqrt(qb);                                          // (a/r)^7, new - old
            drho_mine = ga - gb;
            const double rj = rho[j];
            s[1] = sqrt(rj + drho_mine) - sqrt(rj);
            s[2] = ga;
        }
        block_sum<3, NW, NVMAX>(s, red, parity);
        rho_k_new = s[2];
        dE = p.sc_eps * (s[0] - p.sc_c * (s[1] + sqrt(s[2]) - sqrt(rho[k])));
    }
    // densities of all atoms of the current configuration, O(N^2), one thread per atom (start of an iterative EAM position move)
    __device__ void iter_densities()
    {
        box_consts();
        const double invL = bc_invL, rc2 = p.rc * p.rc, a2 = p.sc_a2;
        __syncthreads(); // rho[] may still be read by the evaluation that ended last
        if (tid < N) {
            const double xj = px[tid], yj = py[tid], zj = pz[tid];
            double r = 0.0;
            for (int i = 0; i < N; ++i) {
                double dx = xj - px[i], dy = yj - py[i], dz = zj - pz[i];
                dx -= L * rint(dx * invL); dy -= L * rint(dy * invL); dz -= L * rint(dz * invL);
                const double r2 = dx * dx + dy * dy + dz * dz;
                const bool in = r2 < rc2 && i != tid;
                const double q = a2 * recip(in ? r2 : 1.0);
                r += in ? q * q * q : 0.0;
            }
            rho[tid] = r;
        }
        __syncthreads();
    }

    // exclusive prefix sum of one value per thread over the workgroup (thread order), and the total; ONE barrier (the two halves
    // of `red` alternate as in block_sum).  Every thread sees the identical total.
    __device__ __forceinline__ double block_scan(double v, double &total)
    {
        const int lane = tid & 63, wv = tid >> 6;
        double incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const double t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        double *r = red.ptr() + parity * (NW * NVMAX);
        if (lane == 63) r[wv * NVMAX] = incl;
        __syncthreads();
        double before = 0.0, all = 0.0;
        for (int w = 0; w < NW; ++w) { const double t = r[w * NVMAX]; if (w < wv) before += t; all += t; }
        parity ^= 1;
        total = all;
        return before + incl - v;
    }

    // The reference's iterative position move, all N trials at once.  In reference mode a trial is NEVER undone (remcmc:522-525:
    // `od` aliases `x`), so the positions do not depend on the decisions: trial k sees the new positions of the atoms before it
    // and the gathered ones of the atoms after it, whatever was accepted.  All N energy differences are therefore independent —
    // two threads per trial, each over every other atom — the running energy the k-th criterion compares with is a prefix sum of
    // them, and the decisions, counters and image-flag increments follow in one pass: ~40 us per move at 256 atoms instead of
    // 256 serial trials of ~1.9 us.  The candidate positions and acceptance draws are in f[] and svx[] (iter_pmc); svy / svz
    // (dead during a position move) take dE, dW.  Summation orders differ from the serial loop (~1e-16 relative).
    __device__ int iter_pmc_all(double et, double &nt, double &na, double &crit)
    {
        box_consts();
        const double invL = bc_invL, rc2 = p.rc * p.rc;
        const double *const pos = px.ptr(), *const cand = fx.ptr(); // x, y, z follow one another NMAX doubles apart in both
        // A cluster shares the trials out like the rows of the pair loop — workgroup q takes the atoms of its own range, TPA threads
        // per trial — and hands the energy differences round (one hop, the position granules' slots).
        const int g = tid / TPA, sub = tid - g * TPA;
        double *xg = xb ? xb + (size_t)(gen & 1) * C::XBUF_DOUBLES : nullptr;
        const unsigned long long mg = magic(), mgp = my_magic();
        for (int i0 = a0; i0 < a1; i0 += G) { // uniform trip count (group_sum)
            const bool have = i0 + g < a1;
            const int k = have ? i0 + g : a1 - 1;
            const double ox = px[k], oy = py[k], oz = pz[k], nx = fx[k], ny = fy[k], nz = fz[k];
            double dE = 0.0, dW = 0.0;
            for (int j = sub; j < N; j += TPA) {
                const double *const src = j < k ? cand : pos; // atoms before k have moved already
                const double xj = src[j], yj = src[NMAX + j], zj = src[2 * NMAX + j];
                double ax = nx - xj, ay = ny - yj, az = nz - zj, bx = ox - xj, by = oy - yj, bz = oz - zj;
                ax -= L * rint(ax * invL); ay -= L * rint(ay * invL); az -= L * rint(az * invL);
                bx -= L * rint(bx * invL); by -= L * rint(by * invL); bz -= L * rint(bz * invL);
                const double ra = ax * ax + ay * ay + az * az, rb = bx * bx + by * by + bz * bz;
                const bool ina = (j != k) && ra < rc2, inb = (j != k) && rb < rc2;
                const double ia = recip(ina ? ra : 1.0), ib = recip(inb ? rb : 1.0);
                const double a6 = ia * ia * ia, b6 = ib * ib * ib;
                const double sa = ina ? 1.0 : 0.0, sb = inb ? -1.0 : 0.0;
                dE += sa * (a6 * (4.0 * a6 - 4.0)); dE += sb * (b6 * (4.0 * b6 - 4.0));
                dW += sa * (a6 * (48.0 * a6 - 24.0)); dW += sb * (b6 * (48.0 * b6 - 24.0));
            }
            dE = group_sum(dE); dW = group_sum(dW);
            if (have && sub == 0) {
                svy[k] = dE; svz[k] = dW;
                if (Q > 1) { put_granule(xg + 2 * (size_t)k, dE, mgp); put_granule(xg + 2 * (size_t)(NMAX + k), dW, mgp); }
            }
        }
        if (Q > 1) {
            int timeout = 0, poisoned = 0;
            const int nown = a1 - a0, nother = N - nown;
            for (int o = tid; o < nother; o += BLOCK) {
                const int i = o < a0 ? o : o + nown;
                double *const g2[2] = { xg + 2 * (size_t)i, xg + 2 * (size_t)(NMAX + i) };
                double v2[2];
                if (get_granules<2>(g2, mg, v2, timeout, poisoned)) { svy[i] = v2[0]; svz[i] = v2[1]; }
            }
            ++gen;
            const int fl = block_any2<NW, NVMAX>(timeout != 0, poisoned != 0, red, parity);
            if (fl & 1) status |= ST_SYNC_TIMEOUT;
            if (fl & 2) status |= ST_LIST_OVERFLOW;
            if (fl) return 0; // the cluster is leaving the block (the caller looks at the status)
        } else __syncthreads();
        // decisions: U_k = U + sum_{i<k} dE_i is what trial k starts from
        int nacc = 0;
        double Urun = U, Wsum = 0.0;
        int runs0 = 0; // `run 0`s before the trials of this chunk (one per accepted trial, two per rejected one)
        for (int k0 = 0; k0 < N; k0 += BLOCK) {
            const int k = k0 + tid;
            const bool have = k < N;
            const double dE = have ? svy[k] : 0.0, dW = have ? svz[k] : 0.0;
            double totE, totW;
            const double before = block_scan(dE, totE);
            (void)block_scan(dW, totW);
            const double Uk = Urun + before, Unew = Uk + dE;
            const double de = Unew / et - Uk / et;
            const double metcrit = exp(-de);
            const double mm = (metcrit != metcrit) ? metcrit : (metcrit < 1.0 ? metcrit : 1.0);
            const bool acc = have && !isinf(metcrit) && svx[have ? k : 0] <= mm;
            double totA, totR;
            (void)block_scan(acc ? 1.0 : 0.0, totA);
            const double runs_before = block_scan(have ? (acc ? 1.0 : 2.0) : 0.0, totR);
            if (have) {
                const int runs = runs0 + (int)runs_before; // the stale coordinate was remapped by every run 0 so far
                px[k] = fx[k]; py[k] = fy[k]; pz[k] = fz[k];
                im[3 * k] = (short)(im[3 * k] + runs * wn[3 * k]);
                im[3 * k + 1] = (short)(im[3 * k + 1] + runs * wn[3 * k + 1]);
                im[3 * k + 2] = (short)(im[3 * k + 2] + runs * wn[3 * k + 2]);
                if (k == N - 1) red[2 * NW * NVMAX - 1] = de; // the criterion of the move's last trial (trace)
            }
            nacc += (int)totA;
            runs0 += (int)totR;
            Urun += totE; Wsum += totW;
        }
        __syncthreads();
        crit = red[2 * NW * NVMAX - 1];
        nt += (double)N; na += (double)nacc;
        U = uniform(Urun); W = uniform(W + Wsum);
        set_fresh(false); // f[] was used as scratch, the atoms have moved
        __syncthreads();  // (red's last slot is free again before the next reduction could reach it)
        return nacc;
    }

    // iter_position_mc (remcmc:505-549) with single-particle energy differences instead of N full evaluations.
    // Reference mode (iter_revert = 0) follows the reference literally: the coordinates gathered at move start stay on the
    // "Python side" un-remapped, every trial re-sends them and runs `run 0` (twice when the trial is rejected, remcmc:541-542),
    // so LAMMPS remaps a stale out-of-box coordinate again each time: its image flag grows by its wrap count per run until the
    // atom's own trial replaces it.  A rejected trial is not undone (`od` aliases `x`, remcmc:522-525).
    __device__ int iter_pmc(uint32_t m, double et, double dx, double &nt, double &na, double &crit)
    {
        int nacc = 0, runs = 0;
        if (p.iter_revert) wrap(); // corrected mode: one consistent remap at move start
        else
            for (int i = tid; i < N; i += BLOCK) { // wrap counts of the gathered coordinates
                double a = px[i], b = py[i], c = pz[i];
                wn[3 * i] = (signed char)wrap1(a); wn[3 * i + 1] = (signed char)wrap1(b); wn[3 * i + 2] = (signed char)wrap1(c);
            }
        const double boxl = L;
        // Philox streams are counter-based, so the N candidate positions and acceptance draws of a move do not have to be
        // produced one trial at a time: all threads prepare them at once (atom k's own position cannot change before its
        // trial) and park them in f[] and the saved-velocity slot, both dead during a position move.  The trial loop is then
        // one block reduction per trial; it needs no other barrier, because the only position a trial changes is written and
        // later read as a neighbour by the same thread (k mod BLOCK: delta_single gives each atom's two pairs to its owner),
        // and nobody else looks at atom k again in this move.
        const bool pre = (tape == nullptr);
        if (pre)
            for (int k = tid; k < N; k += BLOCK) {
                uint32_t o[4], q[4];
                philox4x32_10((uint32_t)k, S_ITER_XY, m, p.step, p.seed, (uint32_t)gslot, o);
                philox4x32_10((uint32_t)k, S_ITER_Z, m, p.step, p.seed, (uint32_t)gslot, q);
                double nx = px[k] + 2.0 * (u01(o[0], o[1]) - 0.5) * dx * p.lat, ny = py[k] + 2.0 * (u01(o[2], o[3]) - 0.5) * dx * p.lat,
                       nz = pz[k] + 2.0 * (u01(q[0], q[1]) - 0.5) * dx * p.lat;
                nx -= floor(nx / boxl) * boxl; ny -= floor(ny / boxl) * boxl; nz -= floor(nz / boxl) * boxl; // remcmc:524
                fx[k] = nx; fy[k] = ny; fz[k] = nz;
                svx[k] = draw_scalar(S_ITER_ACC, m, (uint32_t)k);
            }
        __syncthreads();
        if constexpr (C::POT == 0) {
            if (pre && !p.iter_revert) return iter_pmc_all(et, nt, na, crit);
        }
        if constexpr (C::POT == 1) iter_densities();
        for (int k = 0; k < N; ++k) {
            nt += 1.0;
            const double pe = U / et;
            const double ox = px[k], oy = py[k], oz = pz[k];
            double nx, ny, nz;
            if (pre) { nx = fx[k]; ny = fy[k]; nz = fz[k]; }
            else {
                const double u0 = draw_scalar(0, 0, 0), u1 = draw_scalar(0, 0, 0), u2 = draw_scalar(0, 0, 0);
                nx = ox + 2.0 * (u0 - 0.5) * dx * p.lat; ny = oy + 2.0 * (u1 - 0.5) * dx * p.lat; nz = oz + 2.0 * (u2 - 0.5) * dx * p.lat;
                nx -= floor(nx / boxl) * boxl; ny -= floor(ny / boxl) * boxl; nz -= floor(nz / boxl) * boxl; // remcmc:524
            }
            double dE, dW = 0.0;
            [[maybe_unused]] double drho = 0.0, rho_k_new = 0.0;
            if constexpr (C::POT == 1) delta_single_sc(k, ox, oy, oz, nx, ny, nz, dE, drho, rho_k_new);
            else delta_single(k, ox, oy, oz, nx, ny, nz, dE, dW);
            const double Unew = U + dE;
            const double de = Unew / et - pe;
            bool acc;
            if (pre) { // metropolis() with the draw made above (unused when exp(-de) overflows, as in the reference)
                const double metcrit = exp(-de);
                const double mm = (metcrit != metcrit) ? metcrit : (metcrit < 1.0 ? metcrit : 1.0);
                acc = !isinf(metcrit) && svx[k] <= mm;
            } else acc = metropolis(de, S_ITER_ACC, m, (uint32_t)k);
            crit = de;
            if (acc) { na += 1.0; ++nacc; }
            if (acc || !p.iter_revert) {
                if (tid == (k % BLOCK)) {
                    px[k] = nx; py[k] = ny; pz[k] = nz;
                    if (!p.iter_revert) { // the stale coordinate was remapped by every run 0 so far
                        im[3 * k] = (short)(im[3 * k] + runs * wn[3 * k]);
                        im[3 * k + 1] = (short)(im[3 * k + 1] + runs * wn[3 * k + 1]);
                        im[3 * k + 2] = (short)(im[3 * k + 2] + runs * wn[3 * k + 2]);
                    }
                }
                if constexpr (C::POT == 1) { // the densities follow the move
                    if (tid < N && tid != k) rho[tid] += drho;
                    if (tid == (k % BLOCK)) rho[k] = rho_k_new;
                }
                U = Unew; W += dW;
                set_fresh(false);
            }
            if constexpr (C::POT == 1) __syncthreads(); // the next trial reads rho[k+1] in every thread
            runs += acc ? 1 : 2;
        }
        __syncthreads();
        set_fresh(false); // f[] was used as scratch
        return nacc;
    }
};

template <class C>
template <bool WANT_E>
__device__ __forceinline__ void Replica<C>::pair_loop_sc(double invL, double &eacc, double &wacc, double &nacc, double &kacc, bool fuse, double dtfm, double h)
{
    const int g = tid / TPA, sub = tid - g * TPA;
    const double rc2 = p.rc * p.rc, a2 = p.sc_a2, eps = p.sc_eps, cc = p.sc_c, mhL = -0.5 * L;
    double *xg = xb ? xb + (size_t)(gen & 1) * C::XBUF_DOUBLES : nullptr;
    const unsigned long long mg = magic();
    // pass 1: densities of this workgroup's atoms.  Both passes walk the byte list like pair_loop does: one conflict-free 8-byte
    // read = eight neighbours, two neighbours' dependency chains interleaved stage by stage.
    constexpr int W = NM_PAIR_W;
    static_assert(sizeof(IdxT) == 1, "the EAM loops read byte lists");
    const unsigned long long *nb64 = (const unsigned long long *)nbr_cur();
    for (int i0 = a0; i0 < a1; i0 += G) {
        const int i = i0 + g;
        double r = 0.0;
        prio_begin();
        if (i < a1) {
            const double xi = __builtin_fma(px[i], invL, 0.5), yi = __builtin_fma(py[i], invL, 0.5), zi = __builtin_fma(pz[i], invL, 0.5); // (minimum image as in pair_pre)
            const int c = cnt[i];
            const int mine = (c - sub + TPA - 1) / TPA;
            unsigned long long wn = nb64[(size_t)lrow(i) * TPA + sub];
            for (int k0 = 0; k0 < mine; k0 += 8) {
                const unsigned long long wd = wn; // (the next word one word ahead, as in pair_loop)
                wn = nb64[((size_t)(min(k0 + 8, MAXNB / TPA - 8) >> 3) * C::NLIST + lrow(i)) * TPA + sub];
#pragma unroll
                for (int e0 = 0; e0 < 8; e0 += W) {
                    if (k0 + e0 == PRIO_SW) prio_swap();
                    if (k0 + e0 < mine) {
                        double dx[W], dy[W], dz[W], r2[W], y[W], t[W];
                        bool in[W]; // (switched off by zeroing 1 / r^2 after the reciprocal, as in pair_pre)
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            in[u] = (k0 + e0 + u) < mine;
                            const int j = in[u] ? (int)((wd >> (8 * (e0 + u))) & 0xFFull) : i;
                            dx[u] = __builtin_fma(-px[j], invL, xi); dy[u] = __builtin_fma(-py[j], invL, yi); dz[u] = __builtin_fma(-pz[j], invL, zi);
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            dx[u] = __builtin_fma(__builtin_amdgcn_fract(dx[u]), L, mhL); dy[u] = __builtin_fma(__builtin_amdgcn_fract(dy[u]), L, mhL);
                            dz[u] = __builtin_fma(__builtin_amdgcn_fract(dz[u]), L, mhL);
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            r2[u] = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
                            in[u] = in[u] && r2[u] < rc2;
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) y[u] = __builtin_amdgcn_rcp(r2[u]);
#pragma unroll
                        for (int u = 0; u < W; ++u) t[u] = __builtin_fma(-r2[u], y[u], 1.0);
#pragma unroll
                        for (int u = 0; u < W; ++u) y[u] = __builtin_fma(y[u], t[u], y[u]);
#pragma unroll
                        for (int u = 0; u < W; ++u) t[u] = __builtin_fma(-r2[u], y[u], 1.0);
#pragma unroll
                        for (int u = 0; u < W; ++u) y[u] = in[u] ? __builtin_fma(y[u], t[u], y[u]) * a2 : 0.0; // (a/r)^2
#pragma unroll
                        for (int u = 0; u < W; ++u) r += y[u] * y[u] * y[u];
                    }
                }
            }
        }
        prio_end();
        r = group_sum(r);
        if (i < a1 && sub == 0) {
            rho[i] = r;
            if (Q > 1) put_granule(xg + 2 * (C::XG_RHO + i), r, my_magic());
        }
    }
    int timeout = 0, poisoned = 0;
    if (Q > 1) { // densities of the atoms the other workgroups own (an exchange of its own generation)
        const int nother = N - (a1 - a0);
        for (int o = tid; o < nother; o += BLOCK) {
            const int i = o < a0 ? o : o + (a1 - a0);
            double *const g1[1] = { xg + 2 * (C::XG_RHO + i) };
            double v1[1];
            if (get_granules<1>(g1, mg, v1, timeout, poisoned)) rho[i] = v1[0];
        }
        ++gen;
    }
    {
        const int fl = block_any2<NW, NVMAX>(timeout != 0, poisoned != 0, red, parity);
        if (fl & 2) status |= ST_LIST_OVERFLOW; // a peer's list overflowed: finish this evaluation with the others, then leave
        if (fl & 1) { status |= ST_SYNC_TIMEOUT; return; }
    }
    double sq_own = 0.0; // sum over own atoms of sqrt(rho_i), by the atom's elementwise owner
    for (int i = tid; i < N; i += BLOCK) {
        const double sr = sqrt(rho[i]);
        if (i >= a0 && i < a1) sq_own += sr;
        rho[i] = 1.0 / sr;
    }
    __syncthreads();
    // pass 2: forces (and energy, virial) of this workgroup's atoms.  1/r comes from v_rsq_f64 + two Newton steps (~1 ulp), which
    // gives (a/r)^2 and a/r at once: no separate division and square root.
    const double a1r = sqrt(a2);
    constexpr bool SPREAD = (NM_SPREAD != 0) && TPA >= 4; // the row's epilogue on three lanes, operands prefetched (see pair_loop)
    for (int i0 = a0; i0 < a1; i0 += G) {
        const int i = i0 + g;
        double ax = 0.0, ay = 0.0, az = 0.0, e = 0.0, w = 0.0, np = 0.0;
        [[maybe_unused]] double vpre = 0.0, ppre = 0.0;
        prio_begin();
        if (i < a1) {
            const double xi = __builtin_fma(px[i], invL, 0.5), yi = __builtin_fma(py[i], invL, 0.5), zi = __builtin_fma(pz[i], invL, 0.5), isi = rho[i];
            const int c = cnt[i];
            if constexpr (SPREAD) { if (fuse && sub < 3) { vpre = vx.ptr()[sub * NMAX + i]; ppre = px.ptr()[sub * NMAX + i]; } } // (as pair_loop)
            const int mine = (c - sub + TPA - 1) / TPA;
            unsigned long long wn = nb64[(size_t)lrow(i) * TPA + sub];
            for (int k0 = 0; k0 < mine; k0 += 8) {
                const unsigned long long wd = wn; // (the next word one word ahead, as in pair_loop)
                wn = nb64[((size_t)(min(k0 + 8, MAXNB / TPA - 8) >> 3) * C::NLIST + lrow(i)) * TPA + sub];
#pragma unroll
                for (int e0 = 0; e0 < 8; e0 += W) {
                    if (k0 + e0 == PRIO_SW) prio_swap();
                    if (k0 + e0 < mine) {
                        double dx[W], dy[W], dz[W], r2[W], y[W], t[W], hh[W], rj[W], q2[W], rm[W], rn[W], fp[W];
                        bool in[W];
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            in[u] = (k0 + e0 + u) < mine;
                            const int j = in[u] ? (int)((wd >> (8 * (e0 + u))) & 0xFFull) : i;
                            dx[u] = __builtin_fma(-px[j], invL, xi); dy[u] = __builtin_fma(-py[j], invL, yi); dz[u] = __builtin_fma(-pz[j], invL, zi);
                            rj[u] = rho[j];
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            dx[u] = __builtin_fma(__builtin_amdgcn_fract(dx[u]), L, mhL); dy[u] = __builtin_fma(__builtin_amdgcn_fract(dy[u]), L, mhL);
                            dz[u] = __builtin_fma(__builtin_amdgcn_fract(dz[u]), L, mhL);
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            r2[u] = dx[u] * dx[u] + dy[u] * dy[u] + dz[u] * dz[u];
                            in[u] = in[u] && r2[u] < rc2;
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) y[u] = __builtin_amdgcn_rsq(r2[u]);
#pragma unroll
                        for (int u = 0; u < 2 * W; ++u) { // two Newton steps per neighbour: y <- y + y (1/2 - (r2 y)(y/2))
                            const int v = u % W;
                            t[v] = r2[v] * y[v]; hh[v] = 0.5 * y[v];
                            t[v] = __builtin_fma(-t[v], hh[v], 0.5);
                            y[v] = __builtin_fma(y[v], t[v], y[v]);
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) y[u] = in[u] ? y[u] : 0.0; // 1 / r, or nothing at all
#pragma unroll
                        for (int u = 0; u < W; ++u) { t[u] = y[u] * y[u]; q2[u] = a2 * t[u]; } // t = 1/r^2, q2 = (a/r)^2
#pragma unroll
                        for (int u = 0; u < W; ++u) { rm[u] = q2[u] * q2[u] * q2[u]; }         // (a/r)^6
#pragma unroll
                        for (int u = 0; u < W; ++u) { rn[u] = rm[u] * (a1r * y[u]); }           // (a/r)^7
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            const double dF = 0.5 * cc * (isi + rj[u]);
                            fp[u] = eps * (7.0 * rn[u] - 6.0 * dF * rm[u]) * t[u];
                        }
#pragma unroll
                        for (int u = 0; u < W; ++u) {
                            ax += dx[u] * fp[u]; ay += dy[u] * fp[u]; az += dz[u] * fp[u];
                            if (WANT_E) { e += eps * rn[u]; w += r2[u] * fp[u]; np += in[u] ? 1.0 : 0.0; }
                        }
                    }
                }
            }
        }
        prio_end();
        ax = group_sum(ax); ay = group_sum(ay); az = group_sum(az);
        if (WANT_E) { e = group_sum(e); w = group_sum(w); np = group_sum(np); }
        if constexpr (SPREAD) {
            const double sy = dpp_mov<0x111>(ay), sz = dpp_mov<0x112>(az); // (every lane, outside any branch)
            const double ac = sub == 0 ? ax : sub == 1 ? sy : sz;
            if (i < a1 && sub == 0) { eacc += e; wacc += w; nacc += np; }
            if (i < a1 && sub < 3) {
                const int ci = sub * NMAX + i;
                if (WANT_E && fuse) {
                    const double u = __builtin_fma(dtfm, ac, vpre);
                    vx.ptr()[ci] = u;
                    kacc += p.mass * (u * u);
                    fx.ptr()[ci] = ac;
                } else if (!WANT_E && fuse) { // (the densities' exchange took generation gen - 1)
                    double u = __builtin_fma(dtfm, ac, vpre);
                    u = __builtin_fma(dtfm, ac, u);
                    vx.ptr()[ci] = u;
                    const double pn = __builtin_fma(h, u, ppre);
                    if (Q > 1) put_granule(xb + (size_t)(gen & 1) * C::XBUF_DOUBLES + 2 * (size_t)ci, pn, my_magic());
                    fx.ptr()[ci] = pn;
                } else fx.ptr()[ci] = ac;
            }
            continue;
        }
        if (i < a1 && sub == 0) {
            eacc += e; wacc += w; nacc += np;
            if (WANT_E && fuse) { // the energy evaluation that ends a trajectory: final_integrate on the spot, kinetic energy along
                const double ux = __builtin_fma(dtfm, ax, vx[i]), uy = __builtin_fma(dtfm, ay, vy[i]), uz = __builtin_fma(dtfm, az, vz[i]);
                vx[i] = ux; vy[i] = uy; vz[i] = uz;
                kacc += p.mass * (ux * ux + uy * uy + uz * uz);
            }
            if (!WANT_E && fuse) { // integrate and publish on the spot, as pair_loop does (the densities' exchange took generation gen-1)
                double ux = __builtin_fma(dtfm, ax, vx[i]), uy = __builtin_fma(dtfm, ay, vy[i]), uz = __builtin_fma(dtfm, az, vz[i]);
                ux = __builtin_fma(dtfm, ax, ux); uy = __builtin_fma(dtfm, ay, uy); uz = __builtin_fma(dtfm, az, uz);
                vx[i] = ux; vy[i] = uy; vz[i] = uz;
                ax = __builtin_fma(h, ux, px[i]); ay = __builtin_fma(h, uy, py[i]); az = __builtin_fma(h, uz, pz[i]);
                if (Q > 1) {
                    double *xp = xb + (size_t)(gen & 1) * C::XBUF_DOUBLES;
                    const unsigned long long mgp = my_magic();
                    put_granule(xp + 2 * (size_t)i, ax, mgp);
                    put_granule(xp + 2 * (size_t)(NMAX + i), ay, mgp);
                    put_granule(xp + 2 * (size_t)(2 * NMAX + i), az, mgp);
                }
            }
            fx[i] = ax; fy[i] = ay; fz[i] = az;
        }
    }
    // the caller halves the summed energy (pair terms are counted twice in a full list): fold the embedding term in as -2 eps c sqrt(rho)
    eacc -= 2.0 * eps * cc * sq_own;
}

// Residency census (clusters only).  The workgroups of a cluster spin on one another, so the whole grid must be resident at once;
// HIP promises nothing of the kind, and a CU taken by another process, a CU mask or a second stream would leave part of the grid
// queued behind workgroups that wait for it.  Every workgroup signs in on one counter and waits (bounded: 200 us) until all have;
// whoever gives up sets an abort bit (by compare-and-swap on the short count it saw) that also fails every later arrival, so the
// verdict is unanimous.  Returns false when the
// grid did not gather; the caller has touched nothing by then.
// A grid of more clusters than the chip holds at once (p.over: twice as many, launched longest block first — the replicas of an 8^3
// grid take 31-73 ms per block at two workgroups each, so a resident grid idles a third of the time behind its slowest member)
// cannot gather as a whole and need not: a cluster only waits for its own members, which the dispatcher places one after the other
// (workgroups are handed to an XCD in index order: the next cluster's Q workgroups get the Q CUs that a finished cluster frees), so
// there every cluster takes its own census on its own counter, with the same bound and the same unanimous verdict.
template <class C>
__device__ __forceinline__ bool residency_census(const KParams &p, int cluster)
{
    constexpr unsigned int ABORT = 0x80000000u;
    int *flag = (int *)(nm_lds + C::OFF_RED);
    if (threadIdx.x == 0) {
        unsigned int *const counter = p.over ? p.census + 1 + cluster : p.census;
        const unsigned int full = (p.over ? p.census_cbase + (unsigned int)p.cus : p.census_base + gridDim.x);
        const unsigned int want = p.inj_census ? full + 1u : full; // (injection: a count nobody can complete)
        atomicAdd(counter, 1u);
        const unsigned long long t0 = wall_clock64();
        int ok = 0;
        for (;;) {
            const unsigned int v = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v == want) { ok = 1; break; }
            if (v & ABORT) break;
            if (wall_clock64() - t0 > 20000ull) { // 200 us of the 100 MHz clock: give up — but only on a count that is still short.
                // compare-and-swap, not an unconditional OR: had the last workgroup signed in meanwhile, an abort bit set on top of the
                // full count would fail the workgroups that have not looked yet while this one passes
                if (atomicCAS(counter, v, v | ABORT) == v) break;
                continue; // the counter moved: look again (a full count passes, somebody else's abort fails)
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *flag = ok;
    }
    __syncthreads();
    const int ok = *flag;
    __syncthreads(); // the slot is reused by the reductions
    return ok != 0;
}

// nm_create's residency probe: a grid of the block kernel's shape (same workgroup size, same LDS request — the LDS alone admits one
// workgroup per CU for every configuration, see the static_assert) that only takes the census.
template <class C>
__global__ void __launch_bounds__(C::BLOCK) nm_probe_kernel(const KParams p)
{
#if NM_AB != 4
    static_assert(2 * C::LDS_BYTES > 160 * 1024, "the probe stands in for the block kernel only while LDS limits both to one workgroup per CU");
#endif
    const int Q = p.cus, b = blockIdx.x;
    const int slot = (b & 7) + 8 * ((b >> 3) / Q); // (the block kernel's mapping; a padding workgroup takes the census and leaves)
    if (p.over && slot >= p.nslots) return;        // (clusters that do not exist have no census of their own)
    if (!residency_census<C>(p, slot) && threadIdx.x == 0 && slot < p.nslots) atomicOr(&p.status[slot], (int)ST_NOT_RESIDENT);
}

// Phases of the per-replica state machine.  The block kernel is written so that eval() — by far the largest
// piece of code and the only one whose cost matters — has exactly ONE call site; every move is split into the
// part before its energy/force evaluation and the part after it.
enum : int { PH_INIT = 0, PH_BULK = 1, PH_VMC = 2, PH_HMC_START = 3, PH_HMC_STEP = 4, PH_ITER_END = 5 }; // PH_HMC_STEP: the trajectory's last
// evaluation; PH_ITER_END: the evaluation that closes an iterative position move of the EAM (its virial is not carried through the trials)

// one workgroup = one replica for MOD moves
#if NM_AB == 4
#define NM_MIN_WAVES , 2 // (256-thread workgroups: without it the compiler takes the 512 registers a lone wave per SIMD may have)
#else
#define NM_MIN_WAVES
#endif
// The block of one workgroup: returns 0 when the replica's block completed and was stored, 1 when this workgroup had nothing to run (halted
// queue, padding workgroup, a slot the re-issue mask leaves out, nm_eval), 2 when the block stopped on an error (reported; state untouched).
// CENSUS: the launch's residency census is taken here (nm_cycles_kernel takes it itself, once, in front of its cycles).
template <class C, bool CENSUS = true>
__device__ __forceinline__ int nm_block_body(const KParams &p)
{
    constexpr bool first = CENSUS;
    constexpr int BLOCK = C::BLOCK;
    const unsigned long long t_entry = wall_clock64(); // (stats column 4)
    // Cluster mapping: the Q members of a cluster share blockIdx % 8, i.e. one XCD (and its L2) under the observed round-robin
    // placement — slot x + 8 s takes the workgroups x + 8 (s Q + q), q < Q.  The grid is 8 Q ceil(nslots / 8) workgroups (nm_grid):
    // when 8 does not divide nslots the workgroups of the slots that do not exist only take the census and leave.  (Round 2 mapped
    // such grids slot = b / Q, which spread every cluster over all XCDs: write-through hand-overs, ~0.5 us more per hop.)
    const int Q = p.cus, b = blockIdx.x;
    const int r_ = b >> 3, qq = r_ % Q;
    const int cluster = (b & 7) + 8 * (r_ / Q);    // (Q = 1: b itself)
    const int slot = (p.order && cluster < p.nslots) ? p.order[cluster] : cluster; // order: the replicas with the longest blocks first
    if (halted(p)) return 1; // an earlier block stopped on an error: nothing runs on its state until the host has dealt with it
    if (slot >= p.nslots) { // padding workgroup
        if (first && Q > 1 && p.census && !p.over) (void)residency_census<C>(p, cluster);
        return 1;
    }
    const int buf = p.slot2buf[slot];
    const int tid = threadIdx.x;
    Replica<C> R(p, slot, qq);
    const bool writer = (tid == 0 && qq == 0); // one workgroup of the cluster writes the replica's results
    const int N = p.N;
    // status[] holds the bits of the LAST launch that ran: the writer, the only thread that ever reports for this slot, clears it first
    // (a memset in front of every launch was one more dependent operation on the stream per cycle).  A launch that found the halt word
    // armed has returned above and leaves the failed block's bits where they are.
    if (writer && !(p.rerun_mask && !p.rerun_mask[slot])) p.status[slot] = 0;

    if (first && Q > 1 && p.census && !residency_census<C>(p, cluster)) { // nothing has been touched yet
        if (writer) report_status(p, slot, ST_NOT_RESIDENT, true);
        return 2;
    }
    if (p.rerun_mask && !p.rerun_mask[slot]) return 1; // re-issue of a block: this replica completed it the first time
    // Which XCD is this workgroup on?  HIP promises no placement; blockIdx % 8 is only the observed round-robin.  The members of a
    // cluster exchange their XCC ids once per block (write-through granules, valid anywhere): sum and sum of squares over the Q
    // members tell every member, identically, whether all ids are equal (Q sum(id^2) == (sum id)^2).
    if (Q > 1 && p.plain_granules) {
        const int xcc = (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu); // HW_REG_XCC_ID[3:0]
        double s4[4] = { (double)xcc, (double)(xcc * xcc), 0.0, 0.0 };
        R.template exchange_sums<4>(s4);
        R.same_xcd = ((double)Q * s4[1] == s4[0] * s4[0]);
    }
#ifdef NM_EXPERIMENT
    const unsigned long long clk_c0 = __builtin_readcyclecounter(), clk_w0 = wall_clock64();
#endif
    R.load(buf);
    // init_lammps (remcmc:459-470): change_box %f, scatter x, v, run 0.  nm_eval uses the box as given.
    R.L = uniform(p.eval_only ? p.box[buf] : q6(p.box[buf]));
    R.wrap();

    // uniform scalars that are touched once per move live in LDS (Replica::ust), not in scalar registers
    double &et = R.ust(0), &pf = R.ust(1), &t = R.ust(2), &dx = R.ust(3), &dv = R.ust(4), &dt = R.ust(5);
    double &ntp = R.ust(6), &nap = R.ust(7), &ntv = R.ust(8), &nav = R.ust(9), &nth = R.ust(10), &nah = R.ust(11);
    double &U0 = R.ust(12), &W0 = R.ust(13), &c_pe = R.ust(14), &c_vol = R.ust(15), &c_volnew = R.ust(16), &c_boxl = R.ust(17);
    et = p.et[slot]; pf = p.pf[slot]; t = p.tq[slot];
    dx = p.steps[3 * buf]; dv = p.steps[3 * buf + 1]; dt = p.steps[3 * buf + 2];
    ntp = p.count[6 * slot]; nap = p.count[6 * slot + 1]; ntv = p.count[6 * slot + 2];
    nav = p.count[6 * slot + 3]; nth = p.count[6 * slot + 4]; nah = p.count[6 * slot + 5];
    U0 = 0.0; W0 = 0.0; c_pe = 0.0; c_vol = 0.0; c_volnew = 0.0; c_boxl = 0.0;
    double &nth_entry = R.ust(22); // (stats column 7)
    nth_entry = nth;
    const int fatal = ST_BOX_TOO_SMALL | ST_LIST_OVERFLOW | ST_SYNC_TIMEOUT;

    // state carried across the evaluation of a move
    int phase = PH_INIT, m = 0;
    bool skip_eval = false;
    double &mv2new = R.ust(18); // kinetic-energy sum delivered with the last evaluation of a trajectory
    mv2new = 0.0;
    bool have_need = false, pre_need = false; // rebuild decision delivered with a position hand-over (HMC steps)
    double &c_h = R.ust(19), &c_dtfm = R.ust(20), &mv2_0 = R.ust(21);
    c_h = 0.0; c_dtfm = 0.0; mv2_0 = 0.0;

    for (;;) {
        const int st_before = R.status;
        if (!skip_eval) {
            // The last evaluation of a trajectory (the only one in phase PH_HMC_STEP) makes the final half kick in its pair-loop
            // epilogue and sums the kinetic energy with U and W.
            R.eval(have_need, pre_need, phase == PH_HMC_STEP, c_dtfm);
            // cluster-wide U, W of an energy evaluation — ONE exchange site; the kinetic energy rides along.
            if (!(st_before & fatal) && !(R.status & (ST_BOX_TOO_SMALL | ST_SYNC_TIMEOUT)))
                mv2new = R.finish_sums(phase == PH_HMC_STEP ? R.psum[3] : 0.0,
                                       phase == PH_HMC_START || p.eval_only ? -1 : phase == PH_INIT ? m : m + 1); // the move that follows this evaluation
        }
        skip_eval = false;
        have_need = false;
        if (R.status & fatal) {
            if (R.Q > 1 && (R.status & ST_LIST_OVERFLOW)) R.publish_poison();
            break;
        }
#ifdef NM_PROF
        unsigned long long &prof_t0 = R.prof_t0; unsigned long long (&prof_acc)[NM_PROF_SLOTS] = R.prof_acc;
        const int prof_phase = phase;
#endif
        PROF_BEGIN();

        // ---------------- part of the move after its evaluation
        bool move_done = false, acc = false;
        double crit = 0.0, branch = 0.0;
        if (phase == PH_INIT) {
            if (p.eval_only) { // nm_eval: batched lj_energy_force on the resident states
                if (writer) {
                    p.evalU[slot] = R.U; p.evalW[slot] = R.W; report_status(p, slot, R.status, false);
                    double *st = p.stats + NM_STATS_COLS * (size_t)slot;
                    st[0] += R.st_evals; st[1] += R.st_rebuilds; st[2] += R.st_eevals; st[3] += R.st_pairs;
                }
                if (p.evalF) // every workgroup holds the forces of its own atoms
                    for (int a = 3 * R.a0 + tid; a < 3 * R.a1; a += BLOCK) {
                        const int i = a / 3, c = a - 3 * i;
                        p.evalF[(size_t)slot * 3 * N + a] = (c == 0 ? R.fx : c == 1 ? R.fy : R.fz)[i];
                    }
                return 1;
            }
        } else if (phase == PH_ITER_END) { // iter_position_mc of the EAM: U, W of the final configuration are in (the reference's last `run 0`)
            if (p.trace && writer) {
                double *tr = p.trace + ((size_t)slot * p.mod + m) * 4;
                tr[0] = 3.0; tr[1] = c_vol; tr[2] = c_volnew; tr[3] = R.U;
            }
            ++m;
        } else if (phase == PH_BULK) { // bulk_position_mc, remcmc:485-500
            const double penew = R.U / et;
            crit = penew - c_pe;
            acc = R.metropolis(crit, S_ACC, (uint32_t)m, 0);
            if (acc) nap += 1.0;
            else { R.restore(false); R.wrap(); R.U = U0; R.W = W0; }
            branch = 0.0; move_done = true;
        } else if (phase == PH_VMC) { // volume_mc, remcmc:574-593
            const double penew = R.U / et;
            crit = (penew - c_pe) + pf * (c_volnew - c_vol) - (double)(N + 1) * log(c_volnew / c_vol); // remcmc:576
            acc = R.metropolis(crit, S_ACC, (uint32_t)m, 0);
            if (acc) nav += 1.0;
            else { R.L = uniform(q6(c_boxl)); R.restore(false); R.wrap(); R.U = U0; R.W = W0; }
            branch = 1.0; move_done = true;
        } else if (phase == PH_HMC_START) { // hamiltonian_mc after its "run 0", remcmc:609-616
            R.save(true);
            U0 = R.U; W0 = R.W;
            c_pe = R.U / et + 0.5 * p.mvv2e * mv2_0 / et; // etot; sum m v.v came with the velocities
            PROF_END(5 + prof_phase);
            // `run NSTPS` (fix nve): initial_integrate of step 1, then the NSTPS - 1 force-only evaluations, each of which
            // integrates in its pair loop (final_integrate of its step + initial_integrate of the next) and hands the new
            // positions over.  The NSTPS-th evaluation also wants the energy: it is the generic one at the top of this loop.
            const double dtfm_ = uniform(c_dtfm), h_ = uniform(c_h);
            PROF_BEGIN();
            pre_need = R.advance_and_share(1, dtfm_, h_);
            PROF_END(9);
            const int nfo = p.nstps - 1;
            for (int s = 0; s < nfo; ++s) {
                if (__builtin_amdgcn_readfirstlane(R.status) & fatal) break;
                R.eval_force(pre_need, dtfm_, h_);
                if (__builtin_amdgcn_readfirstlane(R.status) & fatal) break; // own list overflowed: what the pair loop published is poisoned
                PROF_BEGIN();
                pre_need = R.advance_and_share(C::HALF ? 2 : 3, dtfm_, h_);
                PROF_END(9);
            }
            have_need = true;
            phase = PH_HMC_STEP;
            continue;
        } else { // PH_HMC_STEP: the trajectory is complete, energies at its end are in
            const double etotnew = R.U / et + 0.5 * p.mvv2e * mv2new / et; // remcmc:618-622
            crit = etotnew - c_pe;
            if (p.md_mode) acc = true; // plain NVE run (init_sample -is, remcmc:421-425): nothing to accept
            else {
                acc = R.metropolis(crit, S_ACC, (uint32_t)m, 0);
                if (acc) nah += 1.0;
                else { R.restore(true); R.wrap(); R.U = U0; R.W = W0; }
            }
            branch = 2.0; move_done = true;
        }
        if (move_done) {
            if (p.trace && writer) {
                double *tr = p.trace + ((size_t)slot * p.mod + m) * 4;
                tr[0] = branch; tr[1] = acc ? 1.0 : 0.0; tr[2] = crit; tr[3] = R.U;
            }
            ++m;
        }

        PROF_END(5 + prof_phase);
        // ---------------- start moves until one needs an evaluation (move_mc, remcmc:643-658)
        bool pending = false;
        while (m < p.mod && !pending) {
            PROF_BEGIN();
            const double roll = p.md_mode ? 2.0 : R.draw_scalar(S_ROLL, (uint32_t)m, 0);
            if (roll <= p.ppos && p.bulk) { // bulk_position_mc, remcmc:477-484
                ntp += 1.0;
                R.save(false);
                U0 = R.U; W0 = R.W; c_pe = R.U / et;
                const uint32_t tag = R.draw_tag((uint32_t)m);
                const double a = q6(dx * p.lat);
                for (int i = tid; i < N; i += BLOCK) { // displace_atoms all random a a a seed units box
                    uint32_t o[4], q[4];
                    philox4x32_10((uint32_t)i, S_DISP_XY, tag, p.step, p.seed, (uint32_t)R.gslot, o);
                    philox4x32_10((uint32_t)i, S_DISP_Z, tag, p.step, p.seed, (uint32_t)R.gslot, q);
                    R.px[i] += a * 2.0 * (u01(o[0], o[1]) - 0.5);
                    R.py[i] += a * 2.0 * (u01(o[2], o[3]) - 0.5);
                    R.pz[i] += a * 2.0 * (u01(q[0], q[1]) - 0.5);
                }
                R.set_fresh(false);
                R.wrap();
                phase = PH_BULK; pending = true;
                PROF_END(10);
            } else if (roll <= p.ppos) { // iter_position_mc: local energy differences, no full evaluation
                double c2 = 0.0;
                const int na = R.iter_pmc((uint32_t)m, et, dx, ntp, nap, c2);
                if (__builtin_amdgcn_readfirstlane(R.status) & fatal) break; // (its exchange of energy differences failed)
                if constexpr (C::POT == 1) { // close the move with a full evaluation (c_vol, c_volnew are free during a position move)
                    c_vol = (double)na; c_volnew = c2;
                    phase = PH_ITER_END; pending = true;
                } else {
                    if (p.trace && writer) {
                        double *tr = p.trace + ((size_t)slot * p.mod + m) * 4;
                        tr[0] = 3.0; tr[1] = (double)na; tr[2] = c2; tr[3] = R.U;
                    }
                    ++m;
                }
                PROF_END(13);
            } else if (roll <= p.ppos + p.pvol) { // volume_mc, remcmc:552-573
                ntv += 1.0;
                c_boxl = R.L; c_vol = uniform(pow(c_boxl, 3.0));
                R.save(false);
                U0 = R.U; W0 = R.W; c_pe = R.U / et;
                const double u = R.draw_scalar(S_VOL, (uint32_t)m, 0);
                c_volnew = uniform(exp(log(c_vol) + 2.0 * (u - 0.5) * dv));
                const double boxnew = cbrt(c_volnew);
                const double scale = boxnew / c_boxl;
                for (int i = tid; i < N; i += BLOCK) { R.px[i] = scale * R.sx[i]; R.py[i] = scale * R.sy[i]; R.pz[i] = scale * R.sz[i]; }
                R.set_fresh(false);
                R.L = uniform(q6(boxnew)); // change_box ... %f
                R.wrap();
                phase = PH_VMC; pending = true;
                PROF_END(11);
            } else { // hamiltonian_mc, remcmc:598-608
                if (!p.md_mode) nth += 1.0;
                const uint32_t tag = R.draw_tag((uint32_t)m);
                mv2_0 = R.hmc_velocities(q6(t), tag, R.gauss_for == m ? 1 : 0);
                c_h = uniform(q6(dt)); // timestep %f
                c_dtfm = 0.5 * c_h * p.ftm2v / p.mass;
                R.wrap(); // run 0
                // (mapping) velocity_create and wrap wrote atom i on thread i mod BLOCK; save() and the first kick take the own atoms
                // a0 + tid
                if (R.Q > 1) __syncthreads();
                phase = PH_HMC_START; pending = true;
                skip_eval = R.fresh(); // nothing moved since the last evaluation: same U, W, f
                PROF_END(12);
            }
        }
        if (!pending) break;
    }

#ifdef NM_EXPERIMENT
    if (R.tl && tid == 0 && qq == 0) { // shader cycles and 100 MHz ticks of the whole block: the clock the chip held
        R.tl[(size_t)8 * 8 * 512 * 8 - 2] = __builtin_readcyclecounter() - clk_c0;
        R.tl[(size_t)8 * 8 * 512 * 8 - 1] = wall_clock64() - clk_w0;
    }
#endif
    // lammps_extract (remcmc:377-391) and the acceptance ratios (remcmc:685-688)
    double smv2;
    {
        // (mapping) restore() after a rejected last move wrote v[i] on thread i mod BLOCK; own_mv2() reads atom a0 + tid.  Unordered in
        // round 2: a once-in-ten-runs mismatch of the temp / ke columns of one slot, 8e-5 relative, in
        // test_block_parity_large_cells[6-4-True] — a stale end-of-trajectory velocity in the sum; x, v and pe were never affected.
        __syncthreads();
        double k4[4] = { R.own_mv2(), 0.0, 0.0, 0.0 };
        if (!(R.status & fatal)) R.template exchange_sums<4>(k4); // (a cluster that is leaving on an error is no longer in step)
        smv2 = k4[0];
    }
    // A block that ends on an error leaves the replica's state in HBM as it was when the block started (nothing below runs): the
    // host reports the reason, and a caller that can cure it (fewer workgroups per replica after a hand-over timeout) may
    // re-issue the block.
    if (R.status & fatal) {
        if (writer) report_status(p, slot, R.status, true);
        return 2;
    }
    R.store(buf);
    { // the longest list row any workgroup of the replica built (stats column 8; positive doubles order like their bit patterns)
        int mc = R.st_maxc;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mc = max(mc, __shfl_xor(mc, d, 64));
        if ((tid & 63) == 0 && mc > 0)
            atomicMax((unsigned long long *)(p.stats + NM_STATS_COLS * (size_t)slot + 8), (unsigned long long)__double_as_longlong((double)mc));
    }
    if (writer) {
        const double dof = 3.0 * N - 3.0;
        const double temp = smv2 * p.mvv2e / (dof * p.kB);
        const double vol3 = R.L * R.L * R.L;
        double *th = p.therm + 5 * (size_t)buf;
        th[0] = temp;
        th[1] = R.U;
        th[2] = 0.5 * p.mvv2e * smv2;
        th[3] = (dof * p.kB * temp + R.W) / 3.0 * (1.0 / vol3) * p.nktv2p;
        th[4] = pow(R.L, 3.0);
        p.box[buf] = R.L;
        double *c = p.count + 6 * (size_t)slot;
        c[0] = ntp; c[1] = nap; c[2] = ntv; c[3] = nav; c[4] = nth; c[5] = nah;
        float *r = p.ratio + 3 * (size_t)slot;
        r[0] = (ntp > 0.0) ? (float)nap / (float)ntp : 0.0f;
        r[1] = (ntv > 0.0) ? (float)nav / (float)ntv : 0.0f;
        r[2] = (nth > 0.0) ? (float)nah / (float)nth : 0.0f;
        if (R.tape && R.tpos > R.tlen) R.status |= ST_TAPE_EXHAUSTED;
        report_status(p, slot, R.status, false);
        double *st = p.stats + NM_STATS_COLS * (size_t)slot;
        st[0] += R.st_evals; st[1] += R.st_rebuilds; st[2] += R.st_eevals; st[3] += R.st_pairs;
        const unsigned long long ticks = wall_clock64() - t_entry;
        if (p.last_ticks) p.last_ticks[slot] = ticks;
        st[4] += (double)ticks; st[5] += R.same_xcd ? 1.0 : 0.0; st[6] += 1.0; st[7] += nth - nth_entry; st[9] = (double)C::MAXNB;
#if NM_AB == 8 // dev probe (scripts/probe_rounds.py): when did this slot's block start (100 MHz ticks)
        st[9] = (double)t_entry;
#endif
#ifdef NM_PROF
        if (p.prof) for (int q = 0; q < NM_PROF_SLOTS; ++q) p.prof[(size_t)slot * NM_PROF_SLOTS + q] += R.prof_acc[q];
#endif
    }
    return 0;
}

template <class C>
__global__ void __launch_bounds__(C::BLOCK NM_MIN_WAVES) nm_block_kernel(const KParams p)
{
    (void)nm_block_body<C>(p);
}

// Launch order for grids with more one-workgroup replicas than the chip holds at once (the reference's run.sh setting: 1024 replicas
// on 256 CUs): workgroups are dispatched in index order, and a launch ends with whatever the last wave of workgroups happens to hold.
// Longest-processing-time-first by the duration of each slot's PREVIOUS block (a slot's cost changes slowly: it is set by its pressure
// and temperature) shortens the tail: order[rank] = slot, rank by descending ticks, ties by index.  Scheduling only: results do not
// depend on it.
__global__ void nm_order_kernel(int nslots, const unsigned long long *ticks, int *order)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nslots; k += gridDim.x * blockDim.x) {
        const unsigned long long t = ticks[k];
        int rank = 0;
        for (int j = 0; j < nslots; ++j) {
            const unsigned long long u = ticks[j];
            rank += (u > t || (u == t && j < k)) ? 1 : 0;
        }
        // Workgroup (cluster) c runs on XCD c % 8, whose CUs it cannot leave: dealt out 0..7, 0..7, ... the XCDs' shares of the sorted
        // sequence would differ systematically (XCD 0 the longest of every eight); 0..7, 7..0, ... evens them out
        const int g = rank >> 3, pos = rank & 7;
        order[(g & 1) && 8 * g + 8 <= nslots ? 8 * g + 7 - pos : rank] = k;
    }
}

// nm_snapshot: everything a recorded cycle writes, copied in ONE launch (seven stream-ordered memcpys cost seven dispatch gaps per cycle)
struct SnapArgs { const double *x, *box, *therm, *steps, *count; const float *ratio; const int *slot2buf; double *dst; size_t off[7], n[7]; };
__global__ void nm_snapshot_kernel(const SnapArgs a)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    const double *src[5] = { a.x, a.box, a.therm, a.steps, a.count };
#pragma unroll
    for (int q = 0; q < 5; ++q)
        for (size_t i = t; i < a.n[q]; i += nt) a.dst[a.off[q] + i] = src[q][i];
    float *rf = (float *)(a.dst + a.off[5]);
    for (size_t i = t; i < a.n[5]; i += nt) rf[i] = a.ratio[i];
    int *mi = (int *)(a.dst + a.off[6]);
    for (size_t i = t; i < a.n[6]; i += nt) mi[i] = a.slot2buf[i];
}

// gen_mc_param (remcmc:726-745) of one slot
__device__ __forceinline__ void adapt_slot(int k, const int *slot2buf, double *steps, double *count, float *ratio)
{
    const int b = slot2buf[k];
    for (int c = 0; c < 3; ++c) {
        const float a = ratio[3 * k + c];
        double s = steps[3 * b + c];
        if (a < 0.5f) s = 0.9375 * s;
        if (a > 0.5f) s = 1.0625 * s;
        steps[3 * b + c] = s;
        ratio[3 * k + c] = 0.0f;
    }
    for (int c = 0; c < 6; ++c) count[6 * k + c] = 0.0;
}

// replica_exchange (remcmc:776-803) of one local pressure row r; returns the accepted swaps
__device__ __forceinline__ int exchange_row(int r, int nt, int row0, uint32_t seed, uint32_t step, int *slot2buf, const double *therm, const double *et,
                                            const double *pf, const double *tape, double *crit_out)
{
    const int ppr = nt * (nt - 1) / 2;
    int q = 0, sw = 0;
    for (int vv = nt - 1; vv >= 0; --vv)
        for (int w = 0; w < vv; ++w, ++q) {
            const int i = r * nt + vv, j = r * nt + w;
            const int bi = slot2buf[i], bj = slot2buf[j];
            const double de = (therm[5 * bi + 1] + therm[5 * bi + 2]) - (therm[5 * bj + 1] + therm[5 * bj + 2]);
            const double dvol = therm[5 * bi + 4] - therm[5 * bj + 4];
            const double dh = de * (1.0 / et[i] - 1.0 / et[j]) + (pf[i] - pf[j]) * dvol;
            double u;
            if (tape) u = tape[r * ppr + q];
            else {
                uint32_t o[4];
                philox4x32_10((uint32_t)((row0 + r) * ppr + q), S_EXCH, 0u, step, seed, 0xFFFFFFFFu, o);
                u = u01(o[0], o[1]);
            }
            if (crit_out) crit_out[r * ppr + q] = dh;
            const double e = exp(dh);
            const double mm = (e != e) ? e : (e < 1.0 ? e : 1.0);
            if (u <= mm) { slot2buf[i] = bj; slot2buf[j] = bi; ++sw; }
        }
    return sw;
}

// nm_run_cycles: `ncycles` cycles of the reference's main loop with outputs off (remcmc:977-995: gen_samples, gen_mc_params, replica_exchange) in
// ONE launch.  A cycle's block is nm_block_body as in nm_block_kernel; behind it the workgroups of a pressure ROW meet (every workgroup signs in on
// the row's counter with an agent-scope release, so that its stores — positions, velocities, thermo scalars, counters — are out of its XCD's L2), the
// row's leader (workgroup 0 of the row's first replica) adapts the row's step sizes and runs the row's exchange sweep — the arithmetic of
// nm_adapt_kernel and nm_exchange_kernel, the loads and the uniforms spread over its threads — and releases the row into its next cycle (agent-scope acquire in every wave: stale lines of the other
// XCDs' writes are dropped).  Rows never wait for one another — the exchange never leaves a row (remcmc:782-798) — so a launch of K cycles lasts as
// long as its slowest ROW's K blocks, not K times the slowest replica of the grid: the number of trajectories a block draws is Binomial(128, 3/4),
// i.e. a block's time scatters by 3.5 %, and a launch of 64 replicas waited for the slowest of 64 at every cycle.  Spins are bounded (2 s) and leave
// through cyc_abort, which also whoever stops on an error sets: nobody waits for a row that will not complete.  Requires whole rows, no launch
// order table, no tapes, no trace (nm_api.hip nm_run_cycles falls back to single cycles otherwise).
template <class C>
__global__ void __launch_bounds__(C::BLOCK NM_MIN_WAVES) nm_cycles_kernel(const KParams p0)
{
    const int ncycles = p0.ncycles;
    const uint32_t step0 = p0.step, id0 = p0.launch_id;
    {   // the launch's residency census, once, as nm_block_body takes it (halt word first; a slot's status cleared by its writer; nothing touched before it)
        const int Q = p0.cus, b = blockIdx.x;
        const int r_ = b >> 3, qq = r_ % Q;
        const int slot = (b & 7) + 8 * (r_ / Q); // (nm_block_body's mapping, without a launch-order table)
        const bool real = slot < p0.nslots, writer = threadIdx.x == 0 && qq == 0;
        if (halted(p0)) return;
        if (Q > 1 && p0.census) {
            if (!real) { (void)residency_census<C>(p0, slot); return; } // padding workgroups take the census and leave
            if (writer) p0.status[slot] = 0;
            if (!residency_census<C>(p0, slot)) {
                if (writer) report_status(p0, slot, ST_NOT_RESIDENT, true);
                return;
            }
        } else if (!real) return;
    }
#pragma clang loop unroll(disable)
    for (int cyc = 0; cyc < ncycles; ++cyc) {
        // What keeps the block inside this loop as fast as in nm_block_kernel (within 0.6 %; 3.3 % slower without): nothing that the code behind the
        // block needs is held ACROSS it.  The parameters are fetched from the kernel's argument segment anew in every cycle, the workgroup's place in the
        // grid is worked out again behind the block, the census sits in front of the loop (make resources: 48-65 spilled vector registers, 108-133 before)
        typedef const KParams __attribute__((address_space(4))) *kparams_ptr;
        kparams_ptr kp = (kparams_ptr)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        KParams pc = *(const KParams *)kp;
        const KParams &p = pc;                       // (this cycle's copy)
        pc.step = step0 + (uint32_t)cyc;
        pc.launch_id = id0 + (uint32_t)cyc; // (distinguishes the hand-over granules of successive blocks)
        const int rc = nm_block_body<C, false>(pc);
        // the workgroup's place in the grid, worked out HERE from opaque copies of its indices: computed once in front of the loop, these values (and
        // whatever the compiler derives from them for the code below) would be alive all the way through every block
        int b = blockIdx.x, tid = threadIdx.x;
        asm volatile("" : "+s"(b));
        asm volatile("" : "+v"(tid));
        const int nt = p.nt, Q = p.cus;
        const int r_ = b >> 3, qq = r_ % Q;
        const int slot = (b & 7) + 8 * (r_ / Q), row = slot / nt;
        const bool leader_wg = qq == 0 && slot == row * nt; // the workgroup that adapts and exchanges for its row
        const unsigned int per_cycle = (unsigned int)(nt * Q); // workgroups of a row
        if (rc != 0) {                               // stopped (or nothing to run): the rows must not wait for this workgroup
            if (tid == 0) __hip_atomic_store(p.cyc_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __syncthreads();                             // every thread's stores of this block are issued
        int abort_ = 0;
        volatile int *const flag = (volatile int *)(nm_lds + C::OFF_RED);
        unsigned long long t0 = 0;
        if (tid == 0) {
            (void)__hip_atomic_fetch_add(p.rowbar + row, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            t0 = wall_clock64();
        }
        if (leader_wg) {
            // ---- the row's leader workgroup: wait for the row, then gen_mc_params and replica_exchange of the row.  The sweep itself is
            // sequential (later pairs see earlier swaps), but nothing around it is: the row's energies, volumes and constants and the sweep's
            // uniforms are fetched / drawn by the workgroup's threads side by side into LDS (the block's arrays are dead by now), ONE thread
            // then walks the pairs on those (a single thread doing it all on global memory kept the row's CUs waiting ~0.1 ms per cycle)
            if (tid == 0) {
                const unsigned int want = (unsigned int)(cyc + 1) * per_cycle;
                for (;;) {
                    if (__hip_atomic_load(p.rowbar + row, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= want) break;
                    if (__hip_atomic_load(p.cyc_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abort_ = 1; break; }
                    if (wall_clock64() - t0 > 200000000ull) { // 2 s: a workgroup of the row never arrived
                        abort_ = 1; __hip_atomic_store(p.cyc_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        report_status(pc, slot, ST_SYNC_TIMEOUT, true);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
                *flag = abort_;
            }
            __syncthreads();
            abort_ = *flag;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __syncthreads();
            if (!abort_) {
                const int ppr = nt * (nt - 1) / 2, k0 = row * nt;
                double *const E = (double *)(nm_lds + C::OFF_POS), *const V = E + nt, *const IE = V + nt, *const PF = IE + nt, *const Uu = PF + nt;
                int *const Bf = (int *)(Uu + ppr);
                for (int l = tid; l < nt; l += C::BLOCK) {
                    const int bq = p.slot2buf[k0 + l];
                    E[l] = p.therm[5 * bq + 1] + p.therm[5 * bq + 2]; V[l] = p.therm[5 * bq + 4];
                    IE[l] = 1.0 / p.et[k0 + l]; PF[l] = p.pf[k0 + l]; Bf[l] = bq;
                }
                for (int l = tid; l < ppr; l += C::BLOCK) {
                    uint32_t o[4];
                    philox4x32_10((uint32_t)((p.row0 + row) * ppr + l), S_EXCH, 0u, pc.step, p.seed, 0xFFFFFFFFu, o);
                    Uu[l] = u01(o[0], o[1]);
                }
                for (int l = tid; l < 3 * nt; l += C::BLOCK) { // gen_mc_param (adapt_slot), one (slot, step size) per thread
                    const int k = k0 + l / 3, c3 = l % 3, bq = p.slot2buf[k];
                    const float a = p.ratio[3 * k + c3];
                    double sz = p.steps[3 * bq + c3];
                    if (a < 0.5f) sz = 0.9375 * sz;
                    if (a > 0.5f) sz = 1.0625 * sz;
                    p.steps[3 * bq + c3] = sz;
                    p.ratio[3 * k + c3] = 0.0f;
                }
                for (int l = tid; l < 6 * nt; l += C::BLOCK) p.count[6 * k0 + l] = 0.0;
                __syncthreads();
                if (tid == 0) { // the sweep of exchange_row on the row's copies: the same differences, the same uniforms, the same decisions
                    int q = 0, sw = 0;
                    for (int vv = nt - 1; vv >= 0; --vv)
                        for (int w = 0; w < vv; ++w, ++q) {
                            const double dh = (E[vv] - E[w]) * (IE[vv] - IE[w]) + (PF[vv] - PF[w]) * (V[vv] - V[w]);
                            const double e = exp(dh);
                            const double mm = (e != e) ? e : (e < 1.0 ? e : 1.0);
                            if (Uu[q] <= mm) {
                                const double te = E[vv], tv = V[vv]; const int tb = Bf[vv];
                                E[vv] = E[w]; V[vv] = V[w]; Bf[vv] = Bf[w];
                                E[w] = te; V[w] = tv; Bf[w] = tb;
                                ++sw;
                            }
                        }
                    if (sw && p.nswaps) atomicAdd(p.nswaps, sw);
                }
                __syncthreads();
                for (int l = tid; l < nt; l += C::BLOCK) const_cast<int *>(p.slot2buf)[k0 + l] = Bf[l];
                __syncthreads();
                if (tid == 0) __hip_atomic_store(p.rowgo + row, (unsigned int)(cyc + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (tid == 0) {
            if (!abort_)
                for (;;) {
                    if (__hip_atomic_load(p.rowgo + row, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned int)(cyc + 1)) break;
                    if (__hip_atomic_load(p.cyc_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { abort_ = 1; break; }
                    if (wall_clock64() - t0 > 400000000ull) { // 4 s: the row's leader never released the row
                        abort_ = 1; __hip_atomic_store(p.cyc_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        report_status(pc, slot, ST_SYNC_TIMEOUT, true);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            *flag = abort_;
        }
        __syncthreads();
        const int ab = *(volatile int *)(nm_lds + C::OFF_RED);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); // every wave: what the other XCDs wrote is re-read, not found stale in this one's caches
        __syncthreads();                                    // (the word in LDS is free again before the next block's reductions use it)
        if (ab) return;
    }
}

// gen_mc_param (remcmc:726-745): one thread per slot
__global__ void nm_adapt_kernel(int nslots, const int *slot2buf, double *steps, double *count, float *ratio, const int *halt)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nslots || (halt && *halt)) return; // (a block stopped on an error: its successors wait for the host, nm_api.hip settle)
    adapt_slot(k, slot2buf, steps, count, ratio);
}

// replica_exchange (remcmc:776-803): rows are independent, the sweep inside a row is strictly sequential.
// A swap exchanges entries [0..11] of the two state lists = configuration, thermo scalars and dx,dv,dt; here
// that is one swap of slot->buffer labels, no coordinate moves.
__global__ void __launch_bounds__(64) nm_exchange_kernel(int nrows, int nt, int row0, uint32_t seed, uint32_t step, int *slot2buf,
                                   const double *therm, const double *et, const double *pf, const double *tape,
                                   double *crit_out, int *nswaps, const int *halt)
{
    // ONE wave: rows r = lane, lane + 64, ...; the swap count is summed over the wave and stored (it used to be an atomicAdd onto a
    // word zeroed by a memset in front of every launch: one more dependent stream operation per cycle)
    if (halt && *halt) return;
    int sw = 0;
    for (int r = threadIdx.x; r < nrows; r += 64) sw += exchange_row(r, nt, row0, seed, step, slot2buf, therm, et, pf, tape, crit_out);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sw += __shfl_xor(sw, d, 64);
    if (threadIdx.x == 0) *nswaps = sw;
}

} // namespace nm
