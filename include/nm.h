/*
 * nm.h — C-ABI of the MI355X-native NPT-HMC + replica-exchange engine (libnm_hip.so).
 *
 * This is the drop-in boundary for the data-parallel hot path of walkernr/neuralMelting's
 * scripts/lammps_remcmc.py ("remcmc").  The reference has no FFI of its own for this path; the
 * seam it offers is the set of per-replica Python functions its orchestration maps over
 * (seam S-up).  Each entry point below replaces one of them, batched over all
 * replicas ("slots", k = i*NT + j, i = pressure index, j = temperature index, remcmc:117) held by
 * one GPU:
 *
 *   nm_create / nm_set_state   <- init_constants (remcmc:114-141) + the STATE lists (remcmc:432-433)
 *   nm_run_block               <- gen_samples -> gen_sample -> move_mc   (remcmc:643-719)
 *                                 incl. every LAMMPS call it makes        (remcmc:459-470, 477-640)
 *   nm_get_thermo / nm_get_state <- lammps_extract + the returned state   (remcmc:377-391, 681-691)
 *   nm_adapt                   <- gen_mc_params -> gen_mc_param           (remcmc:726-770)
 *   nm_exchange                <- replica_exchange                        (remcmc:776-803)
 *
 * Conventions: extern "C"; plain pointers and sizes; int status (0 = ok, <0 = error, text via
 * nm_last_error); no exceptions cross the boundary; the caller owns every host buffer, the
 * context owns all device memory, its stream and events; one context per GPU; calls on one
 * context are not re-entrant.  All arrays are C-contiguous; x and v are float64, xyz interleaved
 * in atom-ID order exactly as LAMMPS gather_atoms('x',1,3) returns them (remcmc:381-382), so a
 * .traj frame (remcmc:248-256) is a straight dump.
 */
#ifndef NM_H
#define NM_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_OK 0
#define NM_ERR_ARG (-1)       /* bad argument / configuration                              */
#define NM_ERR_HIP (-2)       /* HIP runtime error (no device, allocation, launch)        */
#define NM_ERR_STATE (-3)     /* a replica left the supported regime (see nm_last_error)  */
#define NM_ERR_UNSUPPORTED (-4)

#define NM_EL_LJ 0            /* units lj, fcc, lattice 1.122, mass 1     (remcmc:873-889) */
#define NM_EL_AL 1            /* units metal, fcc, 4.046 A, mass 29.982  (remcmc:880,886) */

#define NM_THERMO_COLS 17     /* temp pe ke virial vol dx dv dt ntp nap ntv nav nth nah ap av ah (remcmc:208) */
#define NM_TRACE_COLS 4       /* branch (0 bulk PMC, 1 VMC, 2 HMC, 3 iter PMC), accepted, criterion, U after */
#define NM_STATS_COLS 10      /* per slot: evaluations, list rebuilds, energy evaluations, sum of interacting pairs over those,
                                 time of the slot's blocks (ticks of the chip's 100 MHz clock, kernel entry to exit of the replica's
                                 first workgroup), blocks whose cluster handed over inside one XCD's L2, blocks run, HMC moves, longest neighbour-list
                                 row built (a maximum, not a sum), list slots per atom of the kernel in use */

typedef struct nm_ctx nm_ctx;

typedef struct nm_config {
    int32_t size;        /* sizeof(nm_config): ABI check                                        */
    int32_t element;     /* NM_EL_*  (-e, remcmc:58)                                            */
    int32_t natoms;      /* N = 4*SZ^3 for fcc (-ss, remcmc:60, 341-343)                        */
    int32_t np, nt;      /* global P x T grid (-pn, -tn)                                        */
    int32_t row0, nrows; /* pressure rows [row0,row0+nrows) owned by this context (sharding)   */
    int32_t nstps;       /* NSTPS (-ts)                                                         */
    int32_t bulk;        /* BM (-bm): 1 = bulk_position_mc, 0 = iter_position_mc                */
    int32_t iter_revert; /* 0 = reference behaviour of iter PMC, 1 = corrected      */
    int32_t device;      /* HIP device ordinal                                                  */
    uint32_t seed;       /* SEED (256 in the reference, remcmc:851)                             */
    double ppos, pvol;   /* PPOS, PVOL (-pm, -vm)                                               */
    const float *P;      /* [np] float32 pressure grid   (remcmc:895)                           */
    const float *T;      /* [nt] float32 temperature grid (remcmc:897)                          */
    int32_t slot0, nslots; /* optional: an arbitrary range of global slots k = i*nt + j instead of whole rows (nslots > 0
                              overrides row0/nrows).  A context that holds a partial row cannot run nm_exchange: the sweep
                              then spans contexts and is done by the host over RCCL (neuralmelting_amd/exchange.py).    */
} nm_config;

/* life cycle */
int nm_create(const nm_config *cfg, nm_ctx **out);
int nm_destroy(nm_ctx *ctx);
const char *nm_last_error(const nm_ctx *ctx);  /* ctx may be NULL: error of the last failed nm_create; cleared by the next synchronising call that returns NM_OK */
const char *nm_create_note(const nm_ctx *ctx); /* not an error: what the residency probe of nm_create gave up (fewer workgroups per replica
                                                  than asked for), and every block that was re-issued at a lower number later on        */
int nm_nslots(const nm_ctx *ctx);              /* nrows*nt replicas held by this context               */
int nm_natoms(const nm_ctx *ctx);
int nm_cus_per_replica(const nm_ctx *ctx);     /* workgroups (CUs) cooperating on one replica: 1, 2, 4 ...  */
int nm_heal_count(const nm_ctx *ctx);          /* blocks re-issued at fewer workgroups per replica so far (see nm_get_status): a timed region
                                                  during which this grew contained launches that did no work (they are not in nm_timing_get)  */

/* thermodynamic constants per local slot: et = k_B T, pf = P/(k_B T) (init_constant, remcmc:114-132) */
int nm_get_const(const nm_ctx *ctx, double *et, double *pf);

/* replica state, local slots [k0,k0+nk): x[nk][3N], v[nk][3N], box[nk], dxdvdt[nk][3].  Any pointer may be NULL. */
int nm_set_state(nm_ctx *ctx, int k0, int nk, const double *x, const double *v, const double *box,
                 const double *dxdvdt);
int nm_get_state(nm_ctx *ctx, int k0, int nk, double *x, double *v, double *box, double *dxdvdt);
/* the same for the nk local slots listed in slots[] (any order), together with their thermo scalars th[nk][5] (nm_set_thermo's columns):
   one settle, one wait for the whole batch.  Used by the split-row exchange, which re-seats only the replicas that swapped
   (replica_exchange moves entries [0..11] of two state lists, remcmc:798).  Any data pointer may be NULL.  In nm_set_slots a box sets
   the volume column, a th row given as well overrides it. */
int nm_get_slots(nm_ctx *ctx, int nk, const int *slots, double *x, double *v, double *box, double *dxdvdt, double *th);
int nm_set_slots(nm_ctx *ctx, int nk, const int *slots, const double *x, const double *v, const double *box, const double *dxdvdt,
                 const double *th);

/* init_samples (remcmc:394-456) for every local replica, for callers without the Python front end: fcc lattice in create_atoms
   order, box edge statically relaxed to the row's pressure (what fix box/relax + minimize converge to for the perfect crystal),
   uniform random displacement of amplitude dx*LAT, velocities zero, steps (dx, dv, TIMESTEP).  interpolate != 0: the volume
   expansion of the -is branch (remcmc:409-419); follow with nm_run_md(ctx, 1024) for its dynamics.  natoms must be 4*sz^3.
   Host code, once per run; the same states neuralmelting_amd/lattice.py produces. */
int nm_init_lattice(nm_ctx *ctx, double dx, double dv, int interpolate);
/* the same for ONE global slot without a context (no GPU involved): x[3*natoms], *box */
int nm_lattice_state(int element, int sz, int np, int nt, const float *P, uint32_t seed, int gslot, double dx, int interpolate,
                     double *x, double *box);

/* thermo scalars carried by a state list (entries 3,4,5,6,8 of remcmc:432-433): th[nk][5] = temp, pe, ke, virial, vol.
   Needed when states come from a restart file and the first thing the reference does is replica_exchange (remcmc:966-968). */
int nm_set_thermo(nm_ctx *ctx, int k0, int nk, const double *th);

/* cycle index STEP of the main loop (remcmc:977); selects the RNG counter block. nm_run_block does not advance it. */
int nm_set_step(nm_ctx *ctx, uint32_t step);

/* the -is branch of init_sample (remcmc:421-425) for every local replica: velocity all create T, zero linear, zero
   angular, timestep dt, run nsteps (plain NVE, no Metropolis test, counters untouched) */
int nm_run_md(nm_ctx *ctx, int nsteps);
/* gen_samples: MOD moves for every local replica, asynchronous on the context's stream */
int nm_run_block(nm_ctx *ctx, int mod);
/* `ncycles` cycles of the main loop with outputs off (remcmc:977-995: gen_samples, gen_mc_params, replica_exchange; steps STEP .. STEP + ncycles - 1 of
   nm_set_step), asynchronous: the same chains, bit for bit, as ncycles times nm_run_block + nm_adapt + nm_exchange, queued by one call.  Needs whole
   pressure rows (NM_ERR_UNSUPPORTED otherwise).
   Where the grid runs as 4^3 clusters of 2 or 4 workgroups per replica (64-128 replicas per GPU) it is ONE launch per 64 cycles (nm_cycles_kernel): the replicas of a
   pressure row meet after every block and rows never wait for one another (the exchange never leaves a row, remcmc:782-798) — +2.5 % on the 8 x 8 grid.
   Elsewhere, and with tapes or a trace set, it is the loop of single launches (NM_FUSED_CYCLES=0 in the environment: always; =all: one launch wherever
   the kernel exists).  A hand-over that times out in the middle of such a launch leaves the rows at different cycles: reported as NM_ERR_STATE, not
   re-issued. */
int nm_run_cycles(nm_ctx *ctx, int ncycles, int mod);
/* rows[nslots][17] in the column order of remcmc:208 (values of the last nm_run_block; call before nm_adapt) */
int nm_get_thermo(nm_ctx *ctx, double *rows);
/* write_outputs (remcmc:259-286) without stopping the stream.  nm_snapshot, queued right behind nm_run_block and in front of nm_adapt, keeps what a
   recorded cycle writes (the 17 thermo columns, positions, box) as of that point; a side stream brings it to the host while the context's stream goes on with
   nm_adapt, nm_exchange and the next block.  nm_snapshot_fetch hands the OLDEST pending snapshot out — rows[nslots][17], x[nslots][3N], box[nslots], any
   of them NULL — and waits for that copy only.  At most two may be pending (a driver fetches cycle s - 1 after queueing cycle s); NM_ERR_STATE otherwise.
   Neither call looks at the queue's outcome: an error is reported by the next synchronising call as always. */
int nm_snapshot(nm_ctx *ctx);
int nm_snapshot_fetch(nm_ctx *ctx, double *rows, double *x, double *box);
/* gen_mc_params: adapt dx, dv, dt, zero counters and ratios */
int nm_adapt(nm_ctx *ctx);
/* replica_exchange over the local pressure rows.  nswaps may be NULL (stays asynchronous). */
int nm_exchange(nm_ctx *ctx, int *nswaps);
/* waits for everything enqueued on the context; reports replicas that left the supported regime */
int nm_synchronize(nm_ctx *ctx);
/* status[nslots]: 0, or the NM_ST_* bits that stopped the slot's last block.  A block that ends on an error leaves the slot's
   x, v, box and thermo scalars as they were when it started (the reference's LAMMPS would have aborted the process), and nothing
   queued behind it (blocks, nm_adapt, nm_exchange) runs until the host has looked: the next nm_synchronize / nm_get_* / nm_set_*
   - re-issues the failed block and everything queued behind it with fewer workgroups per replica when the reason was the launch
     itself (NM_ST_NOT_RESIDENT, NM_ST_SYNC_TIMEOUT: the analogue of Dask retrying a failed task, remcmc:921-922), notes that in
     nm_create_note and returns NM_OK;
   - otherwise returns NM_ERR_STATE once, with the reason in nm_last_error; the bits stay readable here until the next block and the
     context remains usable (e.g. after nm_set_state of a configuration that fits).
   "nm_get_* / nm_set_*" means every entry point that reads or replaces what the queue works on: state, thermo, slots, stats, trace,
   perm, exchange criteria, counters and both tapes.  nm_get_status itself looks too (a halted queue is re-issued) but always hands
   out the bits and returns NM_OK: it is the call that reports them. */
#define NM_ST_LIST_OVERFLOW 1   /* more neighbours within rc + skin than list slots                           */
#define NM_ST_BOX_TOO_SMALL 2   /* box edge < 2 rc: outside the minimum-image regime                          */
#define NM_ST_TAPE_EXHAUSTED 4  /* test-only rng tape too short                                               */
#define NM_ST_NONFINITE 8       /* non-finite energy                                                          */
#define NM_ST_SYNC_TIMEOUT 16   /* a hand-over between the workgroups of a replica timed out                  */
#define NM_ST_NOT_RESIDENT 32   /* the launch's workgroups were not resident together; nothing was changed    */
int nm_get_status(nm_ctx *ctx, int *status);

/* measurement: HIP-event time of the nm_run_block kernel launches since the last reset */
int nm_timing_reset(nm_ctx *ctx);
int nm_timing_get(nm_ctx *ctx, int *launches, double *total_ms);
/* per-slot work counters accumulated since the last reset: stats[nslots][NM_STATS_COLS] */
int nm_stats_get(nm_ctx *ctx, double *stats, int reset);

/* ---- output formatting on the host (write_outputs, remcmc:235-286) ---------------------------------------
   Byte-identical to the reference's Python formatting: a .thrm row is 17 x ' %.4E' + newline (remcmc:245), a .traj frame is
   '%d %.4E' % (natoms, box) + newline followed by natoms lines of 3 x ' %.4E' (remcmc:254-256).  No GPU involved. */
int nm_format_thrm(const double *row17, char *out, int cap);                         /* returns the length written or <0 */
int nm_format_traj(int natoms, double box, const double *x, char *out, int cap);     /* cap >= 20 + 37*natoms            */
/* appends one row and one frame to the nk replicas' files, nthreads writer threads (0 = hardware concurrency) */
int nm_append_outputs(int nk, int natoms, const char *const *thrm_paths, const char *const *traj_paths,
                      const double *rows, const double *x, const double *box, int nthreads);

/* ---- test-only entry points ------------------------------------------------------------------ */
/* batched lj_energy_force on the current states: U[nslots], W[nslots] (virial sum r.f), f[nslots][3N] (may be NULL) */
int nm_eval(nm_ctx *ctx, double *U, double *W, double *f);
/* externally supplied uniforms replacing the scalar draws of nm_run_block, consumed in the order of the
   reference's np.random calls (remcmc:482,490,523,535,565,581,603,626,645); offsets[nslots+1]; NULL clears */
int nm_set_rng_tape(nm_ctx *ctx, const double *tape, const int *offsets);
/* one uniform per pair of the exchange sweep in sweep order (remcmc:795); NULL clears */
int nm_set_exchange_tape(nm_ctx *ctx, const double *tape, int n);
/* per-move records of the next nm_run_block calls: trace[nslots][mod][NM_TRACE_COLS] */
int nm_set_trace(nm_ctx *ctx, int enable);
int nm_get_trace(nm_ctx *ctx, double *trace, int mod);
/* acceptance counters count[nslots][6] (ntp nap ntv nav nth nah) and ratios ratio[nslots][3] (float32 ap av ah) as gen_sample would
   have left them (remcmc:685-691): lets a test hand nm_adapt the reference's own inputs.  Either pointer may be NULL. */
int nm_set_counters(nm_ctx *ctx, const double *count, const float *ratio);
/* slot -> buffer map after exchanges (which initial configuration sits in slot k) */
int nm_get_perm(nm_ctx *ctx, int *perm);
/* dh of every pair visited by the last nm_exchange, sweep order */
int nm_get_exchange_crit(nm_ctx *ctx, double *crit, int n);

#ifdef __cplusplus
}
#endif
#endif
