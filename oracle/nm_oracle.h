/*
 * nm_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the NPT-HMC + replica-exchange hot path of
 * walkernr/neuralMelting's scripts/lammps_remcmc.py ("remcmc" below) and of the
 * subset of LAMMPS that path drives (lj/cut 2.5, fix nve, velocity create,
 * displace_atoms random, change_box, thermo computes).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (neuralmelting_amd/) never links or imports it.
 *
 * Parity status: the control flow (Metropolis criteria, '%f' quantisation,
 * counter logic, RNG-consumption order, exchange sweep, adaptive steps) is
 * PINNED by golden traces produced by the reference's own Python functions
 * (tests/golden/make_golden.py).  The LAMMPS arithmetic itself is
 * "PARITY UNPINNED": liblammps is not in the reference tree nor in this image;
 * it is restated from its published algorithm and pinned only by known-answer
 * values (perfect-fcc LJ energy/pressure).
 */
#ifndef NM_ORACLE_H
#define NM_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_sim orc_sim;

/* units: 0 = lj, 1 = metal (remcmc:873-877).  pot: 0 = lj/cut 2.5, 1 = Sutton-Chen EAM (Al) */
orc_sim *orc_create(int natoms, int units, double mass, int pot);
void orc_destroy(orc_sim *s);

/* RNG addressing: Philox4x32-10, key = (seed, slot), counter = (index, stream, tag, step). */
void orc_set_rng(orc_sim *s, uint32_t seed, uint32_t slot, uint32_t step);

/* LAMMPS-object primitives (call sites remcmc:380-388, 399-428, 462-469, 480-638) */
void orc_set_box(orc_sim *s, double L);            /* change_box all x final 0.0 L ...   */
double orc_get_box(const orc_sim *s);              /* boxhi - boxlo                       */
void orc_set_x(orc_sim *s, const double *x);       /* scatter_atoms x (images untouched)  */
void orc_set_v(orc_sim *s, const double *v);       /* scatter_atoms v                     */
void orc_get_x(const orc_sim *s, double *x);       /* gather_atoms x                      */
void orc_get_v(const orc_sim *s, double *v);       /* gather_atoms v                      */
void orc_get_f(const orc_sim *s, double *f);
void orc_get_image(const orc_sim *s, int *img);
void orc_set_image(orc_sim *s, const int *img);
int orc_setup(orc_sim *s);                         /* "run 0": wrap, neighbour, U/f/W     */
int orc_eval_allpairs(orc_sim *s, double *U, double *W, double *f); /* O(N^2) cross-check */
void orc_displace(orc_sim *s, double a, uint32_t tag);      /* displace_atoms all random a a a */
void orc_velocity_create(orc_sim *s, double t, uint32_t tag);/* velocity all create t seed dist gaussian */
void orc_zero_linear(orc_sim *s);                  /* velocity all zero linear            */
void orc_zero_angular(orc_sim *s);                 /* velocity all zero angular           */
void orc_set_timestep(orc_sim *s, double h);       /* timestep h                          */
int orc_run(orc_sim *s, int nsteps);               /* run N (setup + velocity-Verlet)     */
double orc_pe(const orc_sim *s);                   /* thermo_pe (extensive)               */
double orc_ke(const orc_sim *s);                   /* thermo_ke                           */
double orc_temp(const orc_sim *s);                 /* thermo_temp, dof = 3N-3             */
double orc_press(const orc_sim *s);                /* thermo_press                        */
double orc_virial(const orc_sim *s);               /* W = sum r.f                         */
int orc_nlist(const orc_sim *s);                   /* listed half pairs (diagnostic)      */
int orc_npairs(const orc_sim *s);                  /* interacting pairs r<rc at last eval */

/* '%f' round trip (remcmc:403,407,466,483,571,604,607) */
double orc_q6(double x);          /* arithmetic form used by the engine */
double orc_q6_printf(double x);   /* snprintf("%f") + strtod, to check the above */

/* Philox known-answer access */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_u01(uint32_t hi, uint32_t lo);

/* ---- composite: one block of MOD moves for one replica (remcmc:665-691) ---- */
typedef struct orc_block_params {
    int mod;          /* MOD                                  */
    int nstps;        /* NSTPS                                */
    int bulk;         /* BM flag                              */
    int iter_revert;  /* 0 = reference behaviour of iter PMC (rejections not undone), 1 = corrected */
    double ppos, pvol;/* PPOS, PVOL                           */
    double lat;       /* LAT[EL][1]                           */
    double t;         /* T[j] (float32-rounded, as double)    */
    double et, pf;    /* CONST[k]                             */
    const double *tape; /* optional externally supplied uniforms (test only) */
    int tape_len;
    double *trace;    /* optional [mod][4]: branch, accepted, criterion, U after */
} orc_block_params;

/* state in/out: x[3N], v[3N], box, dxdvdt[3]; out: thermo[5] = temp,pe,ke,virial,vol;
   counters[6] = ntp,nap,ntv,nav,nth,nah (zero on entry per remcmc:745); ratios[3] float32 */
int orc_run_block(orc_sim *s, const orc_block_params *p, double *x, double *v, double *box,
                  const double *dxdvdt, double *thermo, double *counters, float *ratios,
                  int *tape_used);

/* batch over replicas with OpenMP (CPU baseline leg) */
int orc_run_blocks(int ns, int natoms, int units, double mass, int pot, uint32_t seed,
                   uint32_t slot0, uint32_t step, const orc_block_params *p /*[ns]*/,
                   double *x, double *v, double *box, const double *dxdvdt, double *thermo,
                   double *counters, float *ratios, int nthreads);

/* adaptive step sizes (remcmc:726-745) */
void orc_adapt(const float ratios[3], double dxdvdt[3]);

/* replica exchange sweep (remcmc:776-803) over rows [row0,row0+nrows) of an NP x NT grid.
   etot = pe+ke per slot, vol per slot, et/pf per slot; perm[k] (in/out) = buffer held by slot k.
   tape (optional) = one uniform per pair in sweep order; returns number of swaps. */
int orc_exchange(int np_total, int nt, int row0, int nrows, uint32_t seed, uint32_t step,
                 double *etot, double *vol, const double *et, const double *pf, int *perm,
                 const double *tape, double *crit_out);

#ifdef __cplusplus
}
#endif
#endif
