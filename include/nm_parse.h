/*
 * nm_parse.h — C-ABI of the .thrm / .traj reader, the step between the sampler and the
 * structural histograms: walkernr/neuralMelting's scripts/lammps_parse.py turns the consolidated text files into .npy
 * arrays with np.loadtxt (lammps_parse.py:48-49) and a readlines/split/np.array loop (lammps_parse.py:88-96).
 * Host-only (no GPU involved), multi-threaded over a memory-mapped file; values are converted text -> float64 -> float32
 * exactly as numpy does, so the arrays are bit-identical to the reference's.
 *
 * Both functions are called twice: with null outputs to count, then with buffers of that size.
 */
#ifndef NM_PARSE_H
#define NM_PARSE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* rows[nrows][17] float32 in the column order of remcmc:208; lines starting with '#' are skipped like np.loadtxt does.
 * nthreads 0 = hardware concurrency.  Returns 0 or NM_ERR_ARG (include/nm.h); text via nm_parse_last_error(). */
int nm_parse_thrm(const char *path, float *rows, long cap_rows, long *nrows, int nthreads);

/* lines of two tokens are frame heads ('%d %.4E' % (natoms, box), remcmc:254), lines of three tokens are coordinates
 * (remcmc:256); natoms[nframes] uint16, box[nframes] float32, pos[nposrows][3] float32 (lammps_parse.py:91-94). */
int nm_parse_traj(const char *path, uint16_t *natoms, float *box, float *pos, long cap_frames, long cap_posrows,
                  long *nframes, long *nposrows, int nthreads);

const char *nm_parse_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
