/*
 * nm_distr.h — C-ABI of the structural-histogram kernels, the step right after the sampler:
 * walkernr/neuralMelting's scripts/lammps_distr.py computes, for every recorded sample, the radial distribution over the
 * 27 periodic images (calculate_rdf, lammps_distr.py:123-135) and a 3-D histogram of pair displacement vectors
 * (calculate_cdf, lammps_distr.py:161-171) with np.histogram / np.histogramdd on float32 coordinates.
 *
 * nm_distr_histograms replaces both per-sample functions for a batch of samples.  It returns the raw integer counts the
 * reference accumulates before its division by natoms (bit-exact: same float32 arithmetic for the displacements, same
 * float64 edge comparisons and edge-inclusion rules as numpy), as float32 like the reference's `rd` / `cd` arrays.
 */
#ifndef NM_DISTR_H
#define NM_DISTR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* pos[ns][natoms][3], box[ns] float32 (lammps_parse.py's .pos.npy / .box.npy);
 * r_edges[sbins] float64 = R of calculate_spatial (lammps_distr.py:84-95): sbins-1 bins, result written to rdf[s][1..sbins-1],
 *   rdf[s][0] = 0 (rd[1:] += np.histogram(d, r)[0]);
 * rv_edges[cbins+1] float64 = RV[0] of calculate_spatial (lammps_distr.py:111-113), the same edges in x, y and z;
 * rdf[ns][sbins], cdf[ns][cbins][cbins][cbins] float32 counts summed over the 27 images, NOT yet divided by natoms.
 * Either output may be NULL.  Returns 0 or a negative NM_ERR_* code (include/nm.h); message via nm_distr_last_error(). */
int nm_distr_histograms(int device, int ns, int natoms, const float *pos, const float *box, int sbins,
                        const double *r_edges, int cbins, const double *rv_edges, float *rdf, float *cdf);
const char *nm_distr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
