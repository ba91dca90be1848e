// nm_distr.h — histogram kernels for lammps_distr.py's calculate_rdf / calculate_cdf (lammps_distr.py:123-171).
// One workgroup per (sample, periodic image): positions staged in LDS, float32 displacement arithmetic identical to
// numpy's (no contraction), float64 edge comparisons, integer counts in LDS, one integer atomic per non-empty bin at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nm {

constexpr int DISTR_BLOCK = 256;
constexpr int DISTR_MAXS = 256;  // most spherical bin edges
constexpr int DISTR_MAXC = 32;   // most cartesian bins per axis

// np.histogram with explicit edges: bin k holds e[k] <= d < e[k+1], the last bin also d == e[n-1]; -1 = outside
__device__ __forceinline__ int bin_of(const double *e, int n, double d)
{
    if (!(d >= e[0]) || !(d <= e[n - 1])) return -1;
    int lo = 0, hi = n - 1; // invariant: e[lo] <= d, d < e[hi] or hi == n-1
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (d >= e[mid]) lo = mid; else hi = mid;
    }
    return lo; // d == e[n-1] ends in the last bin n-2
}

__global__ void __launch_bounds__(DISTR_BLOCK)
nm_distr_kernel(int natoms, const float *__restrict__ pos, const float *__restrict__ box, int sbins,
                const double *__restrict__ r_edges, int cbins, const double *__restrict__ rv_edges,
                unsigned int *__restrict__ rdf_cnt, unsigned int *__restrict__ cdf_cnt)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    const int s = blockIdx.x / 27, img = blockIdx.x % 27;
    const int tid = threadIdx.x;
    // br[j] = (b[i], b[j], b[k]) for i, j, k in range(3), b = [-1, 0, 1]  (lammps_distr.py:99-102)
    const float bx = (float)(img / 9 - 1), by = (float)((img / 3) % 3 - 1), bz = (float)(img % 3 - 1);
    float *px = (float *)smem, *py = px + natoms, *pz = py + natoms;
    double *re = (double *)(smem + (((size_t)3 * natoms * sizeof(float) + 7) & ~(size_t)7));
    double *ve = re + sbins;
    unsigned int *hr = (unsigned int *)(ve + (cbins + 1));
    unsigned int *hc = hr + sbins;
    const int nc = cbins * cbins * cbins;
    const float L = box[s];
    const float *ps = pos + (size_t)s * natoms * 3;
    for (int a = tid; a < natoms; a += DISTR_BLOCK) { px[a] = ps[3 * a]; py[a] = ps[3 * a + 1]; pz[a] = ps[3 * a + 2]; }
    for (int k = tid; k < sbins; k += DISTR_BLOCK) { re[k] = r_edges[k]; hr[k] = 0u; }
    for (int k = tid; k <= cbins; k += DISTR_BLOCK) ve[k] = rv_edges[k];
    for (int k = tid; k < nc; k += DISTR_BLOCK) hc[k] = 0u;
    __syncthreads();
    const float sx = L * bx, sy = L * by, sz = L * bz; // box*br[j]
    const bool do_r = rdf_cnt != nullptr, do_c = cdf_cnt != nullptr;
    // dvm[b][a] = pos[a] - (pos[b] + box*br): thread = a, loop over b (LDS broadcast reads)
    for (int a = tid; a < natoms; a += DISTR_BLOCK) {
        const float xa = px[a], ya = py[a], za = pz[a];
        for (int b = 0; b < natoms; ++b) {
            const float qx = px[b] + sx, qy = py[b] + sy, qz = pz[b] + sz;
            const float dx = xa - qx, dy = ya - qy, dz = za - qz;
            if (do_r) {
                float d2 = dx * dx;      // np.sum(np.square(dvm), -1): sequential float32 sum of three terms
                d2 = d2 + dy * dy;
                d2 = d2 + dz * dz;
                const float d = __fsqrt_rn(d2);
                const int k = bin_of(re, sbins, (double)d);
                if (k >= 0) atomicAdd(&hr[k + 1], 1u);
            }
            if (do_c) {
                // np.histogramdd: bin = (#edges <= x) - 1, a value on the last edge goes to the last bin, outside is dropped
                const int kx = bin_of(ve, cbins + 1, (double)dx), ky = bin_of(ve, cbins + 1, (double)dy),
                          kz = bin_of(ve, cbins + 1, (double)dz);
                if (kx >= 0 && ky >= 0 && kz >= 0) atomicAdd(&hc[(kx * cbins + ky) * cbins + kz], 1u);
            }
        }
    }
    __syncthreads();
    if (do_r)
        for (int k = tid; k < sbins; k += DISTR_BLOCK) if (hr[k]) atomicAdd(&rdf_cnt[(size_t)s * sbins + k], hr[k]);
    if (do_c)
        for (int k = tid; k < nc; k += DISTR_BLOCK) if (hc[k]) atomicAdd(&cdf_cnt[(size_t)s * nc + k], hc[k]);
}

} // namespace nm
