// nm_distr.h — histogram kernels for lammps_distr.py's calculate_rdf / calculate_cdf (lammps_distr.py:123-171).
// One workgroup per (sample, periodic image): positions staged in LDS, float32 displacement arithmetic identical to
// numpy's (no contraction), float64 edge comparisons, integer counts in LDS, one float atomic (exact below 2^24) per non-empty bin
// at the end.
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

// the same bin found from a guess: the reference's edges are np.linspace grids, so (d - e[0]) * inv lands on the right bin or
// next to it, and the walk below ends after zero or one step; it is exact for any increasing edges
__device__ __forceinline__ int bin_near(const double *e, int n, double d, double inv)
{
    const double e0 = e[0];
    if (!(d >= e0) || !(d <= e[n - 1])) return -1;
    int k = (int)((d - e0) * inv);
    k = k < 0 ? 0 : (k > n - 2 ? n - 2 : k);
    if (d < e[k]) { do --k; while (d < e[k]); } // stops at k >= 0: e[0] <= d
    else while (k < n - 2 && d >= e[k + 1]) ++k;
    return k;
}

// A pair contributes only if its displacement lies inside the cube |d| <= l/2 (the sphere r <= l/2 of the radial histogram
// sits inside it): 1/27 of all (pair, image) combinations.  The loop over b therefore runs a cheap float pre-test over 64
// candidates at a time, collecting the survivors of each lane in a bit mask, and only then walks the set bits through the exact
// path (float32 arithmetic as numpy, float64 edge searches, LDS atomics).  The pre-test is conservative: `cube` is a float
// strictly outside every edge of both histograms, so a rejected candidate is outside in the exact comparison too.
__global__ void __launch_bounds__(DISTR_BLOCK)
nm_distr_kernel(int natoms, const float *__restrict__ pos, const float *__restrict__ box, int sbins,
                const double *__restrict__ r_edges, int cbins, const double *__restrict__ rv_edges,
                float *__restrict__ rdf_cnt, float *__restrict__ cdf_cnt)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    const int s = blockIdx.x / 27, img = blockIdx.x % 27;
    const int tid = threadIdx.x;
    // br[j] = (b[i], b[j], b[k]) for i, j, k in range(3), b = [-1, 0, 1]  (lammps_distr.py:99-102)
    const float bx = (float)(img / 9 - 1), by = (float)((img / 3) % 3 - 1), bz = (float)(img % 3 - 1);
    const int npad = (natoms + 63) & ~63;
    float *px = (float *)smem, *py = px + natoms, *pz = py + natoms;
    float *qx = pz + natoms, *qy = qx + npad, *qz = qy + npad; // pos[b] + box*br, padded with far-away entries
    double *re = (double *)(smem + (((size_t)3 * (natoms + npad) * sizeof(float) + 7) & ~(size_t)7));
    double *ve = re + sbins;
    unsigned int *hr = (unsigned int *)(ve + (cbins + 1));
    unsigned int *hc = hr + sbins;
    const int nc = cbins * cbins * cbins;
    const float L = box[s];
    const float *ps = pos + (size_t)s * natoms * 3;
    const float sx = L * bx, sy = L * by, sz = L * bz; // box*br[j]
    for (int a = tid; a < npad; a += DISTR_BLOCK) {
        if (a < natoms) {
            const float x = ps[3 * a], y = ps[3 * a + 1], z = ps[3 * a + 2];
            px[a] = x; py[a] = y; pz[a] = z;
            qx[a] = x + sx; qy[a] = y + sy; qz[a] = z + sz;
        } else { qx[a] = 3.0e38f; qy[a] = 3.0e38f; qz[a] = 3.0e38f; }
    }
    for (int k = tid; k < sbins; k += DISTR_BLOCK) { re[k] = r_edges[k]; hr[k] = 0u; }
    for (int k = tid; k <= cbins; k += DISTR_BLOCK) ve[k] = rv_edges[k];
    for (int k = tid; k < nc; k += DISTR_BLOCK) hc[k] = 0u;
    __syncthreads();
    const bool do_r = rdf_cnt != nullptr, do_c = cdf_cnt != nullptr;
    double far = 0.0;
    if (do_r) far = fmax(far, fmax(fabs(re[0]), fabs(re[sbins - 1])));
    if (do_c) far = fmax(far, fmax(fabs(ve[0]), fabs(ve[cbins])));
    const double rinv = do_r ? (double)(sbins - 1) / (re[sbins - 1] - re[0]) : 0.0;
    const double vinv = do_c ? (double)cbins / (ve[cbins] - ve[0]) : 0.0;
    const float cube = nextafterf(nextafterf((float)far, 3.0e38f), 3.0e38f); // > every edge in magnitude, also after the cast's rounding
    // dvm[b][a] = pos[a] - (pos[b] + box*br): thread = a, loop over b (LDS broadcast reads)
    for (int a = tid; a < natoms; a += DISTR_BLOCK) {
        const float xa = px[a], ya = py[a], za = pz[a];
        for (int b0 = 0; b0 < npad; b0 += 64) {
            unsigned long long mask = 0ull;
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                const float dx = xa - qx[b0 + j], dy = ya - qy[b0 + j], dz = za - qz[b0 + j];
                const float m = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
                mask |= (m <= cube) ? (1ull << j) : 0ull;
            }
            while (mask) {
                const int b = b0 + __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const float dx = xa - qx[b], dy = ya - qy[b], dz = za - qz[b];
                if (do_r) {
                    float d2 = dx * dx;      // np.sum(np.square(dvm), -1): sequential float32 sum of three terms
                    d2 = d2 + dy * dy;
                    d2 = d2 + dz * dz;
                    const float d = __fsqrt_rn(d2);
                    const int k = bin_near(re, sbins, (double)d, rinv);
                    if (k >= 0) atomicAdd(&hr[k + 1], 1u);
                }
                if (do_c) {
                    // np.histogramdd: bin = (#edges <= x) - 1, a value on the last edge goes to the last bin, outside is dropped
                    const int kx = bin_near(ve, cbins + 1, (double)dx, vinv), ky = bin_near(ve, cbins + 1, (double)dy, vinv),
                              kz = bin_near(ve, cbins + 1, (double)dz, vinv);
                    if (kx >= 0 && ky >= 0 && kz >= 0) atomicAdd(&hc[(kx * cbins + ky) * cbins + kz], 1u);
                }
            }
        }
    }
    __syncthreads();
    // the 27 image blocks of a sample meet in global memory; counts stay below 2^24, so float atomics are exact and
    // the result is what the reference holds in its float32 `rd` / `cd` arrays before the division by natoms
    if (do_r)
        for (int k = tid; k < sbins; k += DISTR_BLOCK) if (hr[k]) atomicAdd(&rdf_cnt[(size_t)s * sbins + k], (float)hr[k]);
    if (do_c)
        for (int k = tid; k < nc; k += DISTR_BLOCK) if (hc[k]) atomicAdd(&cdf_cnt[(size_t)s * nc + k], (float)hc[k]);
}

} // namespace nm
