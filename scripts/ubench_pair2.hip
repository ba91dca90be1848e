// dev micro-benchmark: the force-only LJ pair loop of round 3 in isolation (LDS-resident positions, byte list as 8-byte words,
// v_fract minimum image, rcp + one Newton step, two neighbours interleaved), at 2 and at 4 waves per SIMD: what would the loop gain
// from twice the occupancy?  Prints neighbour evaluations per cycle and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int N = 256, NB = 136; // (list length of the fcc crystal at rc + skin = 2.9: 134)
template <int BLOCK, int TPA, int VAR, int W = 2>
__global__ void __launch_bounds__(BLOCK) k(const double *gx, const unsigned long long *gl, const int *gcnt, double *out, unsigned long long *cyc, int reps, double L)
{
    __shared__ double px[N], py[N], pz[N];
    __shared__ unsigned long long nb64[(NB / 8) * 64 * 8]; // [word k][row][sub] for 64 rows x 8 subs (TPA 8) or [k][row][16]
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += BLOCK) { px[i] = gx[3 * i]; py[i] = gx[3 * i + 1]; pz[i] = gx[3 * i + 2]; }
    for (int i = tid; i < (NB / 8) * 64 * 8; i += BLOCK) nb64[i] = gl[(size_t)(blockIdx.x & 3) * (NB / 8) * 64 * 8 + i];
    __syncthreads();
    const int g = tid / TPA, sub = tid % TPA;        // 64 rows: BLOCK / TPA = 64
    const int i = (blockIdx.x & 3) * 64 + g;
    const double invL = 1.0 / L, rc2 = 6.25, mhL = -0.5 * L;
    double ax = 0, ay = 0, az = 0;
    // VAR 0: every row has NB entries (uniform trip count); VAR 1: per-row lengths from data, as in the kernel (divergent masks)
    const int c = VAR ? gcnt[i] : NB;
    const int mine = (c - sub + TPA - 1) / TPA;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const double pert = 1.0e-9 * r; // (keeps the pass from being hoisted out of the repetition loop)
        const double xi = __builtin_fma(px[i], invL, 0.5 + pert), yi = __builtin_fma(py[i], invL, 0.5 - pert), zi = __builtin_fma(pz[i], invL, 0.5 + 2.0 * pert);
        // VAR 2: the trip count is the wave's maximum (uniform: scalar branches, no exec masking); a lane's surplus entries are
        // switched off as data
        int wmax = mine;
        if (VAR == 2) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) wmax = max(wmax, __shfl_xor(wmax, d, 64));
            wmax = __builtin_amdgcn_readfirstlane(wmax);
        }
        if (VAR == 3) {
            // software-pipelined: the positions of the NEXT two neighbours are fetched before the arithmetic on the current two
            const int npair = (mine + 1) >> 1;
            unsigned long long wd = nb64[(0 * 64 + g) * 8 + (sub & 7)];
            double cx[2], cy[2], cz[2], nx[2], ny[2], nz[2];
            {
#pragma unroll
                for (int q = 0; q < 2; ++q) { const int j = q < mine ? (int)((wd >> (8 * q)) & 0xFF) : i; cx[q] = px[j]; cy[q] = py[j]; cz[q] = pz[j]; }
            }
            for (int it = 0; it < npair; ++it) {
                const int e = 2 * (it + 1);               // first entry of the next pair
                // (no branches around the loads: the compiler then counts them and waits with lgkmcnt(n), not lgkmcnt(0))
                const unsigned long long wn = nb64[(min(e >> 3, NB / 8 - 1) * 64 + g) * 8 + (sub & 7)];
                wd = (e & 7) == 0 ? wn : wd;
#pragma unroll
                for (int q = 0; q < 2; ++q) { const int j = (e + q) < mine ? (int)((wd >> (8 * ((e + q) & 7))) & 0xFF) : i; nx[q] = px[j]; ny[q] = py[j]; nz[q] = pz[j]; }
                double dx[2], dy[2], dz[2], r2[2], y[2], t[2], fp[2];
                bool in[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    in[q] = (2 * it + q) < mine;
                    dx[q] = __builtin_fma(-cx[q], invL, xi); dy[q] = __builtin_fma(-cy[q], invL, yi); dz[q] = __builtin_fma(-cz[q], invL, zi);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) { dx[q] = __builtin_fma(__builtin_amdgcn_fract(dx[q]), L, mhL); dy[q] = __builtin_fma(__builtin_amdgcn_fract(dy[q]), L, mhL); dz[q] = __builtin_fma(__builtin_amdgcn_fract(dz[q]), L, mhL); }
#pragma unroll
                for (int q = 0; q < 2; ++q) { r2[q] = dx[q] * dx[q] + dy[q] * dy[q] + dz[q] * dz[q]; in[q] = in[q] && r2[q] < rc2; }
#pragma unroll
                for (int q = 0; q < 2; ++q) y[q] = __builtin_amdgcn_rcp(r2[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
                for (int q = 0; q < 2; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) y[q] = in[q] ? y[q] : 0.0;
#pragma unroll
                for (int q = 0; q < 2; ++q) t[q] = y[q] * y[q] * y[q];
#pragma unroll
                for (int q = 0; q < 2; ++q) fp[q] = t[q] * __builtin_fma(2.0, t[q], -1.0) * y[q];
#pragma unroll
                for (int q = 0; q < 2; ++q) { ax += dx[q] * fp[q]; ay += dy[q] * fp[q]; az += dz[q] * fp[q]; }
#pragma unroll
                for (int q = 0; q < 2; ++q) { cx[q] = nx[q]; cy[q] = ny[q]; cz[q] = nz[q]; }
            }
            continue;
        }
        const int bound = VAR == 2 ? wmax : mine;
        for (int k0 = 0; k0 < bound; k0 += 8) {
            const unsigned long long wd = nb64[(((k0 >> 3) >> 0) * 64 + g) * 8 + (sub & 7)];
#pragma unroll
            for (int e0 = 0; e0 < 8; e0 += W) {
                if (k0 + e0 < bound) {
                    double dx[W], dy[W], dz[W], r2[W], y[W], t[W], fp[W];
                    bool in[W];
#pragma unroll
                    for (int q = 0; q < W; ++q) {
                        const bool ok = (k0 + e0 + q) < mine;
                        const int j = ok ? (int)((wd >> (8 * (e0 + q))) & 0xFF) : i;
                        dx[q] = __builtin_fma(-px[j], invL, xi); dy[q] = __builtin_fma(-py[j], invL, yi); dz[q] = __builtin_fma(-pz[j], invL, zi);
                        in[q] = ok;
                    }
#pragma unroll
                    for (int q = 0; q < W; ++q) { dx[q] = __builtin_fma(__builtin_amdgcn_fract(dx[q]), L, mhL); dy[q] = __builtin_fma(__builtin_amdgcn_fract(dy[q]), L, mhL); dz[q] = __builtin_fma(__builtin_amdgcn_fract(dz[q]), L, mhL); }
#pragma unroll
                    for (int q = 0; q < W; ++q) { r2[q] = dx[q] * dx[q] + dy[q] * dy[q] + dz[q] * dz[q]; in[q] = in[q] && r2[q] < rc2; }
#pragma unroll
                    for (int q = 0; q < W; ++q) y[q] = __builtin_amdgcn_rcp(r2[q]);
#pragma unroll
                    for (int q = 0; q < W; ++q) t[q] = __builtin_fma(-r2[q], y[q], 1.0);
#pragma unroll
                    for (int q = 0; q < W; ++q) y[q] = __builtin_fma(y[q], t[q], y[q]);
#pragma unroll
                    for (int q = 0; q < W; ++q) y[q] = in[q] ? y[q] : 0.0;
#pragma unroll
                    for (int q = 0; q < W; ++q) t[q] = y[q] * y[q] * y[q];
#pragma unroll
                    for (int q = 0; q < W; ++q) fp[q] = t[q] * __builtin_fma(2.0, t[q], -1.0) * y[q];
#pragma unroll
                    for (int q = 0; q < W; ++q) { ax += dx[q] * fp[q]; ay += dy[q] * fp[q]; az += dz[q] * fp[q]; }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * BLOCK + tid] = ax + ay + az;
    if ((tid & 63) == 0) cyc[blockIdx.x * (BLOCK / 64) + (tid >> 6)] = t1 - t0;
}
template <int BLOCK, int TPA, int VAR, int W = 2>
void run(const char *name, const double *dx, const unsigned long long *dl, const int *dc, double L)
{
    const int blocks = 256;
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * BLOCK); hipMalloc(&cyc, 8 * blocks * (BLOCK / 64));
    const int reps = 400;
    hipLaunchKernelGGL((k<BLOCK, TPA, VAR, W>), dim3(blocks), dim3(BLOCK), 0, 0, dx, dl, dc, out, cyc, reps, L);
    hipLaunchKernelGGL((k<BLOCK, TPA, VAR, W>), dim3(blocks), dim3(BLOCK), 0, 0, dx, dl, dc, out, cyc, reps, L);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (BLOCK / 64));
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (auto v : h) { sum += v; if (v > mx) mx = v; }
    const double evals = 64.0 * 131 * reps;           // neighbour evaluations of one workgroup (64 rows x 112 entries)
    printf("%-44s %6.1f cycles per pass of 64 rows (slowest wave %6.1f): %.3f neighbour evaluations per cycle and CU\n", name, sum / h.size() / reps, mx / reps, evals / mx);
    hipFree(out); hipFree(cyc);
}
int main(int argc, char **argv)
{
    const bool lattice = argc > 1;     // any argument: a perturbed fcc crystal with its real (sorted) neighbour lists instead of random ones
    const double a = 1.5276, L = lattice ? 4 * a : 6.17;
    std::vector<double> x(3 * N);
    srand(1);
    if (lattice) {
        const double b[4][3] = { {0, 0, 0}, {.5, .5, 0}, {.5, 0, .5}, {0, .5, .5} };
        int n = 0;
        for (int k = 0; k < 4; ++k) for (int j = 0; j < 4; ++j) for (int i = 0; i < 4; ++i) for (int q = 0; q < 4; ++q, ++n) {
            x[3 * n] = (i + b[q][0]) * a + 0.1 * (rand() / (double)RAND_MAX - .5); x[3 * n + 1] = (j + b[q][1]) * a + 0.1 * (rand() / (double)RAND_MAX - .5);
            x[3 * n + 2] = (k + b[q][2]) * a + 0.1 * (rand() / (double)RAND_MAX - .5);
        }
    } else for (auto &v : x) v = L * (rand() / (double)RAND_MAX);
    // lists of all 256 atoms; the kernel's block b reads rows (b & 3) * 64 .. + 63 from a table indexed by the local row, so build
    // the table for workgroup 0's rows only when random, or per workgroup-of-four when realistic (four tables back to back)
    std::vector<int> cn(N, NB);
    std::vector<unsigned long long> l((size_t)4 * (NB / 8) * 64 * 8, 0ull);
    for (int i = 0; i < N; ++i) {
        std::vector<int> nb;
        if (lattice) {
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                double d2 = 0;
                for (int c = 0; c < 3; ++c) { double d = x[3 * i + c] - x[3 * j + c]; d -= L * rint(d / L); d2 += d * d; }
                if (d2 < 2.9 * 2.9) nb.push_back(j);
            }
            if ((int)nb.size() > NB) nb.resize(NB);
            cn[i] = (int)nb.size();
        } else { for (int r = 0; r < NB; ++r) nb.push_back(rand() & 255); cn[i] = 126 + rand() % 11; }
        const int wg = i / 64, g = i % 64;
        for (int r = 0; r < (int)nb.size(); ++r) { // entry r belongs to sub r % 8 as its (r / 8)-th: word (r / 8) / 8, byte (r / 8) % 8
            const int sub = r % 8, kk = r / 8;
            l[(size_t)wg * (NB / 8) * 64 * 8 + ((size_t)(kk >> 3) * 64 + g) * 8 + sub] |= (unsigned long long)nb[r] << (8 * (kk & 7));
        }
    }
    double *dx; unsigned long long *dl;
    hipMalloc(&dx, x.size() * 8); hipMalloc(&dl, l.size() * 8);
    hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dl, l.data(), l.size() * 8, hipMemcpyHostToDevice);
    int *dc; hipMalloc(&dc, N * 4); hipMemcpy(dc, cn.data(), N * 4, hipMemcpyHostToDevice);
    double mean = 0; for (int v : cn) mean += v; mean /= N;
    printf("%s lists, mean length %.1f\n", lattice ? "crystal (sorted, overlapping)" : "random", mean);
    run<512, 8, 1>("512 threads, 8 per row, per-row lengths", dx, dl, dc, L);
    run<512, 8, 1, 4>("512 threads, 8 per row, per-row lengths, W = 4", dx, dl, dc, L);
    run<512, 8, 3>("512 threads, 8 per row, gathers of the next pair issued ahead", dx, dl, dc, L);
    run<512, 8, 0>("512 threads, 8 per row, lengths known at compile time (gathers hoisted out of the repetitions: arithmetic only)", dx, dl, dc, L);
    run<1024, 16, 1>("1024 threads, 16 per row, per-row lengths (random entries: the table is laid out for 8 per row)", dx, dl, dc, L);
    run<256, 4, 1>("256 threads, 4 per row, per-row lengths (random entries)", dx, dl, dc, L);
    return 0;
}
