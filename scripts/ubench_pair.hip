// dev micro-benchmark: the LJ pair loop in isolation (LDS-resident positions and byte list, 512 threads, 8 threads per atom),
// to find what bounds it.  Variants selected by template parameters; prints shader cycles per neighbour per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
constexpr int N = 256, NB = 112, BLOCK = 512, TPA = 8;
__device__ __forceinline__ double recip(double a)
{
    double y = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-a, y, 1.0);
    return __builtin_fma(y, e, y);
}
template <int W, int MODE>
__global__ void __launch_bounds__(BLOCK) k(const double *gx, const unsigned char *gl, double *out, unsigned long long *cyc, int reps, double L)
{
    __shared__ double px[N], py[N], pz[N];
    __shared__ unsigned char nbr[NB * N];
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += BLOCK) { px[i] = gx[3 * i]; py[i] = gx[3 * i + 1]; pz[i] = gx[3 * i + 2]; }
    for (int i = tid; i < NB * N; i += BLOCK) nbr[i] = gl[i];
    __syncthreads();
    const int g = tid / TPA, sub = tid % TPA;
    const int i = (blockIdx.x & 3) * 64 + g;
    const double invL = 1.0 / L, rc2 = 6.25;
    double ax = 0, ay = 0, az = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        const double xi = px[i], yi = py[i], zi = pz[i];
        for (int s = sub; s < NB; s += W * TPA) {
            double dx[W], dy[W], dz[W], r2[W], y[W], t[W], fp[W];
            int j[W];
#pragma unroll
            for (int q = 0; q < W; ++q) j[q] = nbr[(s + q * TPA) * N + i];
#pragma unroll
            for (int q = 0; q < W; ++q) {
                if (MODE == 4) { dx[q] = 1.1 + 0.0011 * j[q]; dy[q] = 1.0 + 0.0013 * j[q]; dz[q] = 0.9 + 0.0017 * j[q]; }       // no gathers
                else if (MODE == 5) { const int jj = (tid + s + q) & 255; dx[q] = xi - px[jj]; dy[q] = yi - py[jj]; dz[q] = zi - pz[jj]; } // conflict-free gathers, no index use
                else { dx[q] = xi - px[j[q]]; dy[q] = yi - py[j[q]]; dz[q] = zi - pz[j[q]]; }
            }
            if (MODE != 1) {
#pragma unroll
                for (int q = 0; q < W; ++q) { dx[q] -= L * rint(dx[q] * invL); dy[q] -= L * rint(dy[q] * invL); dz[q] -= L * rint(dz[q] * invL); }
            }
#pragma unroll
            for (int q = 0; q < W; ++q) r2[q] = dx[q] * dx[q] + dy[q] * dy[q] + dz[q] * dz[q];
            if (MODE == 2) {
#pragma unroll
                for (int q = 0; q < W; ++q) y[q] = r2[q] * 0.37; // no reciprocal
            } else {
#pragma unroll
                for (int q = 0; q < W; ++q) y[q] = recip(r2[q]);
            }
#pragma unroll
            for (int q = 0; q < W; ++q) t[q] = y[q] * y[q] * y[q];
#pragma unroll
            for (int q = 0; q < W; ++q) fp[q] = (r2[q] < rc2) ? t[q] * (48.0 * t[q] - 24.0) * y[q] : 0.0;
#pragma unroll
            for (int q = 0; q < W; ++q) { ax += dx[q] * fp[q]; ay += dy[q] * fp[q]; az += dz[q] * fp[q]; }
        }
        if (MODE == 3) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * BLOCK + tid] = ax + ay + az;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}
template <int W, int MODE>
void run(const char *name, int blocks, const double *dx, const unsigned char *dl, double L)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * BLOCK); hipMalloc(&cyc, 8 * blocks * 8);
    const int reps = 200;
    hipLaunchKernelGGL((k<W, MODE>), dim3(blocks), dim3(BLOCK), 0, 0, dx, dl, out, cyc, reps, L);
    hipLaunchKernelGGL((k<W, MODE>), dim3(blocks), dim3(BLOCK), 0, 0, dx, dl, out, cyc, reps, L);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, 8 * blocks * 8, hipMemcpyDeviceToHost);
    double mx = 0, sum = 0;
    for (auto v : h) { sum += v; if (v > mx) mx = v; }
    const double per = (double)NB / TPA * reps; // neighbours per thread
    printf("%-34s blocks %3d: %7.1f cycles per neighbour per wave (mean), %7.1f (slowest wave)\n", name, blocks, sum / h.size() / per, mx / per);
    hipFree(out); hipFree(cyc);
}
int main()
{
    const double L = 6.17;
    std::vector<double> x(3 * N);
    srand(1);
    for (auto &v : x) v = L * (rand() / (double)RAND_MAX);
    std::vector<unsigned char> l(NB * N);
    for (auto &v : l) v = (unsigned char)(rand() & 255);
    double *dx; unsigned char *dl;
    hipMalloc(&dx, x.size() * 8); hipMalloc(&dl, l.size());
    hipMemcpy(dx, x.data(), x.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dl, l.data(), l.size(), hipMemcpyHostToDevice);
    for (int blocks : {256}) {
        run<1, 0>("W=1 full", blocks, dx, dl, L);
        run<2, 0>("W=2 full", blocks, dx, dl, L);
        run<4, 0>("W=4 full", blocks, dx, dl, L);
        run<2, 1>("W=2 no min-image", blocks, dx, dl, L);
        run<2, 2>("W=2 no reciprocal", blocks, dx, dl, L);
        run<2, 3>("W=2 full + barrier per eval", blocks, dx, dl, L);
        run<2, 4>("W=2 no position gathers", blocks, dx, dl, L);
        run<2, 5>("W=2 conflict-free gathers", blocks, dx, dl, L);
        run<4, 4>("W=4 no position gathers", blocks, dx, dl, L);
    }
    return 0;
}
