// dev micro-benchmark: what does a gather from an LDS table cost per wave instruction when all eight waves of a CU gather at once?
// 256-entry tables of doubles (the positions of a 256-atom replica); indices: consecutive per lane (conflict-free), random, or the
// pair loop's pattern (8 rows x 8 lanes, a row's lanes take consecutive entries of a sorted list).  ds_read_b64 from separate x, y, z
// arrays (what the kernel does), ds_read_b128 from an array of (x, y) pairs, ds_read_b128 from (x, y, z, -) quadruples.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 256, REP = 2048;
template <int MODE, int PAT>
__global__ void __launch_bounds__(512) k(double *out, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) double tab[N * 6 + 16];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < N * 6 + 16; i += 512) tab[i] = 1.0 + i;
    __syncthreads();
    unsigned int s = 12345u + 7919u * tid;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        unsigned int j;
        if (PAT == 0) j = (lane + r) & 255;                                   // consecutive per lane
        else if (PAT == 1) { s = s * 1664525u + 1013904223u; j = s >> 24; }     // random
        else { const unsigned int row = lane >> 3, sub = lane & 7; s = s * 1664525u + 1013904223u;
               unsigned int b = __shfl(s, lane & ~7) >> 24; j = (b + sub * 2 + (row & 1)) & 255; } // a row's lanes near one another
        if (MODE == 0) { acc += tab[j] + tab[N + j] + tab[2 * N + j]; }                                   // 3 x b64, SoA
        else if (MODE == 1) { const double2 xy = *(const double2 *)&tab[2 * j]; acc += xy.x + xy.y + tab[2 * N + j]; } // b128 (x,y) + b64 z
        else if (MODE == 2) { const double2 xy = *(const double2 *)&tab[4 * j]; const double2 zw = *(const double2 *)&tab[4 * j + 2]; acc += xy.x + xy.y + zw.x; } // 2 x b128, 32-byte records
        else if (MODE == 3) { const double2 xy = *(const double2 *)&tab[6 * j]; const double z = tab[6 * j + 2]; acc += xy.x + xy.y + z; } // 48-byte records: b128 + b64
        else if (MODE == 4) { acc += tab[j]; }                                                             // one b64
        else if (MODE == 5) { const double2 xy = *(const double2 *)&tab[2 * j]; acc += xy.x + xy.y; }      // one b128
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = acc;
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}
template <int MODE, int PAT>
void run(const char *name)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 8 * 512 * 256); hipMalloc(&cyc, 8 * 8 * 256);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(256), dim3(512), 0, 0, out, cyc);
    hipLaunchKernelGGL((k<MODE, PAT>), dim3(256), dim3(512), 0, 0, out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(8 * 256);
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double mx = 0; for (auto v : h) if (v > mx) mx = v;
    printf("%-64s %7.1f cycles per gathered atom and wave (8 waves per CU): %.1f per CU\n", name, mx / REP, mx / REP / 8.0);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<4, 0>("one ds_read_b64, consecutive"); run<4, 1>("one ds_read_b64, random"); run<4, 2>("one ds_read_b64, pair-loop pattern");
    run<5, 0>("one ds_read_b128, consecutive"); run<5, 1>("one ds_read_b128, random"); run<5, 2>("one ds_read_b128, pair-loop pattern");
    run<0, 0>("x, y, z: 3 x b64 (SoA), consecutive"); run<0, 1>("x, y, z: 3 x b64 (SoA), random"); run<0, 2>("x, y, z: 3 x b64 (SoA), pair-loop pattern");
    run<1, 1>("x, y, z: b128 (x,y) + b64 z, random"); run<1, 2>("x, y, z: b128 (x,y) + b64 z, pair-loop pattern");
    run<2, 1>("x, y, z: 2 x b128, 32-byte records, random"); run<2, 2>("x, y, z: 2 x b128, 32-byte records, pair-loop pattern");
    run<3, 1>("x, y, z: b128 + b64, 48-byte records, random"); run<3, 2>("x, y, z: b128 + b64, 48-byte records, pair-loop pattern");
    return 0;
}
