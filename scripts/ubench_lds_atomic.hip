// dev micro-benchmark (round 4, the half-list experiment): what does an LDS atomic add cost per wave instruction when all eight waves of a
// CU scatter at once, next to the gather it would replace?  2048-entry tables (the force array of an 8^3 replica); targets: consecutive
// per lane, random, or the one-thread-per-row pattern (lane l's target near l's own index: neighbours of consecutive atoms).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 2048, REP = 1024;
template <int MODE, int PAT>
__global__ void __launch_bounds__(512) k(double *out, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tab[N * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < N * 3; i += 512) tab[i] = 0;
    __syncthreads();
    unsigned int s = 12345u + 7919u * tid;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        unsigned int j;
        s = s * 1664525u + 1013904223u;
        if (PAT == 0) j = (tid + r) & (N - 1);
        else if (PAT == 1) j = (s >> 8) & (N - 1);
        else j = (tid * 4 + ((s >> 20) & 255)) & (N - 1);   // within 256 atoms of the lane's own (spatially near, index-near)
        if (MODE == 0) { acc += __longlong_as_double((long long)tab[j]); }                                                      // ds_read_b64
        else if (MODE == 1) { __hip_atomic_fetch_add((unsigned long long *)&tab[j], (unsigned long long)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }   // ds_add_u64
        else if (MODE == 2) { __hip_atomic_fetch_add((unsigned int *)&tab[j], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }                              // ds_add_u32
        else if (MODE == 3) { __hip_atomic_fetch_add((double *)&tab[j], (double)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }                             // ds_add_f64
        else if (MODE == 4) { __hip_atomic_fetch_add((float *)&tab[j], (float)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }                               // ds_add_f32
        else if (MODE == 5) { tab[j] = s; }                                                                                                                       // ds_write_b64
        else if (MODE == 6) { __hip_atomic_fetch_add((unsigned int *)&tab[j], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                              __hip_atomic_fetch_add((unsigned int *)&tab[j] + 1, s >> 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }                      // two ds_add_u32 (limbs)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    out[blockIdx.x * 512 + tid] = acc + (double)tab[tid];
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
    printf("%-44s %7.1f cycles per wave instruction (8 waves per CU): %.1f per CU\n", name, mx / REP, mx / REP / 8.0);
    hipFree(out); hipFree(cyc);
}
int main()
{
    run<0, 0>("ds_read_b64, consecutive"); run<0, 1>("ds_read_b64, random"); run<0, 2>("ds_read_b64, near");
    run<5, 0>("ds_write_b64, consecutive"); run<5, 1>("ds_write_b64, random"); run<5, 2>("ds_write_b64, near");
    run<1, 0>("ds_add_u64, consecutive"); run<1, 1>("ds_add_u64, random"); run<1, 2>("ds_add_u64, near");
    run<3, 0>("ds_add_f64, consecutive"); run<3, 1>("ds_add_f64, random"); run<3, 2>("ds_add_f64, near");
    run<2, 0>("ds_add_u32, consecutive"); run<2, 1>("ds_add_u32, random"); run<2, 2>("ds_add_u32, near");
    run<4, 0>("ds_add_f32, consecutive"); run<4, 1>("ds_add_f32, random"); run<4, 2>("ds_add_f32, near");
    run<6, 1>("2 x ds_add_u32 (two limbs), random"); run<6, 2>("2 x ds_add_u32 (two limbs), near");
    return 0;
}
