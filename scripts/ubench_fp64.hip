// dev micro-benchmark: issue cost (cycles per wave64 instruction on one SIMD) of the fp64 instructions the pair loop uses
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int K>
__global__ void k(double *out, unsigned long long *cyc, double a, double b)
{
    double v0 = a + threadIdx.x, v1 = a * 1.1, v2 = a * 1.2, v3 = a * 1.3, v4 = a * 1.4, v5 = a * 1.5, v6 = a * 1.6, v7 = a * 1.7;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (K == 0) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v##i) : "v"(b));
            OPS8(S)
#undef S
        } else if (K == 1) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v##i) : "v"(b));
            OPS8(S)
#undef S
        } else if (K == 2) {
#define S(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v##i) : "v"(b));
            OPS8(S)
#undef S
        } else if (K == 3) {
#define S(i) asm volatile("v_rndne_f64 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 4) {
#define S(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 5) {
#define S(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v##i), "v"(b) : "vcc");
            OPS8(S)
#undef S
        } else if (K == 6) {
#define S(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(*(float *)&v##i) : "v"((float)b));
            OPS8(S)
#undef S
        } else if (K == 7) {
#define S(i) asm volatile("v_floor_f64 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 8) {
#define S(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 9) {
#define S(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(*(float *)&v##i) : "v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 10) {
#define S(i) asm volatile("v_fract_f64 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 11) {
#define S(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(*(int *)&v##i) : : );
            OPS8(S)
#undef S
        } else if (K == 12) {
#define S(i) asm volatile("v_fma_f64 %0, %0, %1, %1 clamp" : "+v"(v##i) : "v"(b));
            OPS8(S)
#undef S
        } else if (K == 13) {
#define S(i) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v##i) : "v"(b));
            OPS8(S)
#undef S
        } else if (K == 14) {
#define S(i) asm volatile("v_cmp_lt_f64 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(*(int *)&v##i) : "v"(v##i), "v"(b) : "vcc");
            OPS8(S)
#undef S
        } else if (K == 15) {
#define S(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v##i));
            OPS8(S)
#undef S
        } else if (K == 16) {
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(*(float *)&v##i));
            OPS8(S)
#undef S
        } else if (K == 17) {
#define S(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v##i) : "v"(*(float *)&v##i));
            OPS8(S)
#undef S
        } else if (K == 18) {
#define S(i) asm volatile("v_mov_b32 %0, %1" : "=v"(*(int *)&v##i) : "v"(*(int *)&v##i));
            OPS8(S)
#undef S
        } else if (K == 19) {
#define S(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %0" : "+v"(*(int *)&v##i));
            OPS8(S)
#undef S
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K>
void run(const char *name, int threads)
{
    double *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * 4096); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(k<K>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.000001, 0.999999);
    hipLaunchKernelGGL(k<K>, dim3(1), dim3(threads), 0, 0, out, cyc, 1.000001, 0.999999);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s threads %4d: %6.2f cycles per wave-instruction (per wave)\n", name, threads, (double)h / (REP * 8.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int th : {64, 512}) {
        run<0>("v_fma_f64", th); run<1>("v_add_f64", th); run<2>("v_mul_f64", th); run<3>("v_rndne_f64", th); run<7>("v_floor_f64", th);
        run<4>("v_rcp_f64", th); run<8>("v_rsq_f64", th); run<5>("v_cmp_lt_f64", th); run<6>("v_fma_f32", th); run<9>("v_cvt_f32_f64", th);
        run<10>("v_fract_f64", th); run<11>("v_cndmask_b32", th); run<12>("v_fma_f64 clamp", th); run<13>("v_max_f64", th); run<14>("v_cmp_lt_f64 + v_cndmask_b32", th);
        run<15>("v_ldexp_f64", th); run<16>("v_rcp_f32", th); run<17>("v_cvt_f64_f32", th); run<18>("v_mov_b32", th); run<19>("v_lshl_add_u32", th);
        printf("\n");
    }
    return 0;
}
