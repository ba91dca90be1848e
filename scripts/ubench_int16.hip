// dev micro-benchmark: issue cost (cycles per wave64 instruction on one SIMD) of the integer instructions the list rebuild's
// candidate test uses (Replica::test16), and of two whole-test sequences
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
#define OPS8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int K>
__global__ void k(unsigned int *out, unsigned long long *cyc, unsigned int a, unsigned int b)
{
    unsigned int v0 = a + threadIdx.x, v1 = a * 3, v2 = a * 5, v3 = a * 7, v4 = a * 9, v5 = a * 11, v6 = a * 13, v7 = a * 17;
    unsigned int w0 = b + threadIdx.x, w1 = b * 3, w2 = b * 5, w3 = b * 7, w4 = b * 9, w5 = b * 11, w6 = b * 13, w7 = b * 17, m = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < REP; ++r) {
        if (K == 0) {
#define S(i) asm volatile("v_dot2c_i32_i16 %0, %1, %1" : "+v"(v##i) : "v"(w##i));
            OPS8(S)
#undef S
        } else if (K == 1) {
#define S(i) asm volatile("v_dot2_i32_i16 %0, %1, %1, 0" : "=v"(v##i) : "v"(w##i));
            OPS8(S)
#undef S
        } else if (K == 2) {
#define S(i) asm volatile("v_mad_i32_i16 %0, %1, %1, %0" : "+v"(v##i) : "v"(w##i));
            OPS8(S)
#undef S
        } else if (K == 3) {
#define S(i) asm volatile("v_mad_i32_i16 %0, %1, %1, %0 op_sel:[1,1,0,0]" : "+v"(v##i) : "v"(w##i));
            OPS8(S)
#undef S
        } else if (K == 4) {
#define S(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v##i) : "v"(w##i));
            OPS8(S)
#undef S
        } else if (K == 5) {
#define S(i) asm volatile("v_cmp_gt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v##i), "v"(w##i) : "vcc");
            OPS8(S)
#undef S
        } else if (K == 6) {
#define S(i) asm volatile("v_readlane_b32 s20, %0, " #i "\n\tv_add_u32 %0, s20, %0" : "+v"(v##i) : : "s20");
            OPS8(S)
#undef S
        } else if (K == 7) { // the whole test as the compiler emits it: mov, 2 pk_sub, 2 dot2c, cmp, addc
#define S(i) asm volatile("v_pk_sub_i16 v40, %1, %2\n\tv_pk_sub_i16 v41, %1, %3\n\tv_mov_b32 v42, 0\n\tv_dot2c_i32_i16 v42, v40, v40\n\tv_dot2c_i32_i16 v42, v41, v41\n\ts_nop 2\n\tv_cmp_gt_u32 vcc, %3, v42\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v##i), "v"(w##i), "v"(v0) : "vcc", "v40", "v41", "v42");
            OPS8(S)
#undef S
        } else if (K == 8) { // the whole test with three v_mad_i32_i16 instead
#define S(i) asm volatile("v_pk_sub_i16 v40, %1, %2\n\tv_pk_sub_i16 v41, %1, %3\n\tv_mad_i32_i16 v42, v40, v40, 0\n\tv_mad_i32_i16 v42, v40, v40, v42 op_sel:[1,1,0,0]\n\tv_mad_i32_i16 v42, v41, v41, v42\n\tv_cmp_gt_u32 vcc, %3, v42\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(m) : "v"(v##i), "v"(w##i), "v"(v0) : "vcc", "v40", "v41", "v42");
            OPS8(S)
#undef S
        } else if (K == 9) {
#define S(i) asm volatile("v_fract_f32 %0, %0" : "+v"(v##i));
            OPS8(S)
#undef S
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + m;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K>
void run(const char *name, int threads)
{
    unsigned int *out; unsigned long long *cyc;
    hipMalloc(&out, sizeof(unsigned int) * 4096); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(k<K>, dim3(1), dim3(threads), 0, 0, out, cyc, 12345u, 6789u);
    hipLaunchKernelGGL(k<K>, dim3(1), dim3(threads), 0, 0, out, cyc, 12345u, 6789u);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s threads %4d: %6.2f cycles per statement (per wave)\n", name, threads, (double)h / (REP * 8.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int th : {64, 512}) {
        run<0>("v_dot2c_i32_i16", th); run<1>("v_dot2_i32_i16 (VOP3P, acc 0)", th); run<2>("v_mad_i32_i16", th); run<3>("v_mad_i32_i16 op_sel hi", th);
        run<4>("v_pk_sub_i16", th); run<5>("v_cmp_gt_u32 + v_addc_co_u32", th); run<6>("v_readlane + v_add (s operand)", th);
        run<9>("v_fract_f32", th);
        run<7>("test: mov 2sub 2dot2c nop cmp addc", th); run<8>("test: 2sub 3mad_i32_i16 cmp addc", th);
        printf("\n");
    }
    return 0;
}
