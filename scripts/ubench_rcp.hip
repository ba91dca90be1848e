// dev check: accuracy of v_rcp_f64 and of rcp + one / two Newton steps against IEEE division
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(double *out, int n)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    double m0 = 0, m1 = 0, m2 = 0;
    for (int r = 0; r < n; ++r) {
        unsigned long long s = (unsigned long long)(t * 7919 + r) * 6364136223846793005ull + 1442695040888963407ull;
        double a = 0.6 + 7.3 * (double)(s >> 11) * (1.0 / 9007199254740992.0); // r^2 range of the pair loop
        double y = __builtin_amdgcn_rcp(a), ex = 1.0 / a;
        double e0 = fabs(y - ex) / ex;
        double e = __builtin_fma(-a, y, 1.0); double y1 = __builtin_fma(y, e, y);
        double e1 = fabs(y1 - ex) / ex;
        e = __builtin_fma(-a, y1, 1.0); double y2 = __builtin_fma(y1, e, y1);
        double e2 = fabs(y2 - ex) / ex;
        m0 = fmax(m0, e0); m1 = fmax(m1, e1); m2 = fmax(m2, e2);
    }
    out[3 * t] = m0; out[3 * t + 1] = m1; out[3 * t + 2] = m2;
}
int main()
{
    const int T = 64 * 256;
    double *d; hipMalloc(&d, 3 * T * sizeof(double));
    hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, d, 4000);
    static double h[3 * T];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    double m[3] = {0, 0, 0};
    for (int i = 0; i < T; ++i) for (int c = 0; c < 3; ++c) m[c] = fmax(m[c], h[3 * i + c]);
    printf("max relative error over 6.5e7 samples: v_rcp_f64 %.3e (2^%.1f), +1 Newton %.3e (%.2f ulp), +2 Newton %.3e (%.2f ulp)\n",
           m[0], log2(m[0]), m[1], m[1] / 1.11e-16, m[2], m[2] / 1.11e-16);
    return 0;
}
