// rcp_accuracy.hip -- relative error of v_rcp_f64 / v_rsq_f64 + n Newton steps on gfx950
// (justifies SPH_NEWTON_STEPS in pysph_amd/csrc/sph_pair.h).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double d = x[i];
    double r0 = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r0, 1.0);
    double r1 = fma(r0, e, r0);
    e = fma(-d, r1, 1.0);
    double r2 = fma(r1, e, r1);
    double y = __builtin_amdgcn_rsq(d);
    double g = d * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    double g1 = fma(g, r, g), h1 = fma(h, r, h);
    r = fma(-h1, g1, 0.5);
    double g2 = fma(g1, r, g1), h2 = fma(h1, r, h1);
    o[8 * i + 0] = r0; o[8 * i + 1] = r1; o[8 * i + 2] = r2;
    o[8 * i + 3] = y; o[8 * i + 4] = 2 * h1; o[8 * i + 5] = 2 * h2; o[8 * i + 6] = g1; o[8 * i + 7] = g2;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), o(8 * n);
    srand(1);
    for (int i = 0; i < n; i++) x[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0) * (1.0 + rand() / (double)RAND_MAX);
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 8 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, 8 * n * 8, hipMemcpyDeviceToHost);
    double m[8] = {};
    for (int i = 0; i < n; i++) {
        const long double d = x[i], rc = 1.0L / d, rs = 1.0L / sqrtl(d), sq = sqrtl(d);
        const long double ref[8] = {rc, rc, rc, rs, rs, rs, sq, sq};
        for (int k2 = 0; k2 < 8; k2++) { double e = (double)fabsl((o[8 * i + k2] - ref[k2]) / ref[k2]); if (e > m[k2]) m[k2] = e; }
    }
    printf("max relative error over %d samples:\n rcp  hw %.3e  1 step %.3e  2 steps %.3e\n rsq  hw %.3e  1 step %.3e  2 steps %.3e\n sqrt           1 step %.3e  2 steps %.3e\n",
           n, m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7]);
    return 0;
}
