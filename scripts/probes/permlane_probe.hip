// Probe (run on the GPU box): what v_permlane16_swap / v_permlane32_swap do to two registers, lane by lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    const unsigned lane = threadIdx.x;
    unsigned a = 100 + lane, b = 200 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[lane] = r[0]; out[64 + lane] = r[1]; out[128 + lane] = q[0]; out[192 + lane] = q[1];
}
int main() {
    unsigned *d, h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap vdst(a=100+lane)", "permlane32_swap src (b=200+lane)", "permlane16_swap vdst", "permlane16_swap src "};
    for (int j = 0; j < 4; ++j) {
        printf("%s:", names[j]);
        for (int i = 0; i < 64; ++i) printf(" %u", h[64 * j + i]);
        printf("\n");
    }
    return 0;
}
