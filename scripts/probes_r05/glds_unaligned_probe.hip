// Probe (round 5): does global_load_lds_dwordx4 take a BYTE-unaligned per-lane source address, as the unaligned global_load_dwordx4 of
// the token kernels does?  Each lane DMAs 16 bytes from src + lane * 16 + shift(lane) into the linear LDS image; the image is compared
// with the bytes a plain load of the same address returns.  Build: hipcc --offload-arch=gfx950 -O2 glds_unaligned_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;
__global__ void k(const uint8_t *src, uint8_t *out, int shift_mode) {
    __shared__ __align__(16) uint8_t img[1024];
    const int lane = threadIdx.x;
    const int shift = shift_mode == 0 ? 0 : (shift_mode < 16 ? shift_mode : (lane * 7 + shift_mode) % 16);
    const uint8_t *p = src + lane * 16 + shift;
    __builtin_amdgcn_global_load_lds((glb_void *)p, (lds_void *)img, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = img[lane * 16 + i];
}
int main() {
    std::vector<uint8_t> h(4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = uint8_t(i * 131 + 7);
    uint8_t *d_src, *d_out;
    hipMalloc(&d_src, 4096); hipMalloc(&d_out, 1024);
    hipMemcpy(d_src, h.data(), 4096, hipMemcpyHostToDevice);
    int bad_modes = 0;
    for (int mode : {0, 1, 2, 3, 4, 5, 8, 12, 15, 16, 21}) {
        hipMemset(d_out, 0xEE, 1024);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_src, d_out, mode);
        std::vector<uint8_t> o(1024);
        hipError_t e = hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int shift = mode == 0 ? 0 : (mode < 16 ? mode : (lane * 7 + mode) % 16);
            for (int i = 0; i < 16; ++i) bad += o[lane * 16 + i] != h[lane * 16 + shift + i];
        }
        std::printf("shift mode %2d: %s (%d bad bytes, hip %d)\n", mode, bad ? "MISMATCH" : "ok", bad, int(e));
        bad_modes += bad != 0;
    }
    std::printf(bad_modes ? "GLDS_UNALIGNED_NOT_SUPPORTED\n" : "GLDS_UNALIGNED_OK\n");
    return 0;
}
