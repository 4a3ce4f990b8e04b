// What do gfx950's v_permlane32_swap / v_permlane16_swap return?  (Used by the wavefront sum of the coded SpMV for the
// strides 32 and 16: lane i must see lane i + 32 / i + 16.)  Build: hipcc --offload-arch=gfx950 -O3 permlane_probe.hip -o permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out, const int *in) {
    const int v = in[threadIdx.x];
    auto r32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    out[threadIdx.x] = r32[0];
    out[64 + threadIdx.x] = r32[1];
    out[128 + threadIdx.x] = r16[0];
    out[192 + threadIdx.x] = r16[1];
}
int main() {
    int h[64], o[256], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&di, sizeof(h));
    hipMalloc(&dout, sizeof(o));
    hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
    k<<<1, 64>>>(dout, di);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    const char *names[4] = {"permlane32_swap[0]", "permlane32_swap[1]", "permlane16_swap[0]", "permlane16_swap[1]"};
    for (int a = 0; a < 4; ++a) {
        printf("%s:", names[a]);
        for (int i = 0; i < 64; ++i) printf(" %d", o[64 * a + i]);
        printf("\n");
    }
    return 0;
}
