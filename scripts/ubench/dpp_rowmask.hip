// Probe: does the DPP row mask apply to 64-bit DPP (v_fmac_f64_dpp ... row_newbcast)?  Each lane starts with
// r = lane, c = 1; "r += r[lane 3 of the row] * c" with row_mask 0x2 must change the lanes 16..31 only (to lane + 19).
// Also: v_permlane16_swap / v_permlane32_swap semantics on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out, int* iout) {
    const int l = threadIdx.x;
    double r = (double)l, c = 1.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:2 bank_mask:0xf" : "+v"(r) : "v"(c));
    out[l] = r;
    double a = 0.0, s = 100.0 + l;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:12 bank_mask:0xf" : "+v"(a) : "v"(s), "v"(c));
    out[64 + l] = a;
    int x = l, y = 1000 + l;
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    iout[l] = x; iout[64 + l] = y;
    x = l; y = 1000 + l;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    iout[128 + l] = x; iout[192 + l] = y;
}
// second probe: the compiler's builtin (hazards handled by the compiler) and explicit nops around the asm form
__global__ void probe2(int* iout) {
    const int l = threadIdx.x;
    unsigned x = l, y = 1000 + l;
    auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
    iout[l] = r[0]; iout[64 + l] = r[1];
    int a = l, b = 1000 + l;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 7\n\ts_nop 7" : "+v"(a), "+v"(b));
    iout[128 + l] = a; iout[192 + l] = b;
}
int main() {
    double* d; int* di; hipMalloc(&d, 128 * 8); hipMalloc(&di, 256 * 4);
    probe<<<1, 64>>>(d, di);
    double h[128]; int hi[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hi, di, sizeof hi, hipMemcpyDeviceToHost);
    printf("fmac_dpp r,r,c row_newbcast:3 row_mask:2 (r = lane):\n");
    for (int l = 0; l < 64; ++l) printf("%g%c", h[l], l % 16 == 15 ? '\n' : ' ');
    printf("fmac_dpp a,s,c row_newbcast:5 row_mask:12 (s = 100 + lane, a = 0):\n");
    for (int l = 0; l < 64; ++l) printf("%g%c", h[64 + l], l % 16 == 15 ? '\n' : ' ');
    const char* nm[4] = {"permlane16_swap x (x = lane, y = 1000 + lane)", "permlane16_swap y", "permlane32_swap x", "permlane32_swap y"};
    for (int k = 0; k < 4; ++k) {
        printf("%s:\n", nm[k]);
        for (int l = 0; l < 64; ++l) printf("%d%c", hi[64 * k + l], l % 16 == 15 ? '\n' : ' ');
    }
    probe2<<<1, 64>>>(di);
    hipMemcpy(hi, di, sizeof hi, hipMemcpyDeviceToHost);
    const char* nm2[4] = {"builtin permlane16_swap [0] (old = lane, src = 1000 + lane)", "builtin [1]", "asm with nops x", "asm with nops y"};
    for (int k = 0; k < 4; ++k) {
        printf("%s:\n", nm2[k]);
        for (int l = 0; l < 64; ++l) printf("%d%c", hi[64 * k + l], l % 16 == 15 ? '\n' : ' ');
    }
    return 0;
}
