// Probe (developer tool): operand layout of v_mfma_f64_4x4x4_4b_f64 on gfx950.  For every pair of lanes
// (la, lb) the A operand is 1 in lane la only and the B operand 1 in lane lb only; the lane of D that
// becomes 1 (if any) is printed as a table, from which block / row / column / k of each lane follow.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? (int)__builtin_ctzll(m) + 100 * (__builtin_popcountll(m) - 1) : -1;
        }
}
int main() {
    int* out; (void)hipMalloc(&out, 4096 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out);
    (void)hipDeviceSynchronize();
    static int h[4096];
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb] >= 0) printf(" B%d->D%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
