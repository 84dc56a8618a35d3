// Microbenchmark (developer tool): do v_mfma_f64_16x16x4_f64 and v_fma_f64 of two waves on the same
// SIMD overlap on gfx950?  One 512-thread workgroup per CU (8 waves: two per SIMD); waves 0-3 take
// role A, waves 4-7 role B.  Prints cycles per wave for each role combination.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(int roleA, int roleB, int n, long long* out, double* sink) {
    __shared__ double pad[12000];
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleA : roleB;
    if (threadIdx.x == 0) pad[0] = 0;
    __syncthreads();
    const long long t0 = clock64();
    double r = 0;
    if (role == 1) {                       // MFMA f64: three independent accumulators
        v4d a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0;
        const double x = 1.0 + threadIdx.x * 1e-9;
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
        }
        r = a0[0] + a1[1] + a2[2];
    } else if (role == 2) {                // VALU f64 FMA: twelve independent chains
        double c[12];
        for (int j = 0; j < 12; ++j) c[j] = threadIdx.x + j;
        const double m = 1.0000001, b = 1e-9;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) c[j] = fma(c[j], m, b);
        }
        for (int j = 0; j < 12; ++j) r += c[j];
    } else if (role == 3) {                // VALU 32-bit integer ops (v_mad_u32_u24-class multiply-add)
        int c[12];
        for (int j = 0; j < 12; ++j) c[j] = threadIdx.x + j;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) c[j] = c[j] * 3 + 1;
        }
        for (int j = 0; j < 12; ++j) r += c[j];
    } else if (role == 4) {                // VALU 32-bit xor/add (plain full-rate integer ops)
        unsigned c[12];
        for (int j = 0; j < 12; ++j) c[j] = threadIdx.x + j;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) c[j] = (c[j] ^ (unsigned)i) + 0x9e3779b9u;
        }
        for (int j = 0; j < 12; ++j) r += c[j];
    } else if (role == 5) {                // LDS: twelve independent ds_read_b64 per iteration, address chain through the data
        int a = (threadIdx.x & 63) * 8;
        double acc = 0;
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 12; ++j) acc += pad[(a + j * 64 + (i & 7)) & 8191];
        }
        r = acc;
    } else if (role == 6) {                // the 4x4x4 (4 blocks) f64 MFMA: three independent accumulators
        double a0 = 0, a1 = 0, a2 = 0;
        const double x = 1.0 + threadIdx.x * 1e-9;
        for (int i = 0; i < n; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(x, x, a2, 0, 0, 0);
        }
        r = a0 + a1 + a2;
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
    if (r == 12345.678) sink[0] = r + pad[1];
}
int main() {
    long long* out; double* sink;
    hipMalloc(&out, 256 * 8 * sizeof(long long)); hipMalloc(&sink, 8);
    long long h[2048];
    const char* names[] = {"idle", "mfma_f64", "fma_f64", "int32mad", "int32add", "lds_rd64", "mfma4x4x4"};
    const int n = 2000;
    for (int a = 1; a <= 6; ++a)
        for (int b = 0; b <= 6; ++b) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, n, out, sink);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, a, b, n, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
            double sa = 0, sb = 0;
            for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += h[i * 8 + w];
            printf("A=%-9s B=%-9s  cycles/iter (3 mfma | 12 valu | 12 lds): A %.1f  B %.1f\n", names[a], names[b], sa / 1024 / n, sb / 1024 / n);
        }
    return 0;
}
