#include <hip/hip_runtime.h>
#include <cstdio>
template <int CH, typename T>
__global__ void k_chain(T* out, int n, long long* cyc) {
    T a[CH]; for (int c = 0; c < CH; ++c) a[c] = out[c];
    T b = (T)1.0000001, d = (T)1e-9;
    long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = a[c] * b + d;
    }
    long long c1 = clock64();
    T s = 0; for (int c = 0; c < CH; ++c) s += a[c];
    out[threadIdx.x] = s; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_sincos(double* out, int n, long long* cyc) {
    double a = out[0] + 0.3; long long c0 = clock64();
    for (int i = 0; i < n; ++i) { double s, c; sincos(a, &s, &c); a = s + c * 0.25; }
    long long c1 = clock64(); out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_sqrtdiv(double* out, int n, long long* cyc) {
    double a = out[0] + 3.0; long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a = sqrt(a) + 2.0;
    }
    long long c1 = clock64(); out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_div8(double* out, int n, long long* cyc) {
    double a = out[0] + 3.0; long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) a = 1.0 / a + 1.5;
    }
    long long c1 = clock64(); out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_mfma(double* out, int n, long long* cyc) {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0, 0, 0, 0}; double a = out[threadIdx.x], b = 1.0;
    long long c0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    long long c1 = clock64(); out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3]; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
template <typename K> void run(const char* name, K k, double per, double* out, long long* cyc, int n) {
    k(out, n, cyc); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-40s %.2f cyc per op\n", name, (double)c / n / per);
}
int main() {
    double* out; float* outf; long long* cyc; hipMalloc(&out, 8192); hipMalloc(&outf, 8192); hipMalloc(&cyc, 64); hipMemset(out, 0, 8192); hipMemset(outf, 0, 8192);
    int n = 20000;
#define R(name, kern, per, o) { hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, o, n, cyc); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-44s %.2f cyc\n", name, (double)c / n / per); }
    R("f64 fma dependent (1 chain), per fma", (k_chain<1, double>), 16.0, out)
    R("f64 fma 2 chains, per fma", (k_chain<2, double>), 32.0, out)
    R("f64 fma 4 chains, per fma", (k_chain<4, double>), 64.0, out)
    R("f64 fma 8 chains, per fma", (k_chain<8, double>), 128.0, out)
    R("f32 fma dependent (1 chain), per fma", (k_chain<1, float>), 16.0, outf)
    R("f32 fma 8 chains, per fma", (k_chain<8, float>), 128.0, outf)
    R("f64 sincos + fma chain, per sincos", k_sincos, 1.0, out)
    R("f64 sqrt+add dependent, per op", k_sqrtdiv, 8.0, out)
    R("f64 div+add dependent, per op", k_div8, 8.0, out)
    R("mfma f64 16x16x4 dependent, per mfma", k_mfma, 8.0, out)
    return 0;
}
