// microbenchmark: effective shader clock + dependent-op latencies for small latency-bound kernels
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(double* out, int n, long long* cyc, long long* wall) {
    long long c0 = clock64(), w0 = wall_clock64();
    double a = out[0], b = 1.0000001, c = 1e-9;
    for (int i = 0; i < n; ++i) a = fma(a, b, c);
    long long c1 = clock64(), w1 = wall_clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) { cyc[0] = c1 - c0; wall[0] = w1 - w0; }
}
__global__ void k_lds(double* out, int n, long long* cyc) {
    __shared__ double s[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s[i] = (double)((i * 7 + 1) % 1024);
    __syncthreads();
    long long c0 = clock64();
    int idx = threadIdx.x;
    for (int i = 0; i < n; ++i) idx = (int)s[idx];
    long long c1 = clock64();
    out[threadIdx.x] = idx; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_bar(double* out, int n, long long* cyc) {
    long long c0 = clock64();
    for (int i = 0; i < n; ++i) __syncthreads();
    long long c1 = clock64();
    if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_div(double* out, int n, long long* cyc) {
    long long c0 = clock64();
    double a = out[0] + 3.0;
    for (int i = 0; i < n; ++i) a = 1.0 / a + 1.5;
    long long c1 = clock64();
    out[threadIdx.x] = a; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
__global__ void k_gld(const int* chain, double* out, int n, long long* cyc) {
    long long c0 = clock64();
    int idx = threadIdx.x;
    for (int i = 0; i < n; ++i) idx = chain[idx];
    long long c1 = clock64();
    out[threadIdx.x] = idx; if (threadIdx.x == 0) cyc[0] = c1 - c0;
}
int main() {
    double* out; long long *cyc, *wall; int* chain;
    hipMalloc(&out, 8192); hipMalloc(&cyc, 64); hipMalloc(&wall, 64); hipMalloc(&chain, 4 << 20);
    hipMemset(out, 0, 8192);
    int* hc = new int[1 << 20]; for (int i = 0; i < (1 << 20); ++i) hc[i] = (i * 4097 + 12345) & ((1 << 20) - 1);
    hipMemcpy(chain, hc, 4 << 20, hipMemcpyHostToDevice);
    long long c, w; hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms;
    for (int rep = 0; rep < 3; ++rep) {
        int n = 100000;
        hipEventRecord(e0); hipLaunchKernelGGL(k_fma, dim3(1), dim3(64), 0, 0, out, n, cyc, wall); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, wall, 8, hipMemcpyDeviceToHost);
        printf("fma chain n=%d: %.1f us event, clock64 %lld (%.2f cyc/op), wall_clock64 %lld ticks => shader clk %.0f MHz (if wall=100MHz)\n", n, ms * 1e3, c, (double)c / n, w, (double)c / ((double)w / 100.0));
    }
    int n = 20000;
    hipLaunchKernelGGL(k_lds, dim3(1), dim3(64), 0, 0, out, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("lds dependent read (+cvt): %.1f cyc/op\n", (double)c / n);
    hipLaunchKernelGGL(k_bar, dim3(1), dim3(64), 0, 0, out, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("barrier 1 wave: %.1f cyc\n", (double)c / n);
    hipLaunchKernelGGL(k_bar, dim3(1), dim3(256), 0, 0, out, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("barrier 4 waves: %.1f cyc\n", (double)c / n);
    hipLaunchKernelGGL(k_bar, dim3(1), dim3(1024), 0, 0, out, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("barrier 16 waves: %.1f cyc\n", (double)c / n);
    hipLaunchKernelGGL(k_div, dim3(1), dim3(64), 0, 0, out, n, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("f64 div+add chain: %.1f cyc/op\n", (double)c / n);
    hipLaunchKernelGGL(k_gld, dim3(1), dim3(64), 0, 0, chain, out, 2000, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("global dependent load (4 MB chain, 64 lanes divergent): %.1f cyc/op\n", (double)c / 2000);
    hipLaunchKernelGGL(k_gld, dim3(1), dim3(64), 0, 0, chain, out, 2000, cyc); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("  again (warm): %.1f cyc/op\n", (double)c / 2000);
    // empty-kernel chain latency
    hipEventRecord(e0); for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_bar, dim3(1), dim3(64), 0, 0, out, 0, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); printf("100 empty dependent launches: %.2f us each\n", ms * 10);
    return 0;
}
