// micro-benchmark: FP64 VALU issue rate / dependent latency on gfx950 (calibrates the cost model in DESIGN.md)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CHAINS>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = a + c + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) x[c] = fma(x[c], a, b);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void ksq(double *out, long long *cyc, int iters, double a) {
    double x = a + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) x = sqrt(x) + a;
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
__global__ void kdiv(double *out, long long *cyc, int iters, double a) {
    double x = a + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) x = a / x + a;
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *out; long long *cyc, h;
    (void)hipMalloc(&out, 8 * 1024 * 1024); (void)hipMalloc(&cyc, 8);
    const int iters = 4096;
    auto run = [&](const char *name, auto launch, int ops) {
        launch(); (void)hipDeviceSynchronize(); launch(); (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-44s %8.2f cycles/op (per wave)\n", name, (double)h / iters / ops);
    };
    for (int waves : {1, 2, 4, 8}) {
        int threads = 64 * waves;   // one block on one CU: waves spread over the 4 SIMDs
        char nm[128];
        snprintf(nm, 128, "fma f64 1 chain, %d waves/CU", waves);  run(nm, [&] { k<1><<<1, threads>>>(out, cyc, iters, 1.0000001, 1e-9); }, 1);
        snprintf(nm, 128, "fma f64 4 chains, %d waves/CU", waves); run(nm, [&] { k<4><<<1, threads>>>(out, cyc, iters, 1.0000001, 1e-9); }, 4);
    }
    run("fma f64 8 chains, 16 waves/CU (4/SIMD)", [&] { k<8><<<1, 1024>>>(out, cyc, iters, 1.0000001, 1e-9); }, 8);
    run("sqrt f64 dependent, 1 wave", [&] { ksq<<<1, 64>>>(out, cyc, iters, 1.5); }, 1);
    run("div f64 dependent, 1 wave", [&] { kdiv<<<1, 64>>>(out, cyc, iters, 1.5); }, 1);
    return 0;
}
