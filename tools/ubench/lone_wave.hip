// micro-benchmark: what a LONE wave pays per instruction on gfx950 (the planner / dynamics kernels are one-wave-per-SIMD latency
// chains).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/lone_wave tools/ubench/lone_wave.hip && /tmp/lone_wave
// Times in core clocks from s_memrealtime (100 MHz) scaled by the measured clock of a known-rate loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define U 64
template <int CHAINS, int OP>
__global__ void k(double *out, unsigned long long *cyc, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = a + c + threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
#pragma unroll
            for (int c = 0; c < CHAINS; c++) {
                if (OP == 0) x[c] = fma(x[c], a, b);
                if (OP == 1) x[c] = x[c] * a;
                if (OP == 2) x[c] = x[c] + b;
                if (OP == 3) x[c] = sqrt(x[c]) + b;
                if (OP == 4) x[c] = a / x[c] + b;
                if (OP == 5) { float f = (float)x[c]; f = fmaf(f, (float)a, (float)b); x[c] = f; }
            }
        }
    }
    unsigned long long t1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int CHAINS>
__global__ void kf32(float *out, unsigned long long *cyc, int iters, float a, float b) {
    float x[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) x[c] = a + c + threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int c = 0; c < CHAINS; c++) x[c] = fmaf(x[c], a, b);
    }
    unsigned long long t1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
// dependent LDS reads (pointer chase), b32 and b64
__global__ void klds(int *out, unsigned long long *cyc, int iters) {
    __shared__ int tab[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab[i] = (i * 37 + 11) & 1023;
    __syncthreads();
    int p = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) p = tab[p];
    }
    unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
// dependent global reads (L2-resident pointer chase)
__global__ void kglob(const int *tab, int *out, unsigned long long *cyc, int iters, int mask) {
    int p = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) p = tab[p & mask];
    }
    unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = p;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
// a taken branch per iteration (wave-uniform condition the compiler cannot fold)
__global__ void kbranch(int *out, unsigned long long *cyc, int iters, const int *flags) {
    int acc = threadIdx.x;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters * U; i++) {
        if (flags[i & 7]) acc = acc * 3 + 1;      // s_load + branch
        else acc ^= 5;
    }
    unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
// readlane broadcast chain: v_readlane -> SGPR -> v op (what wave_bcast_f64 + add costs)
__global__ void kreadlane(double *out, unsigned long long *cyc, int iters) {
    double x = threadIdx.x * 0.5;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const long long b = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), u & 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), u & 63);
            x = x + __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        }
    }
    unsigned long long t1 = wall_clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
int main() {
    double *out; unsigned long long *cyc, h; int *tab;
    (void)hipMalloc(&out, 8 * 1024 * 1024); (void)hipMalloc(&cyc, 8); (void)hipMalloc(&tab, 4 << 20);
    int *ht = new int[1 << 20];
    for (int i = 0; i < (1 << 20); i++) ht[i] = (int)(((long long)i * 7919 + 13) & ((1 << 20) - 1));
    (void)hipMemcpy(tab, ht, 4 << 20, hipMemcpyHostToDevice);
    const int iters = 256;
    double mhz = 2400.0;
    auto run = [&](const char *name, auto launch, double ops) {
        launch(); (void)hipDeviceSynchronize(); launch(); (void)hipDeviceSynchronize();
        (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double ns = (double)h * 10.0 / ops;     // 100 MHz ticks
        printf("%-52s %8.2f ns/op = %7.1f clk @%.0f MHz\n", name, ns, ns * mhz / 1000.0, mhz);
        return ns;
    };
    for (int waves : {1, 4, 8}) {
        const int th = 64 * waves;
        char nm[128];
        printf("-- %d wave(s) on one CU (%s)\n", waves, waves == 1 ? "lone wave" : waves == 4 ? "one per SIMD" : "two per SIMD");
        snprintf(nm, 128, "f32 fma, 1 dependent chain"); run(nm, [&] { kf32<1><<<1, th>>>((float *)out, cyc, iters, 1.0000001f, 1e-9f); }, iters * U);
        snprintf(nm, 128, "f32 fma, 8 independent chains"); run(nm, [&] { kf32<8><<<1, th>>>((float *)out, cyc, iters, 1.0000001f, 1e-9f); }, iters * U * 8);
        snprintf(nm, 128, "f64 fma, 1 dependent chain"); run(nm, [&] { k<1, 0><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "f64 fma, 2 independent chains"); run(nm, [&] { k<2, 0><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 2);
        snprintf(nm, 128, "f64 fma, 4 independent chains"); run(nm, [&] { k<4, 0><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 4);
        snprintf(nm, 128, "f64 fma, 8 independent chains"); run(nm, [&] { k<8, 0><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 8);
        snprintf(nm, 128, "f64 mul, 1 dependent chain"); run(nm, [&] { k<1, 1><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "f64 add, 1 dependent chain"); run(nm, [&] { k<1, 2><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "f64 add, 4 independent chains"); run(nm, [&] { k<4, 2><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 4);
        snprintf(nm, 128, "f64 sqrt+add, 1 dependent chain"); run(nm, [&] { k<1, 3><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "f64 sqrt+add, 4 independent chains"); run(nm, [&] { k<4, 3><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 4);
        snprintf(nm, 128, "f64 div+add, 1 dependent chain"); run(nm, [&] { k<1, 4><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "f64 div+add, 4 independent chains"); run(nm, [&] { k<4, 4><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U * 4);
        snprintf(nm, 128, "f64->f32 fma->f64, 1 dependent chain (3 ops)"); run(nm, [&] { k<1, 5><<<1, th>>>(out, cyc, iters, 1.0000001, 1e-9); }, iters * U);
        snprintf(nm, 128, "LDS b32 dependent read chain"); run(nm, [&] { klds<<<1, th>>>((int *)out, cyc, iters); }, iters * U);
        snprintf(nm, 128, "global b32 dependent read chain, 4 KB table"); run(nm, [&] { kglob<<<1, th>>>(tab, (int *)out, cyc, iters, 1023); }, iters * 16);
        snprintf(nm, 128, "global b32 dependent read chain, 4 MB table"); run(nm, [&] { kglob<<<1, th>>>(tab, (int *)out, cyc, iters, (1 << 20) - 1); }, iters * 16);
        snprintf(nm, 128, "s_load + uniform branch + int op"); run(nm, [&] { kbranch<<<1, th>>>((int *)out, cyc, iters, tab); }, iters * U);
        snprintf(nm, 128, "readlane x2 -> f64 add, dependent"); run(nm, [&] { kreadlane<<<1, th>>>(out, cyc, iters); }, iters * U);
    }
    return 0;
}
