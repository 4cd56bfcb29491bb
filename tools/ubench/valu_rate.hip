// VALU issue-rate micro-benchmark (gfx950): 30-long chains of independent-register v_med3_f32 / v_min+v_max / v_fma_f32 / v_max3_f32 /
// v_pk_fma_f32 per loop iteration, 4 wavefronts per SIMD on every CU.  Prints cycles per wave-instruction per SIMD.
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 30
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed) {
    float s[N];
    for (int j = 0; j < N; j++) s[j] = seed + j + threadIdx.x;
    float x = seed * 0.5f + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int j = 0; j + 1 < N; j++) {
            if (OP == 0) s[j] = __builtin_amdgcn_fmed3f(x, s[j], s[j + 1]);
            if (OP == 1) s[j] = fmaxf(fminf(x, s[j]), s[j + 1]);                       // 2 instructions
            if (OP == 2) s[j] = __builtin_fmaf(x, s[j], s[j + 1]);
            if (OP == 3) s[j] = __builtin_fmaxf(__builtin_fmaxf(x, s[j]), s[j + 1]);   // v_max3_f32
            if (OP == 4) s[j] = __int_as_float(max(__float_as_int(x), __float_as_int(s[j + 1])));
        }
        x += 1.0f;
    }
    float r = 0; for (int j = 0; j < N; j++) r += s[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int OP> void run(const char *name, int per_iter) {
    float *out; hipMalloc(&out, 256 * 4 * 256 * 4 * sizeof(float));
    const int blocks = 256 * 4, iters = 20000;      // 4 workgroups of 4 wavefronts per CU = 4 wavefronts per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 100, 1.0f); hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<blocks, 256>>>(out, iters, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double winst = (double)blocks * 4 * iters * per_iter;               // wave-instructions
    printf("%-28s %8.3f ms  %.1f G wave-instr/s chip  -> %.2f ns per wave-instr per SIMD (x clock = cycles)\n", name, ms, winst / ms / 1e6, ms * 1e6 / (winst / 1024));
    hipFree(out);
}
int main() {
    run<0>("v_med3_f32", N - 1); run<1>("v_min_f32 + v_max_f32", 2 * (N - 1)); run<2>("v_fma_f32", N - 1); run<3>("v_max3_f32", N - 1); run<4>("v_max_i32", N - 1);
    return 0;
}
