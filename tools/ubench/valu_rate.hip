// VALU issue-rate micro-benchmark (gfx950): 30-long chains of independent-register v_med3_f32 / v_min+v_max / v_fma_f32 / v_max3_f32 /
// v_max_i32 / v_add_f32 / v_mul_f32 per loop iteration, at 1, 2, 4 and 8 wavefronts per SIMD on every CU.  Prints ns and cycles per wave-instruction per SIMD
// (cycles from the kernel's own s_memtime / s_memrealtime clock ratio).
// build + run on the GPU box: hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// (-fno-slp-vectorize: otherwise the compiler packs neighbouring chains into v_pk_* instructions)
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 30
template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed, unsigned long long *clk) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
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
            if (OP == 5) s[j] = s[j] + x;                                                       // v_add_f32, two VGPR sources
            if (OP == 6) s[j] = s[j] * 1.0001f;                                                 // v_mul_f32, one VGPR source + a literal
            if (OP == 7) s[j] = __builtin_fmaf(x, x, s[j]);                                     // v_fma_f32 with two distinct VGPR sources
            if (OP == 8) s[j] = __builtin_fmaf(s[j], 1.0001f, 0.5f);                            // v_fmaak / v_fma with one VGPR source
        }
        x += 1.0f;
    }
    float r = 0; for (int j = 0; j < N; j++) r += s[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}
template <int OP> void run(const char *name, int per_iter, int wps) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    unsigned long long *clk; hipMalloc(&clk, 16); unsigned long long h[2];
    const int blocks = 256 * wps, iters = 20000;      // wps workgroups of 4 wavefronts per CU = wps wavefronts per SIMD
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<OP><<<blocks, 256>>>(out, 100, 1.0f, clk); hipDeviceSynchronize();
    hipEventRecord(a); k<OP><<<blocks, 256>>>(out, iters, 1.0f, clk); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double ghz = (double)h[0] / ((double)h[1] * 10.0);                   // s_memrealtime ticks at 100 MHz
    const double winst = (double)blocks * 4 * iters * per_iter;               // wave-instructions
    const double ns = ms * 1e6 / (winst / 1024);
    printf("%-24s %d waves/SIMD %8.3f ms  %7.1f G wave-instr/s chip  %.3f ns per wave-instr per SIMD  clock %.2f GHz -> %.2f cycles\n", name, wps, ms, winst / ms / 1e6, ns, ghz, ns * ghz);
    hipFree(out); hipFree(clk);
}
int main() {
    for (int wps : {1, 2, 4, 8}) {
        run<0>("v_med3_f32", N - 1, wps); run<1>("v_min_f32 + v_max_f32", 2 * (N - 1), wps); run<2>("v_fma_f32", N - 1, wps);
        run<3>("v_max3_f32", N - 1, wps); run<4>("v_max_i32", N - 1, wps);
        run<5>("v_add_f32 (2 VGPR)", N - 1, wps); run<6>("v_mul_f32 (1 VGPR + lit)", N - 1, wps); run<7>("v_fma_f32 (x, x, s)", N - 1, wps); run<8>("v_fma (1 VGPR + 2 lit)", N - 1, wps);
    }
    return 0;
}
