// Microbenchmark: issue cost (shader cycles per wave-instruction) of the FP64
// VALU operations the tree walk is made of, one wave per SIMD and 8 waves per
// SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/microbench_f64.hip -o /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, unsigned long long* cyc, double seed)
{
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double b = seed * 0.5 + 1.0;
    float f0 = (float)a0, f1 = (float)a1, f2 = (float)a2, f3 = (float)a3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 1
    for (int i = 0; i < REP; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a0 = __builtin_fma(a0, b, b); a1 = __builtin_fma(a1, b, b); a2 = __builtin_fma(a2, b, b); a3 = __builtin_fma(a3, b, b); }
            if (OP == 1) { a0 = a0 * b; a1 = a1 * b; a2 = a2 * b; a3 = a3 * b; }
            if (OP == 2) { a0 = a0 + b; a1 = a1 + b; a2 = a2 + b; a3 = a3 + b; }
            if (OP == 3) { a0 = __builtin_amdgcn_rsq(a0); a1 = __builtin_amdgcn_rsq(a1); a2 = __builtin_amdgcn_rsq(a2); a3 = __builtin_amdgcn_rsq(a3); }
            if (OP == 4) { f0 = __builtin_amdgcn_rsqf(f0); f1 = __builtin_amdgcn_rsqf(f1); f2 = __builtin_amdgcn_rsqf(f2); f3 = __builtin_amdgcn_rsqf(f3); }
            if (OP == 5) { f0 = __builtin_fmaf(f0, f1, f2); f1 = __builtin_fmaf(f1, f2, f3); f2 = __builtin_fmaf(f2, f3, f0); f3 = __builtin_fmaf(f3, f0, f1); }
            if (OP == 6) { a0 = (double)(float)a0; a1 = (double)(float)a1; a2 = (double)(float)a2; a3 = (double)(float)a3; }
            if (OP == 7) { a0 = (a0 < b) ? a1 : a0; a1 = (a1 < b) ? a2 : a1; a2 = (a2 < b) ? a3 : a2; a3 = (a3 < b) ? a0 : a3; }
            if (OP == 8) { a0 = __builtin_amdgcn_rcp(a0); a1 = __builtin_amdgcn_rcp(a1); a2 = __builtin_amdgcn_rcp(a2); a3 = __builtin_amdgcn_rcp(a3); }
            if (OP == 9) { a0 = __builtin_sqrt(a0); a1 = __builtin_sqrt(a1); a2 = __builtin_sqrt(a2); a3 = __builtin_sqrt(a3); }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, int ops_per_iter, int blocks)
{
    double* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&cyc, blocks * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.000001);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += v;
    s /= h.size();
    printf("%-28s blocks=%5d cycles per wave-instruction = %.2f\n", name, blocks, s / (REP * 8.0 * ops_per_iter));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {256, 2048}) {   // 1 wave per SIMD ; 8 waves per SIMD
        run<0>("v_fma_f64", 4, blocks);
        run<1>("v_mul_f64", 4, blocks);
        run<2>("v_add_f64", 4, blocks);
        run<3>("v_rsq_f64", 4, blocks);
        run<8>("v_rcp_f64", 4, blocks);
        run<9>("sqrt(double) library", 4, blocks);
        run<4>("v_rsq_f32", 4, blocks);
        run<5>("v_fma_f32", 4, blocks);
        run<6>("cvt f64->f32->f64 (2 instr)", 4, blocks);
        run<7>("v_cmp_f64 + 2 cndmask", 4, blocks);
    }
    return 0;
}
