// tools/mfma_round_probe.hip — how does v_mfma_f32_16x16x32_f16 / _bf16 round its fp32 accumulation on gfx950?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/mfma_round_probe.hip -o /tmp/probe && /tmp/probe
// Each case: D = A * B + C with A = one row of constants, B = one column, so D[0][0] = sum_k a_k b_k + c.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// a[k], b[k] for k = 0..31 (row 0 of A, column 0 of B; everything else zero), c = C[0][0]
__global__ void probe(const float* a, const float* b, float c, float* out, int chain)
{
    const int lane = threadIdx.x;
    f16x8 av, bv;
    for (int j = 0; j < 8; j++) {
        const int k = 8 * (lane >> 4) + j;
        av[j] = (lane & 15) == 0 ? (_Float16)a[k] : (_Float16)0.0f;
        bv[j] = (lane & 15) == 0 ? (_Float16)b[k] : (_Float16)0.0f;
    }
    f32x4 acc = {0, 0, 0, 0};
    if (lane == 0) acc[0] = c;
    for (int i = 0; i < chain; i++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static float run(const float* a, const float* b, float c, int chain = 1)
{
    float *da, *db, *dout, r;
    hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 4);
    hipMemcpy(da, a, 128, hipMemcpyHostToDevice); hipMemcpy(db, b, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, c, dout, chain);
    hipMemcpy(&r, dout, 4, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return r;
}

int main()
{
    float a[32], b[32];
    const float u = ldexpf(1.0f, -23);   // ulp of 1.0
    auto clr = [&] { memset(a, 0, sizeof a); memset(b, 0, sizeof b); };
    // 1. one product of 0.75 ulp on top of c = 1: nearest = 1 + ulp, toward zero = 1
    clr(); a[0] = 0.75f; b[0] = u; printf("c=1 + 0.75ulp           -> 1 + %.3f ulp   (nearest 1, truncation 0)\n", (run(a, b, 1.0f) - 1.0f) / u);
    clr(); a[0] = -0.75f; b[0] = u; printf("c=1 - 0.75ulp(1)        -> 1 + %.3f ulp   (exact -0.75: nearest -1.0 or -0.5 [half ulps below 1], truncation -0.5)\n", (run(a, b, 1.0f) - 1.0f) / u);
    clr(); a[0] = 0.75f; b[0] = u; printf("c=-1 + 0.75ulp          -> -1 + %.3f ulp\n", (run(a, b, -1.0f) + 1.0f) / u);
    // 2. 32 products of 1/8 ulp each = 4 ulp exactly: lost if each product is aligned to c and truncated on its own
    clr(); for (int k = 0; k < 32; k++) { a[k] = 0.125f; b[k] = u; } printf("c=1 + 32 x ulp/8        -> 1 + %.3f ulp   (exact 4)\n", (run(a, b, 1.0f) - 1.0f) / u);
    // 3. 32 products of 0.3 ulp = 9.6 ulp: nearest 10, truncation of the total 9, per-product truncation 0
    clr(); for (int k = 0; k < 32; k++) { a[k] = 0.2998046875f; b[k] = u; } printf("c=1 + 32 x 0.2998 ulp   -> 1 + %.3f ulp   (exact %.4f)\n", (run(a, b, 1.0f) - 1.0f) / u, 32 * 0.2998046875);
    // 4. a chain of 1000 MFMAs each adding 0.4 ulp (one product): nearest-even per step never moves; a wider internal sum does not help either; just shows per-instruction rounding
    clr(); a[0] = 0.4f; b[0] = u; printf("c=1, 1000 x (+0.4 ulp)   -> 1 + %.3f ulp\n", (run(a, b, 1.0f, 1000) - 1.0f) / u);
    clr(); a[0] = 0.6f; b[0] = u; printf("c=1, 1000 x (+0.6 ulp)   -> 1 + %.3f ulp   (nearest per step: 1000, truncation: 0)\n", (run(a, b, 1.0f, 1000) - 1.0f) / u);
    // 5. subnormal fp16 operands: a = 2^-20 (fp16 subnormal) x b = 2^10
    clr(); a[0] = ldexpf(1.0f, -20); b[0] = 1024.0f; printf("subnormal fp16 a=2^-20 x 1024 -> %.6g   (exact %.6g; 0 = flushed)\n", run(a, b, 0.0f), ldexpf(1.0f, -10));
    // 6. random-ish: products with mixed signs whose exact sum is known in double
    clr(); double ex = 0.37; for (int k = 0; k < 32; k++) { a[k] = (float)(_Float16)(0.01f * (k + 1) * ((k & 1) ? -1 : 1)); b[k] = (float)(_Float16)(0.03f * (32 - k)); ex += (double)a[k] * b[k]; }
    printf("mixed 32 products + 0.37 -> %.9g   (exact %.12g, fp32 nearest %.9g)\n", run(a, b, 0.37f), ex, (float)ex);
    return 0;
}
