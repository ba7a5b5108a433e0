// tools/mfma_round_probe2.hip — accumulation error of a conv-like chain on v_mfma_f32_16x16x32_f16: 72 MFMAs of 32 random
// products each into one fp32 accumulator, against the exact sum (double) and against a sequential fp32 fmaf chain.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define STEPS 72

// 16 independent dot products (one per column c): A[row 0][k] = a[s][c][k]... simpler: every lane column c gets its own b, row 0 only
__global__ void chain(const _Float16* a, const _Float16* b, float* out)   // a: [STEPS][32], b: [STEPS][16 cols][32]
{
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    f32x4 acc = {0, 0, 0, 0};
    for (int s = 0; s < STEPS; s++) {
        f16x8 av, bv;
        for (int j = 0; j < 8; j++) {
            av[j] = c == 0 ? a[s * 32 + 8 * g + j] : (_Float16)0.0f;       // A: only row 0 non-zero
            bv[j] = b[(s * 16 + c) * 32 + 8 * g + j];                        // B: column c
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    }
    if (g == 0) out[c] = acc[0];   // D[row 0][col c]: rows (lane>>4)*4 + reg -> g = 0, reg 0
}

int main()
{
    srand(7);
    std::vector<_Float16> a(STEPS * 32), b(STEPS * 16 * 32);
    auto rnd = [] { return (float)rand() / RAND_MAX; };
    double sum_rel_mfma = 0, sum_rel_f32 = 0, max_mfma = 0, max_f32 = 0; int n = 0;
    _Float16 *da, *db; float* dout;
    (void)hipMalloc(&da, a.size() * 2); (void)hipMalloc(&db, b.size() * 2); (void)hipMalloc(&dout, 64);
    for (int trial = 0; trial < 64; trial++) {
        for (auto& x : a) x = (_Float16)(rnd() * 2.0f);                    // activations >= 0 (post-ReLU), O(1)
        for (auto& x : b) x = (_Float16)((rnd() * 2.0f - 1.0f) * 0.036f * 8192.0f);   // scaled Glorot weights, random sign
        (void)hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, da, db, dout);
        float out[16]; (void)hipMemcpy(out, dout, 64, hipMemcpyDeviceToHost);
        for (int c = 0; c < 16; c++) {
            double ex = 0, scale = 0; float f = 0;
            for (int s = 0; s < STEPS; s++) for (int k = 0; k < 32; k++) {
                const double p = (double)(float)a[s * 32 + k] * (double)(float)b[(s * 16 + c) * 32 + k];
                ex += p; scale += p * p; f = fmaf((float)a[s * 32 + k], (float)b[(s * 16 + c) * 32 + k], f);
            }
            scale = sqrt(scale);   // typical magnitude of the sum
            const double em = fabs(out[c] - ex) / scale, ef = fabs(f - ex) / scale;
            sum_rel_mfma += em * em; sum_rel_f32 += ef * ef; if (em > max_mfma) max_mfma = em; if (ef > max_f32) max_f32 = ef; n++;
        }
    }
    printf("72 x 32 random products (a in [0,2), w signed): error / rms-sum  MFMA f16: rms %.3g max %.3g   sequential fp32 fma: rms %.3g max %.3g   (2^-24 = %.3g)\n",
           sqrt(sum_rel_mfma / n), max_mfma, sqrt(sum_rel_f32 / n), max_f32, ldexp(1.0, -24));
    return 0;
}
