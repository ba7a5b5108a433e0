// azr_train.hip — the optimiser step of the learn loop on the GPU (SURVEY §8 f-2, C-ABI azr_nn_train*).
//
// Replaces AlphaZeroNN::train (neural_network/alphazero_nn.cpp:351-410), i.e. `session->Run({state, target pi, target v,
// training=true}, {loss_policy, loss_value}, {optimize})` on the graph of python/src/build_graph.py:54-103:
//   forward in training mode (batch-statistics BN, moving averages updated with momentum 0.99), loss =
//   softmax-CE(pi) + MSE(v) + 1e-3 * sum ||kernel||^2, backward, Adam(1e-3, .9, .999, 1e-8) — all fp32 like the
//   reference's TF session.
//
// Data layout (all fp32, row-major): an activation is [M = batch * 42 rows][256 channels], row = board * 42 + y * 6 + x —
// the inference kernels' row order.  A 3x3 SAME convolution is the implicit GEMM  im2col(A) [M][9 * 256]  x  W [9 * 256][256]
// (W = the AZRW kernel [tap][ci][co] as it lies in the flat vector); the im2col matrix is never materialised — the tile
// loader gathers the shifted rows.  Its two gradients are the same kernel with other operand views: dW = im2col(A)^T x
// dY (split-K), dA = im2col-(dY) x W^T (negated taps).  Arithmetic: split bf16 on v_mfma_f32_16x16x32_bf16 — an fp32
// value is the exact sum of three bf16 parts; the forward multiplies all parts that matter (6 MFMA passes, fp32-exact
// products, so ReLU masks and batch statistics are those of an fp32 forward), the two gradient GEMMs use two parts (3
// passes, 1e-5 relative) — with t_gemm on the fp32 MFMA (v_mfma_f32_32x32x2_f32) kept for the stem (K = 144), odd
// batch sizes and AZR_TRAIN_GEMM=f32.  Kernels of the split path: t_conv_rs / t_conv_q (forward on fp16 pairs, backward-data on two
// bf16 parts, with the normalise and statistics steps fused into their staging paths and epilogues) and t_wgrad_g5 (weight gradient).
// Every conv output (pre-BN) and every post-activation is kept for the backward pass: 2 x 22 MB per layer at batch 512,
// 1.8 GB for the 41 conv layers of B = 20 — sized for 288 GB of HBM, nothing is recomputed.
// Reductions (BN statistics, bias / BN / head gradients, split-K) are two-stage and atomic-free: a step is
// bit-reproducible.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <rccl/rccl.h>   // types and enumerators only: the library is bound at run time (rccl_api below)
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "azr_internal.hpp"
#include "azr_rowclass.hpp"

using namespace azr;

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e__ = (call);                                                                \
        if (e__ != hipSuccess) {                                                                \
            (void)hipGetLastError(); /* the runtime's last-error slot is sticky: clear it */        \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                      \
            return AZR_E_HIP;                                                                   \
        }                                                                                       \
    } while (0)

namespace {

constexpr int KC = 9 * NF;        // im2col row length of a tower conv
constexpr int SIN = 16;           // stem input planes padded 13 -> 16
constexpr int KS = 9 * SIN;       // im2col row length of the stem conv
constexpr float BN_EPS = 1e-3f;   // tf.layers.batch_normalization epsilon
constexpr float BN_KEEP = 0.99f;  // momentum
constexpr float L2_C = 1e-3f;     // REGULARIZATION_L2_C (build_graph.py:30)
constexpr float LR = 1e-3f, ADAM_B1 = 0.9f, ADAM_B2 = 0.999f, ADAM_EPS = 1e-8f;  // build_graph.py:31,103
constexpr int RB = 64;            // rows per block in the two-stage reductions

// AZRW offsets (same arithmetic as azr_net.hip)
constexpr size_t LAYER = (size_t)9 * NF * NF + 4 * NF;
constexpr size_t OFF_STEM_BN = 9 * 13 * NF;
constexpr size_t OFF_BLOCK0 = OFF_STEM_BN + 28;
// head section
constexpr int H_PI_W = 0, H_PI_BN = 512, H_PD_W = 520, H_PD_B = 4132, H_V_W = 4175, H_V_BN = 4431, H_V1_W = 4435,
              H_V1_B = 15187, H_V2_W = 15443, H_V2_B = 15699, HEAD_FLOATS = 15700;
// per-board dense-gradient partials (t_head_bwd): pd_w | pd_b | v1_w | v1_b | v2_w | v2_b
constexpr int HP_PD_W = 0, HP_PD_B = 3612, HP_V1_W = 3655, HP_V1_B = 14407, HP_V2_W = 14663, HP_V2_B = 14919, HP_FLOATS = 14920;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =====================================================================================================================
// GEMM  C[M][N] = A[M][K] x B[K][N]  on v_mfma_f32_32x32x2_f32.  128x128 block tile, 4 waves of 64x64 (2x2 MFMA tiles),
// k-tile 16 staged in LDS as [k][m|n] so an MFMA operand read is 32 consecutive floats.  Operand storage is a template
// switch: A_MCONTIG = A stored [K][M] (column access of a row-major matrix, used for col^T), B_KCONTIG = B stored [N][K]
// (W^T).  blockIdx.z = split-K slice writing C + z * strideCz.  All edges are bounds-checked.
// =====================================================================================================================
constexpr int GT = 128, GK = 16, GLD = GT + 4;

// tile loaders: a [GK][T] tile (T = 128 or 64 along m|n), T * GK / 256 floats per thread.  MODE selects the operand view:
//   0  plain matrix: element (mn, k) at P[k * ld + mn] (MN_CONTIG) or P[mn * ld + k]
//   1  implicit im2col of an activation P [rows][256]: the matrix col[row][tap * 256 + c] = P[row + off(tap)][c] inside
//      the board, 0 outside (never materialised); "row" is mn when !MN_CONTIG (forward A) and k when MN_CONTIG (col^T)
//   2  the same with the tap offsets negated (the transposed convolution of the backward-data pass)
//   3  conv kernel W [tap][ci][co] viewed as B[k = tap * 256 + co][n = ci] (backward-data), !MN_CONTIG only
template <bool MN_CONTIG, int T, int MODE>
__device__ __forceinline__ void gt_load(const float* __restrict__ P, int ld, int mn0, int k0, int MN, int Kend, int t, float (&r)[T / 16])
{
    constexpr int V = T / 16;  // 8 or 4 floats per thread
    constexpr int TPR = 16 / V;
    const int mn = MN_CONTIG ? mn0 + (t & 15) * V : mn0 + t / TPR;
    const int k = MN_CONTIG ? k0 + (t >> 4) : k0 + (t % TPR) * V;
    const float* p;
    bool ok;  // the whole run of V elements is inside the matrix (runs never straddle: all extents are multiples of V)
    if constexpr (MODE == 1 || MODE == 2) {
        const int row = MN_CONTIG ? k : mn, kk = MN_CONTIG ? mn : k;  // kk = tap * 256 + c
        const int tap = kk >> 8, c = kk & 255;
        int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        if (MODE == 2) { dy = -dy; dx = -dx; }
        const int pos = row % NPOS, y = pos / 6 + dy, x = pos - (pos / 6) * 6 + dx;
        ok = (MN_CONTIG ? (row < Kend && kk < MN) : (row < MN && kk < Kend)) && y >= 0 && y < 7 && x >= 0 && x < 6;
        p = P + (size_t)(row + dy * 6 + dx) * NF + c;
#pragma unroll
        for (int j = 0; j < V; j++) r[j] = 0.0f;
        if (ok) {
#pragma unroll
            for (int q = 0; q < V / 4; q++) {
                const float4 a = reinterpret_cast<const float4*>(p)[q];
                r[4 * q] = a.x; r[4 * q + 1] = a.y; r[4 * q + 2] = a.z; r[4 * q + 3] = a.w;
            }
        }
        return;
    } else if constexpr (MODE == 3) {
        static_assert(!MN_CONTIG, "weight-tap view is k-contiguous");
        p = P + (size_t)(k >> 8) * (NF * NF) + (size_t)mn * NF + (k & 255);
        ok = mn < MN && k + V - 1 < Kend;
    } else if constexpr (MN_CONTIG) {
        p = P + (size_t)k * ld + mn;
        ok = k < Kend && mn + V - 1 < MN;
    } else {
        p = P + (size_t)mn * ld + k;
        ok = mn < MN && k + V - 1 < Kend;
    }
    if (ok) {
#pragma unroll
        for (int q = 0; q < V / 4; q++) {
            const float4 a = reinterpret_cast<const float4*>(p)[q];
            r[4 * q] = a.x; r[4 * q + 1] = a.y; r[4 * q + 2] = a.z; r[4 * q + 3] = a.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < V; j++) {
            const bool in = MN_CONTIG ? (k < Kend && mn + j < MN) : (mn < MN && k + j < Kend);
            r[j] = (MODE == 0 && in) ? p[j] : 0.0f;
        }
    }
}

template <bool MN_CONTIG, int T>
__device__ __forceinline__ void gt_store(float* S, int t, const float (&r)[T / 16])
{
    constexpr int V = T / 16, LD = T + 4;
    if constexpr (MN_CONTIG) {
        float* p = S + (t >> 4) * LD + (t & 15) * V;
#pragma unroll
        for (int q = 0; q < V / 4; q++) reinterpret_cast<float4*>(p)[q] = make_float4(r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]);
    } else {
        constexpr int TPR = 16 / V;
        float* p = S + ((t % TPR) * V) * LD + t / TPR;
#pragma unroll
        for (int j = 0; j < V; j++) p[j * LD] = r[j];
    }
}

// BM = 128: 4 waves as 2 x 2, each 64 x 64 (2 x 2 MFMA tiles); BM = 64: 2 x 2 waves, each 32 x 64 (1 x 2 tiles) — the
// smaller tile is for launches whose 128-row grid would leave CUs with 1 vs 2 blocks (forward conv: 336 -> 672 blocks)
template <bool A_MCONTIG, bool B_KCONTIG, int BM, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void t_gemm(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                              float* __restrict__ C, int ldc, int M, int N, int K, int kchunk, size_t strideCz)
{
    constexpr int MI = BM / 64, LDA = BM + 4;
    __shared__ __attribute__((aligned(16))) float As[GK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[GK * GLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * GT;
    const int kbeg = blockIdx.z * kchunk, kend = min(K, kbeg + kchunk);
    f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
    float ra[BM / 16], rb[8];
    gt_load<A_MCONTIG, BM, AMODE>(A, lda, m0, kbeg, M, kend, t, ra);
    gt_load<!B_KCONTIG, GT, BMODE>(B, ldb, n0, kbeg, N, kend, t, rb);
    for (int k0 = kbeg; k0 < kend; k0 += GK) {
        __syncthreads();
        gt_store<A_MCONTIG, BM>(As, t, ra);
        gt_store<!B_KCONTIG, GT>(Bs, t, rb);
        __syncthreads();
        if (k0 + GK < kend) {
            gt_load<A_MCONTIG, BM, AMODE>(A, lda, m0, k0 + GK, M, kend, t, ra);
            gt_load<!B_KCONTIG, GT, BMODE>(B, ldb, n0, k0 + GK, N, kend, t, rb);
        }
#pragma unroll
        for (int kk = 0; kk < GK / 2; kk++) {
            const int k = kk * 2 + (lane >> 5);
            float a[MI];
#pragma unroll
            for (int i = 0; i < MI; i++) a[i] = As[k * LDA + wm * (32 * MI) + i * 32 + (lane & 31)];
            const float b0 = Bs[k * GLD + wn * 64 + (lane & 31)], b1 = Bs[k * GLD + wn * 64 + 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < MI; i++) {
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b1, acc[i][1], 0, 0, 0);
            }
        }
    }
    float* Cz = C + (size_t)blockIdx.z * strideCz;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int row = m0 + wm * (32 * MI) + i * 32 + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                if (row < M && col < N) Cz[(size_t)row * ldc + col] = acc[i][j][e];
            }
}

// =====================================================================================================================
// The same GEMM in split bf16 on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate).  An fp32 value is the exact sum of
// three bf16 parts x = h + m + l (8 + 8 + 8 mantissa bits); t_split writes the parts of an operand once, and
//   NP = 3 (forward):  C += Al*Bh + Ah*Bl + Am*Bm + Am*Bh + Ah*Bm + Ah*Bh   — every product term above 2^-24 relative:
//                      fp32-exact products, so the ReLU masks and batch statistics match an fp32 forward;
//   NP = 2 (backward): C += Am*Bh + Ah*Bm + Ah*Bh                           — 16 bits per factor, 1e-5 relative;
// fp32 accumulation in both.  Tile BM x 128, k-tile 32, 4 waves as 2 x 2, LDS rows [m|n][32 + 8 pad] bf16 per part
// (80-byte stride: an MFMA fragment's ds_read_b128 is conflict-free).  Operand views as in gt_load (MODE 0..3); a
// mn-contiguous operand is transposed in registers (8 dword loads down k, v_perm, two 16-byte LDS writes).
// =====================================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int K3 = 32, KP3 = 40;
struct Parts { const uint16_t* p[3]; };

__device__ __forceinline__ uint32_t bf_rne_bits(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
// two floats -> two bf16 (round to nearest even) packed low | high: ONE v_cvt_pk_bf16_f32 where bf_rne_bits spends three integer
// operations per value and two more to pack — the same bits for every finite input (the staging paths of the fused convs run this for
// every element of every layer: profiles/r04_train_step.txt)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t bf_rne_pk(float a, float b)
{
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_t{a, b}, bf16x2_t));
}
// hi and mid bf16 parts of a pair of floats (hi = rne(v), mid = rne(v - hi)), packed
__device__ __forceinline__ void bf_split2(float a, float b, uint32_t& hi, uint32_t& mid)
{
    hi = bf_rne_pk(a, b);
    mid = bf_rne_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split_store4(const float (&v)[4], size_t i4, uint16_t* p0, uint16_t* p1, uint16_t* p2)
{
    if (!p2) {   // the two leading parts only (every caller on the step's hot path)
        uint32_t h01, m01, h23, m23;
        bf_split2(v[0], v[1], h01, m01);
        bf_split2(v[2], v[3], h23, m23);
        reinterpret_cast<uint2*>(p0)[i4] = make_uint2(h01, h23);
        reinterpret_cast<uint2*>(p1)[i4] = make_uint2(m01, m23);
        return;
    }
    uint32_t h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        h[j] = bf_rne_bits(v[j]);
        const float r1 = v[j] - __uint_as_float(h[j] << 16);
        m[j] = bf_rne_bits(r1);
        l[j] = bf_rne_bits(r1 - __uint_as_float(m[j] << 16));
    }
    reinterpret_cast<uint2*>(p0)[i4] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    reinterpret_cast<uint2*>(p1)[i4] = make_uint2(m[0] | (m[1] << 16), m[2] | (m[3] << 16));
    reinterpret_cast<uint2*>(p2)[i4] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// the fp16 pair of 4 consecutive values: hi = rne16(v), lo = rne16(v - hi) (unscaled: the matrix core takes fp16 subnormals),
// 22 significand bits — the operand format of the 3-pass forward conv (t_conv_rs<1, 2, 0, true>)
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
__device__ __forceinline__ void split_store4_f16(const float (&v)[4], size_t i4, uint16_t* q0, uint16_t* q1)
{
    _Float16 h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        h[j] = (_Float16)v[j];
        l[j] = (_Float16)(v[j] - (float)h[j]);
    }
    reinterpret_cast<uint2*>(q0)[i4] = make_uint2(__builtin_bit_cast(uint32_t, f16x2_t{h[0], h[1]}), __builtin_bit_cast(uint32_t, f16x2_t{h[2], h[3]}));
    reinterpret_cast<uint2*>(q1)[i4] = make_uint2(__builtin_bit_cast(uint32_t, f16x2_t{l[0], l[1]}), __builtin_bit_cast(uint32_t, f16x2_t{l[2], l[3]}));
}

// =====================================================================================================================
// Conv GEMMs whose B operand is the layer's kernel (forward, backward-data): N = 256, K = 2304.  The measured limit of
// t_gemm_sb on these shapes is LDS traffic, two thirds of it the weight tile.  Here the weights never touch LDS: t_pack_w
// writes their bf16 parts once per step in MFMA-fragment order ([k-tile][n-tile][lane][8]) and every wave loads the
// fragments of ITS 32 columns straight from global memory (1 KB coalesced per fragment, register double buffer).  Block =
// 64 rows x 128 columns, 4 waves side by side (64 x 32 each); only the activation tile goes through LDS.
//   VIEW 0: forward        B[k = tap*256+ci][n = co] = W[tap][ci][co]
//   VIEW 1: backward-data  B[k = tap*256+co][n = ci] = W[tap][ci][co]
// =====================================================================================================================
constexpr size_t WPACK = (size_t)KC * NF;  // elements per layer, part and view

// wscale > 0: fp16 PAIRS of wscale * W instead of bf16 parts (p0 = hi, p1 = lo; the forward conv on the fp16 MFMA)
template <int NP>
__global__ __launch_bounds__(256) void t_pack_w(const float* __restrict__ flat, int view, uint16_t* __restrict__ p0, uint16_t* __restrict__ p1,
                                                uint16_t* __restrict__ p2, float wscale = 0.0f, int* __restrict__ range_flag = nullptr)
{
    // one thread = one lane's 8 values of one fragment: index = ((kt * 16 + nt) * 64 + lane)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WPACK / 8) return;
    const int l = blockIdx.y;
    const float* W = flat + OFF_BLOCK0 + (size_t)l * LAYER;
    const int lane = (int)(i & 63), nt = (int)((i >> 6) & 15), kt = (int)(i >> 10);
    const int n = nt * 16 + (lane & 15), k0 = kt * 32 + (lane >> 4) * 8, tap = k0 >> 8, c0 = k0 & 255;
    uint32_t h[8], m[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const float v = view == 0 ? W[((size_t)tap * NF + (c0 + j)) * NF + n]    // ci = c0 + j, co = n
                                  : W[((size_t)tap * NF + n) * NF + (c0 + j)];   // ci = n, co = c0 + j
        if (wscale > 0.0f) {
            const float vs = v * wscale;
            // a weight that leaves the fp16 range (|w| >= 64 at the 2^10 scale) or is not a number would turn into inf / NaN here and
            // poison the step silently: raise the step's range flag instead (read by the caller behind the epoch)
            if (!(fabsf(vs) < 65504.0f) && range_flag) atomicOr(range_flag, 1);
            const _Float16 hh = (_Float16)vs, ll = (_Float16)(vs - (float)hh);
            h[j] = __builtin_bit_cast(uint16_t, hh);
            m[j] = __builtin_bit_cast(uint16_t, ll);
            lo[j] = 0u;
            continue;
        }
        h[j] = bf_rne_bits(v);
        const float r1 = v - __uint_as_float(h[j] << 16);
        m[j] = bf_rne_bits(r1);
        lo[j] = bf_rne_bits(r1 - __uint_as_float(m[j] << 16));
    }
    const size_t o = (size_t)l * (WPACK / 8) + i;
    reinterpret_cast<uint4*>(p0)[o] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    reinterpret_cast<uint4*>(p1)[o] = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    if (NP == 3) reinterpret_cast<uint4*>(p2)[o] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

// ---------------------------------------------------------------------------------------------------------------------
// t_conv_rs: the same two conv GEMMs (forward, backward-data) with what the inference tower (azr_tower_sb.hip) taught:
//   * a block owns 2 boards = 84 rows in BORDER-CLASS order (azr_rowclass.hpp, 6 MFMA row tiles): the 9 of 54 (tile, tap)
//     pairs that lie wholly outside the board are not issued (17 % of the MFMAs and fragment reads);
//   * 4 waves x 64 output channels (four 16-wide tiles): an activation fragment read from LDS feeds 4 MFMAs per pass,
//     and MFMA(weights, activations) leaves 4 consecutive channels of one cell in a lane: 16-byte stores;
//   * K order = channel chunk outermost (8 chunks of 32 input channels), tap innermost: only the current 32-channel slice
//     of the 84 rows has to be in LDS (two buffers; the next slice is fetched during the 9 k-steps of the current one):
//     ONE barrier per 9 k-steps; the 9 taps are unrolled with compile-time skip masks, the chunk loop is rolled;
//   * weights straight from global memory in MFMA-fragment order (t_pack_w) through a ring of 3 k-steps, refill loads
//     and fragment re-reads dealt out one per pass instead of as bursts.
// AMODE 1: C[row] = sum_tap A[row + tap] W[tap]; AMODE 2 (backward-data): negated taps, i.e. loop index t reads the
// geometric tap 8 - t, with the transposed kernel view.  `boards` = rows / 42 (the last block may hold one board).
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int NP> struct RsPass;
template <> struct RsPass<3> { static constexpr int N = 6; static constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0}; };
template <> struct RsPass<2> { static constexpr int N = 3; static constexpr int QA[3] = {1, 0, 0}, QB[3] = {0, 1, 0}; };

// geometry of t_conv_rs (below)
struct Rs {
    static constexpr int NB = 2, ROWS = 84, MT = 6, ZR = 96, NT = 4, RING = 3;
    static constexpr int CHB = 80;                       // bytes per row of a 32-channel slice (64 + 16 pad)
    static constexpr int PB = (ZR + 1) * CHB;            // one part of one slice, incl. the shared zero row
    static constexpr uint32_t KB = 16 * 64 * 16;         // bytes of one k-step of packed weights (16 column tiles x 64 lanes x 16 B)
};

// one k-step (one tap of one 32-channel slice) of t_conv_rs; everything that depends on the tap is a compile-time constant
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
template <int AMODE, int NP, int TAP, bool F16 = false>
__device__ __forceinline__ void rs_tap(const uint8_t* bufc, int kc, const __amdgpu_buffer_rsrc_t (&wsrc)[NP], uint32_t loff,
                                       const uint32_t (&arow)[9][Rs::MT], u32x4 (&bq)[Rs::RING][NP][Rs::NT], f32x4 (&acc)[Rs::MT][Rs::NT],
                                       s16x8 (&a)[Rs::MT][NP])
{
    constexpr int NB = Rs::NB, MT = Rs::MT, NT = Rs::NT, RING = Rs::RING, PB = Rs::PB, NPASS = RsPass<NP>::N;
    constexpr uint32_t sk = skip_mask<NB>(AMODE == 2 ? 8 - TAP : TAP);
    constexpr uint32_t skn = TAP < 8 ? skip_mask<NB>(AMODE == 2 ? 7 - TAP : TAP + 1) : 0xffffffffu;
    constexpr int active = MT - __builtin_popcount(sk & ((1u << MT) - 1u));
    constexpr int cur = TAP % RING, ref = (TAP + RING - 1) % RING;
    // the k-step RING - 1 ahead in consumption order (chunk-major): tap + 2 of this chunk or tap - 7 of the next
    constexpr int tap2 = (TAP + RING - 1) % 9;
    const uint32_t koff = (uint32_t)(tap2 * 8 + kc + (TAP + RING - 1 >= 9 ? 1 : 0)) * Rs::KB;   // (past the layer: out of range -> 0)
    constexpr int slots = active * NPASS;
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        if (!((sk >> mt) & 1u)) {
            const int j = __builtin_popcount(~sk & ((1u << mt) - 1u));
#pragma unroll
            for (int p = 0; p < NPASS; p++) {
                const int qa = RsPass<NP>::QA[p], qb = RsPass<NP>::QB[p];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    if constexpr (F16)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bq[cur][qb][nt]),
                                                                               __builtin_bit_cast(f16x8_t, a[mt][qa]), acc[mt][nt], 0, 0, 0);
                    else
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bq[cur][qb][nt]),
                                                                                __builtin_bit_cast(bf16x8, a[mt][qa]), acc[mt][nt], 0, 0, 0);
                }
                // one refill load of the ring slot the previous k-step freed, dealt out over the k-step
                const int s2 = j * NPASS + p;
#pragma unroll
                for (int i = 0; i < NP * NT; i++)
                    if (s2 == ((i + 1) * slots) / (NP * NT) - 1)
                        bq[ref][i / NT][i % NT] = __builtin_amdgcn_raw_buffer_load_b128(wsrc[i / NT], loff + (i % NT) * 1024, (int)koff, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!((skn >> mt) & 1u)) {   // this tile's fragments for the next tap
#pragma unroll
                for (int q = 0; q < NP; q++) a[mt][q] = *reinterpret_cast<const s16x8*>(bufc + q * PB + arow[TAP < 8 ? TAP + 1 : 0][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)   // tiles idle in this tap that run in the next
        if (((sk >> mt) & 1u) && !((skn >> mt) & 1u)) {
#pragma unroll
            for (int q = 0; q < NP; q++) a[mt][q] = *reinterpret_cast<const s16x8*>(bufc + q * PB + arow[TAP < 8 ? TAP + 1 : 0][mt]);
        }
}

// What the backward-data conv of layer l can do for layer l - 1 on its way out (FUSE = 1): its output rows are layer l - 1's
// dOut, still in registers — the shortcut gradient joins them here (first conv of a block: + DS, the job of t_add), and
// stage 1 of layer l - 1's batch-norm backward (t_bn_bwd_stats: per channel sum of dz and of dz * xhat, dz = dOut where the
// post-activation is positive) is taken per block of 2 boards, in double, in a fixed order: cells of a lane, then the 16
// lanes of a channel group.  part[blockIdx][2][256] is what t_bn_bwd_finalize / t_parts_sum read (R = number of blocks).
struct BwdFuse {
    const float* DS;      // shortcut gradient to add to the output rows, or null
    const float* Apost;   // layer l - 1: post-activation, pre-BN conv output, batch mean / 1 / std per channel
    const float* Y;
    const float* mean;
    const float* istd;
    double* part;
};

// What the conv can do on the way IN (PRO): its A operand is an elementwise function of tensors that are complete once the
// batch statistics are — so instead of a kernel that writes the operand parts and this one reading them back, the staging path
// computes them (each block stages every element of its 2 boards exactly once) and writes what later kernels still need:
//   PRO = 1 (forward conv of layer l): A_{l-1} = relu(gamma (Y_{l-1} - mean) istd + beta (+ S)) — t_bn_apply's arithmetic — goes
//            to LDS as fp16 pair; side outputs: A_{l-1} in fp32 (backward masks, shortcut, heads) and its bf16 hi / mid parts (the
//            weight-gradient GEMM's operand);
//   PRO = 2 (backward-data conv of layer l): dY_l = gamma istd (dz - sum(dz)/n - xhat sum(dz xhat)/n), dz = dOut where the
//            post-activation is positive — t_bn_bwd_apply's arithmetic — goes to LDS as bf16 hi / mid; side outputs: those two parts
//            (the weight-gradient GEMM, launched AFTER this kernel) and dz itself where the layer closes a block (the shortcut
//            gradient DS).
struct ProFuse {
    const float* X;       // PRO 1: Y_{l-1}   | PRO 2: dOut_l
    const float* S;       // PRO 1: shortcut input or null | PRO 2: Apost_l
    const float* Y;       // PRO 2: Y_l
    const float* mean;    // per channel [256]
    const float* istd;
    const float* bn;      // gamma | beta
    const float* sums;    // PRO 2: [2][256] sum(dz), sum(dz xhat)
    float inv_count;      // PRO 2
    float* O;             // PRO 1: A_{l-1} (fp32) | PRO 2: dz (DS) or null
    uint16_t* p0;         // bf16 hi / mid parts of the computed operand
    uint16_t* p1;
};

// F16: the operands are fp16 pairs (NP = 2: hi, lo) on v_mfma_f32_16x16x32_f16 and the sums are multiplied by `oscale` on the way
// out (the packed kernel carries a power-of-two scale) — the forward conv in 3 passes instead of the 6 of three bf16 parts.
template <int AMODE, int NP, int FUSE = 0, bool F16 = false, int PRO = 0>
__global__ __launch_bounds__(256, 1) void t_conv_rs(Parts A, Parts Bp, float* __restrict__ C, int boards, BwdFuse F = BwdFuse{}, float oscale = 1.0f,
                                                    ProFuse Pf = ProFuse{})
{
    constexpr int NB = Rs::NB, ROWS = Rs::ROWS, MT = Rs::MT, ZR = Rs::ZR, NT = Rs::NT, RING = Rs::RING, CHB = Rs::CHB, PB = Rs::PB;
    constexpr int UN = (NP * ROWS * 4 + 255) / 256;   // 16-byte units of a slice per thread
    constexpr uint32_t KB = Rs::KB;
    __shared__ __attribute__((aligned(16))) uint8_t img[2 * NP * PB];
    __shared__ uint8_t rowof[ROWS];
    __shared__ uint8_t taprow[9 * ZR];
    __shared__ uint16_t rowcell[ZR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    const int b0 = blockIdx.x * NB, m0 = b0 * NPOS;
    const int nbv = boards - b0 < NB ? boards - b0 : NB;

    // ---- weight ring: the first two k-steps (chunk 0, taps 0 and 1) fly while the tables are built
    __amdgpu_buffer_rsrc_t wsrc[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) wsrc[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp.p[q]), (short)0, (int)(WPACK * 2), 0x00020000);
    const uint32_t loff = (uint32_t)((wave * NT) * 64 + lane) * 16u;
    u32x4 bq[RING][NP][NT];
#pragma unroll
    for (int s2 = 0; s2 < RING - 1; s2++)
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bq[s2][q][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc[q], loff + nt * 1024, (int)((s2 * 8) * KB), 0);

    // ---- tables
    for (int i = tid; i < ZR; i += 256) rowcell[i] = 0xffffu;
    for (int i = tid; i < 2 * NP * (CHB / 4); i += 256) {   // the zero rows of both buffers
        const int bp = i / (CHB / 4), w4 = i % (CHB / 4);
        reinterpret_cast<uint32_t*>(img + bp * PB + ZR * CHB)[w4] = 0u;
    }
    __syncthreads();
    for (int i = tid; i < ROWS; i += 256) {
        const int b = i / NPOS, pos = i - b * NPOS, r = row_of<NB>(b, pos);
        rowof[i] = (uint8_t)r;
        rowcell[r] = (uint16_t)((pos / 6) | ((pos % 6) << 4) | (b << 8));
    }
    __syncthreads();
    for (int i = tid; i < 9 * ZR; i += 256) {
        const int t = i / ZR, r = i - t * ZR, ci = rowcell[r];
        int src = ZR;
        if (ci != 0xffff) {
            const int y = (ci & 15) + t / 3 - 1, x = ((ci >> 4) & 15) + t % 3 - 1;
            if ((unsigned)y < 7u && (unsigned)x < 6u) src = rowof[(ci >> 8) * NPOS + y * 6 + x];
        }
        taprow[i] = (uint8_t)src;
    }
    // this thread's units of a slice.  PRO = 0: (part, cell, 16-byte segment of 8 halfs) -> global element offset (chunk 0) and LDS
    // byte offset.  PRO != 0: (cell, 4 channels): the fp32 sources are fetched, the operand is computed when the slice is stashed.
    constexpr int UNR = PRO ? (ROWS * 8 + 255) / 256 : UN;
    size_t goff[UNR];
    uint32_t loffs[UNR];
    bool uok[UNR];
    __shared__ __attribute__((aligned(16))) float ptab[PRO ? 5 * NF : 4];   // PRO: per-channel parameters of the elementwise function
    if constexpr (PRO == 0) {
#pragma unroll
        for (int i = 0; i < UN; i++) {
            const int u = tid + 256 * i, q = u / (ROWS * 4), rem = u - q * (ROWS * 4), cell = rem >> 2, seg = rem & 3;
            uok[i] = u < NP * ROWS * 4 && cell < nbv * NPOS;
            goff[i] = (size_t)(m0 + cell) * NF + seg * 8;
            loffs[i] = (uint32_t)((u < NP * ROWS * 4 ? q : 0) * PB + (u < NP * ROWS * 4 ? rowof[cell] : 0) * CHB + seg * 16);
        }
    } else {
        static_assert(NP == 2, "the computed operand has two parts");
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            const int u = tid + 256 * i, cell = u >> 3, seg = u & 7;
            uok[i] = u < ROWS * 8 && cell < nbv * NPOS;
            goff[i] = (size_t)(m0 + cell) * NF + seg * 4;
            loffs[i] = (uint32_t)((u < ROWS * 8 ? rowof[cell] : 0) * CHB + seg * 8);
        }
        for (int i = tid; i < NF; i += 256) {
            ptab[i] = Pf.bn[i];                                  // gamma
            ptab[2 * NF + i] = Pf.mean[i];
            ptab[3 * NF + i] = Pf.istd[i];
            if constexpr (PRO == 1) ptab[NF + i] = Pf.bn[NF + i];   // beta
            else { ptab[NF + i] = Pf.sums[i] * Pf.inv_count; ptab[4 * NF + i] = Pf.sums[NF + i] * Pf.inv_count; }
        }
        __syncthreads();
    }
    struct Raw { uint4 a, b, c; };   // PRO = 0: a = 16 bytes of a part.  PRO 1: a = Y, b = S.  PRO 2: a = dOut, b = Apost, c = Y
    auto fetch = [&](int kc, Raw (&r)[UNR]) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            if constexpr (PRO == 0) {
                const int q = (tid + 256 * i) / (ROWS * 4);
                r[i].a = uok[i] ? *reinterpret_cast<const uint4*>(A.p[q < NP ? q : 0] + goff[i] + kc * 32) : make_uint4(0u, 0u, 0u, 0u);
            } else {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                r[i].a = uok[i] ? *reinterpret_cast<const uint4*>(Pf.X + goff[i] + kc * 32) : z;
                r[i].b = (uok[i] && Pf.S) ? *reinterpret_cast<const uint4*>(Pf.S + goff[i] + kc * 32) : z;
                if constexpr (PRO == 2) r[i].c = uok[i] ? *reinterpret_cast<const uint4*>(Pf.Y + goff[i] + kc * 32) : z;
            }
        }
    };
    auto stash = [&](int buf, int kc, const Raw (&r)[UNR]) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            if constexpr (PRO == 0) {
                if (tid + 256 * i < NP * ROWS * 4) *reinterpret_cast<uint4*>(img + buf * NP * PB + loffs[i]) = r[i].a;
            } else {
                if (tid + 256 * i >= ROWS * 8) continue;
                const int ch = kc * 32 + ((tid + 256 * i) & 7) * 4;
                const float4 ga = *reinterpret_cast<const float4*>(ptab + ch), p1 = *reinterpret_cast<const float4*>(ptab + NF + ch),
                             mu = *reinterpret_cast<const float4*>(ptab + 2 * NF + ch), is = *reinterpret_cast<const float4*>(ptab + 3 * NF + ch);
                const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, q4[4] = {p1.x, p1.y, p1.z, p1.w}, m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
                const float xa[4] = {__uint_as_float(r[i].a.x), __uint_as_float(r[i].a.y), __uint_as_float(r[i].a.z), __uint_as_float(r[i].a.w)};
                const float xb[4] = {__uint_as_float(r[i].b.x), __uint_as_float(r[i].b.y), __uint_as_float(r[i].b.z), __uint_as_float(r[i].b.w)};
                float o[4];
                uint2 hi, lo;
                if constexpr (PRO == 1) {   // t_bn_apply<false>
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float v = g4[j] * ((xa[j] - m4[j]) * i4[j]) + q4[j] + xb[j];
                        o[j] = v > 0.0f ? v : 0.0f;
                    }
                    {   // the fp16 pair, two values per conversion (v_cvt_pk_f16_f32, RNE: the bits of the scalar conversions)
                        const f16x2_t h01 = __builtin_convertvector(f32x2_t{o[0], o[1]}, f16x2_t), h23 = __builtin_convertvector(f32x2_t{o[2], o[3]}, f16x2_t);
                        const f16x2_t l01 = __builtin_convertvector(f32x2_t{o[0] - (float)h01[0], o[1] - (float)h01[1]}, f16x2_t);
                        const f16x2_t l23 = __builtin_convertvector(f32x2_t{o[2] - (float)h23[0], o[3] - (float)h23[1]}, f16x2_t);
                        hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
                        lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
                    }
                    if (uok[i]) {
                        *reinterpret_cast<float4*>(Pf.O + goff[i] + kc * 32) = make_float4(o[0], o[1], o[2], o[3]);
                        split_store4(o, (goff[i] + kc * 32) / 4, Pf.p0, Pf.p1, nullptr);
                    }
                } else {                    // t_bn_bwd_apply<false>
                    const float4 s1 = *reinterpret_cast<const float4*>(ptab + 4 * NF + ch);
                    const float t4[4] = {s1.x, s1.y, s1.z, s1.w};
                    const float xc[4] = {__uint_as_float(r[i].c.x), __uint_as_float(r[i].c.y), __uint_as_float(r[i].c.z), __uint_as_float(r[i].c.w)};
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float dz = xb[j] > 0.0f ? xa[j] : 0.0f;
                        const float xh = (xc[j] - m4[j]) * i4[j];
                        z[j] = dz;
                        o[j] = g4[j] * i4[j] * (dz - q4[j] - xh * t4[j]);
                    }
                    bf_split2(o[0], o[1], hi.x, lo.x);
                    bf_split2(o[2], o[3], hi.y, lo.y);
                    if (uok[i]) {
                        const size_t i4x = (goff[i] + kc * 32) / 4;
                        reinterpret_cast<uint2*>(Pf.p0)[i4x] = hi;
                        reinterpret_cast<uint2*>(Pf.p1)[i4x] = lo;
                        if (Pf.O) *reinterpret_cast<float4*>(Pf.O + goff[i] + kc * 32) = make_float4(z[0], z[1], z[2], z[3]);
                    }
                }
                if (!uok[i]) { hi = make_uint2(0u, 0u); lo = hi; }   // rows of a missing second board
                *reinterpret_cast<uint2*>(img + buf * NP * PB + loffs[i]) = hi;
                *reinterpret_cast<uint2*>(img + buf * NP * PB + PB + loffs[i]) = lo;
            }
        }
    };
    {
        Raw r0[UNR];
        fetch(0, r0);
        stash(0, 0, r0);
    }
    __syncthreads();
    // per lane: byte offset of its fragment row for (loop tap, tile) inside a part of a slice
    uint32_t arow[9][MT];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) arow[t][mt] = (uint32_t)taprow[(AMODE == 2 ? 8 - t : t) * ZR + mt * 16 + c] * CHB + g * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[MT][NP];

    for (int kc = 0; kc < 8; kc++) {
        Raw nx[UNR];
        if (kc + 1 < 8) fetch(kc + 1, nx);
        const uint8_t* bufc = img + (kc & 1) * NP * PB;
        {   // the fragments of tap 0 of this slice (the slice became visible with the barrier that ended the previous chunk)
            constexpr uint32_t sk0 = skip_mask<NB>(AMODE == 2 ? 8 : 0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
                if (!((sk0 >> mt) & 1u))
#pragma unroll
                    for (int q = 0; q < NP; q++) a[mt][q] = *reinterpret_cast<const s16x8*>(bufc + q * PB + arow[0][mt]);
        }
        rs_tap<AMODE, NP, 0, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 1, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 2, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 3, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 4, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 5, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 6, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 7, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        rs_tap<AMODE, NP, 8, F16>(bufc, kc, wsrc, loff, arow, bq, acc, a);
        if (kc + 1 < 8) stash((kc + 1) & 1, kc + 1, nx);
        __syncthreads();
    }
    // ---- C rows back in natural order: a lane holds 4 consecutive channels of one cell
    if constexpr (FUSE == 0) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int ci = rowcell[mt * 16 + c];
            if (ci == 0xffff || (ci >> 8) >= nbv) continue;
            float* out = C + (size_t)(m0 + (ci >> 8) * NPOS + (ci & 15) * 6 + ((ci >> 4) & 15)) * NF + wave * 64 + g * 4;
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                if constexpr (F16) acc[mt][nt] *= oscale;
                *reinterpret_cast<float4*>(out + nt * 16) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
            }
        }
    } else if constexpr (FUSE == 2) {
        // forward conv: the batch-norm statistics of its own output (t_bn_stats: per channel sum and sum of squares), per block
        // of 2 boards, in double, cells of a lane first, then the 16 lanes of a channel group -> F.part[blockIdx][2][256]
        double s[NT][4], ss[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < 4; e++) s[nt][e] = ss[nt][e] = 0.0;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int ci = rowcell[mt * 16 + c];
            const bool valid = !(ci == 0xffff || (ci >> 8) >= nbv);
            if (valid) {
                float* out = C + (size_t)(m0 + (ci >> 8) * NPOS + (ci & 15) * 6 + ((ci >> 4) & 15)) * NF + wave * 64 + g * 4;
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    if constexpr (F16) acc[mt][nt] *= oscale;
                    *reinterpret_cast<float4*>(out + nt * 16) = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const double v = (double)acc[mt][nt][e];
                        s[nt][e] += v;
                        ss[nt][e] += v * v;
                    }
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int sft = 1; sft < 16; sft <<= 1) {
                    s[nt][e] += __shfl_xor(s[nt][e], sft);
                    ss[nt][e] += __shfl_xor(ss[nt][e], sft);
                }
                if (c == 0) {
                    const int ch = wave * 64 + nt * 16 + g * 4 + e;
                    F.part[((size_t)blockIdx.x * 2 + 0) * NF + ch] = s[nt][e];
                    F.part[((size_t)blockIdx.x * 2 + 1) * NF + ch] = ss[nt][e];
                }
            }
    } else {
        double s[NT][4], sx[NT][4];
        float4 mu[NT], is[NT];
#pragma unroll
        for (int nt = 0; nt < NT; nt++) {
            mu[nt] = *reinterpret_cast<const float4*>(F.mean + wave * 64 + g * 4 + nt * 16);
            is[nt] = *reinterpret_cast<const float4*>(F.istd + wave * 64 + g * 4 + nt * 16);
#pragma unroll
            for (int e = 0; e < 4; e++) s[nt][e] = sx[nt][e] = 0.0;
        }
        // EPG row tiles at a time: ALL their loads (shortcut gradient, post-activation, pre-BN output of layer l - 1 at the output
        // coordinates: up to 24 x 16 bytes per lane) are issued before the first is used — taken one tile at a time, every tile paid
        // its own round trip to memory (the weight ring and the fragment registers are dead here: the registers are free).
        // The sums run over the tiles in the same order as before: same bits.
        constexpr int EPG = 3;
        static_assert(MT % EPG == 0, "tiles per epilogue group");
#pragma unroll
        for (int m2 = 0; m2 < MT; m2 += EPG) {
            size_t o[EPG];
            bool valid[EPG];
            float4 d4[EPG][NT], a4[EPG][NT], y4[EPG][NT];
#pragma unroll
            for (int u = 0; u < EPG; u++) {
                const int mt = m2 + u;
                const int ci = rowcell[mt * 16 + c];
                valid[u] = !(ci == 0xffff || (ci >> 8) >= nbv);
                o[u] = (size_t)(m0 + (valid[u] ? (ci >> 8) * NPOS + (ci & 15) * 6 + ((ci >> 4) & 15) : 0)) * NF + wave * 64 + g * 4;
                if (valid[u]) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        d4[u][nt] = F.DS ? *reinterpret_cast<const float4*>(F.DS + o[u] + nt * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
                        a4[u][nt] = *reinterpret_cast<const float4*>(F.Apost + o[u] + nt * 16);
                        y4[u][nt] = *reinterpret_cast<const float4*>(F.Y + o[u] + nt * 16);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < EPG; u++) {
                const int mt = m2 + u;
                if (valid[u]) {
#pragma unroll
                    for (int nt = 0; nt < NT; nt++) {
                        float4 v = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
                        if (F.DS) { v.x += d4[u][nt].x; v.y += d4[u][nt].y; v.z += d4[u][nt].z; v.w += d4[u][nt].w; }
                        *reinterpret_cast<float4*>(C + o[u] + nt * 16) = v;
                        const float vv[4] = {v.x, v.y, v.z, v.w}, aa[4] = {a4[u][nt].x, a4[u][nt].y, a4[u][nt].z, a4[u][nt].w},
                                    yy[4] = {y4[u][nt].x, y4[u][nt].y, y4[u][nt].z, y4[u][nt].w};
                        const float mm[4] = {mu[nt].x, mu[nt].y, mu[nt].z, mu[nt].w}, ii[4] = {is[nt].x, is[nt].y, is[nt].z, is[nt].w};
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float dz = aa[e] > 0.0f ? vv[e] : 0.0f;
                            s[nt][e] += (double)dz;
                            sx[nt][e] += (double)dz * (double)((yy[e] - mm[e]) * ii[e]);
                        }
                    }
                }
            }
        }
        // the 16 lanes c = 0..15 of a channel group hold different cells: butterfly over c, lane c = 0 writes
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
#pragma unroll
                for (int sft = 1; sft < 16; sft <<= 1) {
                    s[nt][e] += __shfl_xor(s[nt][e], sft);
                    sx[nt][e] += __shfl_xor(sx[nt][e], sft);
                }
                if (c == 0) {
                    const int ch = wave * 64 + nt * 16 + g * 4 + e;
                    F.part[((size_t)blockIdx.x * 2 + 0) * NF + ch] = s[nt][e];
                    F.part[((size_t)blockIdx.x * 2 + 1) * NF + ch] = sx[nt][e];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// t_conv_q: the same conv GEMMs for SMALL batches — a rank's 64-record share of a data-parallel minibatch, or a small
// minibatch.  t_conv_rs gives a CU 2 boards x 256 channels: 32 blocks at 64 records, an eighth of the chip, and the kernel takes
// as long as for 512 records (a block's serial work sets the time).  Here a block is ONE board x 64 output channels
// (blockIdx = board * 4 + channel group: 256 blocks at 64 records), and its 4 waves split K: wave w owns the 32-channel slices
// w and w + 4 of the board (2 x 9 k-steps), staged privately by the wave itself (no barrier until the end), its partial sums
// [48 rows x 64 channels] meet the other three waves' in LDS and are added in wave order (fixed: bit-reproducible); wave w
// then finishes column tile w (16 channels): store, and the same epilogue / staging-path fusions as t_conv_rs (FUSE, PRO;
// the side outputs of PRO are written by channel group 0 only).  One board = rows in natural order, 3 row tiles, no skipped
// (tile, tap) pairs.  Operands: two parts (fp16 pair with F16, else bf16 hi / mid), 3 passes.
// ---------------------------------------------------------------------------------------------------------------------
struct Rq {
    static constexpr int ROWS = 42, MT = 3, ZR = 48, NT = 4, RING = 3, NP = 2;
    static constexpr int CHB = 80;                       // bytes per row of a 32-channel slice (64 + 16 pad)
    static constexpr int PB = (ZR + 1) * CHB;            // one part of one slice, incl. the zero row
    static constexpr int WIMG = 2 * NP * PB;             // a wave's two slices
    static constexpr int RED = 4 * MT * NT * 64 * 16;    // the four waves' partial sums (f32x4 per lane)
    static constexpr int LDS = (4 * WIMG > RED ? 4 * WIMG : RED);
};

template <int AMODE, int FUSE, bool F16, int PRO>
__global__ __launch_bounds__(256, 1) void t_conv_q(Parts A, Parts Bp, float* __restrict__ C, int boards, BwdFuse F, float oscale, ProFuse Pf)
{
    constexpr int ROWS = Rq::ROWS, MT = Rq::MT, ZR = Rq::ZR, NT = Rq::NT, RING = Rq::RING, NP = Rq::NP, CHB = Rq::CHB, PB = Rq::PB;
    constexpr uint32_t KB = Rs::KB;
    __shared__ __attribute__((aligned(16))) uint8_t img[Rq::LDS];
    __shared__ uint8_t taprow[9 * ZR];
    __shared__ __attribute__((aligned(16))) float ptab[PRO ? 5 * NF : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    // (wave enters the weight loads' scalar offset and the compiler cannot prove it wave-uniform: each of those loads sits in a waterfall loop.
    //  Saying so with readfirstlane removes the loops and 32 VGPRs — and measures SLOWER at 64 records: 43.7 / 25.4 us against 34.6 / 20.2 for
    //  the backward / forward conv; the loops pace the loads between the MFMAs better than the scheduler does without them.  Left as it is.)
    const int board = blockIdx.x >> 2, cq = blockIdx.x & 3, m0 = board * NPOS;
    (void)boards;

    // ---- weight ring: this wave's first two k-steps (slice `wave`, taps 0 and 1) fly while the tables are built
    __amdgpu_buffer_rsrc_t wsrc[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) wsrc[q] = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(Bp.p[q]), (short)0, (int)(WPACK * 2), 0x00020000);
    const uint32_t loff = (uint32_t)((cq * NT) * 64 + lane) * 16u;
    u32x4 bq[RING][NP][NT];
    auto kstep_off = [&](int s2) { return (uint32_t)((s2 % 9) * 8 + wave + 4 * (s2 / 9)) * KB; };   // k-step s2 of this wave: tap s2 % 9 of slice wave + 4 (s2 / 9)
#pragma unroll
    for (int s2 = 0; s2 < RING - 1; s2++)
#pragma unroll
        for (int q = 0; q < NP; q++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bq[s2][q][nt] = __builtin_amdgcn_raw_buffer_load_b128(wsrc[q], loff + nt * 1024, (int)kstep_off(s2), 0);

    // ---- tables: source row of (geometric tap, row); pad rows and out-of-board taps read the zero row
    for (int i = tid; i < 9 * ZR; i += 256) {
        const int t = i / ZR, r = i - t * ZR;
        int src = ZR;
        if (r < ROWS) {
            const int y = r / 6 + t / 3 - 1, x = r % 6 + t % 3 - 1;
            if ((unsigned)y < 7u && (unsigned)x < 6u) src = y * 6 + x;
        }
        taprow[i] = (uint8_t)src;
    }
    if constexpr (PRO != 0) {
        for (int i = tid; i < NF; i += 256) {
            ptab[i] = Pf.bn[i];
            ptab[2 * NF + i] = Pf.mean[i];
            ptab[3 * NF + i] = Pf.istd[i];
            if constexpr (PRO == 1) ptab[NF + i] = Pf.bn[NF + i];
            else { ptab[NF + i] = Pf.sums[i] * Pf.inv_count; ptab[4 * NF + i] = Pf.sums[NF + i] * Pf.inv_count; }
        }
    }
    uint8_t* wimg = img + wave * Rq::WIMG;   // this wave's two slices: [slice][part][row][80 B]
    for (int i = lane; i < 2 * NP * (CHB / 4); i += 64) {   // their zero rows
        const int sp = i / (CHB / 4), w4 = i % (CHB / 4);
        reinterpret_cast<uint32_t*>(wimg + sp * PB + ZR * CHB)[w4] = 0u;
    }
    __syncthreads();

    // ---- the wave stages its two 32-channel slices itself (kc = wave, wave + 4)
#pragma unroll
    for (int sl = 0; sl < 2; sl++) {
        const int kc = wave + 4 * sl;
        uint8_t* dst = wimg + sl * NP * PB;
        if constexpr (PRO == 0) {
            constexpr int UNITS = NP * ROWS * 4;   // (part, row, 16-byte segment)
#pragma unroll
            for (int i = 0; i < (UNITS + 63) / 64; i++) {
                const int u = lane + 64 * i;
                if (u < UNITS) {
                    const int q = u / (ROWS * 4), rem = u - q * (ROWS * 4), r = rem >> 2, seg = rem & 3;
                    *reinterpret_cast<uint4*>(dst + q * PB + r * CHB + seg * 16) =
                        *reinterpret_cast<const uint4*>(A.p[q] + (size_t)(m0 + r) * NF + kc * 32 + seg * 8);
                }
            }
        } else {
            constexpr int UNITS = ROWS * 8;        // (row, 4 channels)
#pragma unroll
            for (int i = 0; i < (UNITS + 63) / 64; i++) {
                const int u = lane + 64 * i;
                if (u >= UNITS) continue;
                const int r = u >> 3, seg = u & 7, ch = kc * 32 + seg * 4;
                const size_t go = (size_t)(m0 + r) * NF + ch;
                const float4 ga = *reinterpret_cast<const float4*>(ptab + ch), p1 = *reinterpret_cast<const float4*>(ptab + NF + ch),
                             mu = *reinterpret_cast<const float4*>(ptab + 2 * NF + ch), is = *reinterpret_cast<const float4*>(ptab + 3 * NF + ch);
                const float g4[4] = {ga.x, ga.y, ga.z, ga.w}, q4[4] = {p1.x, p1.y, p1.z, p1.w}, m4[4] = {mu.x, mu.y, mu.z, mu.w}, i4[4] = {is.x, is.y, is.z, is.w};
                const float4 xa4 = *reinterpret_cast<const float4*>(Pf.X + go);
                const float4 xb4 = Pf.S ? *reinterpret_cast<const float4*>(Pf.S + go) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float xa[4] = {xa4.x, xa4.y, xa4.z, xa4.w}, xb[4] = {xb4.x, xb4.y, xb4.z, xb4.w};
                float o[4];
                uint2 hi, lo;
                if constexpr (PRO == 1) {   // t_bn_apply<false>
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float v = g4[j] * ((xa[j] - m4[j]) * i4[j]) + q4[j] + xb[j];
                        o[j] = v > 0.0f ? v : 0.0f;
                    }
                    {   // the fp16 pair, two values per conversion (v_cvt_pk_f16_f32, RNE: the bits of the scalar conversions)
                        const f16x2_t h01 = __builtin_convertvector(f32x2_t{o[0], o[1]}, f16x2_t), h23 = __builtin_convertvector(f32x2_t{o[2], o[3]}, f16x2_t);
                        const f16x2_t l01 = __builtin_convertvector(f32x2_t{o[0] - (float)h01[0], o[1] - (float)h01[1]}, f16x2_t);
                        const f16x2_t l23 = __builtin_convertvector(f32x2_t{o[2] - (float)h23[0], o[3] - (float)h23[1]}, f16x2_t);
                        hi = make_uint2(__builtin_bit_cast(uint32_t, h01), __builtin_bit_cast(uint32_t, h23));
                        lo = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
                    }
                    if (cq == 0) {
                        *reinterpret_cast<float4*>(Pf.O + go) = make_float4(o[0], o[1], o[2], o[3]);
                        split_store4(o, go / 4, Pf.p0, Pf.p1, nullptr);
                    }
                } else {                    // t_bn_bwd_apply<false>
                    const float4 s1 = *reinterpret_cast<const float4*>(ptab + 4 * NF + ch);
                    const float t4[4] = {s1.x, s1.y, s1.z, s1.w};
                    const float4 xc4 = *reinterpret_cast<const float4*>(Pf.Y + go);
                    const float xc[4] = {xc4.x, xc4.y, xc4.z, xc4.w};
                    float z[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const float dz = xb[j] > 0.0f ? xa[j] : 0.0f;
                        const float xh = (xc[j] - m4[j]) * i4[j];
                        z[j] = dz;
                        o[j] = g4[j] * i4[j] * (dz - q4[j] - xh * t4[j]);
                    }
                    bf_split2(o[0], o[1], hi.x, lo.x);
                    bf_split2(o[2], o[3], hi.y, lo.y);
                    if (cq == 0) {
                        reinterpret_cast<uint2*>(Pf.p0)[go / 4] = hi;
                        reinterpret_cast<uint2*>(Pf.p1)[go / 4] = lo;
                        if (Pf.O) *reinterpret_cast<float4*>(Pf.O + go) = make_float4(z[0], z[1], z[2], z[3]);
                    }
                }
                *reinterpret_cast<uint2*>(dst + r * CHB + seg * 8) = hi;
                *reinterpret_cast<uint2*>(dst + PB + r * CHB + seg * 8) = lo;
            }
        }
    }
    asm volatile("" ::: "memory");   // (a wave's LDS operations execute in program order: its fragment reads follow its own stores)

    // per lane: byte offset of its fragment row for (loop tap, tile) inside a part of a slice
    uint32_t arow[9][MT];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int mt = 0; mt < MT; mt++) arow[t][mt] = (uint32_t)taprow[(AMODE == 2 ? 8 - t : t) * ZR + mt * 16 + c] * CHB + g * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 a[MT][NP];
#pragma unroll
    for (int s2 = 0; s2 < 18; s2++) {
        const int sl = s2 / 9, t = s2 % 9, cur = s2 % RING, ref = (s2 + RING - 1) % RING;
        const uint8_t* bufc = wimg + sl * NP * PB;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int q = 0; q < NP; q++) a[mt][q] = *reinterpret_cast<const s16x8*>(bufc + q * PB + arow[t][mt]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int p = 0; p < 3; p++) {
                const int qa = RsPass<2>::QA[p], qb = RsPass<2>::QB[p];
#pragma unroll
                for (int nt = 0; nt < NT; nt++) {
                    if constexpr (F16)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, bq[cur][qb][nt]), __builtin_bit_cast(f16x8_t, a[mt][qa]), acc[mt][nt], 0, 0, 0);
                    else
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bq[cur][qb][nt]), __builtin_bit_cast(bf16x8, a[mt][qa]), acc[mt][nt], 0, 0, 0);
                }
                // one refill load of the ring slot the previous k-step freed per pass (8 loads over the 9 passes of a k-step)
                const int slot = mt * 3 + p;
                if (slot < NP * NT && s2 + RING - 1 < 18)
                    bq[ref][slot / NT][slot % NT] = __builtin_amdgcn_raw_buffer_load_b128(wsrc[slot / NT], loff + (slot % NT) * 1024, (int)kstep_off(s2 + RING - 1), 0);
            }
        }
    }

    // ---- the four waves' partial sums meet in LDS (over the slices: every wave is done reading), added in wave order
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(img);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) red[((wave * MT + mt) * NT + nt) * 64 + lane] = acc[mt][nt];
    __syncthreads();
    f32x4 out[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        out[mt] = red[((0 * MT + mt) * NT + wave) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; w++) out[mt] += red[((w * MT + mt) * NT + wave) * 64 + lane];
        if constexpr (F16) out[mt] *= oscale;
    }
    // wave w holds column tile w: lane (c, g) = cell mt * 16 + c, channels cq * 64 + wave * 16 + g * 4 ..
    const int ch0 = cq * 64 + wave * 16 + g * 4;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, sx[4] = {0.0, 0.0, 0.0, 0.0};
    float4 mu4 = make_float4(0.f, 0.f, 0.f, 0.f), is4 = mu4;
    if constexpr (FUSE == 1) { mu4 = *reinterpret_cast<const float4*>(F.mean + ch0); is4 = *reinterpret_cast<const float4*>(F.istd + ch0); }
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int r = mt * 16 + c;
        if (r >= ROWS) continue;
        const size_t o = (size_t)(m0 + r) * NF + ch0;
        float4 v = make_float4(out[mt][0], out[mt][1], out[mt][2], out[mt][3]);
        if constexpr (FUSE == 1) {
            if (F.DS) {
                const float4 d = *reinterpret_cast<const float4*>(F.DS + o);
                v.x += d.x; v.y += d.y; v.z += d.z; v.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(C + o) = v;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if constexpr (FUSE == 2) {
#pragma unroll
            for (int e = 0; e < 4; e++) { const double d = (double)vv[e]; s[e] += d; sx[e] += d * d; }
        } else if constexpr (FUSE == 1) {
            const float4 a4 = *reinterpret_cast<const float4*>(F.Apost + o), y4 = *reinterpret_cast<const float4*>(F.Y + o);
            const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, yy[4] = {y4.x, y4.y, y4.z, y4.w};
            const float mm[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, ii[4] = {is4.x, is4.y, is4.z, is4.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const float dz = aa[e] > 0.0f ? vv[e] : 0.0f;
                s[e] += (double)dz;
                sx[e] += (double)dz * (double)((yy[e] - mm[e]) * ii[e]);
            }
        }
    }
    if constexpr (FUSE != 0) {   // per-channel partials of this board: cells of a lane, then the 16 lanes of a channel group
#pragma unroll
        for (int e = 0; e < 4; e++) {
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) {
                s[e] += __shfl_xor(s[e], sft);
                sx[e] += __shfl_xor(sx[e], sft);
            }
            if (c == 0) {
                F.part[((size_t)board * 2 + 0) * NF + ch0 + e] = s[e];
                F.part[((size_t)board * 2 + 1) * NF + ch0 + e] = sx[e];
            }
        }
    }
}

typedef __attribute__((ext_vector_type(4))) short s16x4;

// transposed read of one MFMA operand fragment: two 4-row blocks (rows k..k+3 of the lane's group, then k+4..k+7); the
// arguments are absolute LDS addresses (no base to add), IMM a compile-time byte offset that lands in the instruction
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
template <int IMM>
__device__ __forceinline__ s16x8 lds_tr8(uint32_t a_lo, uint32_t a_hi)
{
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((uintptr_t)a_lo) + IMM / 8);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_s16x4*>((uintptr_t)a_hi) + IMM / 8);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifdef AZR_TEST_HOOKS   // the older formulation of the weight gradient: libazr_hip_test.so only (AZR_TRAIN_WGRAD=rs), what t_wgrad_g5 is compared with
// ---------------------------------------------------------------------------------------------------------------------
// t_wgrad_rs: the weight gradient  dW[tap][ci][co] = sum over rows of A[row + tap][ci] * dY[row][co]  (split bf16, 3 passes).
// The reduction runs over ROWS, the slow index of both operands ([row][channel] in memory): the MFMA wants 8 consecutive
// rows of one channel per lane.  gfx950's transposed LDS read (ds_read_b64_tr_b16) delivers exactly that from row-major
// tiles, and because every lane supplies the ADDRESS of one row of a 4-row block, the tap shift and the board-edge mask
// cost nothing: an out-of-board source row is simply the address of a zero row (no im2col, no register transposes, no masks).
//   block = ONE WAVE = one slice of whole boards (the split-K unit) x one 16-channel ci tile x 64 output channels:
//   9 taps x 4 co tiles = 36 accumulator tiles; a dY fragment feeds 9 taps, an A fragment 4 co tiles;
//   per k-step (32 rows) the wave stages its 32 rows x 64 co of dY and 46 rows (7 halo rows each side) x 16 ci of A, both
//   parts, through registers into its private LDS tile: 108 MFMAs per k-step, no barrier anywhere.
// Blocks of one slice are NS apart in blockIdx (same XCD: the slice's dY is fetched into one L2).
// ---------------------------------------------------------------------------------------------------------------------
struct Wg {
    static constexpr int KR = 32, HALO = 7, AR = KR + 2 * HALO;
    // Tile rows are placed for conflict-free transposed reads: a 32-lane half reads 4 rows r..r+3 and the 4 rows 8 further,
    // 32 bytes (8 banks) each; with a row pitch of 8 banks (mod 64) and 32 more banks in front of every further group of 8
    // rows, the eight rows cover the 64 banks once — for any tap shift of the A rows too.
    static constexpr int AST = 32;                 // bytes per row of the A tile (16 ci)
    static constexpr int APB = (AR + 1) * AST + (AR / 8) * 128;   // one part: rows and gaps, incl. the zero row (row AR)
    static constexpr int GST = 160;                // bytes per row of the dY tile (this wave's 64 co + 32 B pad: 40 banks)
    static constexpr int GPB = KR * GST + (KR / 8) * 128;
    __host__ __device__ static constexpr int arow(int r) { return r * AST + (r >> 3) * 128; }
    __host__ __device__ static constexpr int grow(int r) { return r * GST + (r >> 3) * 128; }
    static constexpr int BUF = 2 * APB + 2 * GPB;  // A part 0 | A part 1 | dY part 0 | dY part 1
    static constexpr int LDS_BYTES = BUF;          // ONE buffer: a wave's LDS operations run in program order (see t_wgrad_rs)
};
// one k-step (32 rows) of t_wgrad_rs.  With one wave per SIMD nothing hides a latency: the A fragments of tap t + 1 are
// requested BEFORE the 12 MFMAs of tap t are issued (two fragment slots, pinned with scheduling barriers — left alone, the
// compiler reuses one slot and waits for every read in front of its MFMAs), and everything else is kept to the reads
// themselves, one v_cndmask per A read (valid source row or the zero row: the row addresses are loop-invariant registers,
// part offsets are instruction immediates) and eight edge tests per k-step whose combinations per tap are scalar.
// (One loop body: two copies of the k-step in one loop made the compiler shuffle all 144 accumulators at the back-edge.)
template <int T>
__device__ __forceinline__ void wg_afrag(int y1, int x1, int y2, int x2, const uint32_t (&aoff)[9][2], uint32_t a_zero, s16x8& ah, s16x8& am)
{
    constexpr int dy = T / 3 - 1, dx = T % 3 - 1;
    const bool v1 = (dy < 0 ? y1 > 0 : dy > 0 ? y1 < 6 : true) && (dx < 0 ? x1 > 0 : dx > 0 ? x1 < 5 : true);
    const bool v2 = (dy < 0 ? y2 > 0 : dy > 0 ? y2 < 6 : true) && (dx < 0 ? x2 > 0 : dx > 0 ? x2 < 5 : true);
    const uint32_t o1 = v1 ? aoff[T][0] : a_zero, o2 = v2 ? aoff[T][1] : a_zero;
    ah = lds_tr8<0>(o1, o2);
    am = lds_tr8<Wg::APB>(o1, o2);
}
template <int T>
__device__ __forceinline__ void wg_tap(int y1, int x1, int y2, int x2, const uint32_t (&aoff)[9][2], uint32_t a_zero, const s16x8 (&gf)[2][4],
                                       s16x8 (&ah)[2], s16x8 (&am)[2], f32x4 (&acc)[9][4])
{
    if constexpr (T + 1 < 9) wg_afrag<T + 1>(y1, x1, y2, x2, aoff, a_zero, ah[(T + 1) & 1], am[(T + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < 4; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, am[T & 1]), __builtin_bit_cast(bf16x8, gf[0][c]), acc[T][c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[T & 1]), __builtin_bit_cast(bf16x8, gf[1][c]), acc[T][c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < 4; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[T & 1]), __builtin_bit_cast(bf16x8, gf[0][c]), acc[T][c], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void wg_kstep(int& pos1, int& pos2, const uint32_t (&aoff)[9][2], uint32_t a_zero, uint32_t g_lo, uint32_t g_hi,
                                         f32x4 (&acc)[9][4])
{
    const int y1 = pos1 / 6, x1 = pos1 - 6 * y1, y2 = pos2 / 6, x2 = pos2 - 6 * y2;
    s16x8 gf[2][4];   // dY fragments [part][co tile]
    s16x8 ah[2], am[2];
    gf[0][0] = lds_tr8<0>(g_lo, g_hi); gf[0][1] = lds_tr8<32>(g_lo, g_hi); gf[0][2] = lds_tr8<64>(g_lo, g_hi); gf[0][3] = lds_tr8<96>(g_lo, g_hi);
    wg_afrag<0>(y1, x1, y2, x2, aoff, a_zero, ah[0], am[0]);
    gf[1][0] = lds_tr8<Wg::GPB>(g_lo, g_hi); gf[1][1] = lds_tr8<Wg::GPB + 32>(g_lo, g_hi);
    gf[1][2] = lds_tr8<Wg::GPB + 64>(g_lo, g_hi); gf[1][3] = lds_tr8<Wg::GPB + 96>(g_lo, g_hi);
    wg_tap<0>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<1>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<2>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<3>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<4>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<5>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<6>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<7>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    wg_tap<8>(y1, x1, y2, x2, aoff, a_zero, gf, ah, am, acc);
    pos1 += Wg::KR; if (pos1 >= NPOS) pos1 -= NPOS;
    pos2 += Wg::KR; if (pos2 >= NPOS) pos2 -= NPOS;
}

// One WAVE per block: a wave's tiles (its 64 co columns of dY, its own copy of the 16-ci A rows) are private, so there is
// nothing to synchronise with — no barrier, and the four waves of a CU (four blocks, 15 KB of LDS each) drift apart and hide
// each other's bubbles.  A single LDS buffer suffices: the next tile travels global -> registers while this k-step
// computes and is stored over the current one AFTER the k-step's last fragment read has been issued — the LDS operations
// of one wave execute in program order.  Measured per k-step on one box (rocprofv3 kernel time, parts removed): the 108
// MFMAs 37 us of the 67, the staging 10, the A fragment reads 7: with one wave per SIMD nothing overlaps for free.
__global__ __launch_bounds__(64, 1) void t_wgrad_rs(Parts A, Parts G, float* __restrict__ out, int M, int NS, int rows_per_slice)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t wg_lds[];
    const int lane = threadIdx.x;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int slice = blockIdx.x % NS, rest = blockIdx.x / NS, cit = rest & 15, wq = rest >> 4;   // wq = which 64 co columns
    const int rbeg = slice * rows_per_slice, rend = min(M, rbeg + rows_per_slice);
    const int nks = (rend - rbeg + Wg::KR - 1) / Wg::KR;   // k-steps; a slice that is not a multiple of 32 rows (8 boards = 10.5 k-steps) ends
                                                          // inside one: the dY rows past the slice read as zero (range of gsrc below)

    // zero rows of the two A parts
    if (lane < 2 * (Wg::AST / 4))
        reinterpret_cast<uint32_t*>(wg_lds + (lane / (Wg::AST / 4)) * Wg::APB + Wg::arow(Wg::AR))[lane % (Wg::AST / 4)] = 0u;

    // staging units of this lane: 8 of the dY tile (4 (row, 16-byte segment) pairs x 2 parts: always inside the slice) and
    // up to 4 of the A tile (2 per part; halo rows before row 0 or after row M - 1 are out of range of the buffer resource
    // and read as 0).  Buffer loads: the k-step advances a scalar offset.
    const __amdgpu_buffer_rsrc_t gsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G.p[0]), (short)0, rend * NF * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G.p[1]), (short)0, rend * NF * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t asrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A.p[0]), (short)0, M * NF * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t asrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A.p[1]), (short)0, M * NF * 2, 0x00020000);
    uint32_t goffs[4], gl[4], aoffs[2], al[2];
    bool a_unit[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int row = i * 8 + (lane >> 3), seg = lane & 7;
        goffs[i] = (uint32_t)((rbeg + row) * NF + wq * 64 + seg * 8) * 2u;
        gl[i] = (uint32_t)(2 * Wg::APB + Wg::grow(row) + seg * 16);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int v = lane + 64 * j, row = v >> 1, seg = v & 1;
        a_unit[j] = v < 2 * Wg::AR;
        // (as a vector offset, so that the range check sees it: rows before 0 wrap to huge offsets, rows past M - 1 exceed M * 512)
        aoffs[j] = a_unit[j] ? (uint32_t)(((rbeg - Wg::HALO + row) * NF + cit * 16 + seg * 8) * 2) : 0xfffffff0u;
        al[j] = (uint32_t)(Wg::arow(a_unit[j] ? row : Wg::AR - 1) + seg * 16);
    }
    u32x4 sg[8], sa[4];
    auto fetch = [&](int ks) {
        const int so = ks * (Wg::KR * NF * 2);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            sg[i] = __builtin_amdgcn_raw_buffer_load_b128(gsrc0, goffs[i], so, 0);
            sg[i + 4] = __builtin_amdgcn_raw_buffer_load_b128(gsrc1, goffs[i], so, 0);
        }
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const uint32_t vo = a_unit[j] ? aoffs[j] + (uint32_t)so : 0xfffffff0u;
            sa[j] = __builtin_amdgcn_raw_buffer_load_b128(asrc0, vo, 0, 0);
            sa[j + 2] = __builtin_amdgcn_raw_buffer_load_b128(asrc1, vo, 0, 0);
        }
    };
    auto stash = [&]() {
        uint8_t* b = wg_lds;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            *reinterpret_cast<u32x4*>(b + gl[i]) = sg[i];
            *reinterpret_cast<u32x4*>(b + Wg::GPB + gl[i]) = sg[i + 4];
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            if (a_unit[j]) {
                *reinterpret_cast<u32x4*>(b + al[j]) = sa[j];
                *reinterpret_cast<u32x4*>(b + Wg::APB + al[j]) = sa[j + 2];
            }
    };
    fetch(0);
    stash();

    f32x4 acc[9][4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int c = 0; c < 4; c++) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the two tile rows this lane addresses in a transposed read: k1 = 8g + q and k1 + 4; their board cells and, per tap, the
    // LDS addresses of their source rows (loop-invariant: the tile moves, the lane's place in it does not)
    const int k1 = 8 * g + q;
    int pos1 = k1 % NPOS, pos2 = (k1 + 4) % NPOS;      // (slices start on a board boundary)
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)wg_lds;   // absolute LDS addresses
    uint32_t aoff[9][2];
#pragma unroll
    for (int t = 0; t < 9; t++) {
        const int sh = Wg::HALO + (t / 3 - 1) * 6 + (t % 3 - 1);   // source tile row = k + sh
        aoff[t][0] = lbase + (uint32_t)(Wg::arow(k1 + sh) + p * 8);
        aoff[t][1] = lbase + (uint32_t)(Wg::arow(k1 + 4 + sh) + p * 8);
    }
    const uint32_t a_zero = lbase + (uint32_t)(Wg::arow(Wg::AR) + p * 8);
    const uint32_t g_lo = lbase + (uint32_t)(2 * Wg::APB + Wg::grow(k1) + p * 8);
    const uint32_t g_hi = lbase + (uint32_t)(2 * Wg::APB + Wg::grow(k1 + 4) + p * 8);

    for (int ks = 0; ks < nks; ks++) {
        if (ks + 1 < nks) fetch(ks + 1);
        asm volatile("" ::: "memory");   // (the fragment reads below follow this wave's own tile stores in program order ...
        wg_kstep(pos1, pos2, aoff, a_zero, g_lo, g_hi, acc);
        asm volatile("" ::: "memory");   //  ... and the stores of the next tile follow the reads)
        if (ks + 1 < nks) stash();
    }
    float* o = out + (size_t)slice * KC * NF;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int e = 0; e < 4; e++)
                o[(size_t)(t * NF + cit * 16 + 4 * g + e) * NF + wq * 64 + c * 16 + i16] = acc[t][c][e];
}
#endif   // AZR_TEST_HOOKS

// ---------------------------------------------------------------------------------------------------------------------
// t_wgrad_g5: the same weight gradient with the reduction index laid out so that the 3 x 3 taps share operand fragments.
// t_wgrad_rs reduces over rows in memory order: every tap is its own shift of the A rows, so a k-step reads 9 x 2 A fragments from LDS
// for 108 MFMAs, and its 52 transposed reads per wave (4 waves per CU) hold the matrix pipe at one half.  Here a k-step is ONE BOARD ROW
// y OF FIVE BOARDS: k = 6 j + x (board j of the group, column x; k = 30, 31 are zero).  Then
//   * the dy shift of a tap is a shift by whole k-steps: the A fragments of board row y + dy are those read for k-step y + dy — a
//     fragment is read from LDS ONCE and serves three k-steps out of a ring of three rows in registers (3 dx x 2 parts x 3 rows);
//   * the dx shift is the lane's source-row address, loop-invariant (x = k mod 6 belongs to the lane): no edge tests in the loop;
//   * taps that leave the board vertically are whole k-steps of zeros and are skipped (y = 0: dy = -1, y = 6: dy = +1): 57 of 63
//     tap-rows per group, which pays for the 2 idle k of 32;
//   * per k-step 12 A reads + 16 dY reads instead of 36 + 16, 10 KB of tile stores instead of 12, no halo rows.
// The k-steps of a slice form one flat sequence s (7 per group of 5 boards; "row 7" of a group is row 0 of the next, and the taps that
// would mix them are the skipped ones).  In k-step s the LDS tile holds {dY(s + 1), A(s + 2)}: stored at the start of the k-step (its
// global loads were issued one k-step earlier), read into the NEXT fragment registers while this k-step's MFMAs run from registers —
// the dY fragments at once, the A fragments into the ring slot of row s - 1 once that row's taps (dy = -1) are done.  Ring slots and
// the dY double buffer are compile-time: the loop body is six k-steps.
// Measured (batch 512, one box, rocprofv3 averages over 1000 launches; profiles/r04_train_step.txt): 62.2 us against 68.9 for t_wgrad_rs
// (64.3 before the memory operations were dealt between the MFMAs).  Of the 62: 39 are the 4788 MFMAs of a block (7 groups x 7 k-steps),
// ~9 the 35 MB of split-K partials that all 960 waves write at the same moment, ~3 the prologue's three dependent round trips.
// ---------------------------------------------------------------------------------------------------------------------
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int NC>   // NC = co tiles of 16 per wave: 4 (64 output channels, 64 blocks per slice) or 2 (32 channels, 128 blocks per slice)
struct Wg5 {
    static constexpr int GB = 5, KR = 32, ROWS_Y = 7;
    static constexpr int AST = 32;                                     // bytes per row of the A tile (16 ci)
    __host__ __device__ static constexpr int arow(int r) { return r * AST + (r >> 3) * 128; }   // rows 0 .. 31, row 32 = the zero row
    static constexpr int APB = 33 * AST + 5 * 128;
    // dY tile rows: NC x 32 bytes + pad so that the pitch is 8 banks mod 16 — with 32 banks in front of every further group of 8 rows the
    // eight rows of a transposed read (r .. r + 3 and r + 8 .. r + 11, 8 banks each) cover the 64 banks once (40 banks / 24 banks)
    static constexpr int GST = NC == 4 ? 160 : 96;
    __host__ __device__ static constexpr int grow(int r) { return r * GST + (r >> 3) * 128; }
    static constexpr int GPB = KR * GST + (KR / 8) * 128;
    static constexpr int LDS_BYTES = 2 * APB + 2 * GPB;               // one wave's tile: A part 0 | A part 1 | dY part 0 | dY part 1
    static constexpr int LDS_BLOCK = 4 * LDS_BYTES > 2 * 9 * NC * 1024 ? 4 * LDS_BYTES : 2 * 9 * NC * 1024;   // four waves' tiles, or two accumulator sets in the closing sum
    static constexpr int COW = 16 * NC;                                // output channels per wave
    static constexpr int BLOCKS_PER_SLICE = 16 * (NF / COW);
    // the memory operations of a k-step, in dependence order: tile stores (GU dY units x 2 parts, 2 A units), global loads (the same
    // units), dY fragments (2 parts x NC, two reads each), A fragments (3 dx x 2 parts, two reads each)
    static constexpr int GU = NC;                                      // 16-byte dY units per lane and part (32 rows x 2 NC segments / 64 lanes)
    static constexpr int NW = 2 * GU + 2, NL = 2 * GU + 2, NG = 2 * NC, NA = 6, NOPS = NW + NL + NG + NA;
    static constexpr int BUDGET = NC == 4 ? 6 : 4;                     // memory instructions dealt into one tap's slot (3 NC MFMAs)
    __host__ __device__ static constexpr int cost(int k) { return k < NW + NL ? 1 : 2; }
    __host__ __device__ static constexpr int slot_lo(int slot)          // first operation of a slot: greedy fill in order
    {
        int k = 0;
        for (int sl = 0; sl < slot; sl++) {
            int b = 0;
            while (k < NOPS && b + cost(k) <= BUDGET) { b += cost(k); k++; }
        }
        return k;
    }
    static_assert(slot_lo(9) == NOPS, "nine slots take every operation");
    static_assert(slot_lo(3) <= NW + NL + NG, "the A fragment reads stand behind the dy = -1 taps (slots 0 - 2)");
};

template <int T, int NC>
__device__ __forceinline__ void g5_tap(const s16x8 (&a)[2], const s16x8 (&gf)[2][NC], f32x4 (&acc)[9][NC])
{
#pragma unroll
    for (int c = 0; c < NC; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[1]), __builtin_bit_cast(bf16x8, gf[0][c]), acc[T][c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < NC; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, gf[1][c]), acc[T][c], 0, 0, 0);
#pragma unroll
    for (int c = 0; c < NC; c++)
        acc[T][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0]), __builtin_bit_cast(bf16x8, gf[0][c]), acc[T][c], 0, 0, 0);
}

// A block is FOUR waves = four slices of one (ci tile, co range): each wave runs its slice alone (private tile, no barrier in the loop),
// and the four accumulator sets are summed through LDS before anything is written — a quarter of the split-K partials leave the chip and
// come back into the slice sum (512 records: 4 instead of 15 per weight; t_sum_slices_fin 10.6 -> 6.4 us, the kernel itself unchanged:
// the two rounds through LDS cost what the smaller write saves).  Block id -> (quad of slices, rest): the blocks of a quad are NQ apart,
// i.e. on the same two XCDs, whose L2s then hold that quad's dY.
template <int NC>
__global__ __launch_bounds__(256, 1) void t_wgrad_g5(Parts A, Parts G, float* __restrict__ out, int boards, int NS, int boards_per_slice)
{
    using W = Wg5<NC>;
    extern __shared__ __attribute__((aligned(16))) uint8_t wg_lds_all[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;   // (uniform by construction: say so, or every buffer load gets a waterfall loop around its descriptor)
    uint8_t* wg_lds = wg_lds_all + wave * W::LDS_BYTES;
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, p = i16 & 3;
    const int NQ = (NS + 3) / 4;
    const int quad = blockIdx.x % NQ, rest = blockIdx.x / NQ, cit = rest & 15, wq = rest >> 4;   // wq = which COW output channels
    const int slice = quad * 4 + wave;
    const int bbeg = slice * boards_per_slice, bend = min(boards, bbeg + boards_per_slice);
    const int S = slice < NS ? W::ROWS_Y * ((bend - bbeg + W::GB - 1) / W::GB) : 0;   // k-steps of the slice (a quad past the last slice: none)

    if (lane < 2 * (W::AST / 4))   // zero rows of the two A parts
        reinterpret_cast<uint32_t*>(wg_lds + (lane / (W::AST / 4)) * W::APB + W::arow(32))[lane % (W::AST / 4)] = 0u;

    // Staging units of this lane: GU of the dY tile per part ((tile row, 16-byte segment) pairs) and one of the A tile per part.
    // Tile row k = board 6 j + column x of the group; rows 30, 31 and the boards past the slice are out of range of the buffer resources
    // (everything is in the vector offset, which the range check sees) and arrive as zeros.
    const uint32_t range = (uint32_t)bend * NPOS * NF * 2u;
    const __amdgpu_buffer_rsrc_t gsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G.p[0]), (short)0, range, 0x00020000);
    const __amdgpu_buffer_rsrc_t gsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(G.p[1]), (short)0, range, 0x00020000);
    const __amdgpu_buffer_rsrc_t asrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A.p[0]), (short)0, range, 0x00020000);
    const __amdgpu_buffer_rsrc_t asrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A.p[1]), (short)0, range, 0x00020000);
    constexpr uint32_t OOR = 0xfffffff0u;
    constexpr int SEGS = 2 * NC;   // 16-byte segments of a dY tile row
    uint32_t gvo[W::GU], gl[W::GU], avo, al;
#pragma unroll
    for (int i = 0; i < W::GU; i++) {
        const int u = lane + 64 * i, k = u / SEGS, seg = u % SEGS;
        gvo[i] = k < 30 ? (uint32_t)((((bbeg + k / 6) * NPOS + k % 6) * NF + wq * W::COW + seg * 8) * 2) : OOR;
        gl[i] = (uint32_t)(2 * W::APB + W::grow(k) + seg * 16);
    }
    {
        const int k = lane >> 1, seg = lane & 1;
        avo = k < 30 ? (uint32_t)((((bbeg + k / 6) * NPOS + k % 6) * NF + cit * 16 + seg * 8) * 2) : OOR;
        al = (uint32_t)(W::arow(k) + seg * 16);
    }
    // byte offset of flat k-step s: group s / 7 (5 boards further each), board row s % 7
    auto step_off = [](int s) -> uint32_t { return (uint32_t)(((s / W::ROWS_Y) * W::GB * NPOS + (s % W::ROWS_Y) * 6) * NF * 2); };
    u32x4 sg[2 * W::GU], sa[2];

    // the two tile rows this lane addresses in a transposed read (k1 = 8 g + q and k1 + 4), their columns, and per dx the LDS address of
    // the source row: k + dx inside the board row, the zero row outside it (and for the idle k = 30, 31)
    const int k1 = 8 * g + q, k2 = k1 + 4;
    const uint32_t lbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)wg_lds;
    const uint32_t a_zero = lbase + (uint32_t)(W::arow(32) + p * 8);
    uint32_t aoff[3][2];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const int x1 = k1 % 6 + d - 1, x2 = k2 % 6 + d - 1;
        aoff[d][0] = (k1 < 30 && x1 >= 0 && x1 < 6) ? lbase + (uint32_t)(W::arow(k1 + d - 1) + p * 8) : a_zero;
        aoff[d][1] = (k2 < 30 && x2 >= 0 && x2 < 6) ? lbase + (uint32_t)(W::arow(k2 + d - 1) + p * 8) : a_zero;
    }
    const uint32_t g_lo = lbase + (uint32_t)(2 * W::APB + W::grow(k1) + p * 8);
    const uint32_t g_hi = lbase + (uint32_t)(2 * W::APB + W::grow(k2) + p * 8);

    f32x4 acc[9][NC];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int c = 0; c < NC; c++) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    s16x8 R[3][3][2];     // A fragments [ring slot = board-row index mod 3][dx][part]
    s16x8 gf[2][2][NC];   // dY fragments [k-step mod 2][part][co tile]

    // Operation K of a k-step that stores tile TS + 1, requests tile TS + 2 (vg, va: its vector offsets), reads the dY fragments into gf[GN]
    // and the A fragments into ring slot RM:  tile stores | global loads | dY fragments | A fragments (Wg5: NW, NL, NG, NA)
    uint32_t vg[W::GU], va = OOR;
    auto op = [&](auto kk, auto gnn, auto rmm) {
        constexpr int K = decltype(kk)::value, GN = decltype(gnn)::value, RM = decltype(rmm)::value;
        constexpr int KW = K, KL = K - W::NW, KG = K - W::NW - W::NL, KA = K - W::NW - W::NL - W::NG;
        if constexpr (KW < W::GU) *reinterpret_cast<u32x4*>(wg_lds + gl[KW]) = sg[KW];
        else if constexpr (KW < 2 * W::GU) *reinterpret_cast<u32x4*>(wg_lds + W::GPB + gl[KW - W::GU]) = sg[KW];
        else if constexpr (KW == 2 * W::GU) *reinterpret_cast<u32x4*>(wg_lds + al) = sa[0];
        else if constexpr (KW == 2 * W::GU + 1) *reinterpret_cast<u32x4*>(wg_lds + W::APB + al) = sa[1];
        else if constexpr (KL < W::GU) sg[KL] = __builtin_amdgcn_raw_buffer_load_b128(gsrc0, vg[KL], 0, 0);
        else if constexpr (KL < 2 * W::GU) sg[KL] = __builtin_amdgcn_raw_buffer_load_b128(gsrc1, vg[KL - W::GU], 0, 0);
        else if constexpr (KL == 2 * W::GU) sa[0] = __builtin_amdgcn_raw_buffer_load_b128(asrc0, va, 0, 0);
        else if constexpr (KL == 2 * W::GU + 1) sa[1] = __builtin_amdgcn_raw_buffer_load_b128(asrc1, va, 0, 0);
        else if constexpr (KG < W::NG) gf[GN][KG / NC][KG % NC] = lds_tr8<(KG / NC) * W::GPB + (KG % NC) * 32>(g_lo, g_hi);
        else R[RM][KA / 2][KA % 2] = lds_tr8<(KA % 2) * W::APB>(aoff[KA / 2][0], aoff[KA / 2][1]);
    };
    auto offsets = [&](int sd, int sa_) {   // vector offsets of the loads of {dY(sd), A(sa_)}
        const uint32_t so_g = step_off(sd), so_a = step_off(sa_);
#pragma unroll
        for (int i = 0; i < W::GU; i++) vg[i] = gvo[i] == OOR ? OOR : gvo[i] + so_g;
        va = avo == OOR ? OOR : avo + so_a;
    };
#define IC(n) std::integral_constant<int, (n)>{}
    // prologue: A(0) -> ring slot 0; tile 0 = {dY(0), A(1)} -> gf[0], ring slot 1; tile 1 on its way
    offsets(0, 0);
    op(IC(W::NW + 2 * W::GU), IC(0), IC(0)); op(IC(W::NW + 2 * W::GU + 1), IC(0), IC(0));                   // load A(0)
    op(IC(2 * W::GU), IC(0), IC(0)); op(IC(2 * W::GU + 1), IC(0), IC(0));                                   // store it
    asm volatile("" ::: "memory");
    static_for<W::NW + W::NL + W::NG, W::NOPS>([&](auto k) { op(k, IC(0), IC(0)); });                       // -> R[0]
    offsets(0, 1);
    static_for<W::NW, W::NW + W::NL>([&](auto k) { op(k, IC(0), IC(0)); });                                 // load tile 0
    asm volatile("" ::: "memory");
    static_for<0, W::NW>([&](auto k) { op(k, IC(0), IC(0)); });                                             // store it
    asm volatile("" ::: "memory");
    static_for<W::NW + W::NL, W::NOPS>([&](auto k) { op(k, IC(0), IC(1)); });                               // -> gf[0], R[1]
    offsets(1, 2);
    static_for<W::NW, W::NW + W::NL>([&](auto k) { op(k, IC(0), IC(0)); });                                 // load tile 1

    // One k-step = one straight-line piece of code per ring phase, cut into SLOTS of one tap (3 NC MFMAs) each.  The memory instructions of
    // the k-step are dealt over the slots in dependence order (Wg5::slot_lo) and inside a slot one is issued behind each of the first MFMAs
    // (sched_group_barrier; a slot is one scheduling region): a lone wave issues in order, and ten stores or sixteen reads in a row in
    // front of the MFMAs leave the matrix pipe idle for as long as they take to issue.
    int y = 0;
    auto kstep = [&](auto ph, int s) {
        constexpr int PH = decltype(ph)::value;
        constexpr int rm = (PH + 2) % 3, r0 = PH % 3, rp = (PH + 1) % 3, gc = PH % 2, gn = (PH + 1) % 2;
        offsets(s + 2, s + 3);   // tile s + 2 = {dY(s + 2), A(s + 3)} (past the slice: zeros or the next slice's rows — nobody multiplies them)
        auto slot = [&](auto tt) {
            constexpr int T = decltype(tt)::value, LO = W::slot_lo(T), HI = W::slot_lo(T + 1);
            constexpr int NI = []() { int n = 0; for (int k = LO; k < HI; k++) n += W::cost(k); return n; }();
            constexpr int RS = T / 3 == 0 ? rm : T / 3 == 1 ? r0 : rp;
            auto ops = [&](auto k) { op(k, IC(gn), IC(rm)); };   // (the A fragments: A(s + 2) into the slot row s - 1 has left)
            __builtin_amdgcn_sched_barrier(0);
            // (the taps that leave the board vertically are whole k-steps of zeros: skipped; their slots' memory operations are not)
            if (T / 3 == 1 || (T / 3 == 0 ? y > 0 : y < W::ROWS_Y - 1)) {
                static_for<LO, HI>(ops);
                g5_tap<T, NC>(R[RS][T % 3], gf[gc], acc);
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x0a0, 1, 0);   // one LDS or global-load instruction
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * NC - NI, 0);
            } else {
                asm volatile("; slot without its tap" ::: "memory");   // (keeps the two branches' common operations from being hoisted in front of the branch)
                static_for<LO, HI>(ops);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 9>(slot);
        y = y == W::ROWS_Y - 1 ? 0 : y + 1;
    };
    for (int s = 0; s < S; s += 6) {
        kstep(IC(0), s);
        if (s + 1 < S) kstep(IC(1), s + 1);
        if (s + 2 < S) kstep(IC(2), s + 2);
        if (s + 3 < S) kstep(IC(3), s + 3);
        if (s + 4 < S) kstep(IC(4), s + 4);
        if (s + 5 < S) kstep(IC(5), s + 5);
    }
#undef IC
    // (w0 + w2) + (w1 + w3): two rounds through LDS (the tiles are dead behind the first barrier), 16 bytes per lane and accumulator tile
    f32x4* red = reinterpret_cast<f32x4*>(wg_lds_all) + lane;
    constexpr int TILES = 9 * NC;
    __syncthreads();
    if (wave >= 2) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int c = 0; c < NC; c++) red[((wave - 2) * TILES + t * NC + c) * 64] = acc[t][c];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int c = 0; c < NC; c++) acc[t][c] += red[(wave * TILES + t * NC + c) * 64];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int t = 0; t < 9; t++)
#pragma unroll
            for (int c = 0; c < NC; c++) red[(t * NC + c) * 64] = acc[t][c];
    }
    __syncthreads();
    if (wave != 0) return;
    float* o = out + (size_t)quad * KC * NF;
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const f32x4 v = acc[t][c] + red[(t * NC + c) * 64];
#pragma unroll
            for (int e = 0; e < 4; e++) o[(size_t)(t * NF + cit * 16 + 4 * g + e) * NF + wq * W::COW + c * 16 + i16] = v[e];
        }
}

// out[i] = sum_z part[z][i]
__global__ void t_sum_slices(const float* __restrict__ part, int nz, size_t n, float* __restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int z = 0; z < nz; z++) s += part[(size_t)z * n + i];
    out[i] = s;
}

// =====================================================================================================================
// data movement
// =====================================================================================================================
// records [n][265] (i8 player | in88 | f32 z | f32 pi[43]; alphazero_nn_data.h:111-141) -> minibatch tensors
// `cur` = {offset of this minibatch in perm, Adam step count}: device-resident so that one captured graph serves every step
__global__ void t_gather(const uint8_t* __restrict__ rec, const int* __restrict__ perm, const int* __restrict__ cur, int BS,
                         uint8_t* __restrict__ in88, float* __restrict__ pit, float* __restrict__ zt)
{
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t* r = rec + (size_t)perm[cur[0] + cur[2] + b] * 265;   // cur[2] = this rank's offset inside the global minibatch
    for (int i = t; i < 88; i += blockDim.x) in88[b * 88 + i] = r[1 + i];
    for (int i = t; i < 44; i += blockDim.x) {
        float f;
        memcpy(&f, r + 89 + 4 * i, 4);
        if (i == 0) zt[b] = f; else pit[b * 43 + i - 1] = f;
    }
}

// setInStateTensor (alphazero_nn.cpp:31-67): in88 -> [M][16] planes (13 used)
__global__ void t_planes(const uint8_t* __restrict__ in88, int M, float* __restrict__ X0)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * SIN) return;
    const int r = i / SIN, c = i % SIN, b = r / NPOS, pos = r % NPOS;
    const uint8_t* in = in88 + (size_t)b * 88;
    float v = 0.0f;
    if (c < 13) {
        const uint32_t la = in[pos];
        const int army = la & 63, owner = la >> 6, cur = in[42], enemy = cur == 0 ? 1 : 0;
        const float fa = (float)army / 32.0f;
        float f[10];
        memcpy(f, in + 48, 40);
        switch (c) {
        case 0: v = owner == cur ? fa : 0.0f; break;
        case 1: v = owner == enemy ? fa : 0.0f; break;
        case 2: v = owner == 2 ? fa : 0.0f; break;
        case 3: v = f[9]; break;
        case 4: v = f[0]; break;
        case 5: v = f[1]; break;
        case 6: v = f[2]; break;
        default: v = f[3 + (c - 7)]; break;
        }
    }
    X0[i] = v;
}

// stem kernel [9][13][256] <-> padded [9][16][256]
__global__ void t_stem_pad(const float* __restrict__ w, float* __restrict__ wp)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= KS * NF) return;
    const int co = i % NF, ci = (i / NF) % SIN, tap = i / (NF * SIN);
    wp[i] = ci < 13 ? w[((size_t)tap * 13 + ci) * NF + co] : 0.0f;
}
__global__ void t_stem_unpad(const float* __restrict__ gp, float* __restrict__ g)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * 13 * NF) return;
    const int co = i % NF, ci = (i / NF) % 13, tap = i / (NF * 13);
    g[i] = gp[((size_t)tap * SIN + ci) * NF + co];
}

// col[r][tap][c] = A[r + dy*6 + dx][c] inside the board, else 0   (tap = (dy+1)*3 + (dx+1))
template <int C>
__global__ __launch_bounds__(256) void t_im2col(const float* __restrict__ A, float* __restrict__ col, int M)
{
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= M * 9) return;
    const int r = idx / 9, tap = idx % 9, dy = tap / 3 - 1, dx = tap % 3 - 1;
    const int pos = r % NPOS, y = pos / 6 + dy, x = pos % 6 + dx;
    const bool ok = y >= 0 && y < 7 && x >= 0 && x < 6;
    const float4* src = reinterpret_cast<const float4*>(A) + (size_t)(r + dy * 6 + dx) * (C / 4);
    float4* dst = reinterpret_cast<float4*>(col) + (size_t)idx * (C / 4);
    for (int q = lane; q < C / 4; q += 64) dst[q] = ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// a += b
__global__ __launch_bounds__(256) void t_add(float* __restrict__ a, const float* __restrict__ b, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 x = reinterpret_cast<float4*>(a)[i];
    const float4 y = reinterpret_cast<const float4*>(b)[i];
    x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
    reinterpret_cast<float4*>(a)[i] = x;
}

// =====================================================================================================================
// batch normalisation, training mode.  STEM = the conv_bn layer normalising over axis 1 = board row y (7 groups,
// build_graph.py:68); otherwise per channel.  Stage 1: per block of RB rows, thread c accumulates in double; stage 2: one
// block sums the partials.
// =====================================================================================================================
constexpr int NG = 7;  // stem groups

__device__ __forceinline__ uint32_t bf_rne_bits(float f);
// the bf16 parts of 4 consecutive values (see t_split), written as one 8-byte store per part
__device__ __forceinline__ void split_store4(const float (&v)[4], size_t i4, uint16_t* p0, uint16_t* p1, uint16_t* p2);

// sum of v over the 4 row-groups q = t >> 8 of a 1024-thread block, per channel c = t & 255 (result valid where q == 0)
__device__ __forceinline__ double reduce_q4(double v, double* sh)
{
    const int t = threadIdx.x;
    __syncthreads();
    sh[t] = v;
    __syncthreads();
    if (t < 256) v = sh[t] + sh[t + 256] + sh[t + 512] + sh[t + 768];
    return v;
}

// sum of v over the 32 row-groups q = t >> 5 of a 1024-thread block, per channel slot t & 31 (valid where q == 0)
__device__ __forceinline__ double reduce_q32(double v, double* sh)
{
    const int t = threadIdx.x;
    __syncthreads();
    sh[t] = v;
    __syncthreads();
    for (int o = 512; o >= 32; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    return sh[t & 31];
}

__device__ __forceinline__ double block_sum_1024(double v, double* sh)
{
    const int t = threadIdx.x;
    __syncthreads();
    sh[t] = v;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    return sh[0];
}

template <bool STEM>
__global__ __launch_bounds__(1024) void t_bn_stats(const float* __restrict__ Y, int M, double* __restrict__ part)
{
    __shared__ double sh[1024];
    const int c = threadIdx.x & 255, q = threadIdx.x >> 8, r0 = blockIdx.x * RB, r1 = min(M, r0 + RB);
    if constexpr (!STEM) {
        double s = 0.0, ss = 0.0;
#pragma unroll 4
        for (int r = r0 + q; r < r1; r += 4) { const double v = Y[(size_t)r * NF + c]; s += v; ss += v * v; }
        s = reduce_q4(s, sh);
        ss = reduce_q4(ss, sh);
        if (q == 0) {
            part[((size_t)blockIdx.x * 2 + 0) * NF + c] = s;
            part[((size_t)blockIdx.x * 2 + 1) * NF + c] = ss;
        }
    } else {
        double s[NG], ss[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) s[g] = ss[g] = 0.0;
        for (int r = r0 + q; r < r1; r += 4) {
            const double v = Y[(size_t)r * NF + c];
            const int y = (r % NPOS) / 6;
#pragma unroll
            for (int g = 0; g < NG; g++) { s[g] += y == g ? v : 0.0; ss[g] += y == g ? v * v : 0.0; }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const double a = reduce_q4(s[g], sh), b = reduce_q4(ss[g], sh);
            if (q == 0) {
                part[((size_t)blockIdx.x * 2 * NG + g) * NF + c] = a;
                part[((size_t)blockIdx.x * 2 * NG + NG + g) * NF + c] = b;
            }
        }
    }
}

// mean / 1/sqrt(var+eps) of the batch + moving-average update (TF fused BN: moving variance gets Bessel's correction)
template <bool STEM>
__global__ __launch_bounds__(1024) void t_bn_finalize(const double* __restrict__ part, int R, double count, float* __restrict__ mean,
                                                      float* __restrict__ istd, float* __restrict__ bn /* g|b|mu|var */)
{
    __shared__ double sh[1024];
    int c = threadIdx.x & 255, q = threadIdx.x >> 8;
    if constexpr (!STEM) {
        // grid of 8 blocks: block = 32 channels x 32 groups of partial rows
        c = blockIdx.x * 32 + (threadIdx.x & 31);
        q = threadIdx.x >> 5;
        double s = 0.0, ss = 0.0;
        for (int b = q; b < R; b += 32) { s += part[((size_t)b * 2 + 0) * NF + c]; ss += part[((size_t)b * 2 + 1) * NF + c]; }
        s = reduce_q32(s, sh);
        ss = reduce_q32(ss, sh);
        if (q == 0) {
            const double mu = s / count, var = fmax(ss / count - mu * mu, 0.0);
            mean[c] = (float)mu;
            istd[c] = (float)(1.0 / sqrt(var + (double)BN_EPS));
            bn[2 * NF + c] = bn[2 * NF + c] * BN_KEEP + (float)mu * (1.0f - BN_KEEP);
            bn[3 * NF + c] = bn[3 * NF + c] * BN_KEEP + (float)(var * count / (count - 1.0)) * (1.0f - BN_KEEP);
        }
    } else {
        {   // grid of NG blocks: one board row each (the sums of a row keep their order)
            const int g = blockIdx.x;
            double s = 0.0, ss = 0.0;
#pragma unroll 8
            for (int b = q; b < R; b += 4) {
                s += part[((size_t)b * 2 * NG + g) * NF + c];
                ss += part[((size_t)b * 2 * NG + NG + g) * NF + c];
            }
            s = block_sum_1024(s, sh);
            ss = block_sum_1024(ss, sh);
            if (threadIdx.x == 0) {
                const double mu = s / count, var = fmax(ss / count - mu * mu, 0.0);
                mean[g] = (float)mu;
                istd[g] = (float)(1.0 / sqrt(var + (double)BN_EPS));
                bn[2 * NG + g] = bn[2 * NG + g] * BN_KEEP + (float)mu * (1.0f - BN_KEEP);
                bn[3 * NG + g] = bn[3 * NG + g] * BN_KEEP + (float)(var * count / (count - 1.0)) * (1.0f - BN_KEEP);
            }
        }
    }
}

// A = relu(gamma * (Y - mean) * istd + beta (+ S))
template <bool STEM>
__global__ __launch_bounds__(256) void t_bn_apply(const float* __restrict__ Y, const float* __restrict__ mean, const float* __restrict__ istd,
                                                  const float* __restrict__ bn, const float* __restrict__ S, float* __restrict__ A, int M,
                                                  uint16_t* __restrict__ p0, uint16_t* __restrict__ p1, uint16_t* __restrict__ p2,
                                                  uint16_t* __restrict__ q0 = nullptr, uint16_t* __restrict__ q1 = nullptr)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // float4 index
    if (i >= (size_t)M * (NF / 4)) return;
    const int r = (int)(i / (NF / 4)), c4 = (int)(i % (NF / 4)) * 4;
    const int CH = STEM ? NG : NF;
    const float4 y = reinterpret_cast<const float4*>(Y)[i];
    const float4 s = S ? reinterpret_cast<const float4*>(S)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float yy[4] = {y.x, y.y, y.z, y.w}, sv[4] = {s.x, s.y, s.z, s.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ch = STEM ? (r % NPOS) / 6 : c4 + j;
        const float v = bn[ch] * ((yy[j] - mean[ch]) * istd[ch]) + bn[CH + ch] + sv[j];
        o[j] = v > 0.0f ? v : 0.0f;
    }
    reinterpret_cast<float4*>(A)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (p0) split_store4(o, i, p0, p1, p2);  // bf16 parts: the weight-gradient GEMM's operand (and, with p2, the 6-pass forward conv's)
    if (q0) split_store4_f16(o, i, q0, q1);  // fp16 pair: the next layer's 3-pass forward conv
}

// backward stage 1: dz = dOut * (Apost > 0); partial sums of dz and dz * xhat
template <bool STEM>
__global__ __launch_bounds__(1024) void t_bn_bwd_stats(const float* __restrict__ dOut, const float* __restrict__ Apost,
                                                       const float* __restrict__ Y, const float* __restrict__ mean,
                                                       const float* __restrict__ istd, int M, double* __restrict__ part)
{
    __shared__ double sh[1024];
    const int c = threadIdx.x & 255, q = threadIdx.x >> 8, r0 = blockIdx.x * RB, r1 = min(M, r0 + RB);
    if constexpr (!STEM) {
        const float mu = mean[c], is = istd[c];
        double s = 0.0, sx = 0.0;
#pragma unroll 4
        for (int r = r0 + q; r < r1; r += 4) {
            const size_t i = (size_t)r * NF + c;
            const float dz = Apost[i] > 0.0f ? dOut[i] : 0.0f;
            s += dz;
            sx += (double)dz * (double)((Y[i] - mu) * is);
        }
        s = reduce_q4(s, sh);
        sx = reduce_q4(sx, sh);
        if (q == 0) {
            part[((size_t)blockIdx.x * 2 + 0) * NF + c] = s;
            part[((size_t)blockIdx.x * 2 + 1) * NF + c] = sx;
        }
    } else {
        double s[NG], sx[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) s[g] = sx[g] = 0.0;
        for (int r = r0 + q; r < r1; r += 4) {
            const size_t i = (size_t)r * NF + c;
            const int y = (r % NPOS) / 6;
            const float dz = Apost[i] > 0.0f ? dOut[i] : 0.0f;
            const double x = (double)dz * (double)((Y[i] - mean[y]) * istd[y]);
#pragma unroll
            for (int g = 0; g < NG; g++) { s[g] += y == g ? (double)dz : 0.0; sx[g] += y == g ? x : 0.0; }
        }
#pragma unroll
        for (int g = 0; g < NG; g++) {
            const double a = reduce_q4(s[g], sh), b = reduce_q4(sx[g], sh);
            if (q == 0) {
                part[((size_t)blockIdx.x * 2 * NG + g) * NF + c] = a;
                part[((size_t)blockIdx.x * 2 * NG + NG + g) * NF + c] = b;
            }
        }
    }
}

// data-parallel step: this rank's per-block partials -> one [K][256] slab of doubles, which the ranks then all-reduce; the
// finalize kernels read the reduced slab as "R = 1 block of partials"
__global__ __launch_bounds__(256) void t_parts_sum(const double* __restrict__ part, int R, int K, double* __restrict__ red)
{
    const int c = threadIdx.x, k = blockIdx.x;
    double s = 0.0;
    for (int b = 0; b < R; b++) s += part[((size_t)b * K + k) * NF + c];
    red[(size_t)k * NF + c] = s;
}

// backward stage 2: d(beta) = sum dz, d(gamma) = sum dz * xhat -> gradient vector; sums[0|1][ch] kept for stage 3.
// gscale = 1 / world in a data-parallel step (the sums are already global; the closing all-reduce of the gradient vector
// adds the `world` copies up again)
template <bool STEM>
__global__ __launch_bounds__(1024) void t_bn_bwd_finalize(const double* __restrict__ part, int R, float* __restrict__ gbn, float* __restrict__ sums,
                                                          float gscale)
{
    __shared__ double sh[1024];
    int c = threadIdx.x & 255, q = threadIdx.x >> 8;
    if constexpr (!STEM) {
        c = blockIdx.x * 32 + (threadIdx.x & 31);
        q = threadIdx.x >> 5;
        double s = 0.0, sx = 0.0;
        for (int b = q; b < R; b += 32) { s += part[((size_t)b * 2 + 0) * NF + c]; sx += part[((size_t)b * 2 + 1) * NF + c]; }
        s = reduce_q32(s, sh);
        sx = reduce_q32(sx, sh);
        if (q == 0) {
            gbn[c] = (float)sx * gscale;
            gbn[NF + c] = (float)s * gscale;
            sums[c] = (float)s;
            sums[NF + c] = (float)sx;
        }
    } else {
        {   // grid of NG blocks: one board row each
            const int g = blockIdx.x;
            double s = 0.0, sx = 0.0;
#pragma unroll 8
            for (int b = q; b < R; b += 4) {
                s += part[((size_t)b * 2 * NG + g) * NF + c];
                sx += part[((size_t)b * 2 * NG + NG + g) * NF + c];
            }
            s = block_sum_1024(s, sh);
            sx = block_sum_1024(sx, sh);
            if (threadIdx.x == 0) { gbn[g] = (float)sx * gscale; gbn[NG + g] = (float)s * gscale; sums[g] = (float)s; sums[NF + g] = (float)sx; }
        }
    }
}

// Two small kernels of the backward chain in ONE launch: the slice sum of layer l + 1's weight gradient (out[i] = sum_z part[z][i],
// the job of t_sum_slices) and stage 2 of layer l's batch-norm backward (t_bn_bwd_finalize<false>, blocks 0 - 7).  They are neighbours in
// the stream and independent of each other — the finalize reads the block partials the backward-data conv of layer l + 1 left, the
// slice sum what that layer's weight-gradient kernel left — so one launch saves a kernel boundary (~3 us on the device) and hides the
// 5-us finalize under the 10-us sum: ~8 us per layer.
__global__ __launch_bounds__(1024) void t_sum_slices_fin(const float* __restrict__ wpart, int nz, size_t n, float* __restrict__ out,
                                                         const double* __restrict__ part, int R, float* __restrict__ gbn, float* __restrict__ sums,
                                                         float gscale)
{
    if (blockIdx.x < 8) {
        __shared__ double sh[1024];
        const int c = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
        double s = 0.0, sx = 0.0;
        for (int b = q; b < R; b += 32) { s += part[((size_t)b * 2 + 0) * NF + c]; sx += part[((size_t)b * 2 + 1) * NF + c]; }
        s = reduce_q32(s, sh);
        sx = reduce_q32(sx, sh);
        if (q == 0) {
            gbn[c] = (float)sx * gscale;
            gbn[NF + c] = (float)s * gscale;
            sums[c] = (float)s;
            sums[NF + c] = (float)sx;
        }
        return;
    }
    const size_t i = (size_t)(blockIdx.x - 8) * 1024 + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int z = 0; z < nz; z++) s += wpart[(size_t)z * n + i];
    out[i] = s;
}

// backward stage 3: dY = gamma * istd * (dz - sum(dz)/n - xhat * sum(dz xhat)/n); dZ (optional) = dz for the shortcut
template <bool STEM>
__global__ __launch_bounds__(256) void t_bn_bwd_apply(const float* __restrict__ dOut, const float* __restrict__ Apost,
                                                      const float* __restrict__ Y, const float* __restrict__ mean,
                                                      const float* __restrict__ istd, const float* __restrict__ bn,
                                                      const float* __restrict__ sums, float inv_count, float* __restrict__ dY,
                                                      float* __restrict__ dZ, int M, uint16_t* __restrict__ p0, uint16_t* __restrict__ p1)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * (NF / 4)) return;
    const int r = (int)(i / (NF / 4)), c4 = (int)(i % (NF / 4)) * 4;
    const float4 d4 = reinterpret_cast<const float4*>(dOut)[i], a4 = reinterpret_cast<const float4*>(Apost)[i],
                 y4 = reinterpret_cast<const float4*>(Y)[i];
    const float d[4] = {d4.x, d4.y, d4.z, d4.w}, a[4] = {a4.x, a4.y, a4.z, a4.w}, y[4] = {y4.x, y4.y, y4.z, y4.w};
    float o[4], z[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int ch = STEM ? (r % NPOS) / 6 : c4 + j;
        const float dz = a[j] > 0.0f ? d[j] : 0.0f;
        const float xh = (y[j] - mean[ch]) * istd[ch];
        z[j] = dz;
        o[j] = bn[ch] * istd[ch] * (dz - sums[ch] * inv_count - xh * (sums[NF + ch] * inv_count));
    }
    if (dY) reinterpret_cast<float4*>(dY)[i] = make_float4(o[0], o[1], o[2], o[3]);
    if (dZ) reinterpret_cast<float4*>(dZ)[i] = make_float4(z[0], z[1], z[2], z[3]);
    if (p0) split_store4(o, i, p0, p1, nullptr);  // operand parts of the two gradient GEMMs
}

// =====================================================================================================================
// heads (build_graph.py:76-98)
// =====================================================================================================================
// 1x1 convs: pv0[r] = { H[r] . pi_w[:,0], H[r] . pi_w[:,1], H[r] . v_w, 0 }; one wave per row
__global__ __launch_bounds__(256) void t_head_conv(const float* __restrict__ H, const float* __restrict__ hp, float* __restrict__ pv0, int M)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float4 h = reinterpret_cast<const float4*>(H)[(size_t)r * 64 + lane];
    const float hv[4] = {h.x, h.y, h.z, h.w};
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = lane * 4 + j;
        s0 += hv[j] * hp[H_PI_W + c * 2];
        s1 += hv[j] * hp[H_PI_W + c * 2 + 1];
        s2 += hv[j] * hp[H_V_W + c];
    }
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) reinterpret_cast<float4*>(pv0)[r] = make_float4(s0, s1, s2, 0.0f);
}

// batch statistics of the 3 head channels (bn_pi x2, bn_v) + moving averages; single block of 1024 threads.
// Two stages so that a data-parallel step can all-reduce the six sums in between: hsum = {s[3], ss[3]} (doubles).
__global__ __launch_bounds__(1024) void t_head_bn_sums(const float* __restrict__ pv0, int M, double* __restrict__ hsum)
{
    __shared__ double sh[1024];
    for (int ch = 0; ch < 3; ch++) {
        double s = 0.0, ss = 0.0;
        for (int r = threadIdx.x; r < M; r += 1024) { const double v = pv0[(size_t)r * 4 + ch]; s += v; ss += v * v; }
        s = block_sum_1024(s, sh);
        ss = block_sum_1024(ss, sh);
        if (threadIdx.x == 0) { hsum[ch] = s; hsum[3 + ch] = ss; }
    }
}
__global__ void t_head_bn_stats(const double* __restrict__ hsum, double n, float* __restrict__ hp, float* __restrict__ hstat /* mean[3] istd[3] */)
{
    const int ch = threadIdx.x;
    if (ch >= 3 || blockIdx.x != 0) return;
    const double mu = hsum[ch] / n, var = fmax(hsum[3 + ch] / n - mu * mu, 0.0);
    hstat[ch] = (float)mu;
    hstat[3 + ch] = (float)(1.0 / sqrt(var + (double)BN_EPS));
    float* bn = ch < 2 ? hp + H_PI_BN : hp + H_V_BN;
    const int C = ch < 2 ? 2 : 1, k = ch < 2 ? ch : 0;
    bn[2 * C + k] = bn[2 * C + k] * BN_KEEP + (float)mu * (1.0f - BN_KEEP);
    bn[3 * C + k] = bn[3 * C + k] * BN_KEEP + (float)(var * n / (n - 1.0)) * (1.0f - BN_KEEP);
}

__device__ __forceinline__ float head_bn_relu(const float* hp, const float* hstat, float x, int ch)
{
    const float* bn = ch < 2 ? hp + H_PI_BN : hp + H_V_BN;
    const int C = ch < 2 ? 2 : 1, k = ch < 2 ? ch : 0;
    const float v = bn[k] * ((x - hstat[ch]) * hstat[3 + ch]) + bn[C + k];
    return v > 0.0f ? v : 0.0f;
}

// dense parts + losses; one block of 256 threads per board.  Saves fpi[84], fv[42], h1[256], v, prob[43].
__global__ __launch_bounds__(256) void t_head_fwd(const float* __restrict__ pv0, const float* __restrict__ hp, const float* __restrict__ hstat,
                                                  const float* __restrict__ pit, const float* __restrict__ zt, float* __restrict__ fpi,
                                                  float* __restrict__ fv, float* __restrict__ h1, float* __restrict__ vout,
                                                  float* __restrict__ prob, float* __restrict__ lossb)
{
    __shared__ float s_pi[84], s_v[42], s_h[256], s_l[44];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 126) {
        const int cell = t / 3, ch = t % 3;
        const float f = head_bn_relu(hp, hstat, pv0[((size_t)b * NPOS + cell) * 4 + ch], ch);
        if (ch < 2) { s_pi[cell * 2 + ch] = f; fpi[b * 84 + cell * 2 + ch] = f; }
        else { s_v[cell] = f; fv[b * 42 + cell] = f; }
    }
    __syncthreads();
    {   // dense_1 42 -> 256 + ReLU
        float a = hp[H_V1_B + t];
        for (int k = 0; k < 42; k++) a += s_v[k] * hp[H_V1_W + k * 256 + t];
        a = a > 0.0f ? a : 0.0f;
        s_h[t] = a;
        h1[b * 256 + t] = a;
    }
    if (t < 43) {  // dense 84 -> 43
        float a = hp[H_PD_B + t];
        for (int k = 0; k < 84; k++) a += s_pi[k] * hp[H_PD_W + k * 43 + t];
        s_l[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        float mx = s_l[0];
        for (int j = 1; j < 43; j++) mx = fmaxf(mx, s_l[j]);
        float se = 0.0f;
        for (int j = 0; j < 43; j++) se += expf(s_l[j] - mx);
        const float lse = mx + logf(se);
        float lp = 0.0f;
        for (int j = 0; j < 43; j++) {
            prob[b * 43 + j] = expf(s_l[j] - lse);
            lp -= pit[b * 43 + j] * (s_l[j] - lse);
        }
        float a = hp[H_V2_B];
        for (int j = 0; j < 256; j++) a += s_h[j] * hp[H_V2_W + j];
        const float v = tanhf(a), dv = zt[b] - v;
        vout[b] = v;
        lossb[b * 2] = lp;
        lossb[b * 2 + 1] = dv * dv;
    }
}

// batch means of the two losses (softmax_cross_entropy / mean_squared_error reduce over the batch) -> loss[0..1];
// acc[0..1] += them (the epoch sums of alphazero_nn.cpp:393-394, float like the reference)
// (a data-parallel step sums its own boards, divides by the GLOBAL batch, all-reduces loss[0..1], then accumulates)
__global__ void t_loss(const float* __restrict__ lossb, int BS, int BS_global, float* __restrict__ loss)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float lp = 0.0f, lv = 0.0f;
    for (int b = 0; b < BS; b++) { lp += lossb[b * 2]; lv += lossb[b * 2 + 1]; }
    loss[0] = lp / (float)BS_global; loss[1] = lv / (float)BS_global;
}
__global__ void t_loss_acc(const float* __restrict__ loss, float* __restrict__ acc)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { acc[0] += loss[0]; acc[1] += loss[1]; }
}

// backward of the dense parts; writes dz of the three head BN outputs (dpv [M][4]) and the per-board parameter partials
__global__ __launch_bounds__(256) void t_head_bwd(const float* __restrict__ hp, const float* __restrict__ pit, const float* __restrict__ zt,
                                                  const float* __restrict__ fpi, const float* __restrict__ fv, const float* __restrict__ h1,
                                                  const float* __restrict__ vout, const float* __restrict__ prob, int BS,
                                                  float* __restrict__ dpv, float* __restrict__ hpart)
{
    __shared__ float s_dl[43], s_dh[256], s_pi[84], s_v[42];
    __shared__ float s_dv;
    const int b = blockIdx.x, t = threadIdx.x;
    const float inv = 1.0f / (float)BS;
    float* hpb = hpart + (size_t)b * HP_FLOATS;
    if (t < 84) s_pi[t] = fpi[b * 84 + t];
    if (t < 42) s_v[t] = fv[b * 42 + t];
    if (t == 0) {
        float sp = 0.0f;
        for (int j = 0; j < 43; j++) sp += pit[b * 43 + j];
        for (int j = 0; j < 43; j++) s_dl[j] = (prob[b * 43 + j] * sp - pit[b * 43 + j]) * inv;
        const float v = vout[b];
        s_dv = 2.0f * (v - zt[b]) * inv * (1.0f - v * v);
    }
    __syncthreads();
    const float dv = s_dv;
    {
        const float h = h1[b * 256 + t];
        const float dh = h > 0.0f ? hp[H_V2_W + t] * dv : 0.0f;
        s_dh[t] = dh;
        hpb[HP_V1_B + t] = dh;
        hpb[HP_V2_W + t] = h * dv;
        if (t == 0) hpb[HP_V2_B] = dv;
        if (t < 43) hpb[HP_PD_B + t] = s_dl[t];
    }
    __syncthreads();
    for (int i = t; i < 84 * 43; i += 256) hpb[HP_PD_W + i] = s_pi[i / 43] * s_dl[i % 43];
    for (int i = t; i < 42 * 256; i += 256) hpb[HP_V1_W + i] = s_v[i / 256] * s_dh[i % 256];
    if (t < 84) {  // d fpi -> dz of bn_pi
        float a = 0.0f;
        for (int j = 0; j < 43; j++) a += hp[H_PD_W + t * 43 + j] * s_dl[j];
        dpv[((size_t)b * NPOS + t / 2) * 4 + (t & 1)] = s_pi[t] > 0.0f ? a : 0.0f;
    } else if (t >= 128 && t < 128 + 42) {  // d fv -> dz of bn_v
        const int k = t - 128;
        float a = 0.0f;
        for (int j = 0; j < 256; j++) a += hp[H_V1_W + k * 256 + j] * s_dh[j];
        dpv[((size_t)b * NPOS + k) * 4 + 2] = s_v[k] > 0.0f ? a : 0.0f;
    }
}

// g[param] = sum over boards of the per-board partials
__global__ void t_head_reduce(const float* __restrict__ hpart, int BS, float* __restrict__ ghead)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HP_FLOATS) return;
    float s = 0.0f;
#pragma unroll 16   // (loads in flight together; the additions keep their order)
    for (int b = 0; b < BS; b++) s += hpart[(size_t)b * HP_FLOATS + i];
    int o;
    if (i < HP_PD_B) o = H_PD_W + i;
    else if (i < HP_V1_W) o = H_PD_B + (i - HP_PD_B);
    else if (i < HP_V1_B) o = H_V1_W + (i - HP_V1_W);
    else if (i < HP_V2_W) o = H_V1_B + (i - HP_V1_B);
    else if (i < HP_V2_B) o = H_V2_W + (i - HP_V2_W);
    else o = H_V2_B;
    ghead[o] = s;
}

// BN backward of the 3 head channels: dpv (dz) -> gradients of gamma/beta and dpv := d(conv output).  Two stages (sums,
// apply) so that a data-parallel step can all-reduce hsum = {s[3], sx[3]} in between; n = rows of the GLOBAL batch.
__global__ __launch_bounds__(1024) void t_head_bn_bwd_sums(const float* __restrict__ pv0, const float* __restrict__ hstat, int M,
                                                           const float* __restrict__ dpv, double* __restrict__ hsum)
{
    __shared__ double sh[1024];
    for (int ch = 0; ch < 3; ch++) {
        const float mu = hstat[ch], is = hstat[3 + ch];
        double s = 0.0, sx = 0.0;
        for (int r = threadIdx.x; r < M; r += 1024) {
            const float dz = dpv[(size_t)r * 4 + ch];
            s += dz;
            sx += (double)dz * (double)((pv0[(size_t)r * 4 + ch] - mu) * is);
        }
        s = block_sum_1024(s, sh);
        sx = block_sum_1024(sx, sh);
        if (threadIdx.x == 0) { hsum[ch] = s; hsum[3 + ch] = sx; }
    }
}
__global__ __launch_bounds__(1024) void t_head_bn_bwd(const float* __restrict__ pv0, const float* __restrict__ hp, const float* __restrict__ hstat,
                                                      int M, const double* __restrict__ hsum, float n, float gscale, float* __restrict__ dpv,
                                                      float* __restrict__ ghead)
{
    for (int ch = 0; ch < 3; ch++) {
        const float mu = hstat[ch], is = hstat[3 + ch];
        const double s = hsum[ch], sx = hsum[3 + ch];
        const int C = ch < 2 ? 2 : 1, k = ch < 2 ? ch : 0, base = ch < 2 ? H_PI_BN : H_V_BN;
        if (threadIdx.x == 0 && blockIdx.x == 0) { ghead[base + k] = (float)sx * gscale; ghead[base + C + k] = (float)s * gscale; }
        const float gamma = hp[base + k], fs = (float)s / n, fsx = (float)sx / n;
        for (int r = blockIdx.x * 1024 + threadIdx.x; r < M; r += gridDim.x * 1024) {
            const float dz = dpv[(size_t)r * 4 + ch];
            const float xh = (pv0[(size_t)r * 4 + ch] - mu) * is;
            dpv[(size_t)r * 4 + ch] = gamma * is * (dz - fs - xh * fsx);
        }
    }
}

// dH[r][c] = dp0 * pi_w[c][0] + dp1 * pi_w[c][1] + dv * v_w[c]; partial d(pi_w), d(v_w) per block of RB rows
__global__ __launch_bounds__(256) void t_head_conv_bwd(const float* __restrict__ H, const float* __restrict__ dpv, const float* __restrict__ hp,
                                                       int M, float* __restrict__ dH, float* __restrict__ part /* [R][3][256] */)
{
    const int c = threadIdx.x, r0 = blockIdx.x * RB, r1 = min(M, r0 + RB);
    const float w0 = hp[H_PI_W + c * 2], w1 = hp[H_PI_W + c * 2 + 1], w2 = hp[H_V_W + c];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int r = r0; r < r1; r++) {
        const float4 d = reinterpret_cast<const float4*>(dpv)[r];
        const float h = H[(size_t)r * NF + c];
        dH[(size_t)r * NF + c] = d.x * w0 + d.y * w1 + d.z * w2;
        g0 += h * d.x; g1 += h * d.y; g2 += h * d.z;
    }
    part[((size_t)blockIdx.x * 3 + 0) * NF + c] = g0;
    part[((size_t)blockIdx.x * 3 + 1) * NF + c] = g1;
    part[((size_t)blockIdx.x * 3 + 2) * NF + c] = g2;
}
__global__ __launch_bounds__(256) void t_head_conv_bwd_finalize(const float* __restrict__ part, int R, float* __restrict__ ghead)
{
    const int c = threadIdx.x;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll 8
    for (int b = 0; b < R; b++) {
        g0 += part[((size_t)b * 3 + 0) * NF + c];
        g1 += part[((size_t)b * 3 + 1) * NF + c];
        g2 += part[((size_t)b * 3 + 2) * NF + c];
    }
    ghead[H_PI_W + c * 2] = g0;
    ghead[H_PI_W + c * 2 + 1] = g1;
    ghead[H_V_W + c] = g2;
}

// =====================================================================================================================
// Adam (tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - b2^t) / (1 - b1^t); w -= lr_t * m / (sqrt(v) + eps)); kind 1 adds
// the L2 regulariser's gradient 2 * L2_C * w (keras l2 = l * sum w^2), kind 0 (BN moving statistics) is not trained
// =====================================================================================================================
// step t := t + 1 and its bias-corrected learning rate; afterwards the minibatch offset advances
__global__ void t_tick_lr(int* __restrict__ cur, float* __restrict__ lr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double t = (double)(++cur[1]);
    *lr = (float)((double)LR * sqrt(1.0 - pow((double)ADAM_B2, t)) / (1.0 - pow((double)ADAM_B1, t)));
}
__global__ void t_tick_batch(int* __restrict__ cur, int BS /* the GLOBAL minibatch */)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) cur[0] += BS;
}

__global__ void t_adam(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       const uint8_t* __restrict__ kind, size_t n, const float* __restrict__ lr)
{
    const float lr_t = *lr;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = kind[i];
    if (k == 0) return;
    float gi = g[i];
    if (k == 1) gi += 2.0f * L2_C * w[i];
    const float mi = ADAM_B1 * m[i] + (1.0f - ADAM_B1) * gi;
    const float vi = ADAM_B2 * v[i] + (1.0f - ADAM_B2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    w[i] -= lr_t * mi / (sqrtf(vi) + ADAM_EPS);
}

// =====================================================================================================================
// host
// =====================================================================================================================
// conv GEMM arithmetic: split bf16 (default; 6-pass forward, 3-pass backward) or the fp32 MFMA (AZR_TRAIN_GEMM=f32)
// (all five are test hooks: run-time switches in libazr_hip_test.so, compile-time constants in the product library)
#ifdef AZR_TEST_HOOKS
#define AZR_HOOK_FLAG bool
#else
#define AZR_HOOK_FLAG constexpr bool
#endif
AZR_HOOK_FLAG g_gemm_bf16x3 = true;
// forward conv arithmetic: fp16 pairs, 3 passes (default) or three bf16 parts, 6 passes (AZR_TRAIN_FWD=bf16)
AZR_HOOK_FLAG g_fwd_f16 = true;
constexpr float FWD_WSCALE = 1024.0f;   // the packed forward kernels are 2^10 * W: |w| < 64 stays inside fp16, a weight of 1e-4 keeps a normal low part
AZR_HOOK_FLAG g_fuse_bwd = true;
#ifndef WG5_NC
#define WG5_NC 4   // co tiles per wave of t_wgrad_g5 (2: half the split-K partials but 20 fragment reads per 54 MFMAs instead of 28 per 108 — 86.9 us against 61.2)
#endif
AZR_HOOK_FLAG g_wgrad_g5 = true;    // t_wgrad_g5 (k-step = a board row of five boards); AZR_TRAIN_WGRAD=rs: t_wgrad_rs (rows in memory order), the older formulation
AZR_HOOK_FLAG g_conv_q = true;       // t_conv_q for batches of up to 128 records (AZR_TRAIN_CONVQ=0: t_conv_rs at every size)
AZR_HOOK_FLAG g_fuse_apply = true;   // t_conv_rs<.., PRO>: the normalise kernels (t_bn_apply / t_bn_bwd_apply) computed in the consuming conv's staging path (AZR_TRAIN_FUSE_APPLY=0: separate kernels; same bits)   // t_conv_rs<2, 2, 1>: shortcut add + BN-backward stage 1 in the backward-data conv's epilogue (AZR_TRAIN_FUSE=0: separate kernels)

struct TrainCtx {
    int BS = 0, blocks = 0, M = 0, L = 0, R = 0, nz = 0, kchunk = 0;
    int wg_slices = 0, wg_rows = 0;          // weight gradient: slices (split-K units of whole boards) and rows per slice
    int wg_bps = 0;                          // ... boards per slice
    int wg_parts = 0;                        // ... split-K partials that reach memory (t_wgrad_g5: one per four slices, summed in the block)
    size_t count = 0;
    long step = 0;
    float *g = nullptr, *m = nullptr, *v = nullptr;
    uint8_t* kind = nullptr;
    float *X0 = nullptr, *col0 = nullptr, *wpad = nullptr, *gpad = nullptr;
    float *Y = nullptr, *A = nullptr;        // [L][M][256]
    float *G = nullptr, *DS = nullptr, *DT = nullptr, *dY = nullptr;
    float *mean = nullptr, *istd = nullptr;  // [L][256]
    float* sums = nullptr;                   // [2][256]
    double* part = nullptr;                  // [R][2*NG][256]
    float* wpart = nullptr;                  // split-K partials [nz][KC][256]
    uint16_t *ap[3] = {nullptr, nullptr, nullptr}, *dyp[2] = {nullptr, nullptr};          // bf16 parts of activations / gradients
    uint16_t* af[2] = {nullptr, nullptr};    // fp16 pair (hi, lo) of the newest post-activation: the next forward conv's operand
    uint16_t *wpf[3] = {nullptr, nullptr, nullptr}, *wpb[2] = {nullptr, nullptr};        // packed kernels: forward / backward-data view
    float *pv0 = nullptr, *dpv = nullptr, *hstat = nullptr, *fpi = nullptr, *fv = nullptr, *h1 = nullptr, *vout = nullptr,
          *prob = nullptr, *lossb = nullptr, *hpart = nullptr, *cpart = nullptr, *loss = nullptr;
    int* cur = nullptr;    // device: {minibatch offset in perm, Adam step count, this rank's offset inside the minibatch}
    // data-parallel step (azr_nn_train_dp): this rank's shard of every minibatch; sums that span the batch are all-reduced
    int world = 1, rank = 0;
    azr_allreduce_fn ar = nullptr;
    void* ar_ctx = nullptr;
    bool native = false;       // the sums go through the handle's own RCCL communicator, in stream order (azr_dp_init)
    double* red = nullptr;     // [2 * NG][256] reduced BN partials
    double* hsum = nullptr;    // [6] head BN sums
    float* lr = nullptr;   // device: this step's bias-corrected learning rate
    // the weight-gradient branch of the backward pass (t_wgrad_g5 + t_sum_slices of every layer) hangs off the gradient chain: nothing
    // but Adam waits for it.  At small batches (<= 128 records: a rank's share of a data-parallel minibatch) it runs on a second,
    // low-priority stream beside the chain's kernels, which leave most of the chip idle there: 4.22 -> 3.87 ms per step at 64 records.
    hipStream_t side = nullptr;
    hipEvent_t ev_conv[2] = {nullptr, nullptr};   // main -> side: the backward-data conv of a layer has left its dY parts (by layer parity)
    hipEvent_t ev_wg[2] = {nullptr, nullptr};     // side -> main: that layer's weight gradient has read them
    uint16_t* dyp2[2] = {nullptr, nullptr};       // the dY parts of odd layers (a second set: the chain runs ahead of the branch)
    uint8_t* rec = nullptr;
    size_t rec_cap = 0;
    int* perm = nullptr;
    size_t perm_cap = 0;
    uint8_t* in88 = nullptr;
    float *pit = nullptr, *zt = nullptr;
    std::vector<void*> allocs;
};

TrainCtx* ctx_of(azr_engine* h) { return static_cast<TrainCtx*>(h->train); }

template <typename T>
int dalloc(azr_engine* h, TrainCtx* c, T** p, size_t n)
{
    HIPCHK(h, hipMalloc((void**)p, n * sizeof(T)));
    c->allocs.push_back(*p);
    return AZR_OK;
}

void ctx_free(TrainCtx* c)
{
    if (!c) return;
    for (int q = 0; q < 2; q++) { if (c->ev_conv[q]) hipEventDestroy(c->ev_conv[q]); if (c->ev_wg[q]) hipEventDestroy(c->ev_wg[q]); }
    if (c->side) hipStreamDestroy(c->side);
    for (void* p : c->allocs) hipFree(p);
    if (c->rec) hipFree(c->rec);
    if (c->perm) hipFree(c->perm);
    delete c;
}

#define TRY(x)                 \
    do {                       \
        int rc__ = (x);        \
        if (rc__) return rc__; \
    } while (0)

// the weight gradient of one tower conv: 64 one-wave blocks (16 ci tiles x 4 co quarters) per slice of whole boards -> c->wpart[slice]
static void launch_wgrad(TrainCtx* c, hipStream_t st, const Parts& apP, const Parts& dyP, int M)
{
#ifdef AZR_TEST_HOOKS
    if (!g_wgrad_g5) {
        hipLaunchKernelGGL(t_wgrad_rs, dim3(64 * c->wg_slices), dim3(64), Wg::LDS_BYTES, st, apP, dyP, c->wpart, M, c->wg_slices, c->wg_rows);
        return;
    }
#endif
    hipLaunchKernelGGL(t_wgrad_g5<WG5_NC>, dim3(Wg5<WG5_NC>::BLOCKS_PER_SLICE * c->wg_parts), dim3(256), Wg5<WG5_NC>::LDS_BLOCK, st, apP, dyP, c->wpart, M / NPOS,
                       c->wg_slices, c->wg_bps);
}

int ctx_ensure(azr_engine* h, int BS)
{
    TrainCtx* c = ctx_of(h);
    if (c && c->BS == BS) return AZR_OK;
    // tuning switch, read when a training context is (re)built — never in the step path
#ifdef AZR_TEST_HOOKS
    g_gemm_bf16x3 = !(hook_env("AZR_TRAIN_GEMM") && strcmp(hook_env("AZR_TRAIN_GEMM"), "f32") == 0);
    g_fuse_bwd = hook_env_int("AZR_TRAIN_FUSE", 1) != 0;
    g_fwd_f16 = !(hook_env("AZR_TRAIN_FWD") && strcmp(hook_env("AZR_TRAIN_FWD"), "bf16") == 0);
    g_fuse_apply = hook_env_int("AZR_TRAIN_FUSE_APPLY", 1) != 0;
    g_conv_q = hook_env_int("AZR_TRAIN_CONVQ", 1) != 0;
    g_wgrad_g5 = !(hook_env("AZR_TRAIN_WGRAD") && strcmp(hook_env("AZR_TRAIN_WGRAD"), "rs") == 0);
#endif
#ifdef AZR_TEST_HOOKS
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(t_wgrad_rs), hipFuncAttributeMaxDynamicSharedMemorySize, Wg::LDS_BYTES));
#endif
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(t_wgrad_g5<WG5_NC>), hipFuncAttributeMaxDynamicSharedMemorySize, Wg5<WG5_NC>::LDS_BLOCK));
    // a different batch size rebuilds the activation slabs but keeps the optimiser state
    std::vector<float> keep_m, keep_v;
    long keep_step = 0;
    if (c) {
        keep_m.resize(c->count); keep_v.resize(c->count);
        HIPCHK(h, hipMemcpy(keep_m.data(), c->m, c->count * 4, hipMemcpyDeviceToHost));
        HIPCHK(h, hipMemcpy(keep_v.data(), c->v, c->count * 4, hipMemcpyDeviceToHost));
        keep_step = c->step;
        ctx_free(c);
        h->train = nullptr;
    }
    c = new TrainCtx();
    h->train = c;
    const int B = h->net.blocks;
    c->BS = BS; c->blocks = B; c->M = BS * NPOS; c->L = 2 * B + 1;
    c->R = (c->M + RB - 1) / RB;
    c->count = net_param_count(B);
    // split-K of the weight-gradient GEMM (36 output tiles of 128 x 128): as many slices as keep <= 2 blocks per CU
    c->nz = std::max(1, std::min(512 / ((KC / GT) * (NF / GT)), (c->M + 1023) / 1024));
    c->kchunk = (((c->M + c->nz - 1) / c->nz) + K3 - 1) / K3 * K3;
    c->step = keep_step;
    const size_t M = c->M, act = M * NF;
    TRY(dalloc(h, c, &c->g, c->count)); TRY(dalloc(h, c, &c->m, c->count)); TRY(dalloc(h, c, &c->v, c->count));
    TRY(dalloc(h, c, &c->kind, c->count));
    TRY(dalloc(h, c, &c->X0, M * SIN)); TRY(dalloc(h, c, &c->col0, M * KS));
    TRY(dalloc(h, c, &c->wpad, (size_t)KS * NF)); TRY(dalloc(h, c, &c->gpad, (size_t)KS * NF));
    TRY(dalloc(h, c, &c->Y, act * c->L)); TRY(dalloc(h, c, &c->A, act * c->L));
    TRY(dalloc(h, c, &c->G, act)); TRY(dalloc(h, c, &c->DS, act)); TRY(dalloc(h, c, &c->DT, act)); TRY(dalloc(h, c, &c->dY, act));
    TRY(dalloc(h, c, &c->mean, (size_t)c->L * NF)); TRY(dalloc(h, c, &c->istd, (size_t)c->L * NF));
    TRY(dalloc(h, c, &c->sums, (size_t)2 * NF));
    TRY(dalloc(h, c, &c->part, (size_t)c->R * 2 * NG * NF));
    // t_wgrad_rs: slices of 16 j boards (16 boards = 672 rows = 21 k-steps), about 256 blocks = 16 ci tiles x slices
    {
        // (small batches — a rank's share of a data-parallel minibatch: 8-board slices, so that 64 records are 512 one-wave blocks)
        int bps = (BS <= 128 && BS % 8 == 0) ? 8 : 16 * std::max(1, BS / 256);
        // t_wgrad_g5: whole groups of 5 boards, as many slices as make the chip's 1024 SIMDs one block each
        if (g_wgrad_g5) {
            using W = Wg5<WG5_NC>;
            const int want = 1024 / W::BLOCKS_PER_SLICE;
            bps = std::max(W::GB, ((BS + want - 1) / want + W::GB - 1) / W::GB * W::GB);
        }
        c->wg_slices = (BS + bps - 1) / bps;
        c->wg_rows = bps * NPOS;
        c->wg_bps = bps;
        c->wg_parts = g_wgrad_g5 ? (c->wg_slices + 3) / 4 : c->wg_slices;
    }
    TRY(dalloc(h, c, &c->wpart, (size_t)std::max(c->nz, c->wg_slices) * KC * NF));
    // bf16 parts of the post-activations: the two leading parts are kept PER LAYER (the weight-gradient GEMM of the backward
    // pass wants exactly them: no second split), the third one only until the next forward conv has read it
    for (int q = 0; q < 3; q++) { TRY(dalloc(h, c, &c->ap[q], q < 2 ? act * c->L : act)); TRY(dalloc(h, c, &c->wpf[q], (size_t)2 * B * KC * NF)); }
    for (int q = 0; q < 2; q++) TRY(dalloc(h, c, &c->wpb[q], (size_t)2 * B * KC * NF));
    for (int q = 0; q < 2; q++) { TRY(dalloc(h, c, &c->dyp[q], act)); TRY(dalloc(h, c, &c->dyp2[q], act)); }
    {
        int lo = 0, hi = 0;
        HIPCHK(h, hipDeviceGetStreamPriorityRange(&lo, &hi));   // (lo = the numerically greatest = least urgent)
        HIPCHK(h, hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, lo));
        for (int q = 0; q < 2; q++) {
            HIPCHK(h, hipEventCreateWithFlags(&c->ev_conv[q], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&c->ev_wg[q], hipEventDisableTiming));
        }
    }
    for (int q = 0; q < 2; q++) TRY(dalloc(h, c, &c->af[q], act));
    TRY(dalloc(h, c, &c->pv0, M * 4)); TRY(dalloc(h, c, &c->dpv, M * 4)); TRY(dalloc(h, c, &c->hstat, (size_t)8));
    TRY(dalloc(h, c, &c->fpi, (size_t)BS * 84)); TRY(dalloc(h, c, &c->fv, (size_t)BS * 42)); TRY(dalloc(h, c, &c->h1, (size_t)BS * 256));
    TRY(dalloc(h, c, &c->vout, (size_t)BS)); TRY(dalloc(h, c, &c->prob, (size_t)BS * 43)); TRY(dalloc(h, c, &c->lossb, (size_t)BS * 2));
    TRY(dalloc(h, c, &c->hpart, (size_t)BS * HP_FLOATS)); TRY(dalloc(h, c, &c->cpart, (size_t)c->R * 3 * NF));
    TRY(dalloc(h, c, &c->loss, (size_t)4));
    TRY(dalloc(h, c, &c->cur, (size_t)4)); TRY(dalloc(h, c, &c->lr, (size_t)1));
    TRY(dalloc(h, c, &c->red, (size_t)2 * NG * NF)); TRY(dalloc(h, c, &c->hsum, (size_t)8));
    {
        const int init[4] = {0, (int)keep_step, 0, 0};
        HIPCHK(h, hipMemcpy(c->cur, init, sizeof init, hipMemcpyHostToDevice));
    }
    TRY(dalloc(h, c, &c->in88, (size_t)BS * 88)); TRY(dalloc(h, c, &c->pit, (size_t)BS * 43)); TRY(dalloc(h, c, &c->zt, (size_t)BS));
    HIPCHK(h, hipMemset(c->dpv, 0, M * 4 * sizeof(float)));
    HIPCHK(h, hipMemset(c->g, 0, c->count * 4));
    if (keep_m.empty()) {
        HIPCHK(h, hipMemset(c->m, 0, c->count * 4));
        HIPCHK(h, hipMemset(c->v, 0, c->count * 4));
    } else {
        HIPCHK(h, hipMemcpy(c->m, keep_m.data(), c->count * 4, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(c->v, keep_v.data(), c->count * 4, hipMemcpyHostToDevice));
    }
    // parameter kinds: 1 = kernel (L2-regularised), 2 = BN gamma/beta and dense biases, 0 = BN moving statistics
    std::vector<uint8_t> kind(c->count, 0);
    auto fill = [&](size_t off, size_t n, uint8_t k) { std::fill(kind.begin() + off, kind.begin() + off + n, k); };
    fill(0, OFF_STEM_BN, 1);
    fill(OFF_STEM_BN, 14, 2);
    for (int l = 0; l < 2 * B; l++) {
        const size_t o = OFF_BLOCK0 + (size_t)l * LAYER;
        fill(o, (size_t)9 * NF * NF, 1);
        fill(o + (size_t)9 * NF * NF, 2 * NF, 2);
    }
    const size_t hh = OFF_BLOCK0 + (size_t)2 * B * LAYER;
    if (hh + HEAD_FLOATS != c->count) { h->err = "azr_nn_train: AZRW layout mismatch"; return AZR_E_STATE; }
    fill(hh + H_PI_W, 512, 1); fill(hh + H_PI_BN, 4, 2);
    fill(hh + H_PD_W, 3612, 1); fill(hh + H_PD_B, 43, 2);
    fill(hh + H_V_W, 256, 1); fill(hh + H_V_BN, 2, 2);
    fill(hh + H_V1_W, 10752, 1); fill(hh + H_V1_B, 256, 2);
    fill(hh + H_V2_W, 256, 1); fill(hh + H_V2_B, 1, 2);
    HIPCHK(h, hipMemcpy(c->kind, kind.data(), c->count, hipMemcpyHostToDevice));
    return AZR_OK;
}

template <bool A_MCONTIG, bool B_KCONTIG, int BM = 128, int AMODE = 0, int BMODE = 0>
void gemm(hipStream_t st, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K, int nz = 1,
          int kchunk = 0, size_t strideCz = 0)
{
    if (nz == 1) kchunk = K;
    hipLaunchKernelGGL((t_gemm<A_MCONTIG, B_KCONTIG, BM, AMODE, BMODE>), dim3((N + GT - 1) / GT, (M + BM - 1) / BM, nz), dim3(256), 0, st, A, lda, B, ldb,
                       C, ldc, M, N, K, kchunk, strideCz);
}

inline dim3 grid1(size_t n, int bs) { return dim3((unsigned)((n + bs - 1) / bs)); }

// RCCL, bound at run time: dlopen("librccl.so.1") returns the copy a host process has already loaded (PyTorch-ROCm ships one under
// the same soname) or /opt/rocm's — one RCCL, one HIP runtime per process, and libazr_hip.so itself carries no link dependency.
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static void rccl_bind(RcclApi& api);
RcclApi* rccl_api()   // bound once, whichever host thread asks first (the host CLI runs one thread per GPU)
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] { rccl_bind(api); });
    return &api;
}
static void rccl_bind(RcclApi& api)
{
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("RCCL not found: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(dlsym(api.lib, "ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GetErrorString) {
        api.err = "RCCL: a symbol is missing from the loaded library";
        api.lib = nullptr;
    }
}

// one all-reduce (sum, in place) of a device buffer over the ranks of a data-parallel step.  Native (azr_dp_init): ncclAllReduce on
// the engine's own stream — stream-ordered, the host never waits (86 of them per step at B = 20: with a host hand-over each they
// cost more than the step's kernels).  Otherwise through the caller's callback (a torch.distributed rehearsal over gloo, or any other
// transport): the engine's stream is drained first, the callback returns when the result is in place.
int dp_allreduce(azr_engine* h, TrainCtx* c, void* dev, size_t count, int dtype)
{
    if (c->native) {
        RcclApi* R = rccl_api();
        const ncclResult_t rc = R->AllReduce(dev, dev, count, dtype ? ncclDouble : ncclFloat, ncclSum, static_cast<ncclComm_t>(h->dp_comm), h->stream);
        if (rc != ncclSuccess) { h->err = std::string("ncclAllReduce: ") + R->GetErrorString(rc); return AZR_E_HIP; }
        return AZR_OK;
    }
    if (!c->ar) return AZR_OK;
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int rc = c->ar(c->ar_ctx, dev, count, dtype);
    if (rc) { h->err = "azr_nn_train_dp: the all-reduce callback failed with code " + std::to_string(rc); return AZR_E_STATE; }
    return AZR_OK;
}

// one optimiser step on the minibatch already gathered into c->in88 / pit / zt
int train_step(azr_engine* h, TrainCtx* c, float* d_acc)
{
    hipStream_t st = h->stream;
    const int M = c->M, B = c->blocks, R = c->R, BS = c->BS;
    // data-parallel: BS / M are this rank's shard, BSg / Mg the whole minibatch every statistic and mean refers to
    const int W = c->world, BSg = BS * W, Mg = M * W;
    const bool dp = c->ar != nullptr || c->native;   // the data-parallel code path (a callback was supplied, or the handle has a communicator), also with one rank
    const float gscale = 1.0f / (float)W;
    // sums over the batch: [local partials -> one slab] -> all-reduce over the ranks -> the finalize kernel reads the slab
    auto reduce_parts = [&](int K) -> int {
        hipLaunchKernelGGL(t_parts_sum, dim3(K), dim3(256), 0, st, c->part, R, K, c->red);
        return dp_allreduce(h, c, c->red, (size_t)K * NF, 1);
    };
    float* w = h->net.d_flat;
    float* g = c->g;
    const size_t act = (size_t)M * NF, wn_ = (size_t)KC * NF;
    const size_t hh = OFF_BLOCK0 + (size_t)2 * B * LAYER;
    float* hp = w + hh;
    float* gh = g + hh;
    auto Wl = [&](int l) { return w + OFF_BLOCK0 + (size_t)(l - 1) * LAYER; };       // conv layer l >= 1: kernel, then bn
    auto Gl = [&](int l) { return g + OFF_BLOCK0 + (size_t)(l - 1) * LAYER; };
    auto Yl = [&](int l) { return c->Y + act * l; };
    auto Al = [&](int l) { return c->A + act * l; };
    const unsigned g4 = (unsigned)((act / 4 + 255) / 256);
    const bool sb = g_gemm_bf16x3 && M % K3 == 0;  // split-bf16 conv GEMMs (the stem, K = 144, stays on the fp32 MFMA)
    const bool f16 = sb && g_fwd_f16;              // forward conv: fp16 pairs, 3 passes
    // small batches (a rank's share of a data-parallel minibatch): one board x 64 channels per block (t_conv_q) instead of 2 boards x 256
    const bool convq = f16 && g_fuse_bwd && g_fuse_apply && g_conv_q && BS <= 128;
    auto Wpf = [&](int l) { const size_t o = (size_t)(l - 1) * KC * NF; return Parts{{c->wpf[0] + o, c->wpf[1] + o, c->wpf[2] + o}}; };
    auto Wpb = [&](int l) { const size_t o = (size_t)(l - 1) * KC * NF; return Parts{{c->wpb[0] + o, c->wpb[1] + o, nullptr}}; };
    if (sb) {
        const dim3 pg((unsigned)((wn_ / 8 + 255) / 256), 2 * B);
        hipLaunchKernelGGL((t_pack_w<3>), pg, dim3(256), 0, st, w, 0, c->wpf[0], c->wpf[1], c->wpf[2], f16 ? FWD_WSCALE : 0.0f, c->cur + 3);
        hipLaunchKernelGGL((t_pack_w<2>), pg, dim3(256), 0, st, w, 1, c->wpb[0], c->wpb[1], (uint16_t*)nullptr);
    }

    // ---------------- forward, training mode
    hipLaunchKernelGGL(t_planes, grid1((size_t)M * SIN, 256), dim3(256), 0, st, c->in88, M, c->X0);
    hipLaunchKernelGGL(t_stem_pad, grid1((size_t)KS * NF, 256), dim3(256), 0, st, w, c->wpad);
    hipLaunchKernelGGL((t_im2col<SIN>), grid1((size_t)M * 9, 4), dim3(256), 0, st, c->X0, c->col0, M);
    gemm<false, false, 64>(st, c->col0, KS, c->wpad, NF, Yl(0), NF, M, NF, KS);
    hipLaunchKernelGGL((t_bn_stats<true>), dim3(R), dim3(1024), 0, st, Yl(0), M, c->part);
    if (dp) TRY(reduce_parts(2 * NG));
    hipLaunchKernelGGL((t_bn_finalize<true>), dim3(NG), dim3(1024), 0, st, dp ? c->red : c->part, dp ? 1 : R, (double)BSg * 6 * NF, c->mean, c->istd,
                       w + OFF_STEM_BN);
    uint16_t* const nil16 = nullptr;
    // (each layer's normalise kernel also writes the three bf16 parts of its output: the next conv's A operand)
    auto Ap = [&](int l) { return Parts{{c->ap[0] + act * l, c->ap[1] + act * l, c->ap[2]}}; };   // parts of the post-activation of layer l
    hipLaunchKernelGGL((t_bn_apply<true>), dim3(g4), dim3(256), 0, st, Yl(0), c->mean, c->istd, w + OFF_STEM_BN, (const float*)nullptr, Al(0), M,
                       sb ? c->ap[0] : nil16, c->ap[1], f16 ? nil16 : c->ap[2], f16 ? c->af[0] : nil16, c->af[1]);
    // Fused mode (fp16 forward + epilogue statistics + staging-path normalise): layer l's normalise step is not a kernel of its own
    // — the forward conv of layer l + 1 computes A_l = relu(BN(Y_l) (+ S)) while it stages its operand and writes A_l and its
    // bf16 parts out; only the stem (row-wise BN) and the last layer (the heads read it) keep t_bn_apply.
    const bool fap = f16 && g_fuse_bwd && g_fuse_apply;
    for (int l = 1; l < c->L; l++) {
        float* bn = Wl(l) + (size_t)9 * NF * NF;
        const float* S = (l % 2 == 0) ? Al(l - 2) : nullptr;  // second conv of a block adds the block input
        int fwd_parts = 0;
        if (sb) {  // conv = implicit im2col x W in split precision (fp32-exact products)
            if (fap && l >= 2) {   // operand computed on the way in (layer l - 1's normalise step), statistics of the output on the way out
                const int m = l - 1;
                const BwdFuse bf{nullptr, nullptr, nullptr, nullptr, nullptr, c->part};
                const ProFuse pf{Yl(m), (m % 2 == 0) ? (const float*)Al(m - 2) : (const float*)nullptr, nullptr, c->mean + m * NF, c->istd + m * NF,
                                 Wl(m) + (size_t)9 * NF * NF, nullptr, 0.0f, Al(m), const_cast<uint16_t*>(Ap(m).p[0]), const_cast<uint16_t*>(Ap(m).p[1])};
                if (convq) {
                    fwd_parts = BS;
                    hipLaunchKernelGGL((t_conv_q<1, 2, true, 1>), dim3(4 * BS), dim3(256), 0, st, Parts{{nullptr, nullptr, nullptr}}, Wpf(l), Yl(l), BS, bf, 1.0f / FWD_WSCALE, pf);
                } else {
                    fwd_parts = (BS + 1) / 2;
                    hipLaunchKernelGGL((t_conv_rs<1, 2, 2, true, 1>), dim3(fwd_parts), dim3(256), 0, st, Parts{{nullptr, nullptr, nullptr}}, Wpf(l), Yl(l), BS, bf, 1.0f / FWD_WSCALE, pf);
                }
            } else if (convq) {    // (layer 1: the stem's normalise kernel wrote the fp16 pair)
                fwd_parts = BS;
                hipLaunchKernelGGL((t_conv_q<1, 2, true, 0>), dim3(4 * BS), dim3(256), 0, st, Parts{{c->af[0], c->af[1], nullptr}}, Wpf(l), Yl(l), BS,
                                   BwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, c->part}, 1.0f / FWD_WSCALE, ProFuse{});
            } else if (f16 && g_fuse_bwd) {   // + the batch statistics of the output, per block of 2 boards
                fwd_parts = (BS + 1) / 2;
                hipLaunchKernelGGL((t_conv_rs<1, 2, 2, true>), dim3(fwd_parts), dim3(256), 0, st, Parts{{c->af[0], c->af[1], nullptr}}, Wpf(l), Yl(l), BS,
                                   BwdFuse{nullptr, nullptr, nullptr, nullptr, nullptr, c->part}, 1.0f / FWD_WSCALE, ProFuse{});
            }
#ifdef AZR_TEST_HOOKS   // older formulations of the same conv (AZR_TRAIN_FUSE=0, AZR_TRAIN_FWD=bf16): compiled into libazr_hip_test.so only
            else if (f16) hipLaunchKernelGGL((t_conv_rs<1, 2, 0, true>), dim3((BS + 1) / 2), dim3(256), 0, st, Parts{{c->af[0], c->af[1], nullptr}}, Wpf(l), Yl(l), BS,
                                        BwdFuse{}, 1.0f / FWD_WSCALE, ProFuse{});
            else hipLaunchKernelGGL((t_conv_rs<1, 3>), dim3((BS + 1) / 2), dim3(256), 0, st, Ap(l - 1), Wpf(l), Yl(l), BS, BwdFuse{}, 1.0f, ProFuse{});
#endif
        } else gemm<false, false, 64, 1, 0>(st, Al(l - 1), KC, Wl(l), NF, Yl(l), NF, M, NF, KC);
        const int Rf = fwd_parts ? fwd_parts : R;
        if (!fwd_parts) hipLaunchKernelGGL((t_bn_stats<false>), dim3(R), dim3(1024), 0, st, Yl(l), M, c->part);
        if (dp) {
            hipLaunchKernelGGL(t_parts_sum, dim3(2), dim3(256), 0, st, c->part, Rf, 2, c->red);
            TRY(dp_allreduce(h, c, c->red, (size_t)2 * NF, 1));
        }
        hipLaunchKernelGGL((t_bn_finalize<false>), dim3(8), dim3(1024), 0, st, dp ? c->red : c->part, dp ? 1 : Rf, (double)Mg, c->mean + l * NF,
                           c->istd + l * NF, bn);
        if (fap && l + 1 < c->L) continue;   // the next conv normalises this layer's output itself
        hipLaunchKernelGGL((t_bn_apply<false>), dim3(g4), dim3(256), 0, st, Yl(l), c->mean + l * NF, c->istd + l * NF, bn, S, Al(l), M,
                           (sb && l + 1 < c->L) ? const_cast<uint16_t*>(Ap(l).p[0]) : nil16, const_cast<uint16_t*>(Ap(l).p[1]), f16 ? nil16 : c->ap[2],
                           (f16 && l + 1 < c->L) ? c->af[0] : nil16, c->af[1]);
    }
    const float* H = Al(c->L - 1);
    hipLaunchKernelGGL(t_head_conv, grid1((size_t)M, 4), dim3(256), 0, st, H, hp, c->pv0, M);
    hipLaunchKernelGGL(t_head_bn_sums, dim3(1), dim3(1024), 0, st, c->pv0, M, c->hsum);
    if (dp) TRY(dp_allreduce(h, c, c->hsum, 6, 1));
    hipLaunchKernelGGL(t_head_bn_stats, dim3(1), dim3(64), 0, st, (const double*)c->hsum, (double)Mg, hp, c->hstat);
    hipLaunchKernelGGL(t_head_fwd, dim3(BS), dim3(256), 0, st, c->pv0, hp, c->hstat, c->pit, c->zt, c->fpi, c->fv, c->h1, c->vout, c->prob, c->lossb);
    hipLaunchKernelGGL(t_loss, dim3(1), dim3(1), 0, st, c->lossb, BS, BSg, c->loss);
    if (dp) TRY(dp_allreduce(h, c, c->loss, 2, 0));
    hipLaunchKernelGGL(t_loss_acc, dim3(1), dim3(1), 0, st, (const float*)c->loss, d_acc);

    // ---------------- backward
    hipLaunchKernelGGL(t_head_bwd, dim3(BS), dim3(256), 0, st, hp, c->pit, c->zt, c->fpi, c->fv, c->h1, c->vout, c->prob, BSg, c->dpv, c->hpart);
    hipLaunchKernelGGL(t_head_reduce, grid1(HP_FLOATS, 256), dim3(256), 0, st, c->hpart, BS, gh);
    hipLaunchKernelGGL(t_head_bn_bwd_sums, dim3(1), dim3(1024), 0, st, c->pv0, c->hstat, M, (const float*)c->dpv, c->hsum);
    if (dp) TRY(dp_allreduce(h, c, c->hsum, 6, 1));
    hipLaunchKernelGGL(t_head_bn_bwd, dim3((M + 1023) / 1024), dim3(1024), 0, st, c->pv0, hp, c->hstat, M, (const double*)c->hsum, (float)Mg, gscale, c->dpv, gh);
    hipLaunchKernelGGL(t_head_conv_bwd, dim3(R), dim3(256), 0, st, H, c->dpv, hp, M, c->G, c->cpart);
    hipLaunchKernelGGL(t_head_conv_bwd_finalize, dim3(1), dim3(256), 0, st, c->cpart, R, gh);
    const float invM = 1.0f / (float)Mg;
    const size_t wn = wn_;
    // (t_conv_rs<2, 2, 1>, the backward-data conv of layer l, leaves stage 1 of layer l - 1's batch-norm backward behind: its
    //  block partials are then already in c->part, `fused_parts` blocks of them)
    int fused_parts = 0, side_used = 0;
    const bool fuse = sb && g_fuse_bwd;
    float* pending_sum = nullptr;   // the slice sum of the layer above is launched together with this layer's BN-backward stage 2 (t_sum_slices_fin)
    for (int l = c->L - 1; l >= 1; l--) {
        // gradient w.r.t. this layer's post-activation output: G for the second conv of a block, DT for the first
        const bool second = (l % 2 == 0);
        const float* dOut = second ? c->G : c->DT;
        float* bn = Wl(l) + wn;
        float* gbn = Gl(l) + wn;
        const int Rl = fused_parts ? fused_parts : R;
        if (!fused_parts)
            hipLaunchKernelGGL((t_bn_bwd_stats<false>), dim3(R), dim3(1024), 0, st, dOut, Al(l), Yl(l), c->mean + l * NF, c->istd + l * NF, M, c->part);
        if (dp) {
            hipLaunchKernelGGL(t_parts_sum, dim3(2), dim3(256), 0, st, c->part, Rl, 2, c->red);
            TRY(dp_allreduce(h, c, c->red, (size_t)2 * NF, 1));
        }
        if (pending_sum && !dp && fused_parts) {
            hipLaunchKernelGGL(t_sum_slices_fin, dim3(8 + (unsigned)((wn + 1023) / 1024)), dim3(1024), 0, st, c->wpart, c->wg_parts, wn, pending_sum, c->part,
                               Rl, gbn, c->sums, gscale);
            pending_sum = nullptr;
        } else {
            if (pending_sum) { hipLaunchKernelGGL(t_sum_slices, grid1(wn, 256), dim3(256), 0, st, c->wpart, c->wg_parts, wn, pending_sum); pending_sum = nullptr; }
            hipLaunchKernelGGL((t_bn_bwd_finalize<false>), dim3(8), dim3(1024), 0, st, dp ? c->red : c->part, dp ? 1 : Rl, gbn, c->sums, gscale);
        }
        float* dIn = second ? c->DT : c->G;
        const Parts apP{{Ap(l - 1).p[0], Ap(l - 1).p[1], nullptr}};
        if (fuse && g_fuse_apply) {
            // dY is computed in the backward-data conv's staging path (t_bn_bwd_apply's arithmetic; its two bf16 parts and, where the
            // layer closes a block, dz = the shortcut gradient DS are written out on the way), so that conv runs FIRST and the
            // weight-gradient GEMM reads the parts it left behind — on the side stream (TrainCtx::side), from the set of its layer parity
            const bool beside = convq && !c->native;   // (in-stream RCCL collectives and cross-stream edges do not mix: 16 ms per step measured)
            const int q = beside ? (l & 1) : 0;
            uint16_t* const* dq = q ? c->dyp2 : c->dyp;
            const Parts dyP{{dq[0], dq[1], nullptr}};
            if (beside && l + 2 <= c->L - 1) HIPCHK(h, hipStreamWaitEvent(st, c->ev_wg[q], 0));   // layer l + 2's weight gradient has read this set
            const ProFuse pf{dOut, Al(l), Yl(l), c->mean + l * NF, c->istd + l * NF, bn, c->sums, invM, second ? c->DS : (float*)nullptr, dq[0], dq[1]};
            fused_parts = 0;
            const Parts none{{nullptr, nullptr, nullptr}};
            if (l >= 2) {
                const BwdFuse bf{second ? (const float*)nullptr : (const float*)c->DS, Al(l - 1), Yl(l - 1), c->mean + (l - 1) * NF, c->istd + (l - 1) * NF, c->part};
                if (convq) {
                    fused_parts = BS;
                    hipLaunchKernelGGL((t_conv_q<2, 1, false, 2>), dim3(4 * BS), dim3(256), 0, st, none, Wpb(l), dIn, BS, bf, 1.0f, pf);
                } else {
                    fused_parts = (BS + 1) / 2;
                    hipLaunchKernelGGL((t_conv_rs<2, 2, 1, false, 2>), dim3(fused_parts), dim3(256), 0, st, none, Wpb(l), dIn, BS, bf, 1.0f, pf);
                }
            } else {
                if (convq) hipLaunchKernelGGL((t_conv_q<2, 0, false, 2>), dim3(4 * BS), dim3(256), 0, st, none, Wpb(l), dIn, BS, BwdFuse{}, 1.0f, pf);
                else hipLaunchKernelGGL((t_conv_rs<2, 2, 0, false, 2>), dim3((BS + 1) / 2), dim3(256), 0, st, none, Wpb(l), dIn, BS, BwdFuse{}, 1.0f, pf);
                if (!second) hipLaunchKernelGGL(t_add, dim3(g4), dim3(256), 0, st, dIn, c->DS, act / 4);
            }
            if (!beside) {  // large batches: the chain's kernels and the weight gradient each fill the register files on their own (368 and 280
                            // VGPRs: no SIMD holds a wave of both) and nothing overlaps: one stream.  (Measured at batch 512: the branch on the
                            // side stream 11.16 ms per step, only its slice sums there 11.28, one stream 11.1 — a cross-stream edge costs the
                            // chain a barrier packet per layer, about what hiding the 10-us slice sum saves.)
                launch_wgrad(c, st, apP, dyP, M);
                pending_sum = Gl(l);   // summed by the next layer's launch (t_sum_slices_fin), or behind the loop
                continue;
            }
            // small batches (a rank's share of a data-parallel minibatch): the chain's kernels leave most of the chip idle, the branch runs beside them
            HIPCHK(h, hipEventRecord(c->ev_conv[q], st));
            HIPCHK(h, hipStreamWaitEvent(c->side, c->ev_conv[q], 0));
            launch_wgrad(c, c->side, apP, dyP, M);
            hipLaunchKernelGGL(t_sum_slices, grid1(wn, 256), dim3(256), 0, c->side, c->wpart, c->wg_parts, wn, Gl(l));
            HIPCHK(h, hipEventRecord(c->ev_wg[q], c->side));
            side_used |= 1 << q;
            continue;
        }
        const Parts dyP{{c->dyp[0], c->dyp[1], nullptr}};
        // (the split-bf16 kernels read the two parts of dY; its fp32 image is only written for the fp32-MFMA GEMMs)
        hipLaunchKernelGGL((t_bn_bwd_apply<false>), dim3(g4), dim3(256), 0, st, dOut, Al(l), Yl(l), c->mean + l * NF, c->istd + l * NF, bn, c->sums,
                           invM, sb ? (float*)nullptr : c->dY, second ? c->DS : (float*)nullptr, M, sb ? c->dyp[0] : nil16, c->dyp[1]);
        // dW = col(input)^T x dY  (implicit im2col, split-K over the M rows)
        // (In the product library this point is reached only by batches whose row count is no multiple of the 32-deep k-tile: the fp32-MFMA
        //  GEMMs.  The split-bf16 kernels WITHOUT the staging-path fusions are older formulations, compiled into libazr_hip_test.so only.)
#ifdef AZR_TEST_HOOKS
        if (sb) {
            launch_wgrad(c, st, apP, dyP, M);
        } else
#endif
        gemm<true, false, 128, 1, 0>(st, Al(l - 1), KC, c->dY, NF, c->wpart, NF, KC, NF, M, c->nz, c->kchunk, wn);
        hipLaunchKernelGGL(t_sum_slices, grid1(wn, 256), dim3(256), 0, st, c->wpart, sb ? c->wg_parts : c->nz, wn, Gl(l));
        // d(input) = transposed conv of dY with W: the same implicit GEMM with negated taps and W read as [tap][co] x [ci]
        fused_parts = 0;
#ifdef AZR_TEST_HOOKS
        if (fuse && l >= 2) {   // + the shortcut gradient (first conv of a block), + stage 1 of layer l - 1's BN backward
            fused_parts = (BS + 1) / 2;
            hipLaunchKernelGGL((t_conv_rs<2, 2, 1>), dim3(fused_parts), dim3(256), 0, st, dyP, Wpb(l), dIn, BS,
                               BwdFuse{second ? (const float*)nullptr : (const float*)c->DS, Al(l - 1), Yl(l - 1), c->mean + (l - 1) * NF,
                                       c->istd + (l - 1) * NF, c->part}, 1.0f, ProFuse{});
            continue;
        }
        if (sb) hipLaunchKernelGGL((t_conv_rs<2, 2>), dim3((BS + 1) / 2), dim3(256), 0, st, dyP, Wpb(l), dIn, BS, BwdFuse{}, 1.0f, ProFuse{});
        else
#endif
        gemm<false, true, 64, 2, 3>(st, c->dY, KC, Wl(l), NF, dIn, NF, M, NF, KC);
        if (!second) hipLaunchKernelGGL(t_add, dim3(g4), dim3(256), 0, st, dIn, c->DS, act / 4);  // + shortcut gradient
    }
    if (pending_sum) hipLaunchKernelGGL(t_sum_slices, grid1(wn, 256), dim3(256), 0, st, c->wpart, c->wg_parts, wn, pending_sum);
    // the weight-gradient branch joins: the stem below reuses its split-K buffer, and the gradient vector is complete behind it
    for (int q = 0; q < 2; q++) if ((side_used >> q) & 1) HIPCHK(h, hipStreamWaitEvent(st, c->ev_wg[q], 0));
    {   // stem: parameters only
        hipLaunchKernelGGL((t_bn_bwd_stats<true>), dim3(R), dim3(1024), 0, st, c->G, Al(0), Yl(0), c->mean, c->istd, M, c->part);
        if (dp) TRY(reduce_parts(2 * NG));
        hipLaunchKernelGGL((t_bn_bwd_finalize<true>), dim3(NG), dim3(1024), 0, st, dp ? c->red : c->part, dp ? 1 : R, g + OFF_STEM_BN, c->sums, gscale);
        hipLaunchKernelGGL((t_bn_bwd_apply<true>), dim3(g4), dim3(256), 0, st, c->G, Al(0), Yl(0), c->mean, c->istd, w + OFF_STEM_BN, c->sums,
                           1.0f / ((float)BSg * 6 * NF), c->dY, (float*)nullptr, M, nil16, nil16);
        gemm<true, false>(st, c->col0, KS, c->dY, NF, c->wpart, NF, KS, NF, M, c->nz, c->kchunk, (size_t)KS * NF);
        hipLaunchKernelGGL(t_sum_slices, grid1((size_t)KS * NF, 256), dim3(256), 0, st, c->wpart, c->nz, (size_t)KS * NF, c->gpad);
        hipLaunchKernelGGL(t_stem_unpad, grid1((size_t)9 * 13 * NF, 256), dim3(256), 0, st, c->gpad, g);
    }
    // ---------------- data-parallel: the ranks' gradient vectors add up to the gradient of the whole minibatch (one RCCL
    //                  all-reduce of count floats: 94.7 MB at B = 20); every rank then takes the same Adam step
    if (dp) TRY(dp_allreduce(h, c, g, c->count, 0));
    // ---------------- Adam
    hipLaunchKernelGGL(t_tick_lr, dim3(1), dim3(1), 0, st, c->cur, c->lr);
    hipLaunchKernelGGL(t_adam, grid1(c->count, 256), dim3(256), 0, st, w, g, c->m, c->v, c->kind, c->count, (const float*)c->lr);
    hipLaunchKernelGGL(t_tick_batch, dim3(1), dim3(1), 0, st, c->cur, BSg);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

// One minibatch step = gather + forward + backward + Adam.  Everything that changes from step to step (minibatch offset, Adam step
// count, learning rate) lives in device memory; the launches are plain stream launches (replaying the step as a captured hipGraph
// measured the same: the ~3 us between consecutive kernels is device-side, not host launch cost).
int run_step(azr_engine* h, TrainCtx* c)
{
    c->step++;
    hipLaunchKernelGGL(t_gather, dim3(c->BS), dim3(64), 0, h->stream, c->rec, c->perm, (const int*)c->cur, c->BS, c->in88, c->pit, c->zt);
    return train_step(h, c, c->loss + 2);
}

// Behind an epoch (or a single step): did a conv weight leave the range of the fp16-pair forward conv (t_pack_w's flag: |w| >= 64, or
// not a number), or did the losses stop being numbers?  Then the device copy of the weights and the optimiser state are poisoned: the
// call fails loudly, the weights the handle had before the call (its host AZRW copy) are put back and the optimiser state is dropped.
int step_health(azr_engine* h, TrainCtx* c, float loss_pi, float loss_v)
{
    int flag = 0;
    HIPCHK(h, hipMemcpyAsync(&flag, c->cur + 3, sizeof flag, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (!flag && std::isfinite(loss_pi) && std::isfinite(loss_v)) return AZR_OK;
    HIPCHK(h, hipMemcpyAsync(h->net.d_flat, h->flat.data(), h->flat.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    azr::train_free(h);
    h->err = flag ? "azr_nn_train: a conv weight left the range of the fp16-pair forward conv (|w| must stay below 64) or is not a number; "
                    "the weights from before the call have been restored and the optimiser state dropped"
                  : "azr_nn_train: the loss is not a number (diverged step); the weights from before the call have been restored and the "
                    "optimiser state dropped";
    return AZR_E_INVALID_ARGUMENT;
}

// after training: device master copy -> host AZRW copy -> refold / repack for inference
int finish(azr_engine* h)
{
    HIPCHK(h, hipMemcpyAsync(h->flat.data(), h->net.d_flat, h->flat.size() * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return net_upload(h);
}

int upload_records(azr_engine* h, TrainCtx* c, const void* rec, size_t n)
{
    if (n > c->rec_cap) {
        if (c->rec) hipFree(c->rec);
        c->rec = nullptr; c->rec_cap = 0;
        HIPCHK(h, hipMalloc((void**)&c->rec, n * 265));
        c->rec_cap = n;
    }
    if (n > c->perm_cap) {
        if (c->perm) hipFree(c->perm);
        c->perm = nullptr; c->perm_cap = 0;
        HIPCHK(h, hipMalloc((void**)&c->perm, n * sizeof(int)));
        c->perm_cap = n;
    }
    HIPCHK(h, hipMemcpyAsync(c->rec, rec, n * 265, hipMemcpyHostToDevice, h->stream));
    return AZR_OK;
}

}  // namespace

namespace azr {
void train_free(azr_engine* h)
{
    ctx_free(ctx_of(h));
    h->train = nullptr;
}
}  // namespace azr
extern "C" int azr_dp_shutdown(azr_engine* h);
namespace azr {
void dp_free(azr_engine* h) { azr_dp_shutdown(h); }
}  // namespace azr

#define ENTER(h)                                 \
    if (!(h)) return AZR_E_BAD_HANDLE;           \
    HIPCHK(h, hipSetDevice((h)->cfg.device))

extern "C" int azr_nn_train_batch(azr_engine* h, const void* rec265_host, int n, float* loss_pi, float* loss_v)
{
    ENTER(h);
    if (!h->weights_set) { h->err = "azr_nn_train_batch: no weights"; return AZR_E_STATE; }
    if (!rec265_host || n < 2) { h->err = "azr_nn_train_batch: need a minibatch of at least 2 records"; return AZR_E_INVALID_ARGUMENT; }
    TRY(ctx_ensure(h, n));
    TrainCtx* c = ctx_of(h);
    c->world = 1; c->rank = 0; c->ar = nullptr; c->ar_ctx = nullptr; c->native = false;   // (a data-parallel call that failed half-way must not linger)
    TRY(upload_records(h, c, rec265_host, (size_t)n));
    std::vector<int> id(n);
    for (int i = 0; i < n; i++) id[i] = i;
    HIPCHK(h, hipMemcpyAsync(c->perm, id.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
    {
        const int cur4[4] = {0, (int)c->step, 0, 0};
        HIPCHK(h, hipMemcpyAsync(c->cur, cur4, sizeof cur4, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    HIPCHK(h, hipMemsetAsync(c->loss + 2, 0, 2 * sizeof(float), h->stream));
    TRY(run_step(h, c));
    float l[2];
    HIPCHK(h, hipMemcpyAsync(l, c->loss, sizeof l, hipMemcpyDeviceToHost, h->stream));
    TRY(step_health(h, c, l[0], l[1]));
    TRY(finish(h));
    if (loss_pi) *loss_pi = l[0];
    if (loss_v) *loss_v = l[1];
    return AZR_OK;
}

// AlphaZeroNN::train for rank `rank` of `world` data-parallel ranks (world = 1: the reference's single-GPU training).  Every
// rank holds ALL n records and the same shuffle stream; of each minibatch of batch_size records rank r takes the slice
// [r * batch_size / world, (r + 1) * batch_size / world).
static int train_impl(azr_engine* h, const void* rec265_host, size_t n, int epochs, int batch_size, uint32_t* shuffle_rng_state, int rank,
                      int world, azr_allreduce_fn ar, void* ar_ctx, float* loss_pi_host, float* loss_v_host, bool dp_call = false)
{
    if (!h->weights_set) { h->err = "azr_nn_train: no weights"; return AZR_E_STATE; }
    if (!rec265_host || epochs < 0 || batch_size < 2) { h->err = "azr_nn_train: bad arguments"; return AZR_E_INVALID_ARGUMENT; }
    // no callback: the handle's own communicator (azr_dp_init) carries the sums.  (Test hook AZR_DP_LOOPBACK=1 — libazr_hip_test.so only, a
    // timing aid of tools/train_bench.py: a ONE-rank communicator stands in for `world` ranks — rank 0's share of the step with every
    // collective in the stream, sums stay local, so the weights it leaves are NOT those of a real step.  The product library has no such
    // switch: a communicator of another world size is refused.)
    const bool loopback = hook_env_int("AZR_DP_LOOPBACK", 0) != 0;
    const bool native = dp_call && !ar && h->dp_comm && ((h->dp_world == world && h->dp_rank == rank) || (loopback && h->dp_world == 1 && rank == 0));
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !ar && !native) || batch_size % world != 0 || batch_size / world < 2) {
        h->err = "azr_nn_train_dp: need 0 <= rank < world, batch_size a multiple of world with >= 2 records per rank, and either an all-reduce "
                 "callback or a communicator of exactly this rank / world (azr_dp_init)";
        return AZR_E_INVALID_ARGUMENT;
    }
    const int local_bs = batch_size / world;
    const size_t batches = n / (size_t)batch_size;  // the remainder of an epoch is dropped (alphazero_nn.cpp:374)
    std::minstd_rand0 eng;                          // RNG.getEngine() (src/rng.h:5-50): the caller's stream continues here
    if (shuffle_rng_state) {
        // a raw engine state, as azr_engine_set_rng takes it; 0 is not a state of minstd_rand0
        if (*shuffle_rng_state == 0 || *shuffle_rng_state >= 2147483647u) { h->err = "azr_nn_train: bad engine state"; return AZR_E_INVALID_ARGUMENT; }
        eng.seed(*shuffle_rng_state);
    }
    std::vector<int> order(n);
    for (size_t i = 0; i < n; i++) order[i] = (int)i;
    TrainCtx* c = nullptr;
    if (batches > 0 && epochs > 0) {
        TRY(ctx_ensure(h, local_bs));
        c = ctx_of(h);
        c->world = world; c->rank = rank; c->ar = ar; c->ar_ctx = ar_ctx; c->native = native;
        TRY(upload_records(h, c, rec265_host, n));
    }
    int rc = AZR_OK;
    for (int e = 0; e < epochs && rc == AZR_OK; e++) {
        std::shuffle(order.begin(), order.end(), eng);  // alphazero_nn.cpp:372 (same libstdc++ algorithm, same engine)
        float l[2] = {NAN, NAN};
        if (batches > 0) {
            const int cur4[4] = {0, (int)c->step, rank * local_bs, 0};   // minibatch offset, Adam step count, this rank's slice, range flag
            HIPCHK(h, hipMemcpyAsync(c->perm, order.data(), n * sizeof(int), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemsetAsync(c->loss + 2, 0, 2 * sizeof(float), h->stream));
            HIPCHK(h, hipMemcpyAsync(c->cur, cur4, sizeof cur4, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            for (size_t b = 0; b < batches && rc == AZR_OK; b++) rc = run_step(h, c);
            if (rc) break;
            HIPCHK(h, hipMemcpyAsync(l, c->loss + 2, sizeof l, hipMemcpyDeviceToHost, h->stream));
            rc = step_health(h, c, l[0], l[1]);   // (synchronises the stream)
            if (rc) { c = nullptr; break; }       // (the training context is gone with the poisoned optimiser state)
            l[0] /= (float)batches;
            l[1] /= (float)batches;
        }
        if (loss_pi_host) loss_pi_host[e] = l[0];
        if (loss_v_host) loss_v_host[e] = l[1];
    }
    if (c) { c->world = 1; c->rank = 0; c->ar = nullptr; c->ar_ctx = nullptr; c->native = false; }
    if (rc) return rc;
    if (shuffle_rng_state) {
        // minstd_rand0 has no state accessor; operator<< prints the state as decimal text
        std::ostringstream os;
        os << eng;
        *shuffle_rng_state = (uint32_t)std::stoul(os.str());
    }
    if (batches > 0 && epochs > 0) TRY(finish(h));
    return AZR_OK;
}

extern "C" int azr_nn_train(azr_engine* h, const void* rec265_host, size_t n, int epochs, int batch_size, uint32_t* shuffle_rng_state,
                            float* loss_pi_host, float* loss_v_host)
{
    ENTER(h);
    return train_impl(h, rec265_host, n, epochs, batch_size, shuffle_rng_state, 0, 1, nullptr, nullptr, loss_pi_host, loss_v_host);
}

extern "C" int azr_nn_train_dp(azr_engine* h, const void* rec265_host, size_t n, int epochs, int batch_size, uint32_t* shuffle_rng_state,
                               int rank, int world, azr_allreduce_fn allreduce, void* ctx, float* loss_pi_host, float* loss_v_host)
{
    ENTER(h);
    return train_impl(h, rec265_host, n, epochs, batch_size, shuffle_rng_state, rank, world, allreduce, ctx, loss_pi_host, loss_v_host, true);
}

// ---- the handle's own RCCL communicator (one process per GPU; the unique id travels by whatever the launcher has: torch.distributed,
//      MPI, a file) ----------------------------------------------------------------------------------------------------------------
extern "C" int azr_dp_unique_id(void* id128)
{
    if (!id128) return AZR_E_INVALID_ARGUMENT;
    RcclApi* R = rccl_api();
    if (!R->lib) return AZR_E_STATE;
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return AZR_E_HIP;
    static_assert(sizeof id == AZR_DP_ID_BYTES, "ncclUniqueId size");
    memcpy(id128, &id, sizeof id);
    return AZR_OK;
}

extern "C" int azr_dp_shutdown(azr_engine* h)
{
    if (!h) return AZR_E_BAD_HANDLE;
    if (h->dp_comm) {
        (void)hipSetDevice(h->cfg.device);
        if (h->stream) (void)hipStreamSynchronize(h->stream);
        rccl_api()->CommDestroy(static_cast<ncclComm_t>(h->dp_comm));
        h->dp_comm = nullptr;
    }
    h->dp_rank = 0; h->dp_world = 0;
    return AZR_OK;
}

extern "C" int azr_dp_init(azr_engine* h, int rank, int world, const void* id128)
{
    ENTER(h);
    if (!id128 || world < 1 || rank < 0 || rank >= world) { h->err = "azr_dp_init: need 0 <= rank < world and the 128-byte id of azr_dp_unique_id"; return AZR_E_INVALID_ARGUMENT; }
    RcclApi* R = rccl_api();
    if (!R->lib) { h->err = R->err; return AZR_E_STATE; }
    azr_dp_shutdown(h);
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t rc = R->CommInitRank(&comm, world, id, rank);   // collective over the `world` processes; binds to the current device
    if (rc != ncclSuccess) { h->err = std::string("ncclCommInitRank: ") + R->GetErrorString(rc); return AZR_E_HIP; }
    h->dp_comm = comm; h->dp_rank = rank; h->dp_world = world;
    return AZR_OK;
}

extern "C" int azr_nn_train_grads(azr_engine* h, float* flat_host, size_t count)
{
    ENTER(h);
    TrainCtx* c = ctx_of(h);
    if (!c) { h->err = "azr_nn_train_grads: no training step has run"; return AZR_E_STATE; }
    if (!flat_host || count != c->count) return AZR_E_INVALID_ARGUMENT;
    HIPCHK(h, hipMemcpy(flat_host, c->g, count * sizeof(float), hipMemcpyDeviceToHost));
    return AZR_OK;
}

extern "C" int azr_nn_train_reset(azr_engine* h)
{
    if (!h) return AZR_E_BAD_HANDLE;
    azr::train_free(h);
    return AZR_OK;
}

