// azr_net.hip — policy/value residual net of python/src/build_graph.py:54-90 as HIP kernels (gfx950).
//
// Host side: the AZRW flat fp32 parameter vector (DESIGN.md "weights"), BN folding, packing, checkpoints, and
// the dispatcher net_forward().  Device side in this file: the fp32 (exact, VALU) kernels — the precise
// reference path on the GPU and the tolerance anchor for the bf16 MFMA tower in azr_net_bf16.hip.
//
// AZRW flat layout (floats), B = blocks, F = 256 (shapes as in python/model/model_txt_V2_5.pb):
//   stem   : conv W[3][3][13][F] (HWIO) ; conv_bn gamma[7] beta[7] mean[7] var[7]  (BN over the board ROW y:
//            build_graph.py:68 `axis=1` on an NHWC tensor)
//   block i: res2a W[3][3][F][F] ; bn2a gamma,beta,mean,var [F] ; res2b W ; bn2b
//   policy : pi W[F][2] ; bn_pi g,b,m,v [2] ; dense W[84][43] ; bias[43]
//   value  : v W[F][1] ; bn_v g,b,m,v [1] ; dense_1 W[42][256] ; bias[256] ; dense_2 W[256][1] ; bias[1]
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "azr_internal.hpp"

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

namespace azr {
int net_bf16_alloc(azr_engine* h);
void net_bf16_free(azr_engine* h);
int net_bf16_upload(azr_engine* h, const float* fold_host);
int net_bf16_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
bool net_bf16_counted_ok(azr_engine* h, int n_max);
int net_bf16_forward_counted(azr_engine* h, const uint8_t* d_in88, int in_stride, int n_max, const int* n_dev, const int* n_other, float* d_pi, float* d_v, const int* d_map, hipStream_t st);
}  // namespace azr

constexpr float BN_EPS = 1e-3f;  // tf.layers.batch_normalization default epsilon
constexpr int NIN = 13;

static size_t off_stem_w() { return 0; }
static size_t off_stem_bn() { return 9 * NIN * NF; }
static size_t off_block(int i) { return off_stem_bn() + 28 + (size_t)i * 2 * (9 * NF * NF + 4 * NF); }
static size_t off_heads(int blocks) { return off_block(blocks); }
constexpr size_t HEAD_FLOATS = NF * 2 + 8 + 84 * 43 + 43 + NF + 4 + 42 * 256 + 256 + 256 + 1;

size_t azr::net_param_count(int blocks) { return off_heads(blocks) + HEAD_FLOATS; }

extern "C" size_t azr_nn_param_count(int blocks) { return net_param_count(blocks); }

// `init` op of the graph: Glorot-uniform kernels (tf.layers default), zero biases, BN gamma 1 / beta 0 /
// moving mean 0 / moving variance 1.  Generator: splitmix64 (any fixed stream is "random init").
static uint64_t splitmix64(uint64_t* s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
static float* glorot(float* p, size_t n, int fan_in, int fan_out, uint64_t* s)
{
    const float lim = sqrtf(6.0f / (float)(fan_in + fan_out));
    for (size_t i = 0; i < n; i++) p[i] = (2.0f * (float)((splitmix64(s) >> 40) * (1.0 / 16777216.0)) - 1.0f) * lim;
    return p + n;
}
static float* bn_identity(float* p, int c)
{
    for (int i = 0; i < c; i++) { p[i] = 1.0f; p[c + i] = 0.0f; p[2 * c + i] = 0.0f; p[3 * c + i] = 1.0f; }
    return p + 4 * c;
}
void azr::net_init_random(float* flat, int blocks, uint64_t seed)
{
    uint64_t s = seed;
    float* p = flat;
    p = glorot(p, 9 * NIN * NF, 9 * NIN, 9 * NF, &s);
    p = bn_identity(p, 7);
    for (int b = 0; b < 2 * blocks; b++) { p = glorot(p, 9 * NF * NF, 9 * NF, 9 * NF, &s); p = bn_identity(p, NF); }
    p = glorot(p, NF * 2, NF, 2, &s);
    p = bn_identity(p, 2);
    p = glorot(p, 84 * 43, 84, 43, &s);
    memset(p, 0, 43 * 4); p += 43;
    p = glorot(p, NF, NF, 1, &s);
    p = bn_identity(p, 1);
    p = glorot(p, 42 * 256, 42, 256, &s);
    memset(p, 0, 256 * 4); p += 256;
    p = glorot(p, 256, 256, 1, &s);
    *p++ = 0.0f;
}

// ================================================================================================
// fp32 kernels: one workgroup (256 threads = 256 output channels) per board; the board's input
// activation is staged in LDS inside a zero-bordered 9x8 frame so the 3x3 taps need no bounds tests.
// ================================================================================================
constexpr int FRAME_W = 8, FRAME_H = 9, FRAME = FRAME_W * FRAME_H;  // 72 cells, board cell (y,x) at (y+1,x+1)

__device__ __forceinline__ int frame_cell(int pos) { return (pos / 6 + 1) * FRAME_W + (pos % 6 + 1); }

// setInStateTensor (alphazero_nn.cpp:31-67): in88 -> 13 planes for one board cell
__device__ __forceinline__ float plane_value(const uint8_t* in88, int pos, int c)
{
    const uint32_t b = in88[pos];
    const int army = b & 63, owner = b >> 6, cur = in88[42], enemy = cur == 0 ? 1 : 0;
    const float fa = (float)army / 32.0f;
    const float* f = reinterpret_cast<const float*>(in88 + 48);
    switch (c) {
    case 0: return owner == cur ? fa : 0.0f;     // IF_CURRENT_PLAYER
    case 1: return owner == enemy ? fa : 0.0f;   // IF_ENEMY_PLAYER
    case 2: return owner == 2 ? fa : 0.0f;       // IF_NEUTRAL_PLAYER
    case 3: return f[9];                          // IF_ARMY_SHARE
    case 4: return f[0];                          // IF_REINFORCEMENT_SHARE
    case 5: return f[1];                          // IF_ATTACKS_DURING_TURN
    case 6: return f[2];                          // IF_CAN_DRAW_CARD
    default: return f[3 + (c - 7)];               // IF_PHASE_*
    }
}

__global__ __launch_bounds__(256) void k_stem_f32(const uint8_t* in88, int in_stride, const float* W,
                                                  const float* scale7, const float* shift7, float* out, const int* slot_map)
{
    __shared__ float x[FRAME][16];
    __shared__ __attribute__((aligned(16))) uint8_t in[96];
    const int b = blockIdx.x, co = threadIdx.x;
    if (co < 88) in[co] = in88[(size_t)(slot_map ? slot_map[b] : b) * in_stride + co];
    for (int i = co; i < FRAME * 16; i += 256) (&x[0][0])[i] = 0.0f;
    __syncthreads();
    for (int i = co; i < NPOS * NIN; i += 256) {
        int pos = i / NIN, c = i % NIN;
        x[frame_cell(pos)][c] = plane_value(in, pos, c);
    }
    __syncthreads();
    float acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; p++) acc[p] = 0.0f;
    for (int tap = 0; tap < 9; tap++) {
        const int off = (tap / 3 - 1) * FRAME_W + (tap % 3 - 1);
        for (int ci = 0; ci < NIN; ci++) {
            const float w = W[((size_t)tap * NIN + ci) * NF + co];
#pragma unroll
            for (int p = 0; p < NPOS; p++) acc[p] = fmaf(x[frame_cell(p) + off][ci], w, acc[p]);
        }
    }
#pragma unroll
    for (int p = 0; p < NPOS; p++) {
        const int y = p / 6;  // conv_bn normalises over axis 1 = board row
        float v = fmaf(acc[p], scale7[y], shift7[y]);
        out[((size_t)b * NPOS + p) * NF + co] = v > 0.0f ? v : 0.0f;
    }
}

// 3x3 conv F->F + folded BN (+ residual) + ReLU.  `out` may alias `res` (each element is read then written by
// the same thread) but not `in`.
__global__ __launch_bounds__(256) void k_conv_f32(const float* in, const float* W, const float* scale, const float* shift,
                                                  const float* res, float* out)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];  // [FRAME][NF]
    const int b = blockIdx.x, co = threadIdx.x;
    for (int i = co; i < FRAME * NF; i += 256) xs[i] = 0.0f;
    __syncthreads();
    const float* src = in + (size_t)b * NPOS * NF;
    for (int i = co; i < NPOS * NF; i += 256) xs[frame_cell(i / NF) * NF + (i % NF)] = src[i];
    __syncthreads();
    float acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; p++) acc[p] = 0.0f;
    for (int tap = 0; tap < 9; tap++) {
        const int off = ((tap / 3 - 1) * FRAME_W + (tap % 3 - 1)) * NF;
        const float* wt = W + (size_t)tap * NF * NF + co;
        for (int ci = 0; ci < NF; ci += 4) {
            const float w0 = wt[(size_t)(ci + 0) * NF], w1 = wt[(size_t)(ci + 1) * NF];
            const float w2 = wt[(size_t)(ci + 2) * NF], w3 = wt[(size_t)(ci + 3) * NF];
#pragma unroll
            for (int p = 0; p < NPOS; p++) {
                const float4 xv = *reinterpret_cast<const float4*>(&xs[frame_cell(p) * NF + off + ci]);
                float a = acc[p];
                a = fmaf(xv.x, w0, a); a = fmaf(xv.y, w1, a); a = fmaf(xv.z, w2, a); a = fmaf(xv.w, w3, a);
                acc[p] = a;
            }
        }
    }
    const float sc = scale[co], sh = shift[co];
#pragma unroll
    for (int p = 0; p < NPOS; p++) {
        const size_t o = ((size_t)b * NPOS + p) * NF + co;
        float v = fmaf(acc[p], sc, sh);
        if (res) v += res[o];
        out[o] = v > 0.0f ? v : 0.0f;
    }
}

// both heads for one board (build_graph.py:76-90).  `hp` = the head section of the AZRW vector.
// X is the tower output [n][42][256] in fp32 (XT = float) or bf16 bits (XT = uint16_t).
template <typename XT>
__device__ __forceinline__ float load_act(const XT* p);
template <>
__device__ __forceinline__ float load_act<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_act<uint16_t>(const uint16_t* p) { return __uint_as_float((uint32_t)*p << 16); }

template <typename XT>
__global__ __launch_bounds__(256) void k_heads(const XT* X, const float* hp, float* pi_out, float* v_out, const int* slot_map)
{
    __shared__ float feat[128];  // 84 policy features, then 42 value features
    __shared__ float hid[256];
    __shared__ float logit[44];
    const int b = blockIdx.x, t = threadIdx.x;
    const float* wpi = hp;              // [256][2]
    const float* bnpi = wpi + NF * 2;   // g[2] b[2] m[2] v[2]
    const float* wd = bnpi + 8;         // [84][43]
    const float* bd = wd + 84 * 43;     // [43]
    const float* wv = bd + 43;          // [256]
    const float* bnv = wv + NF;         // g b m v
    const float* w1 = bnv + 4;          // [42][256]
    const float* b1 = w1 + 42 * 256;    // [256]
    const float* w2 = b1 + 256;         // [256]
    const float* b2 = w2 + 256;         // [1]
    const XT* x = X + (size_t)b * NPOS * NF;
    if (t < 126) {  // 42 positions x {pi0, pi1, v}
        const int pos = t / 3, c = t % 3;
        float s = 0.0f;
        if (c < 2) for (int ci = 0; ci < NF; ci++) s = fmaf(load_act<XT>(x + pos * NF + ci), wpi[ci * 2 + c], s);
        else for (int ci = 0; ci < NF; ci++) s = fmaf(load_act<XT>(x + pos * NF + ci), wv[ci], s);
        const float* bn = c < 2 ? bnpi : bnv;
        const int nc = c < 2 ? 2 : 1, k = c < 2 ? c : 0;
        float y = (s - bn[2 * nc + k]) * (bn[k] / sqrtf(bn[3 * nc + k] + BN_EPS)) + bn[nc + k];
        y = y > 0.0f ? y : 0.0f;
        if (c < 2) feat[pos * 2 + c] = y;  // NHWC flatten: (y*6+x)*2 + c
        else feat[84 + pos] = y;
    }
    __syncthreads();
    if (t < 43) {
        float s = 0.0f;
        for (int i = 0; i < 84; i++) s = fmaf(feat[i], wd[i * 43 + t], s);
        logit[t] = s + bd[t];
    }
    {
        float s = 0.0f;
        for (int i = 0; i < 42; i++) s = fmaf(feat[84 + i], w1[i * 256 + t], s);
        s += b1[t];
        hid[t] = (s > 0.0f ? s : 0.0f) * w2[t];
    }
    __syncthreads();
    if (t < 64) {  // softmax over 43 logits by one wave
        float lv = t < 43 ? logit[t] : -INFINITY;
        float mx = lv;
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        float e = t < 43 ? expf(lv - mx) : 0.0f;
        float se = e;
        for (int m = 32; m >= 1; m >>= 1) se += __shfl_xor(se, m);
        const int slot = slot_map ? slot_map[b] : b;
        if (t < 43) pi_out[(size_t)slot * PI_STRIDE + t] = e / se;
        if (t == 43) pi_out[(size_t)slot * PI_STRIDE + 43] = 0.0f;
    } else if (t < 128) {  // value: sum of 256 products by one wave
        const int l = t - 64;
        float s = hid[l] + hid[l + 64] + hid[l + 128] + hid[l + 192];
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (l == 0) v_out[slot_map ? slot_map[b] : b] = tanhf(s + b2[0]);
    }
}

template __global__ void k_heads<float>(const float*, const float*, float*, float*, const int*);
template __global__ void k_heads<uint16_t>(const uint16_t*, const float*, float*, float*, const int*);

namespace azr {
void launch_heads_bf16(hipStream_t st, int n, const uint16_t* X, const float* hp, float* pi, float* v)
{
    hipLaunchKernelGGL(k_heads<uint16_t>, dim3(n), dim3(256), 0, st, X, hp, pi, v, (const int*)nullptr);
}
}  // namespace azr

// ================================================================================================
// host: allocation, folding, upload, dispatch
// ================================================================================================
static NetDev* nh(azr_engine* h) { return &h->net; }

int azr::net_alloc(azr_engine* h)
{
    NetDev& n = h->net;
    memset(&n, 0, sizeof n);
    n.blocks = h->cfg.blocks;
    NetDev* x = &n;
    const size_t count = net_param_count(n.blocks);
    h->flat.assign(count, 0.0f);
    HIPCHK(h, hipMalloc((void**)&x->d_flat, count * sizeof(float)));
    HIPCHK(h, hipMalloc((void**)&x->d_fold, (14 + (size_t)2 * n.blocks * 2 * NF) * sizeof(float)));
    if (h->cfg.net_dtype == AZR_NET_F32) {
        HIPCHK(h, hipMalloc((void**)&n.actX, (size_t)h->d.G * h->d.T * NPOS * NF * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&n.actT, (size_t)h->d.G * h->d.T * NPOS * NF * sizeof(float)));
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_f32), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      FRAME * NF * (int)sizeof(float)));
    } else if (h->cfg.net_dtype == AZR_NET_F32X) {
        int rc = net_fx_alloc(h);
        if (rc) return rc;
    } else {
        int rc = net_bf16_alloc(h);
        if (rc) return rc;
    }
    return AZR_OK;
}

void azr::net_free(azr_engine* h)
{
    NetDev* x = nh(h);
    if (x->d_flat) hipFree(x->d_flat);
    if (x->d_fold) hipFree(x->d_fold);
    if (h->net.actX) hipFree(h->net.actX);
    if (h->net.actT) hipFree(h->net.actT);
    net_bf16_free(h);
    net_fx_free(h);
    if (h->net.pred_dev) hipFree(h->net.pred_dev);
    if (h->net.pred_host) hipHostFree(h->net.pred_host);
    h->net.pred_dev = h->net.pred_host = nullptr;
    h->net.head = nullptr;
}

static void fold_bn(const float* bn, int c, float* scale, float* shift)
{
    for (int i = 0; i < c; i++) {
        const float g = bn[i], b = bn[c + i], m = bn[2 * c + i], v = bn[3 * c + i];
        const float s = g / sqrtf(v + BN_EPS);
        scale[i] = s;
        shift[i] = b - m * s;
    }
}

int azr::net_upload(azr_engine* h)
{
    NetDev* x = nh(h);
    const int B = h->net.blocks;
    const size_t count = net_param_count(B);
    std::vector<float> fold(14 + (size_t)2 * B * 2 * NF);
    fold_bn(h->flat.data() + off_stem_bn(), 7, fold.data(), fold.data() + 7);
    for (int l = 0; l < 2 * B; l++) {
        const float* bn = h->flat.data() + off_block(0) + (size_t)l * (9 * NF * NF + 4 * NF) + 9 * NF * NF;
        fold_bn(bn, NF, fold.data() + 14 + (size_t)l * 2 * NF, fold.data() + 14 + (size_t)l * 2 * NF + NF);
    }
    HIPCHK(h, hipMemcpyAsync(x->d_flat, h->flat.data(), count * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(x->d_fold, fold.data(), fold.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->net.stem_w = x->d_flat + off_stem_w();
    h->net.stem_scale = x->d_fold;
    h->net.stem_shift = x->d_fold + 7;
    h->net.tower_w = x->d_flat + off_block(0);
    h->net.tower_scale = x->d_fold + 14;
    h->net.tower_shift = x->d_fold + 14 + NF;
    h->net.head = x->d_flat + off_heads(B);
    if (h->cfg.net_dtype == AZR_NET_BF16 || h->cfg.net_dtype == AZR_NET_F16) {
        int rc = net_bf16_upload(h, fold.data());
        if (rc) return rc;
    }
    if (h->cfg.net_dtype == AZR_NET_F32X) {
        int rc = net_fx_upload(h, fold.data());
        if (rc) return rc;
    }
    h->weights_set = true;
    return AZR_OK;
}

namespace azr {
const float* net_head_params(azr_engine* h) { return nh(h)->d_flat + off_heads(h->net.blocks); }
const float* net_fold(azr_engine* h) { return nh(h)->d_fold; }
}  // namespace azr

static int net_f32_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    NetDev& N = h->net;
    NetDev* x = nh(h);
    hipLaunchKernelGGL(k_stem_f32, dim3(n), dim3(256), 0, st, d_in88, in_stride, N.stem_w, N.stem_scale,
                       N.stem_shift, N.actX, d_map);
    const size_t layer = (size_t)9 * NF * NF + 4 * NF;
    const size_t lds = FRAME * NF * sizeof(float);
    for (int b = 0; b < N.blocks; b++) {
        const float* wa = N.tower_w + (size_t)(2 * b) * layer;
        const float* wb = N.tower_w + (size_t)(2 * b + 1) * layer;
        const float* fa = x->d_fold + 14 + (size_t)(2 * b) * 2 * NF;
        const float* fb = x->d_fold + 14 + (size_t)(2 * b + 1) * 2 * NF;
        hipLaunchKernelGGL(k_conv_f32, dim3(n), dim3(256), lds, st, (const float*)N.actX, wa, fa, fa + NF,
                           (const float*)nullptr, N.actT);
        hipLaunchKernelGGL(k_conv_f32, dim3(n), dim3(256), lds, st, (const float*)N.actT, wb, fb, fb + NF,
                           (const float*)N.actX, N.actX);
    }
    hipLaunchKernelGGL(k_heads<float>, dim3(n), dim3(256), 0, st, (const float*)N.actX,
                       x->d_flat + off_heads(N.blocks), d_pi, d_v, d_map);
    HIPCHK(h, hipGetLastError());
    return AZR_OK;
}

int azr::net_forward(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v)
{
    if (n > h->d.G * h->d.T) { h->err = "net_forward: batch larger than the engine's leaf slots"; return AZR_E_INVALID_ARGUMENT; }
    return net_forward_ex(h, d_in88, in_stride, n, d_pi, d_v, nullptr, h->stream);
}

// forward of h's network on caller-supplied buffers and stream; d_map (optional, [n]) = leaf slot of board i, for input
// and output.  The two-net arena runs the opponent handle's weights on the arena handle's leaves this way.
int azr::net_forward_ex(azr_engine* h, const uint8_t* d_in88, int in_stride, int n, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    if (n <= 0) return AZR_OK;
    if (h->cfg.net_dtype == AZR_NET_F32) {
        if (n > h->d.G * h->d.T) { h->err = "net_forward: batch larger than the fp32 activation buffers"; return AZR_E_INVALID_ARGUMENT; }
        return net_f32_forward(h, d_in88, in_stride, n, d_pi, d_v, d_map, st);
    }
    if (h->cfg.net_dtype == AZR_NET_F32X) return net_fx_forward(h, d_in88, in_stride, n, d_pi, d_v, d_map, st);
    return net_bf16_forward(h, d_in88, in_stride, n, d_pi, d_v, d_map, st);
}

// forward of a batch whose size is the device word *n_dev (at most n_max boards), without a read-back: the 16-bit split-channel tower only
bool azr::net_forward_counted_ok(azr_engine* h, int n_max)
{
    return (h->cfg.net_dtype == AZR_NET_BF16 || h->cfg.net_dtype == AZR_NET_F16) && h->net.bf16ctx && net_bf16_counted_ok(h, n_max);
}

int azr::net_forward_counted(azr_engine* h, const uint8_t* d_in88, int in_stride, int n_max, const int* n_dev, const int* n_other, float* d_pi, float* d_v, const int* d_map, hipStream_t st)
{
    if (!net_forward_counted_ok(h, n_max)) { h->err = "net_forward_counted: not available for this net / batch"; return AZR_E_STATE; }
    return net_bf16_forward_counted(h, d_in88, in_stride, n_max, n_dev, n_other, d_pi, d_v, d_map, st);
}

namespace azr { int tower_sc_fallbacks(azr_engine* h, unsigned long long* out); }   // azr_tower_sc.hip
// launches of the split-channel tower that gave up a hand-off and were recomputed by the guarded launch behind them
int azr::net_fallbacks(azr_engine* h, unsigned long long* out)
{
    *out = 0;
    if ((h->cfg.net_dtype == AZR_NET_BF16 || h->cfg.net_dtype == AZR_NET_F16) && h->net.bf16ctx) return tower_sc_fallbacks(h, out);
    return AZR_OK;
}

// ---- C-ABI: AlphaZeroNNId ----------------------------------------------------------------------------
#define ENTER(h)                                 \
    if (!(h)) return AZR_E_BAD_HANDLE;           \
    HIPCHK(h, hipSetDevice((h)->cfg.device))

extern "C" int azr_nn_init_random(azr_engine* h, uint64_t seed)
{
    ENTER(h);
    net_init_random(h->flat.data(), h->net.blocks, seed);
    return net_upload(h);
}

extern "C" int azr_nn_set_weights(azr_engine* h, const float* flat, size_t count)
{
    ENTER(h);
    if (!flat || count != net_param_count(h->net.blocks)) { h->err = "azr_nn_set_weights: wrong parameter count"; return AZR_E_INVALID_ARGUMENT; }
    memcpy(h->flat.data(), flat, count * sizeof(float));
    return net_upload(h);
}

extern "C" int azr_nn_get_weights(azr_engine* h, float* flat, size_t count)
{
    if (!h) return AZR_E_BAD_HANDLE;
    if (!flat || count != net_param_count(h->net.blocks)) return AZR_E_INVALID_ARGUMENT;
    memcpy(flat, h->flat.data(), count * sizeof(float));
    return AZR_OK;
}

// checkpoint file: "AZRW" | u32 version | u32 blocks | u64 count | floats   (our own container; the reference's
// TF Saver files cannot be read without TensorFlow — INTEGRATION.md)
extern "C" int azr_nn_save(azr_engine* h, const char* path)
{
    if (!h) return AZR_E_BAD_HANDLE;
    // written next to the target and renamed into place: a concurrent reader never sees a partial checkpoint
    const std::string tmp = std::string(path) + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f) { h->err = std::string("cannot open ") + tmp; return AZR_E_IO; }
    const uint32_t ver = 1, blocks = (uint32_t)h->net.blocks;
    const uint64_t count = h->flat.size();
    bool ok = fwrite("AZRW", 1, 4, f) == 4 && fwrite(&ver, 4, 1, f) == 1 && fwrite(&blocks, 4, 1, f) == 1 &&
              fwrite(&count, 8, 1, f) == 1 && fwrite(h->flat.data(), 4, count, f) == count;
    ok = fclose(f) == 0 && ok;
    if (!ok) { remove(tmp.c_str()); h->err = "short write"; return AZR_E_IO; }
    if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); h->err = std::string("cannot rename to ") + path; return AZR_E_IO; }
    return AZR_OK;
}

extern "C" int azr_nn_load(azr_engine* h, const char* path)
{
    ENTER(h);
    FILE* f = fopen(path, "rb");
    if (!f) { h->err = std::string("cannot open ") + path; return AZR_E_IO; }
    char magic[4];
    uint32_t ver = 0, blocks = 0;
    uint64_t count = 0;
    // read into a temporary: a short or mismatching file must leave the handle's weights (host copy and device) untouched
    std::vector<float> tmp;
    bool ok = fread(magic, 1, 4, f) == 4 && memcmp(magic, "AZRW", 4) == 0 && fread(&ver, 4, 1, f) == 1 &&
              fread(&blocks, 4, 1, f) == 1 && fread(&count, 8, 1, f) == 1 && ver == 1 &&
              blocks == (uint32_t)h->net.blocks && count == h->flat.size();
    if (ok) {
        tmp.resize(count);
        ok = fread(tmp.data(), 4, count, f) == count && fgetc(f) == EOF;
    }
    fclose(f);
    if (!ok) { h->err = std::string("bad, truncated or mismatching checkpoint: ") + path; return AZR_E_IO; }
    h->flat.swap(tmp);
    return net_upload(h);
}

extern "C" int azr_nn_predict(azr_engine* h, const void* in88, int n, float* pi, float* v)
{
    ENTER(h);
    if (!h->weights_set) { h->err = "azr_nn_predict: no weights"; return AZR_E_STATE; }
    if (!in88 || n < 0 || (!pi && !v)) return AZR_E_INVALID_ARGUMENT;
    // batches of up to G boards through staging buffers that live as long as the handle (pinned host side: the copies are
    // real async DMA, no allocation in the call path — the reference's predict() is timed per call, alphazero_gpu_cluster.cpp:54-65)
    const int G = h->d.G;
    const size_t in_b = (size_t)G * LEAF_STRIDE, pi_b = (size_t)G * PI_STRIDE * 4, v_b = (size_t)G * 4;
    if (!h->net.pred_dev) {
        HIPCHK(h, hipMalloc((void**)&h->net.pred_dev, in_b + pi_b + v_b));
        HIPCHK(h, hipMemsetAsync(h->net.pred_dev, 0, in_b + pi_b + v_b, h->stream));
        HIPCHK(h, hipHostMalloc((void**)&h->net.pred_host, in_b + pi_b + v_b, hipHostMallocDefault));
        memset(h->net.pred_host, 0, in_b + pi_b + v_b);
    }
    uint8_t* d_in = h->net.pred_dev;
    float* d_pi = reinterpret_cast<float*>(h->net.pred_dev + in_b);
    float* d_v = reinterpret_cast<float*>(h->net.pred_dev + in_b + pi_b);
    uint8_t* s_in = h->net.pred_host;
    float* s_pi = reinterpret_cast<float*>(h->net.pred_host + in_b);
    float* s_v = reinterpret_cast<float*>(h->net.pred_host + in_b + pi_b);
    int rc = AZR_OK;
    for (int base = 0; base < n && rc == AZR_OK; base += G) {
        const int m = n - base < G ? n - base : G;
        for (int i = 0; i < m; i++) memcpy(s_in + (size_t)i * LEAF_STRIDE, (const uint8_t*)in88 + (size_t)(base + i) * 88, 88);
        HIPCHK(h, hipMemcpyAsync(d_in, s_in, (size_t)m * LEAF_STRIDE, hipMemcpyHostToDevice, h->stream));
        rc = net_forward(h, d_in, LEAF_STRIDE, m, d_pi, d_v);
        if (rc) break;
        if (pi) HIPCHK(h, hipMemcpyAsync(s_pi, d_pi, (size_t)m * PI_STRIDE * 4, hipMemcpyDeviceToHost, h->stream));
        if (v) HIPCHK(h, hipMemcpyAsync(s_v, d_v, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
        if (hipStreamSynchronize(h->stream) != hipSuccess) { (void)hipGetLastError(); h->err = "azr_nn_predict: stream error"; rc = AZR_E_HIP; break; }
        if (pi) for (int i = 0; i < m; i++) memcpy(pi + (size_t)(base + i) * 43, s_pi + (size_t)i * PI_STRIDE, 43 * 4);
        if (v) memcpy(v + base, s_v, (size_t)m * 4);
    }
    return rc;
}
