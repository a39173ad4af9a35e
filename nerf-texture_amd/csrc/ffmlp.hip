// Fully-fused tiny MLP on the gfx950 matrix cores (v_mfma_f32_16x16x32_f16, fp32 accumulation).
//
// Replaces the reference's ffmlp/src/ffmlp.cu (kernel_mlp_fused :331-407, kernel_mlp_fused_backward
// :410-518, ffmlp_backward :749-895 with its CUTLASS split-K weight-gradient GEMMs on side streams)
// behind include/nerftex_hip.h.  Same data contract (fp16 tensors, flat [out,in] row-major weights,
// forward_buffer / backward_buffer [num_layers, B, hidden]), different machine mapping:
//
//  * Orientation.  A wave computes H^T = W . X^T, i.e. the WEIGHTS are the MFMA A operand and a 16-row
//    batch tile is the B operand.  The 16x16 result tile then has the batch index on the lane axis --
//    which is exactly where the next layer's B operand wants it.  With a fixed permutation of the
//    contraction index (two result tiles interleave into one 32-deep operand: slot j<4 <- tile 2s,
//    j>=4 <- tile 2s+1), activations go register -> register through the whole network; the weight
//    fragments are staged ONCE per workgroup into LDS already in that permuted, lane-ordered form
//    (one conflict-free ds_read_b128 per fragment).  No activation ever touches LDS; the reference
//    round-trips every layer through padded shared memory because warp-32 WMMA cannot do this.
//  * Weight gradients  dW[o,i] = sum_n dPre[n,o] In[n,i]  contract over the BATCH, so both operands
//    need the batch index in registers.  They are read from global in the natural row-major form and
//    transposed by the matrix core itself (one MFMA against a 0/1 selection matrix per 16x16 tile,
//    exact in fp32) instead of through shared memory.  Partial sums stay in fp32 registers, are
//    combined per workgroup in LDS, written once per workgroup and reduced by a tiny second kernel:
//    one stream, no split-K side streams / events (ffmlp.cu:711-740), fp32 instead of fp16 accumulation.
//
// Supported: hidden_dim in {16,32,64,128,256}, input_dim % 16 == 0, output_dim == 16 (padded),
// num_layers >= 2, B % 128 == 0, every activation of ffmlp.py:89-96.  The weights of all layers must
// fit the 160 KB LDS of a CU (true for every width <= 128 and for 256 with num_layers == 2).
#include "common.hpp"
#include "workspace.hpp"

namespace nerftex {
namespace {

constexpr int kBlockThreads = 256;  // 4 waves, one per SIMD
constexpr int kTilesPerWave = 2;    // 2 x 16 batch rows per wave iteration (two independent MFMA chains)
constexpr int kRowsPerBlock = 4 * 16 * kTilesPerWave;  // 128, the reference's batch granule too
constexpr float kAct = 10.0f;       // K_ACT of utils.h:41

enum Act : uint32_t { kRelu = 0, kExp = 1, kSine = 2, kSigmoid = 3, kSquareplus = 4, kSoftplus = 5, kNone = 6 };

__device__ __forceinline__ float act_forward(uint32_t a, float x) {
    switch (a) {
        // integer max on the bit pattern: negative floats (sign bit set, -0 included) are negative ints -> +0, positive ones pass through.
        // One v_max_i32; the float compare-select form comes out as v_max_f32 plus a canonicalising v_max_f32 x, x per element
        case kRelu: return __builtin_bit_cast(float, max(__builtin_bit_cast(int, x), 0));
        case kExp: return expf(x);
        case kSine: return sinf(x);
        case kSigmoid: return 1.0f / (1.0f + expf(-x));
        case kSquareplus: { const float s = x * kAct; return 0.5f * (s + sqrtf(s * s + 4.0f)) / kAct; }
        case kSoftplus: return logf(expf(x * kAct) + 1.0f) / kAct;
        default: return x;
    }
}
// gradient through the activation, expressed with the stored POST-activation value y (utils.h:537-582)
__device__ __forceinline__ float act_backward(uint32_t a, float g, float y) {
    switch (a) {
        case kRelu: return y > 0.0f ? g : 0.0f;
        case kExp: return g * y;
        case kSigmoid: return g * (float)(half_t)(y * (1.0f - y));
        case kSquareplus: { const float s = y * kAct; return g * (float)(half_t)(s * s / (s * s + 1.0f)); }
        case kSoftplus: return g * (float)(half_t)(1.0f - expf(-y * kAct));
        default: return g;  // none, and sine (the reference leaves the gradient untouched, utils.h:552-556)
    }
}

// ACT >= 0: activation known at compile time (the switch folds away -- with a run-time id every element drags a chain of scalar
// compares and branches through all seven formulas: 1100 branches per 32-row step of the fused backward); ACT < 0: run-time id
template <int ACT>
__device__ __forceinline__ float act_fwd(uint32_t act_rt, float x) {
    if constexpr (ACT >= 0) return act_forward((uint32_t)ACT, x);
    else return act_forward(act_rt, x);
}
template <int ACT>
__device__ __forceinline__ float act_bwd(uint32_t act_rt, float g, float y) {
    if constexpr (ACT >= 0) return act_backward((uint32_t)ACT, g, y);
    else return act_backward(act_rt, g, y);
}

__device__ __forceinline__ float4_t mfma16(const half8_t& a, const half8_t& b, const float4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// contraction-index map of a 32-deep operand slot (lane group g = lane>>4, element j<8)
//   natural : operand comes from row-major memory, 8 consecutive k per lane
//   permuted: operand is rebuilt from two 16x16 result tiles held in registers (see file header)
__device__ __forceinline__ int kmap(bool permuted, int ks, int g, int j) {
    return permuted ? 32 * ks + 16 * (j >> 2) + 4 * g + (j & 3) : 32 * ks + 8 * g + j;
}

// Stage the A-operand fragments of a matrix view A[m][k] (M x K, M % 16 == 0) into LDS.
//   element (m,k) = transposed ? W[k*ldw + m] : W[m*ldw + k];  k >= K pads with zero.
// Layout: fragment (mt, ks) at ((mt*KS + ks)*64 + lane) * 16 bytes -> one ds_read_b128 per lane, no conflicts.
// Loads are 8- or 16-byte pieces of the row-major weights (rows start 32-byte aligned: ldw % 16 == 0, checked at the entry points):
// element-wise 2-byte gathers go through the texture-address path at about one LANE per clock -- 7-14 k of them per workgroup, four
// workgroups per CU, were 12 us of a 24 us inference launch.
//   as stored  : a lane's 8 values are one 16-byte run of row m (natural k order) or two 8-byte runs (permuted)
//   transposed : a thread takes 8 consecutive m of one k (one 16-byte load) and scatters them into the table with 2-byte LDS stores
__device__ void stage_fragments(half8_t* __restrict__ dst, const half_t* __restrict__ W, int ldw, bool transposed, int M, int K,
                                bool permuted) {
    const int KS = (K + 31) / 32;
    const half8_t zero8{0, 0, 0, 0, 0, 0, 0, 0};
    if (!transposed) {
        const int total = (M / 16) * KS * 64;
        for (int s = threadIdx.x; s < total; s += kBlockThreads) {
            const int frag = s >> 6, lane = s & 63;
            const int mt = frag / KS, ks = frag - mt * KS;
            const int m = 16 * mt + (lane & 15), g = lane >> 4;
            const half_t* row = W + (size_t)m * ldw;
            half8_t v;
            if (!permuted) {
                const int k0 = 32 * ks + 8 * g;
                v = k0 < K ? *reinterpret_cast<const half8_t*>(row + k0) : zero8;
            } else {
                const int ka = 32 * ks + 4 * g, kb = ka + 16;
                const half4_t z4{0, 0, 0, 0};
                const half4_t lo = ka < K ? *reinterpret_cast<const half4_t*>(row + ka) : z4;
                const half4_t hi = kb < K ? *reinterpret_cast<const half4_t*>(row + kb) : z4;
                v = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            dst[s] = v;
        }
    } else {
        half_t* d = reinterpret_cast<half_t*>(dst);
        const int MP = M / 8;
        for (int s = threadIdx.x; s < KS * 32 * MP; s += kBlockThreads) {
            const int k = s / MP, m0 = 8 * (s - k * MP);
            const half8_t v = k < K ? *reinterpret_cast<const half8_t*>(W + (size_t)k * ldw + m0) : zero8;
            const int ks = k >> 5, kk = k & 31;
            const int g = permuted ? (kk & 15) >> 2 : kk >> 3;        // inverse of kmap
            const int j = permuted ? 4 * (kk >> 4) + (kk & 3) : kk & 7;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int m = m0 + i;
                d[((size_t)((m >> 4) * KS + ks) * 64 + (m & 15) + 16 * g) * 8 + j] = v[i];
            }
        }
    }
}

__host__ __device__ constexpr int frag_count(int M, int K) { return (M / 16) * ((K + 31) / 32); }

// two fp32 result tiles (rows 4g+j of tiles 2s, 2s+1) -> the permuted 32-deep B operand of the next layer
__device__ __forceinline__ half8_t pack_operand(const float4_t& lo, const float4_t& hi) {
    half8_t b;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        b[j] = (half_t)lo[j];
        b[4 + j] = (half_t)hi[j];
    }
    return b;
}

__device__ __forceinline__ void store4(half_t* p, const float4_t& v) {
    half4_t h;
#pragma unroll
    for (int j = 0; j < 4; j++) h[j] = (half_t)v[j];
    *reinterpret_cast<half4_t*>(p) = h;
}

// ------------------------------------------------------------------------------------------------
// forward / inference
// ------------------------------------------------------------------------------------------------
template <int HIDDEN, bool INFERENCE, bool STAGED, int ACT, int OUT_ACT>  // STAGED: forward_buffer rows leave through an LDS patch; ACT: see act_fwd
__global__ __launch_bounds__(kBlockThreads) void ffmlp_forward_kernel(const half_t* __restrict__ X, const half_t* __restrict__ W,
                                                                      half_t* __restrict__ fwd, half_t* __restrict__ out, uint32_t B,
                                                                      uint32_t IN, uint32_t NL, uint32_t act, uint32_t out_act) {
    constexpr int OT = HIDDEN / 16;         // result tiles per hidden layer
    constexpr int KSH = (HIDDEN + 31) / 32;  // 32-deep steps over a hidden layer
    constexpr int NT = kTilesPerWave;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8_t* frags = reinterpret_cast<half8_t*>(smem);

    const int KS0 = (IN + 31) / 32;
    const int base_hidden = OT * KS0;             // first fragment of matrix 1
    const int per_hidden = OT * KSH;
    const int base_out = base_hidden + (NL - 1) * per_hidden;
    const size_t patch_offset = (size_t)(base_out + KSH) * 1024;  // after the last fragment (16 x HIDDEN output matrix)
    stage_fragments(frags, W, IN, false, HIDDEN, IN, false);
    for (uint32_t l = 1; l < NL; l++)
        stage_fragments(frags + (size_t)(base_hidden + (l - 1) * per_hidden) * 64, W + (size_t)HIDDEN * IN + (size_t)(l - 1) * HIDDEN * HIDDEN,
                        HIDDEN, false, HIDDEN, HIDDEN, true);
    stage_fragments(frags + (size_t)base_out * 64, W + (size_t)HIDDEN * IN + (size_t)(NL - 1) * HIDDEN * HIDDEN, HIDDEN, false, 16, HIDDEN,
                    true);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const size_t layer_stride = (size_t)B * HIDDEN;
    // training: a wave's 32 x HIDDEN activation block goes through a private LDS patch so that it leaves as whole rows
    // (1 KiB contiguous per store instruction) instead of 8-B pieces of 16 different rows
    constexpr int kRowPitch = HIDDEN + 8;  // halfs; +16 B keeps ds_read_b128 alignment and staggers the banks
    half_t* patch = reinterpret_cast<half_t*>(smem + patch_offset) + (size_t)wave * 16 * NT * kRowPitch;

    for (uint32_t row0 = blockIdx.x * kRowsPerBlock + wave * 16 * NT; row0 < B; row0 += gridDim.x * kRowsPerBlock) {
        float4_t acc[NT][OT];
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int ot = 0; ot < OT; ot++) acc[t][ot] = float4_t{0, 0, 0, 0};

        // ---- layer 0: B operand straight from the row-major input (16 B per lane, 1 KiB per wave load)
        for (int ks = 0; ks < KS0; ks++) {
            half8_t b[NT];
            const uint32_t k0 = 32 * ks + 8 * g;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (k0 < IN) b[t] = *reinterpret_cast<const half8_t*>(X + (size_t)(row0 + 16 * t + r) * IN + k0);
                else b[t] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                const half8_t a = frags[(size_t)(ot * KS0 + ks) * 64 + lane];
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ot] = mfma16(a, b[t], acc[t][ot]);
            }
        }

        half8_t bop[NT][KSH];
        for (uint32_t l = 0;; l++) {
            // activation, optional write-out of the post-activation values, repack as next B operand
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int ot = 0; ot < OT; ot++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[t][ot][j] = act_fwd<ACT>(act, acc[t][ot][j]);
                    if constexpr (!INFERENCE) {
                        if constexpr (STAGED) store4(patch + (size_t)(16 * t + r) * kRowPitch + 16 * ot + 4 * g, acc[t][ot]);
                        else store4(fwd + l * layer_stride + (size_t)(row0 + 16 * t + r) * HIDDEN + 16 * ot + 4 * g, acc[t][ot]);
                    }
                }
#pragma unroll
                for (int s = 0; s < KSH; s++) {
                    const float4_t zero{0, 0, 0, 0};
                    bop[t][s] = pack_operand(acc[t][2 * s], (2 * s + 1 < OT) ? acc[t][(2 * s + 1 < OT) ? 2 * s + 1 : 0] : zero);
                }
            }
            if constexpr (!INFERENCE && STAGED) {  // patch -> forward_buffer[l], 16 B per lane, rows are contiguous in memory
                constexpr int kPieces = HIDDEN / 8;  // 16-B pieces per row
                half_t* dst = fwd + l * layer_stride + (size_t)row0 * HIDDEN;
#pragma unroll
                for (int it = 0; it < 16 * NT * kPieces / 64; it++) {
                    const int c = it * 64 + lane, row = c / kPieces, piece = c % kPieces;
                    *reinterpret_cast<half8_t*>(dst + (size_t)row * HIDDEN + 8 * piece) =
                        *reinterpret_cast<const half8_t*>(patch + (size_t)row * kRowPitch + 8 * piece);
                }
            }
            if (l + 1 >= NL) break;
            // ---- hidden matrix l+1
            const half8_t* fl = frags + (size_t)(base_hidden + l * per_hidden) * 64;
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                float4_t c[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) c[t] = float4_t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSH; ks++) {
                    const half8_t a = fl[(size_t)(ot * KSH + ks) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < NT; t++) c[t] = mfma16(a, bop[t][ks], c[t]);
                }
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ot] = c[t];
            }
        }

        // ---- output layer: 16 (padded) outputs = one result tile
        {
            const half8_t* fo = frags + (size_t)base_out * 64;
            float4_t c[NT];
#pragma unroll
            for (int t = 0; t < NT; t++) c[t] = float4_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < KSH; ks++) {
                const half8_t a = fo[(size_t)ks * 64 + lane];
#pragma unroll
                for (int t = 0; t < NT; t++) c[t] = mfma16(a, bop[t][ks], c[t]);
            }
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int j = 0; j < 4; j++) c[t][j] = act_fwd<OUT_ACT>(out_act, c[t][j]);
                store4(out + (size_t)(row0 + 16 * t + r) * 16 + 4 * g, c[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 1: activation gradients (dL/d pre-activation of every hidden layer) + dL/d input
//   bb[j] holds the gradient at hidden activation NL-1-j (the reference's backward_buffer order)
// ------------------------------------------------------------------------------------------------
template <int HIDDEN>
__global__ __launch_bounds__(kBlockThreads) void ffmlp_dgrad_kernel(const half_t* __restrict__ grad, const half_t* __restrict__ W,
                                                                    const half_t* __restrict__ fwd, half_t* __restrict__ bb,
                                                                    half_t* __restrict__ grad_inputs, uint32_t B, uint32_t IN, uint32_t NL,
                                                                    uint32_t act) {
    constexpr int OT = HIDDEN / 16;
    constexpr int KSH = (HIDDEN + 31) / 32;
    constexpr int NT = kTilesPerWave;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8_t* frags = reinterpret_cast<half8_t*>(smem);

    const half_t* W_hidden = W + (size_t)HIDDEN * IN;
    const half_t* W_out = W_hidden + (size_t)(NL - 1) * HIDDEN * HIDDEN;
    // fragment table: [W_out^T (HIDDEN x 16)] [W_l^T for l = NL-1 .. 1] [W_0^T (IN x HIDDEN), optional]
    const int per_hidden = OT * KSH;
    const int base_hidden = OT;  // W_out^T has K = 16 -> one 32-deep step per tile
    const int base_in = base_hidden + (NL - 1) * per_hidden;
    stage_fragments(frags, W_out, HIDDEN, true, HIDDEN, 16, false);
    for (uint32_t j = 1; j < NL; j++)
        stage_fragments(frags + (size_t)(base_hidden + (j - 1) * per_hidden) * 64, W_hidden + (size_t)(NL - 1 - j) * HIDDEN * HIDDEN, HIDDEN, true,
                        HIDDEN, HIDDEN, true);
    if (grad_inputs) stage_fragments(frags + (size_t)base_in * 64, W, IN, true, IN, HIDDEN, true);
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const size_t layer_stride = (size_t)B * HIDDEN;

    for (uint32_t row0 = blockIdx.x * kRowsPerBlock + wave * 16 * NT; row0 < B; row0 += gridDim.x * kRowsPerBlock) {
        // B operand of the first product: grad^T (16 outputs = lane groups 0,1; groups 2,3 are zero padding)
        half8_t bg[NT];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (g < 2) bg[t] = *reinterpret_cast<const half8_t*>(grad + (size_t)(row0 + 16 * t + r) * 16 + 8 * g);
            else bg[t] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
        float4_t acc[NT][OT];
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const half8_t a = frags[(size_t)ot * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t][ot] = mfma16(a, bg[t], float4_t{0, 0, 0, 0});
        }

        half8_t bop[NT][KSH];
        for (uint32_t j = 0;; j++) {
            const half_t* f = fwd + (size_t)(NL - 1 - j) * layer_stride;  // post-activations of the layer being crossed
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int ot = 0; ot < OT; ot++) {
                    const size_t off = (size_t)(row0 + 16 * t + r) * HIDDEN + 16 * ot + 4 * g;
                    const half4_t y = *reinterpret_cast<const half4_t*>(f + off);
#pragma unroll
                    for (int q = 0; q < 4; q++) acc[t][ot][q] = act_backward(act, (float)(half_t)acc[t][ot][q], (float)y[q]);
                    store4(bb + j * layer_stride + off, acc[t][ot]);
                }
#pragma unroll
                for (int s = 0; s < KSH; s++) {
                    const float4_t zero{0, 0, 0, 0};
                    bop[t][s] = pack_operand(acc[t][2 * s], (2 * s + 1 < OT) ? acc[t][(2 * s + 1 < OT) ? 2 * s + 1 : 0] : zero);
                }
            }
            if (j + 1 >= NL) break;
            const half8_t* fl = frags + (size_t)(base_hidden + j * per_hidden) * 64;  // W_{NL-1-j}^T
#pragma unroll
            for (int ot = 0; ot < OT; ot++) {
                float4_t c[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) c[t] = float4_t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSH; ks++) {
                    const half8_t a = fl[(size_t)(ot * KSH + ks) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < NT; t++) c[t] = mfma16(a, bop[t][ks], c[t]);
                }
#pragma unroll
                for (int t = 0; t < NT; t++) acc[t][ot] = c[t];
            }
        }

        if (grad_inputs) {  // dL/dX = W_0^T . dPre_0, no activation (ffmlp.cu:880-887)
            const half8_t* fi = frags + (size_t)base_in * 64;
            for (uint32_t it = 0; it < IN / 16; it++) {
                float4_t c[NT];
#pragma unroll
                for (int t = 0; t < NT; t++) c[t] = float4_t{0, 0, 0, 0};
#pragma unroll
                for (int ks = 0; ks < KSH; ks++) {
                    const half8_t a = fi[(size_t)(it * KSH + ks) * 64 + lane];
#pragma unroll
                    for (int t = 0; t < NT; t++) c[t] = mfma16(a, bop[t][ks], c[t]);
                }
#pragma unroll
                for (int t = 0; t < NT; t++) store4(grad_inputs + (size_t)(row0 + 16 * t + r) * IN + 16 * it + 4 * g, c[t]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: weight gradients  dW[o,i] = sum_n dPre[n,o] * In[n,i]   (contraction over the batch)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxLayers = 12;
struct WgradLayer {
    const half_t* dpre;  // [B, ld_d] rows, O valid columns
    const half_t* in;    // [B, ld_i] rows, K valid columns
    uint32_t ld_d, ld_i, O, K;
    uint32_t w_off;      // offset of this matrix in the flat weight vector
};
struct WgradArgs {
    WgradLayer layer[kMaxLayers];
};

// 16 batch rows x 32 features, natural A-operand form (lane r = batch row, 8 consecutive features)
__device__ __forceinline__ half8_t load_rows(const half_t* p, uint32_t ld, uint32_t row, uint32_t col0, uint32_t ncols, int g) {
    const uint32_t c = col0 + 8 * g;
    if (c < ncols) return *reinterpret_cast<const half8_t*>(p + (size_t)row * ld + c);
    return half8_t{0, 0, 0, 0, 0, 0, 0, 0};
}

__global__ __launch_bounds__(kBlockThreads) void ffmlp_wgrad_kernel(const WgradArgs args, uint32_t B, float* __restrict__ partials,
                                                                    uint32_t n_params) {
    constexpr int G = 4;  // tiles per group in each direction -> 16 accumulator tiles (64 VGPRs)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [4 waves][G*G tiles][256] fp32, per-workgroup combine

    const WgradLayer L = args.layer[blockIdx.y];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const uint32_t OT = L.O / 16, IT = L.K / 16;

    // selection matrices: B operand with B[k][col] = (k == col) / (k == 16 + col); A x Sel = transpose into "batch in registers"
    half8_t sel0, sel1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sel0[j] = (8 * g + j == r) ? (half_t)1.0f : (half_t)0.0f;
        sel1[j] = (8 * g + j == 16 + r) ? (half_t)1.0f : (half_t)0.0f;
    }
    const float4_t zero{0, 0, 0, 0};
    float* out = partials + (size_t)blockIdx.x * n_params + L.w_off;

    for (uint32_t og = 0; og < OT; og += G) {
        for (uint32_t ig = 0; ig < IT; ig += G) {
            float4_t acc[G][G];
#pragma unroll
            for (int a = 0; a < G; a++)
#pragma unroll
                for (int b = 0; b < G; b++) acc[a][b] = zero;

            // each wave walks 32-row steps of the batch
            for (uint32_t row0 = (blockIdx.x * 4 + wave) * 32; row0 < B; row0 += gridDim.x * 4 * 32) {
                half8_t A[G], Bm[G];
                // transposed dPre tiles -> A operands (lane = output neuron, slots = 2 x 4 batch rows)
#pragma unroll
                for (int a = 0; a < G; a += 2) {
                    const uint32_t col0 = 16 * (og + a);  // 32 features feed tiles a, a+1
                    const half8_t x0 = load_rows(L.dpre, L.ld_d, row0 + r, col0, L.O, g);
                    const half8_t x1 = load_rows(L.dpre, L.ld_d, row0 + 16 + r, col0, L.O, g);
                    A[a] = pack_operand(mfma16(x0, sel0, zero), mfma16(x1, sel0, zero));
                    A[a + 1] = pack_operand(mfma16(x0, sel1, zero), mfma16(x1, sel1, zero));
                }
#pragma unroll
                for (int b = 0; b < G; b += 2) {
                    const uint32_t col0 = 16 * (ig + b);
                    const half8_t x0 = load_rows(L.in, L.ld_i, row0 + r, col0, L.K, g);
                    const half8_t x1 = load_rows(L.in, L.ld_i, row0 + 16 + r, col0, L.K, g);
                    Bm[b] = pack_operand(mfma16(x0, sel0, zero), mfma16(x1, sel0, zero));
                    Bm[b + 1] = pack_operand(mfma16(x0, sel1, zero), mfma16(x1, sel1, zero));
                }
#pragma unroll
                for (int a = 0; a < G; a++)
#pragma unroll
                    for (int b = 0; b < G; b++) acc[a][b] = mfma16(A[a], Bm[b], acc[a][b]);
            }

            // combine the 4 waves in LDS, then one fp32 write per element per workgroup
            __syncthreads();
#pragma unroll
            for (int a = 0; a < G; a++)
#pragma unroll
                for (int b = 0; b < G; b++)
#pragma unroll
                    for (int j = 0; j < 4; j++) red[((wave * G * G + a * G + b) * 4 + j) * 64 + lane] = acc[a][b][j];
            __syncthreads();
            for (int e = threadIdx.x; e < G * G * 256; e += kBlockThreads) {
                const int tile = e >> 8, j = (e >> 6) & 3, ln = e & 63;
                const int a = tile / G, b = tile % G;
                const uint32_t o = 16 * (og + a) + 4 * (ln >> 4) + j, i = 16 * (ig + b) + (ln & 15);
                if (o < L.O && i < L.K) {
                    float s = 0.0f;
#pragma unroll
                    for (int w = 0; w < 4; w++) s += red[((w * G * G + tile) * 4 + j) * 64 + ln];
                    out[(size_t)o * L.K + i] = s;
                }
            }
        }
    }
}

// per-workgroup combine of one weight-gradient matrix held as NA x NB result tiles per wave: LDS sum over the 4 waves, one fp32
// write per element.  Everything is indexed at compile time so the accumulators stay in registers.
// One layer's products for NB 16-row batch tiles: out[t][ot] = sum_ks A(ot, ks) . b[t][ks], the A fragments read from LDS (fr = table
// base + lane).  The fragments of tile ot+1 are requested BEFORE the MFMAs of tile ot and the order is pinned: left alone, the
// scheduler (at the register limit) reloads the same registers right after their last use and waits out the full LDS latency in
// front of every tile -- ~30 exposed round trips per 32-row step of the fused backward.
template <int NTILES, int KS, int NB>
__device__ __forceinline__ void layer_products(const half8_t* __restrict__ fr, const half8_t (&b)[NB][KS], float4_t (&out)[NB][NTILES]) {
    half8_t a[2][KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) a[0][ks] = fr[(size_t)ks * 64];
#pragma unroll
    for (int ot = 0; ot < NTILES; ot++) {
        if (ot + 1 < NTILES) {
#pragma unroll
            for (int ks = 0; ks < KS; ks++) a[(ot + 1) & 1][ks] = fr[(size_t)((ot + 1) * KS + ks) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        float4_t c[NB];
#pragma unroll
        for (int t = 0; t < NB; t++) c[t] = float4_t{0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int t = 0; t < NB; t++) c[t] = mfma16(a[ot & 1][ks], b[t][ks], c[t]);
#pragma unroll
        for (int t = 0; t < NB; t++) out[t][ot] = c[t];
    }
}

template <int NA, int NB>
__device__ __forceinline__ void flush_tiles(const float4_t (&tiles)[NA][NB], float* __restrict__ red, float* __restrict__ out, uint32_t K, int wave,
                                            int lane) {
    static_assert(NA * NB <= 16, "combine buffer holds 16 tiles per wave");
    __syncthreads();
#pragma unroll
    for (int a = 0; a < NA; a++)
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) red[((wave * 16 + a * NB + b) * 4 + q) * 64 + lane] = tiles[a][b][q];
    __syncthreads();
    for (int e = threadIdx.x; e < NA * NB * 256; e += kBlockThreads) {
        const int tile = e >> 8, q = (e >> 6) & 3, ln = e & 63;
        const int a = tile / NB, b = tile % NB;
        const uint32_t o = 16 * a + 4 * (ln >> 4) + q, i = 16 * b + (ln & 15);
        float sum = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; w++) sum += red[((w * 16 + tile) * 4 + q) * 64 + ln];
        out[(size_t)o * K + i] = sum;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, fused: activation gradients AND weight gradients in one pass over the batch
// ------------------------------------------------------------------------------------------------
// The dgrad chain already holds, per 32 batch rows, every dPre tile (fp32 result tiles -> the permuted 32-deep operand `bop`)
// and loads every saved activation tile (for the activation derivative) in the same (lane = batch row, slots = features)
// form.  That is exactly the input form of the transposing MFMAs of the weight-gradient kernel above -- so the weight gradients
// are accumulated right here, in fp32 registers that live across the whole batch loop, and
//   * backward_buffer is never written nor read (it is scratch in the reference's contract),
//   * forward_buffer and the inputs are read once instead of twice:  ~52 % less HBM traffic for the whole MLP backward.
// One wave per SIMD (the dW accumulators are 112-176 registers); the loads of a 32-row step are issued together.
// RECOMPUTE: forward_buffer is not read at all -- the saved activations are rebuilt from the inputs with the forward kernel's own
// chain (bit-identical halfs), which costs a few dozen MFMAs per 32 rows and removes 70 % of this kernel's loads; the training
// forward then has no forward_buffer to write either.
template <int HIDDEN, int NL, int IT, bool RECOMPUTE, int ACT>  // IT = input_dim / 16; ACT: see act_fwd
__global__ __launch_bounds__(kBlockThreads, 1) void ffmlp_backward_fused_kernel(const half_t* __restrict__ grad, const half_t* __restrict__ X,
                                                                               const half_t* __restrict__ W, const half_t* __restrict__ fwd,
                                                                               half_t* __restrict__ grad_inputs, uint32_t B, uint32_t act,
                                                                               float* __restrict__ partials, uint32_t n_params) {
    constexpr int OT = HIDDEN / 16;
    constexpr int KSH = (HIDDEN + 31) / 32;
    constexpr int NT = 2;               // 32 batch rows per wave step = one 32-deep contraction step of the weight gradients
    constexpr int IN = 16 * IT;
    constexpr int KS0 = (IN + 31) / 32;
    static_assert(HIDDEN % 32 == 0 && kTilesPerWave == NT, "fused backward: hidden width must be a multiple of 32");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8_t* frags = reinterpret_cast<half8_t*>(smem);

    const half_t* W_hidden = W + (size_t)HIDDEN * IN;
    const half_t* W_out = W_hidden + (size_t)(NL - 1) * HIDDEN * HIDDEN;
    constexpr int per_hidden = OT * KSH;
    constexpr int base_hidden = OT;
    constexpr int base_in = base_hidden + (NL - 1) * per_hidden;
    stage_fragments(frags, W_out, HIDDEN, true, HIDDEN, 16, false);
    for (int j = 1; j < NL; j++)
        stage_fragments(frags + (size_t)(base_hidden + (j - 1) * per_hidden) * 64, W_hidden + (size_t)(NL - 1 - j) * HIDDEN * HIDDEN, HIDDEN, true,
                        HIDDEN, HIDDEN, true);
    if (grad_inputs) stage_fragments(frags + (size_t)base_in * 64, W, IN, true, IN, HIDDEN, true);
    // forward-orientation fragments for the recomputation: [W_0 (HIDDEN x IN)] [W_l, l = 1 .. NL-1]
    constexpr int base_f0 = base_in + IT * KSH;
    constexpr int base_fh = base_f0 + OT * KS0;
    if constexpr (RECOMPUTE) {
        stage_fragments(frags + (size_t)base_f0 * 64, W, IN, false, HIDDEN, IN, false);
        for (int l = 1; l < NL; l++)
            stage_fragments(frags + (size_t)(base_fh + (l - 1) * per_hidden) * 64, W_hidden + (size_t)(l - 1) * HIDDEN * HIDDEN, HIDDEN, false, HIDDEN,
                            HIDDEN, true);
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const size_t layer_stride = (size_t)B * HIDDEN;
    const float4_t zero{0, 0, 0, 0};

    // 0/1 selection operands: X . Sel moves 16 of the 32 features of X onto the lane axis (batch rows into registers)
    //   natural order (operand read from row-major memory): slot (g, j) <-> feature 8g + j
    //   permuted order (operand packed from two result tiles / two half4 activation loads): j < 4 <-> tile 2s feature 4g + j,
    //                                                                                       j >= 4 <-> tile 2s+1 feature 4g + j - 4
    half8_t sel0, sel1, selP0, selP1;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        sel0[j] = (8 * g + j == r) ? (half_t)1.0f : (half_t)0.0f;
        sel1[j] = (8 * g + j == 16 + r) ? (half_t)1.0f : (half_t)0.0f;
        selP0[j] = (j < 4 && 4 * g + j == r) ? (half_t)1.0f : (half_t)0.0f;
        selP1[j] = (j >= 4 && 4 * g + j - 4 == r) ? (half_t)1.0f : (half_t)0.0f;
    }

    float4_t gw_out[1][OT];                         // dW_out [16 x HIDDEN]
    float4_t gw_hid[NL > 1 ? NL - 1 : 1][OT][OT];   // dW_l, l = 1..NL-1 [HIDDEN x HIDDEN]
    float4_t gw_in[OT][IT];                         // dW_0 [HIDDEN x IN]
#pragma unroll
    for (int b = 0; b < OT; b++) gw_out[0][b] = zero;
#pragma unroll
    for (int l = 0; l < NL - 1; l++)
#pragma unroll
        for (int a = 0; a < OT; a++)
#pragma unroll
            for (int b = 0; b < OT; b++) gw_hid[l][a][b] = zero;
#pragma unroll
    for (int a = 0; a < OT; a++)
#pragma unroll
        for (int b = 0; b < IT; b++) gw_in[a][b] = zero;

    // One wave per SIMD: nobody hides the latency of a step's loads, and asking for the NEXT step's operands in registers costs 64
    // of them (tried: the 3-layer instantiation spills, 130 -> 160 us).  With the activations recomputed a step needs only its
    // gradient and input rows -- 4 KiB per wave -- and those are prefetched one step ahead straight into LDS (global_load_lds: no
    // registers, the wave's own counted vmcnt orders its later ds_read), two slots per wave.
    constexpr int kPieces = NT + NT * KS0;  // 1-KiB pieces per step: grad tiles, then input tiles
    constexpr size_t kRingOffset = ((size_t)(base_fh + (NL - 1) * per_hidden) * 1024 > 64 * 1024) ? (size_t)(base_fh + (NL - 1) * per_hidden) * 1024 : 64 * 1024;
    half8_t* ring = reinterpret_cast<half8_t*>(smem + kRingOffset) + (size_t)wave * 2 * kPieces * 64;
    const uint32_t row_step = gridDim.x * kRowsPerBlock;
    auto prefetch = [&](uint32_t rw, int slot) {
        half8_t* dst = ring + (size_t)slot * kPieces * 64;
#pragma unroll
        for (int t = 0; t < NT; t++)  // lane groups 2, 3 have no gradient columns: they fetch a copy of groups 0, 1 and drop it
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(grad + (size_t)(rw + 16 * t + r) * 16 + 8 * (g & 1)),
                                             (__attribute__((address_space(3))) void*)(dst + t * 64), 16, 0, 0);
#pragma unroll
        for (int ks = 0; ks < KS0; ks++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int k0 = 32 * ks + 8 * g;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(X + (size_t)(rw + 16 * t + r) * IN + (k0 < IN ? k0 : 0)),
                                                 (__attribute__((address_space(3))) void*)(dst + (NT + ks * NT + t) * 64), 16, 0, 0);
            }
    };
    uint32_t step = 0;
    if constexpr (RECOMPUTE) {
        const uint32_t first = blockIdx.x * kRowsPerBlock + wave * 16 * NT;
        if (first < B) prefetch(first, 0);
    }
    for (uint32_t row0 = blockIdx.x * kRowsPerBlock + wave * 16 * NT; row0 < B; row0 += row_step, step++) {
        // ---- all loads of this step first: output gradient, inputs, saved activations of every layer
        half8_t bg[NT];
        half8_t xin[NT][KS0];
        if constexpr (RECOMPUTE) {
            const uint32_t next = row0 + row_step;
            prefetch(next < B ? next : row0, (step + 1) & 1);  // past the end: a harmless re-fetch keeps the count below uniform
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kPieces) : "memory");  // everything but the pieces just requested has landed
            const half8_t* src = ring + (size_t)(step & 1) * kPieces * 64;
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const half8_t v = src[t * 64 + lane];
                bg[t] = g < 2 ? v : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int ks = 0; ks < KS0; ks++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    const half8_t v = src[(NT + ks * NT + t) * 64 + lane];
                    xin[t][ks] = (32 * ks + 8 * g < IN) ? v : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                }
        } else {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (g < 2) bg[t] = *reinterpret_cast<const half8_t*>(grad + (size_t)(row0 + 16 * t + r) * 16 + 8 * g);
                else bg[t] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
#pragma unroll
            for (int ks = 0; ks < KS0; ks++)
#pragma unroll
                for (int t = 0; t < NT; t++) {
                    if (32 * ks + 8 * g < IN) xin[t][ks] = *reinterpret_cast<const half8_t*>(X + (size_t)(row0 + 16 * t + r) * IN + 32 * ks + 8 * g);
                    else xin[t][ks] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
                }
        }
        // yop[j][t][s] = post-activations of hidden layer NL-1-j for 16 rows x 32 features, already in the packed-operand form
        // (slot q < 4: feature 16(2s) + 4g + q, slot 4 + q: feature 16(2s+1) + 4g + q of row r): what the recomputation produces anyway,
        // what the transposing MFMAs consume, and what the activation derivative reads element by element
        half8_t yop[NL][NT][KSH];
        float4_t acc[NT][OT];
        if constexpr (RECOMPUTE) {
            // the forward kernel's chain, verbatim (layer 0 from the row-major input, then register to register), one 16-row tile
            // at a time to keep the transient registers at one tile's worth
#pragma unroll
            for (int t = 0; t < NT; t++) {
                asm volatile("" ::: "memory");  // re-read the weight fragments from LDS per tile instead of keeping 20 of them in registers
                float4_t fa[1][OT];
                half8_t x1[1][KS0];
#pragma unroll
                for (int ks = 0; ks < KS0; ks++) x1[0][ks] = xin[t][ks];
                layer_products<OT, KS0, 1>(frags + (size_t)base_f0 * 64 + lane, x1, fa);
#pragma unroll
                for (int l = 0; l < NL; l++) {
                    half8_t fop[1][KSH];
#pragma unroll
                    for (int ot = 0; ot < OT; ot++)
#pragma unroll
                        for (int q = 0; q < 4; q++) fa[0][ot][q] = act_fwd<ACT>(act, fa[0][ot][q]);
#pragma unroll
                    for (int s = 0; s < KSH; s++) {
                        fop[0][s] = pack_operand(fa[0][2 * s], fa[0][2 * s + 1]);
                        yop[NL - 1 - l][t][s] = fop[0][s];
                    }
                    if (l + 1 < NL) layer_products<OT, KSH, 1>(frags + (size_t)(base_fh + l * per_hidden) * 64 + lane, fop, fa);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NL; j++)
#pragma unroll
                for (int t = 0; t < NT; t++)
#pragma unroll
                    for (int s = 0; s < KSH; s++) {
                        const half_t* row = fwd + (size_t)(NL - 1 - j) * layer_stride + (size_t)(row0 + 16 * t + r) * HIDDEN + 4 * g;
                        const half4_t lo = *reinterpret_cast<const half4_t*>(row + 16 * (2 * s));
                        const half4_t hi = *reinterpret_cast<const half4_t*>(row + 16 * (2 * s + 1));
                        yop[j][t][s] = half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
        }

        // ---- dL/d(last hidden activation) = W_out^T . grad^T
#pragma unroll
        for (int ot = 0; ot < OT; ot++) {
            const half8_t a = frags[(size_t)ot * 64 + lane];
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t][ot] = mfma16(a, bg[t], zero);
        }
        // A operands (lane = output neuron, slots = the 32 batch rows) of the gradient whose matrix comes next: first dL/dOut
        half8_t Aprev[OT];
        Aprev[0] = pack_operand(mfma16(bg[0], sel0, zero), mfma16(bg[1], sel0, zero));

        half8_t bop[NT][KSH];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            // B operands (lane = input neuron, slots = batch rows) from the activations of layer NL-1-j = the inputs of the matrix
            // whose gradient Aprev holds
            half8_t Bm[OT];
#pragma unroll
            for (int s = 0; s < KSH; s++) {
                Bm[2 * s] = pack_operand(mfma16(yop[j][0][s], selP0, zero), mfma16(yop[j][1][s], selP0, zero));
                Bm[2 * s + 1] = pack_operand(mfma16(yop[j][0][s], selP1, zero), mfma16(yop[j][1][s], selP1, zero));
            }
            if (j == 0) {
#pragma unroll
                for (int b = 0; b < OT; b++) gw_out[0][b] = mfma16(Aprev[0], Bm[b], gw_out[0][b]);
            } else {
#pragma unroll
                for (int a = 0; a < OT; a++)
#pragma unroll
                    for (int b = 0; b < OT; b++) gw_hid[NL - 1 - j][a][b] = mfma16(Aprev[a], Bm[b], gw_hid[NL - 1 - j][a][b]);
            }

            // through the activation of layer NL-1-j (same arithmetic as ffmlp_dgrad_kernel), repack as next operand
#pragma unroll
            for (int t = 0; t < NT; t++) {
#pragma unroll
                for (int ot = 0; ot < OT; ot++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const half_t yv = yop[j][t][ot / 2][4 * (ot & 1) + q];
                        // ReLU: a 0/1 mask commutes with the rounding to half that pack_operand applies next, so the incoming gradient
                        // need not be narrowed and widened first (two conversions per element saved); other activations keep the
                        // dgrad kernel's order (narrow, multiply, narrow)
                        if constexpr (ACT == (int)kRelu) acc[t][ot][q] = yv > (half_t)0.0f ? acc[t][ot][q] : 0.0f;
                        else acc[t][ot][q] = act_bwd<ACT>(act, (float)(half_t)acc[t][ot][q], (float)yv);
                    }
#pragma unroll
                for (int s = 0; s < KSH; s++) bop[t][s] = pack_operand(acc[t][2 * s], acc[t][2 * s + 1]);
            }
#pragma unroll
            for (int s = 0; s < KSH; s++) {
                Aprev[2 * s] = pack_operand(mfma16(bop[0][s], selP0, zero), mfma16(bop[1][s], selP0, zero));
                Aprev[2 * s + 1] = pack_operand(mfma16(bop[0][s], selP1, zero), mfma16(bop[1][s], selP1, zero));
            }
            if (j + 1 < NL) layer_products<OT, KSH, NT>(frags + (size_t)(base_hidden + j * per_hidden) * 64 + lane, bop, acc);  // W_{NL-1-j}^T
        }

        // ---- first matrix: dW_0 += dPre_0^T . X
        {
            half8_t Bx[2 * KS0];
#pragma unroll
            for (int ks = 0; ks < KS0; ks++) {
                Bx[2 * ks] = pack_operand(mfma16(xin[0][ks], sel0, zero), mfma16(xin[1][ks], sel0, zero));
                Bx[2 * ks + 1] = pack_operand(mfma16(xin[0][ks], sel1, zero), mfma16(xin[1][ks], sel1, zero));
            }
#pragma unroll
            for (int a = 0; a < OT; a++)
#pragma unroll
                for (int b = 0; b < IT; b++) gw_in[a][b] = mfma16(Aprev[a], Bx[b], gw_in[a][b]);
        }

        if (grad_inputs) {  // dL/dX = W_0^T . dPre_0, no activation (ffmlp.cu:880-887)
            float4_t gi[NT][IT];
            layer_products<IT, KSH, NT>(frags + (size_t)base_in * 64 + lane, bop, gi);
#pragma unroll
            for (int it = 0; it < IT; it++)
#pragma unroll
                for (int t = 0; t < NT; t++) store4(grad_inputs + (size_t)(row0 + 16 * t + r) * IN + 16 * it + 4 * g, gi[t][it]);
        }
    }

    if constexpr (RECOMPUTE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the last step's spare prefetch
    // ---- combine the 4 waves in LDS (one matrix at a time, <= 16 tiles = 64 KiB), one fp32 partial row per workgroup
    float* red = reinterpret_cast<float*>(smem);
    float* out = partials + (size_t)blockIdx.x * n_params;
    flush_tiles<OT, IT>(gw_in, red, out, IN, wave, lane);
#pragma unroll
    for (int l = 0; l < NL - 1; l++) flush_tiles<OT, OT>(gw_hid[l], red, out + HIDDEN * IN + l * HIDDEN * HIDDEN, HIDDEN, wave, lane);
    flush_tiles<1, OT>(gw_out, red, out + HIDDEN * IN + (NL - 1) * HIDDEN * HIDDEN, HIDDEN, wave, lane);
}

// sum the per-workgroup partials: kRedParams consecutive parameters x (256 / kRedParams) slices of the partial list per workgroup
// (128-B coalesced reads, twice the workgroups of a 64 x 4 split: the pass is latency-bound), LDS combine of the slices, one
// fp16 store per parameter.  Fixed order: deterministic.
constexpr uint32_t kRedParams = 32, kRedSlices = kBlockThreads / kRedParams;
__global__ __launch_bounds__(kBlockThreads) void ffmlp_wgrad_reduce_kernel(const float* __restrict__ partials, uint32_t n_parts,
                                                                           uint32_t n_params, half_t* __restrict__ grad_weights) {
    __shared__ float red[kRedSlices][kRedParams];
    const uint32_t lane = threadIdx.x % kRedParams, slice = threadIdx.x / kRedParams;
    const uint32_t p = blockIdx.x * kRedParams + lane;
    float s = 0.0f;
    if (p < n_params) {
        float s0 = 0.0f, s1 = 0.0f;  // two independent chains: more loads in flight
        uint32_t k = slice;
        for (; k + kRedSlices < n_parts; k += 2 * kRedSlices) {
            s0 += partials[(size_t)k * n_params + p];
            s1 += partials[(size_t)(k + kRedSlices) * n_params + p];
        }
        if (k < n_parts) s0 += partials[(size_t)k * n_params + p];
        s = s0 + s1;
    }
    red[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && p < n_params) {
        float t = 0.0f;
#pragma unroll
        for (uint32_t i = 0; i < kRedSlices; i++) t += red[i][lane];
        grad_weights[p] = (half_t)t;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int g_num_cus = 0;
int num_cus() {
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

int validate(uint32_t B, uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
             uint32_t output_activation) {
    if (hidden_dim != 16 && hidden_dim != 32 && hidden_dim != 64 && hidden_dim != 128 && hidden_dim != 256) {
        set_error("hidden_dim should in [16, 32, 64, 128, 256]");
        return NERFTEX_ERR_INVALID;
    }
    if (input_dim == 0 || input_dim % 16 != 0) {
        set_error("FFMLP input_dim should be 16 * m (m  > 0), but got %u", input_dim);
        return NERFTEX_ERR_INVALID;
    }
    if (output_dim != 16) {
        set_error("FFMLP current only supports output dim <= 16 (padded to 16), but got %u", output_dim);
        return NERFTEX_ERR_INVALID;
    }
    if (num_layers < 2 || num_layers + 1 > (uint32_t)kMaxLayers) {
        set_error("FFMLP num_layers should be in [2, %d], but got %u", kMaxLayers - 1, num_layers);
        return NERFTEX_ERR_INVALID;
    }
    if (B % kRowsPerBlock != 0) {
        set_error("ffmlp batch size must be 128 * m (m > 0), but got %u.", B);
        return NERFTEX_ERR_INVALID;
    }
    if (activation > kNone || output_activation > kNone) {
        set_error("FFMLP: unknown activation id");
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}

size_t lds_bytes_forward(uint32_t H, uint32_t IN, uint32_t NL, bool training) {
    const size_t frags = (size_t)(frag_count(H, IN) + (NL - 1) * frag_count(H, H) + frag_count(16, H)) * 1024;
    return frags + (training ? (size_t)4 * 16 * kTilesPerWave * (H + 8) * sizeof(half_t) : 0);  // + the per-wave store patches
}
size_t lds_bytes_dgrad(uint32_t H, uint32_t IN, uint32_t NL, bool with_inputs) {
    return (size_t)(frag_count(H, 16) + (NL - 1) * frag_count(H, H) + (with_inputs ? frag_count(IN, H) : 0)) * 1024;
}
constexpr size_t kLdsLimit = 160 * 1024;

int lds_check(size_t bytes) {
    if (bytes > kLdsLimit) {
        set_error("FFMLP: the weights of this network (%zu KB as MFMA fragments) exceed the 160 KB LDS of a gfx950 CU", bytes / 1024);
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}

template <typename K>
int set_lds(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes), "hipFuncSetAttribute");
    return NERFTEX_OK;
}

uint32_t persistent_grid(uint32_t B, size_t lds) {
    const uint32_t blocks_needed = B / kRowsPerBlock;
    uint32_t per_cu = lds > 0 ? (uint32_t)(kLdsLimit / lds) : 8;
    const uint32_t wg_cap = knob(kKnobFfmlpWgPerCu) > 0 ? (uint32_t)knob(kKnobFfmlpWgPerCu) : 4u;
    per_cu = per_cu < 1 ? 1 : (per_cu > wg_cap ? wg_cap : per_cu);
    const uint32_t cap = (uint32_t)num_cus() * per_cu;
    return blocks_needed < cap ? (blocks_needed ? blocks_needed : 1) : cap;
}

template <int H, bool INF>
int launch_forward(const void* inputs, const void* weights, uint32_t B, uint32_t IN, uint32_t NL, uint32_t act, uint32_t out_act,
                   void* fwd, void* outputs, hipStream_t st) {
    const bool staged = !INF && lds_bytes_forward(H, IN, NL, true) <= kLdsLimit / 2;  // keep two workgroups per CU
    const size_t lds = lds_bytes_forward(H, IN, NL, staged);
    int rc = lds_check(lds);
    if (rc != NERFTEX_OK) return rc;
    auto kernel = staged ? ffmlp_forward_kernel<H, INF, true, -1, -1> : ffmlp_forward_kernel<H, INF, false, -1, -1>;
    if constexpr (H == 64) {  // the field's networks: ReLU inside, no output activation -> both folded in at compile time
        if (act == kRelu && out_act == kNone)
            kernel = staged ? ffmlp_forward_kernel<H, INF, true, (int)kRelu, (int)kNone> : ffmlp_forward_kernel<H, INF, false, (int)kRelu, (int)kNone>;
    }
    rc = set_lds(kernel, lds);
    if (rc != NERFTEX_OK) return rc;
    {
        KernelTimer kt(INF ? "ffmlp_inference_kernel" : "ffmlp_forward_kernel", st);
        hipLaunchKernelGGL(kernel, dim3(persistent_grid(B, lds)), dim3(kBlockThreads), lds, st, (const half_t*)inputs, (const half_t*)weights,
                           (half_t*)fwd, (half_t*)outputs, B, IN, NL, act, out_act);
    }
    return check_launch(INF ? "ffmlp_inference" : "ffmlp_forward");
}

template <bool INF>
int dispatch_forward(const void* inputs, const void* weights, uint32_t B, uint32_t IN, uint32_t H, uint32_t NL, uint32_t act,
                     uint32_t out_act, void* fwd, void* outputs, hipStream_t st) {
    switch (H) {
        case 16: return launch_forward<16, INF>(inputs, weights, B, IN, NL, act, out_act, fwd, outputs, st);
        case 32: return launch_forward<32, INF>(inputs, weights, B, IN, NL, act, out_act, fwd, outputs, st);
        case 64: return launch_forward<64, INF>(inputs, weights, B, IN, NL, act, out_act, fwd, outputs, st);
        case 128: return launch_forward<128, INF>(inputs, weights, B, IN, NL, act, out_act, fwd, outputs, st);
        default: return launch_forward<256, INF>(inputs, weights, B, IN, NL, act, out_act, fwd, outputs, st);
    }
}

template <int H>
int launch_dgrad(const void* grad, const void* weights, const void* fwd, void* bb, void* grad_inputs, uint32_t B, uint32_t IN, uint32_t NL,
                 uint32_t act, hipStream_t st) {
    const size_t lds = lds_bytes_dgrad(H, IN, NL, grad_inputs != nullptr);
    int rc = lds_check(lds);
    if (rc != NERFTEX_OK) return rc;
    auto kernel = ffmlp_dgrad_kernel<H>;
    rc = set_lds(kernel, lds);
    if (rc != NERFTEX_OK) return rc;
    {
        KernelTimer kt("ffmlp_dgrad_kernel", st);
        hipLaunchKernelGGL(kernel, dim3(persistent_grid(B, lds)), dim3(kBlockThreads), lds, st, (const half_t*)grad, (const half_t*)weights,
                           (const half_t*)fwd, (half_t*)bb, (half_t*)grad_inputs, B, IN, NL, act);
    }
    return check_launch("ffmlp_backward(dgrad)");
}

template <int H, int NL, int IT, int ACT>
int launch_fused(const void* grad, const void* inputs, const void* weights, const void* fwd, void* grad_inputs, uint32_t B, uint32_t act,
                 uint32_t n_params, void* grad_weights, hipStream_t st) {
    const bool recompute = fwd == nullptr;
    size_t lds = lds_bytes_dgrad(H, 16 * IT, NL, true) + (recompute ? (size_t)(frag_count(H, 16 * IT) + (NL - 1) * frag_count(H, H)) * 1024 : 0);
    if (lds < 64 * 1024) lds = 64 * 1024;  // the end-of-kernel combine reuses the fragment area
    if (recompute) lds += (size_t)4 * 2 * (2 + 2 * ((16 * IT + 31) / 32)) * 1024;  // prefetch ring: 4 waves x 2 slots x (grad + input pieces) KiB
    int rc = lds_check(lds);
    if (rc != NERFTEX_OK) return rc;
    auto kernel = recompute ? ffmlp_backward_fused_kernel<H, NL, IT, true, ACT> : ffmlp_backward_fused_kernel<H, NL, IT, false, ACT>;
    NERFTEX_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "hipFuncSetAttribute");
    uint32_t n_parts = B / kRowsPerBlock;
    if (n_parts > (uint32_t)num_cus()) n_parts = (uint32_t)num_cus();  // one workgroup (4 waves, one per SIMD) per CU
    float* partials = static_cast<float*>(workspace(kWsMlp, sizeof(float) * (size_t)n_parts * n_params));
    if (!partials) return NERFTEX_ERR_HIP;
    {
        KernelTimer kt(recompute ? "ffmlp_backward_recompute_kernel" : "ffmlp_backward_fused_kernel", st);
        hipLaunchKernelGGL(kernel, dim3(n_parts), dim3(kBlockThreads), lds, st, (const half_t*)grad, (const half_t*)inputs, (const half_t*)weights,
                           (const half_t*)fwd, (half_t*)grad_inputs, B, act, partials, n_params);
    }
    rc = check_launch("ffmlp_backward(fused)");
    if (rc != NERFTEX_OK) return rc;
    {
        KernelTimer kt("ffmlp_wgrad_reduce_kernel", st);
        hipLaunchKernelGGL(ffmlp_wgrad_reduce_kernel, dim3(div_up(n_params, kRedParams)), dim3(kBlockThreads), 0, st, partials, n_parts, n_params,
                           static_cast<half_t*>(grad_weights));
    }
    return check_launch("ffmlp_backward(reduce)");
}

// -1: no fused instantiation for this shape
int launch_backward_fused(const void* grad, const void* inputs, const void* weights, const void* fwd, void* grad_inputs, uint32_t B, uint32_t IN,
                          uint32_t H, uint32_t NL, uint32_t act, uint32_t n_params, void* grad_weights, hipStream_t st) {
    if (H != 64 || IN % 16 != 0 || IN > 64 || NL < 2 || NL > 4) return -1;
    // the two networks of the ngp field (32 inputs, 2 or 3 hidden layers, ReLU) get the activation folded in at compile time
    if (act == kRelu && IN == 32 && NL == 2) return launch_fused<64, 2, 2, (int)kRelu>(grad, inputs, weights, fwd, grad_inputs, B, act, n_params, grad_weights, st);
    if (act == kRelu && IN == 32 && NL == 3) return launch_fused<64, 3, 2, (int)kRelu>(grad, inputs, weights, fwd, grad_inputs, B, act, n_params, grad_weights, st);
#define NERFTEX_FUSED_CASE(nl, it) \
    if (NL == nl && IN == 16 * it) return launch_fused<64, nl, it, -1>(grad, inputs, weights, fwd, grad_inputs, B, act, n_params, grad_weights, st);
    NERFTEX_FUSED_CASE(2, 1) NERFTEX_FUSED_CASE(2, 2) NERFTEX_FUSED_CASE(2, 3) NERFTEX_FUSED_CASE(2, 4)
    NERFTEX_FUSED_CASE(3, 1) NERFTEX_FUSED_CASE(3, 2) NERFTEX_FUSED_CASE(3, 3) NERFTEX_FUSED_CASE(3, 4)
    NERFTEX_FUSED_CASE(4, 1) NERFTEX_FUSED_CASE(4, 2) NERFTEX_FUSED_CASE(4, 3) NERFTEX_FUSED_CASE(4, 4)
#undef NERFTEX_FUSED_CASE
    return -1;
}

}  // namespace
}  // namespace nerftex

using namespace nerftex;

namespace {
// the weight fragments are staged with 16-byte loads of the row-major layers (every layer starts a multiple of 256 halfs into the vector)
int weights_aligned(const void* weights) {
    if (reinterpret_cast<uintptr_t>(weights) & 15) {
        set_error("FFMLP: the weight vector must be 16-byte aligned");
        return NERFTEX_ERR_INVALID;
    }
    return NERFTEX_OK;
}
}  // namespace

extern "C" int nerftex_ffmlp_forward(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                     uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                     void* forward_buffer, void* outputs, void* stream) {
    clear_error();
    int rc = validate(B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc != NERFTEX_OK || B == 0) return rc;
    if ((rc = weights_aligned(weights)) != NERFTEX_OK) return rc;
    if (!forward_buffer) {
        set_error("ffmlp_forward: forward_buffer must not be NULL (use ffmlp_inference)");
        return NERFTEX_ERR_INVALID;
    }
    return dispatch_forward<false>(inputs, weights, B, input_dim, hidden_dim, num_layers, activation, output_activation, forward_buffer, outputs,
                                   as_stream(stream));
}

extern "C" int nerftex_ffmlp_inference(const void* inputs, const void* weights, uint32_t B, uint32_t input_dim, uint32_t output_dim,
                                       uint32_t hidden_dim, uint32_t num_layers, uint32_t activation, uint32_t output_activation,
                                       void* inference_buffer, void* outputs, void* stream) {
    (void)inference_buffer;  // the reference needs a [B, hidden] scratch; activations stay in registers here
    clear_error();
    int rc = validate(B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc != NERFTEX_OK || B == 0) return rc;
    if ((rc = weights_aligned(weights)) != NERFTEX_OK) return rc;
    return dispatch_forward<true>(inputs, weights, B, input_dim, hidden_dim, num_layers, activation, output_activation, nullptr, outputs,
                                  as_stream(stream));
}

extern "C" int nerftex_ffmlp_backward(const void* grad, const void* inputs, const void* weights, const void* forward_buffer, uint32_t B,
                                      uint32_t input_dim, uint32_t output_dim, uint32_t hidden_dim, uint32_t num_layers, uint32_t activation,
                                      uint32_t output_activation, int calc_grad_inputs, void* backward_buffer, void* grad_inputs,
                                      void* grad_weights, void* stream) {
    clear_error();
    int rc = validate(B, input_dim, output_dim, hidden_dim, num_layers, activation, output_activation);
    if (rc != NERFTEX_OK || B == 0) return rc;
    if ((rc = weights_aligned(weights)) != NERFTEX_OK) return rc;

    hipStream_t st = as_stream(stream);
    const uint32_t H = hidden_dim, IN = input_dim, NL = num_layers;
    void* gi = calc_grad_inputs ? grad_inputs : nullptr;
    const uint32_t n_params = H * (IN + H * (NL - 1) + 16);

    {   // fused activation + weight gradients (the default where instantiated): backward_buffer stays untouched
        if (!knob(kKnobFfmlpBwdSplit)) {  // ffmlp_bwd_split = 1: dgrad kernel, then wgrad kernel through backward_buffer
            rc = launch_backward_fused(grad, inputs, weights, forward_buffer, gi, B, IN, H, NL, activation, n_params, grad_weights, st);
            if (rc >= 0) return rc;  // rc < 0: shape not instantiated -> split path below
        }
    }
    if (!backward_buffer || !forward_buffer) {  // forward_buffer == NULL (recompute the activations) exists in the fused kernel only
        set_error("ffmlp_backward: forward_buffer and backward_buffer must not be NULL for this shape / mode");
        return NERFTEX_ERR_INVALID;
    }

    switch (H) {
        case 16: rc = launch_dgrad<16>(grad, weights, forward_buffer, backward_buffer, gi, B, IN, NL, activation, st); break;
        case 32: rc = launch_dgrad<32>(grad, weights, forward_buffer, backward_buffer, gi, B, IN, NL, activation, st); break;
        case 64: rc = launch_dgrad<64>(grad, weights, forward_buffer, backward_buffer, gi, B, IN, NL, activation, st); break;
        case 128: rc = launch_dgrad<128>(grad, weights, forward_buffer, backward_buffer, gi, B, IN, NL, activation, st); break;
        default: rc = launch_dgrad<256>(grad, weights, forward_buffer, backward_buffer, gi, B, IN, NL, activation, st); break;
    }
    if (rc != NERFTEX_OK) return rc;

    // weight gradients: one launch over (batch chunks, matrices), then the cross-workgroup reduction
    const size_t LS = (size_t)B * H;
    const half_t* fb = static_cast<const half_t*>(forward_buffer);
    const half_t* bb = static_cast<const half_t*>(backward_buffer);
    WgradArgs args{};
    // matrix 0: dPre = bb[NL-1], In = X
    args.layer[0] = WgradLayer{bb + (size_t)(NL - 1) * LS, static_cast<const half_t*>(inputs), H, IN, H, IN, 0};
    for (uint32_t l = 1; l < NL; l++)  // hidden matrix l: dPre = bb[NL-1-l], In = fwd[l-1]
        args.layer[l] = WgradLayer{bb + (size_t)(NL - 1 - l) * LS, fb + (size_t)(l - 1) * LS, H, H, H, H, H * IN + (l - 1) * H * H};
    args.layer[NL] = WgradLayer{static_cast<const half_t*>(grad), fb + (size_t)(NL - 1) * LS, 16, H, 16, H, H * IN + (NL - 1) * H * H};

    uint32_t n_parts = B / 128;  // every wave gets at least one 32-row step
    const uint32_t cap = (uint32_t)num_cus() / 2;
    if (n_parts > cap) n_parts = cap;
    if (n_parts == 0) n_parts = 1;
    float* partials = static_cast<float*>(workspace(kWsMlp, sizeof(float) * (size_t)n_parts * n_params));
    if (!partials) return NERFTEX_ERR_HIP;
    const size_t red_bytes = sizeof(float) * 4 * 16 * 256;  // 64 KiB
    {
        KernelTimer kt("ffmlp_wgrad_kernel", st);
        hipLaunchKernelGGL(ffmlp_wgrad_kernel, dim3(n_parts, NL + 1), dim3(kBlockThreads), red_bytes, st, args, B, partials, n_params);
    }
    rc = check_launch("ffmlp_backward(wgrad)");
    if (rc != NERFTEX_OK) return rc;
    {
        KernelTimer kt("ffmlp_wgrad_reduce_kernel", st);
        hipLaunchKernelGGL(ffmlp_wgrad_reduce_kernel, dim3(div_up(n_params, kRedParams)), dim3(kBlockThreads), 0, st, partials, n_parts,
                           n_params, static_cast<half_t*>(grad_weights));
    }
    return check_launch("ffmlp_backward(reduce)");
}

// ffmlp.cu:711-740 creates num_layers+1 side streams + events for the split-K GEMMs.  Nothing to create here:
// the weight-gradient partials live in the library workspace, sized on demand.
extern "C" int nerftex_ffmlp_allocate_splitk(size_t size) {
    (void)size;
    clear_error();
    return NERFTEX_OK;
}

extern "C" int nerftex_ffmlp_free_splitk(void) {
    clear_error();
    return NERFTEX_OK;
}
