// Which instructions give a wrong result while ANOTHER wave on the same CU runs MFMAs?  (round 5)
//
// tools/probes/k3d_reduce.hip showed the round-4 fault of the hash-grid record builder with nothing but loads + packed-fp32 arithmetic in the
// victim, and ONLY beside a neighbour that issues v_mfma (not fp32 / fp64 / packed-fp32 / integer / transcendental / LDS / memory neighbours).
// This probe asks the question per instruction: every lane executes the SAME instruction twice on the same operands (two asm volatile statements)
// and counts the times the two results differ -- a transient fault shows as a difference whatever the instruction computes, no reference
// arithmetic needed.  Each instruction form runs alone and beside an MFMA kernel on a second stream of this process.
//   build:  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/probes/pk_mfma_probe.hip -o tools/probes/_bin/pk_mfma_probe
//   run:    pk_mfma_probe [seconds per case, default 1.5]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

enum Op {
    PK_MUL, PK_MUL_OPSEL, PK_ADD, PK_ADD_NEG, PK_ADD_ONE, PK_FMA, PK_MOV,   // the packed-fp32 family (target feature packed-fp32-ops)
    MUL_F32, FMA_F32, FMA_F64, MUL_F64, ADD_F64,                              // plain fp32 (control) and fp64
    PK_FMA_F16, PK_MUL_F16, PK_ADD_F16, CVT_PK_F16, CVT_PK_BF16, FMA_MIX, FMA_MIXLO, DOT2C,  // 16-bit forms the library contains
    MUL_LO_U32, MAD_U64, EXP_F32, RCP_F32, DPP_ROW_SHL, CVT_F64_F32,
    CHAIN_K3D, CHAIN_K3D_NOPS, CHAIN_K3D_FRESH_REGS, CHAIN_SCALAR, CHAIN_K3D_AFTER_LOAD, CHAIN_PK3_PLAIN,   // dependent chains (round 5, second pass)
    // third pass: producer -> consumer PAIRS, which link of the chain is it?  (all beside the MFMA neighbours; N = s_nop between the two)
    PAIR_ADD_MULSWAP, PAIR_ADD_MULSWAP_NOP7, PAIR_ADD_MULSWAP_NOP15, PAIR_ADD_MULSWAP_INDEP, PAIR_ADD_MULPLAIN, PAIR_ADD_ADDONE, PAIR_FLOOR_ADD, PAIR_ADD_SCALARMUL,
    PAIR_FMA_ADD, PAIR_MOVS_MULSWAP, N_OPS
};
static const char* kNames[N_OPS] = {
    "v_pk_mul_f32", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32", "v_pk_add_f32 neg_lo/neg_hi", "v_pk_add_f32 v, 1.0 op_sel_hi:[1,0] neg", "v_pk_fma_f32",
    "v_pk_mov_b32 op_sel:[1,0]", "v_mul_f32 (control)", "v_fma_f32 (control)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_pk_fma_f16", "v_pk_mul_f16", "v_pk_add_f16",
    "v_cvt_pk_f16_f32", "v_cvt_pk_bf16_f32", "v_fma_mix_f32", "v_fma_mixlo_f16 (writes 16 of 32 bits: differs alone too, an artefact)", "v_dot2c_f32_f16", "v_mul_lo_u32", "v_mad_u64_u32", "v_exp_f32", "v_rcp_f32",
    "v_mov_b32_dpp row_shl:1 (inline asm: no hazard nops -- differs alone too, an artefact)", "v_cvt_f64_f32",
    "CHAIN as compiled into K3d: pk_fma(sgpr,0.5) floor floor pk_add(neg) pk_add(1.0) 3 x pk_mul(op_sel), registers reused", "the same chain, s_nop 1 after every instruction",
    "the same chain, every result in a fresh register pair", "the same arithmetic with single instructions (control)", "the K3d chain right behind the global load of its input + s_waitcnt",
    "three dependent plain v_pk_mul_f32",
    "PAIR pk_add(neg) -> pk_mul op_sel:[0,1] op_sel_hi:[1,0] of its result (halves swapped)", "the same pair, s_nop 7 between", "the same pair, 2 x s_nop 7 between",
    "pk_add(neg) then pk_mul op_sel of an OLDER register (no dependency)", "PAIR pk_add(neg) -> plain pk_mul of its result", "PAIR pk_add(neg) -> pk_add(1.0 op_sel_hi:[1,0]) of its result",
    "PAIR 2 x v_floor_f32 -> pk_add(neg) of their results", "PAIR pk_add(neg) -> v_mul_f32 hi, lo of its result", "PAIR pk_fma(sgpr pair, 0.5 op_sel_hi:[1,0,0]) -> pk_add(neg) of its result",
    "PAIR 2 x v_mov_b32 -> pk_mul op_sel of the moved pair (single-instruction producers)"};

template <typename T>
__device__ __forceinline__ bool bits_differ(const T& x, const T& y) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(uint32_t, x) != __builtin_bit_cast(uint32_t, y);
    else return __builtin_bit_cast(unsigned long long, x) != __builtin_bit_cast(unsigned long long, y);
}

template <int OP>
__device__ __forceinline__ bool twice_differs(f2 a, f2 b, f2 c) {
    // 64-bit views of the operands for the fp64 / u64 forms
    const double da = (double)a[0] * 1.0000001 + (double)a[1], db = (double)b[0] + (double)b[1] * 0.999999, dc = (double)c[0];
    const uint32_t ua = __float_as_uint(a[0]), ub = __float_as_uint(b[1]);
    const h2 ha = {(_Float16)a[0], (_Float16)a[1]}, hb = {(_Float16)b[0], (_Float16)b[1]}, hc = {(_Float16)c[0], (_Float16)c[1]};
#define TWICE2(T, INS, ...) { T r1, r2; asm volatile(INS : "=v"(r1) : __VA_ARGS__); asm volatile(INS : "=v"(r2) : __VA_ARGS__); return bits_differ(r1, r2); }
    if constexpr (OP == PK_MUL) TWICE2(f2, "v_pk_mul_f32 %0, %1, %2", "v"(a), "v"(b))
    else if constexpr (OP == PK_MUL_OPSEL) TWICE2(f2, "v_pk_mul_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]", "v"(a))
    else if constexpr (OP == PK_ADD) TWICE2(f2, "v_pk_add_f32 %0, %1, %2", "v"(a), "v"(b))
    else if constexpr (OP == PK_ADD_NEG) TWICE2(f2, "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]", "v"(a), "v"(b))
    else if constexpr (OP == PK_ADD_ONE) TWICE2(f2, "v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]", "v"(a))
    else if constexpr (OP == PK_FMA) TWICE2(f2, "v_pk_fma_f32 %0, %1, %2, %3", "v"(a), "v"(b), "v"(c))
    else if constexpr (OP == PK_MOV) TWICE2(f2, "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]", "v"(a), "v"(b))
    else if constexpr (OP == MUL_F32) TWICE2(float, "v_mul_f32 %0, %1, %2", "v"(a[0]), "v"(b[0]))
    else if constexpr (OP == FMA_F32) TWICE2(float, "v_fma_f32 %0, %1, %2, %3", "v"(a[0]), "v"(b[0]), "v"(c[0]))
    else if constexpr (OP == FMA_F64) TWICE2(double, "v_fma_f64 %0, %1, %2, %3", "v"(da), "v"(db), "v"(dc))
    else if constexpr (OP == MUL_F64) TWICE2(double, "v_mul_f64 %0, %1, %2", "v"(da), "v"(db))
    else if constexpr (OP == ADD_F64) TWICE2(double, "v_add_f64 %0, %1, %2", "v"(da), "v"(db))
    else if constexpr (OP == PK_FMA_F16) TWICE2(h2, "v_pk_fma_f16 %0, %1, %2, %3", "v"(ha), "v"(hb), "v"(hc))
    else if constexpr (OP == PK_MUL_F16) TWICE2(h2, "v_pk_mul_f16 %0, %1, %2", "v"(ha), "v"(hb))
    else if constexpr (OP == PK_ADD_F16) TWICE2(h2, "v_pk_add_f16 %0, %1, %2", "v"(ha), "v"(hb))
    else if constexpr (OP == CVT_PK_F16) TWICE2(uint32_t, "v_cvt_pk_f16_f32 %0, %1, %2", "v"(a[0]), "v"(a[1]))
    else if constexpr (OP == CVT_PK_BF16) TWICE2(uint32_t, "v_cvt_pk_bf16_f32 %0, %1, %2", "v"(a[0]), "v"(a[1]))
    else if constexpr (OP == FMA_MIX) TWICE2(float, "v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]", "v"(ha), "v"(hb), "v"(c[0]))
    else if constexpr (OP == FMA_MIXLO) TWICE2(uint32_t, "v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,1,0]", "v"(ha), "v"(hb), "v"(c[0]))
    else if constexpr (OP == DOT2C) { float r1 = c[0], r2 = c[0]; asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r1) : "v"(ha), "v"(hb)); asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r2) : "v"(ha), "v"(hb)); return __float_as_uint(r1) != __float_as_uint(r2); }
    else if constexpr (OP == MUL_LO_U32) TWICE2(uint32_t, "v_mul_lo_u32 %0, %1, %2", "v"(ua), "v"(ub))
    else if constexpr (OP == MAD_U64) { unsigned long long r1, r2, cc = ((unsigned long long)ub << 32) | ua; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r1) : "v"(ua), "v"(ub), "v"(cc) : "vcc"); asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r2) : "v"(ua), "v"(ub), "v"(cc) : "vcc"); return r1 != r2; }
    else if constexpr (OP == EXP_F32) TWICE2(float, "v_exp_f32 %0, %1", "v"(c[0]))
    else if constexpr (OP == RCP_F32) TWICE2(float, "v_rcp_f32 %0, %1", "v"(a[0]))
    else if constexpr (OP == DPP_ROW_SHL) TWICE2(float, "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", "v"(a[0]))
    else TWICE2(double, "v_cvt_f64_f32 %0, %1", "v"(a[0]))
#undef TWICE2
}

// ---- dependent chains, as one asm block each, executed twice.  Fixed registers v[40:53] (clobbered); %3 = the per-level scale pair in SGPRs
#define K3D_BODY(NOP)                                                                                   \
    "v_pk_fma_f32 v[42:43], v[40:41], %3, 0.5 op_sel_hi:[1,0,0]\n" NOP                                  \
    "v_floor_f32 v44, v42\n" NOP "v_floor_f32 v45, v43\n" NOP                                           \
    "v_pk_add_f32 v[46:47], v[42:43], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n" NOP                         \
    "v_pk_add_f32 v[42:43], v[46:47], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n" NOP              \
    "v_pk_mul_f32 v[48:49], v[42:43], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n" NOP                      \
    "v_pk_mul_f32 v[50:51], v[42:43], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n" NOP                      \
    "v_pk_mul_f32 v[46:47], v[46:47], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n" NOP
#define K3D_OUT "s_nop 4\n v_mov_b32 %0, v48\n v_mov_b32 %1, v50\n v_mov_b32 %2, v46\n"
#define K3D_CLOB "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53"
template <int OP>
__device__ __forceinline__ bool chain_differs(f2 a, f2 sc, const float* mem) {
    float r[2][3];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        if constexpr (OP == CHAIN_K3D)
            asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n s_nop 4\n" K3D_BODY("") K3D_OUT
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(a[0]), "v"(a[1]) : K3D_CLOB);
        else if constexpr (OP == CHAIN_K3D_NOPS)
            asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n s_nop 4\n" K3D_BODY("s_nop 1\n") K3D_OUT
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(a[0]), "v"(a[1]) : K3D_CLOB);
        else if constexpr (OP == CHAIN_K3D_FRESH_REGS)
            asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n s_nop 4\n"
                         "v_pk_fma_f32 v[42:43], v[40:41], %3, 0.5 op_sel_hi:[1,0,0]\n"
                         "v_floor_f32 v44, v42\n v_floor_f32 v45, v43\n"
                         "v_pk_add_f32 v[46:47], v[42:43], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n"
                         "v_pk_add_f32 v[52:53], v[46:47], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n"
                         "v_pk_mul_f32 v[48:49], v[52:53], v[52:53] op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_mul_f32 v[50:51], v[52:53], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "v_pk_mul_f32 v[40:41], v[46:47], v[46:47] op_sel:[0,1] op_sel_hi:[1,0]\n"
                         "s_nop 4\n v_mov_b32 %0, v48\n v_mov_b32 %1, v50\n v_mov_b32 %2, v40\n"
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(a[0]), "v"(a[1]) : K3D_CLOB);
        else if constexpr (OP == CHAIN_SCALAR)
            asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n s_nop 4\n"
                         "v_fma_f32 v42, v40, %3, 0.5\n v_fma_f32 v43, v41, %3, 0.5\n v_floor_f32 v44, v42\n v_floor_f32 v45, v43\n"
                         "v_sub_f32 v46, v42, v44\n v_sub_f32 v47, v43, v45\n v_sub_f32 v42, 1.0, v46\n v_sub_f32 v43, 1.0, v47\n"
                         "v_mul_f32 v48, v42, v43\n v_mul_f32 v50, v42, v47\n v_mul_f32 v46, v46, v47\n" K3D_OUT
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc[0]), "v"(a[0]), "v"(a[1]) : K3D_CLOB);
        else if constexpr (OP == CHAIN_K3D_AFTER_LOAD)
            asm volatile("global_load_dwordx2 v[40:41], %4, off\n s_waitcnt vmcnt(0)\n" K3D_BODY("") K3D_OUT
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(mem), "v"(a[1]) : K3D_CLOB, "memory");
        else if constexpr (OP >= PAIR_ADD_MULSWAP) {
#define PAIR(BODY) asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n v_mov_b32 v44, %5\n v_mov_b32 v45, %4\n s_nop 7\n" BODY "s_nop 7\n v_mov_b32 %0, v48\n v_mov_b32 %1, v49\n v_mov_b32 %2, v42\n" \
                                : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(a[0]), "v"(a[1]) : K3D_CLOB)
#define SWAPMUL "v_pk_mul_f32 v[48:49], v[42:43], v[42:43] op_sel:[0,1] op_sel_hi:[1,0]\n"
#define ADDNEG "v_pk_add_f32 v[42:43], v[40:41], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n"
            if constexpr (OP == PAIR_ADD_MULSWAP) PAIR(ADDNEG SWAPMUL);
            else if constexpr (OP == PAIR_ADD_MULSWAP_NOP7) PAIR(ADDNEG "s_nop 7\n" SWAPMUL);
            else if constexpr (OP == PAIR_ADD_MULSWAP_NOP15) PAIR(ADDNEG "s_nop 7\n s_nop 7\n" SWAPMUL);
            else if constexpr (OP == PAIR_ADD_MULSWAP_INDEP) PAIR(ADDNEG "v_pk_mul_f32 v[48:49], v[40:41], v[40:41] op_sel:[0,1] op_sel_hi:[1,0]\n");
            else if constexpr (OP == PAIR_ADD_MULPLAIN) PAIR(ADDNEG "v_pk_mul_f32 v[48:49], v[42:43], v[42:43]\n");
            else if constexpr (OP == PAIR_ADD_ADDONE) PAIR(ADDNEG "v_pk_add_f32 v[48:49], v[42:43], 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n");
            else if constexpr (OP == PAIR_FLOOR_ADD) PAIR("v_floor_f32 v42, v40\n v_floor_f32 v43, v41\n v_pk_add_f32 v[48:49], v[40:41], v[42:43] neg_lo:[0,1] neg_hi:[0,1]\n");
            else if constexpr (OP == PAIR_ADD_SCALARMUL) PAIR(ADDNEG "v_mul_f32 v48, v43, v42\n v_mov_b32 v49, v48\n");
            else if constexpr (OP == PAIR_FMA_ADD) PAIR("v_pk_fma_f32 v[42:43], v[40:41], %3, 0.5 op_sel_hi:[1,0,0]\n v_pk_add_f32 v[48:49], v[42:43], v[44:45] neg_lo:[0,1] neg_hi:[0,1]\n");
            else PAIR("v_mov_b32 v42, v41\n v_mov_b32 v43, v40\n" SWAPMUL);
#undef PAIR
#undef SWAPMUL
#undef ADDNEG
        } else
            asm volatile("v_mov_b32 v40, %4\n v_mov_b32 v41, %5\n s_nop 4\n"
                         "v_pk_mul_f32 v[42:43], v[40:41], %3\n v_pk_mul_f32 v[44:45], v[42:43], v[40:41]\n v_pk_mul_f32 v[46:47], v[44:45], v[42:43]\n"
                         "s_nop 4\n v_mov_b32 %0, v46\n v_mov_b32 %1, v47\n v_mov_b32 %2, v44\n"
                         : "=v"(r[k][0]), "=v"(r[k][1]), "=v"(r[k][2]) : "s"(sc), "v"(a[0]), "v"(a[1]) : K3D_CLOB);
    }
    return __float_as_uint(r[0][0]) != __float_as_uint(r[1][0]) || __float_as_uint(r[0][1]) != __float_as_uint(r[1][1]) || __float_as_uint(r[0][2]) != __float_as_uint(r[1][2]);
}

template <int OP>
__global__ __launch_bounds__(1024) void victim(uint32_t iters, uint32_t* __restrict__ stats, uint32_t* __restrict__ first, const float* __restrict__ mem, const f2 scale_arg) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    f2 a = {0.37f + (float)(t & 1023) * 1e-3f, 0.81f + (float)(t >> 10) * 1e-4f}, b = {1.25f, 0.61f + (float)(t & 63) * 1e-2f}, c = {1e-3f * (float)(t & 255), -0.5f};
    uint32_t bad = 0, when = 0;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            bool d;
            if constexpr (OP >= CHAIN_K3D) d = chain_differs<OP>(a, scale_arg, mem + 2 * ((t + i * 8 + u) & 0xfffff));
            else d = twice_differs<OP>(a, b, c);
            if (d) { bad++; when = i * 8 + u; }
            a[0] += 0.37f; a[1] = a[1] * 0.999f + 0.013f; b[0] -= 1e-3f; b[1] += 0.61f; c[0] = c[0] * 0.5f + 0.1f; c[1] += 0.01f;
            if (a[0] > 100.0f) { a[0] -= 99.5f; b[1] -= 60.0f; }
        }
    }
    if (bad) {
        const uint32_t k = atomicAdd(stats, bad);
        atomicAdd(stats + 1, 1u);
        if (k < 8) { first[2 * k] = t; first[2 * k + 1] = when; }
    }
}

template <int KIND>
__global__ __launch_bounds__(256) void mfma_busy(float* __restrict__ sink, uint32_t iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc;
    if (KIND == 0) {
        h8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(0.01f * (float)((t + k) & 7)); b[k] = (_Float16)(0.02f * (float)((t * 3 + k) & 7)); }
        f4 c0 = {0, 0, 0, 0}, c1 = {1, 1, 1, 1};
        for (uint32_t i = 0; i < iters; i++) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0); }
        acc = c0[0] + c0[3] + c1[1] + c1[2];
    } else if (KIND == 1) {
        h8 a, b;
        for (int k = 0; k < 8; k++) { a[k] = (_Float16)(0.01f * (float)((t + k) & 7)); b[k] = (_Float16)(0.02f * (float)((t * 3 + k) & 7)); }
        f16v c0 = {0};
        for (uint32_t i = 0; i < iters; i++) c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        acc = c0[0] + c0[15];
    } else {
        f4 c0 = {0, 0, 0, 0};
        const float a = 0.01f * (float)(t & 7), b = 0.02f * (float)(t & 3);
        for (uint32_t i = 0; i < iters; i++) c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);  // the fp32 MFMA
        acc = c0[0] + c0[3];
    }
    if (acc == 12345.678f) sink[0] = acc;
}

static float* g_mem = nullptr;  // 2 M floats in [0, 1): the inputs of CHAIN_K3D_AFTER_LOAD
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

template <int OP>
static void run_case(double secs, int neighbour, hipStream_t mainst, hipStream_t side, float* sink, uint32_t* stats, uint32_t* first) {
    CK(hipMemsetAsync(stats, 0, 8, mainst));
    const double t0 = now();
    long launches = 0;
    while (now() - t0 < secs) {
        for (int l = 0; l < 8; l++) {
            if (neighbour >= 0 && (l & 1) == 0) {
                if (neighbour == 0) hipLaunchKernelGGL(mfma_busy<0>, dim3(1024), dim3(256), 0, side, sink, 20000u);
                else if (neighbour == 1) hipLaunchKernelGGL(mfma_busy<1>, dim3(1024), dim3(256), 0, side, sink, 10000u);
                else hipLaunchKernelGGL(mfma_busy<2>, dim3(1024), dim3(256), 0, side, sink, 20000u);
            }
            hipLaunchKernelGGL(victim<OP>, dim3(2048), dim3(1024), 0, mainst, 64u, stats, first, g_mem, f2{127.0f, 127.0f});
            launches++;
        }
        CK(hipStreamSynchronize(mainst));
    }
    CK(hipDeviceSynchronize());
    uint32_t st[2], f[16];
    CK(hipMemcpy(st, stats, 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(f, first, 64, hipMemcpyDeviceToHost));
    const char* nb = neighbour < 0 ? "none" : neighbour == 0 ? "v_mfma_f32_16x16x32_f16" : neighbour == 1 ? "v_mfma_f32_32x32x16_f16" : "v_mfma_f32_16x16x4_f32";
    const double execs = (double)launches * 2048.0 * 1024.0 * 64.0 * 8.0 * 2.0;
    printf("{\"instruction\": \"%s\", \"neighbour\": \"%s\", \"launches\": %ld, \"lane_executions\": %.3g, \"differing_pairs\": %u, \"threads_with_a_difference\": %u",
           kNames[OP], nb, launches, execs, st[0], st[1]);
    if (st[0]) printf(", \"first_thread\": %u, \"first_lane\": %u", f[0], f[0] % 64);
    printf("}\n");
    fflush(stdout);
}

template <int OP>
static void run_all(double secs, bool all_neighbours, hipStream_t mainst, hipStream_t side, float* sink, uint32_t* stats, uint32_t* first) {
    run_case<OP>(secs * 0.3, -1, mainst, side, sink, stats, first);
    run_case<OP>(secs, 0, mainst, side, sink, stats, first);
    if (all_neighbours) {
        run_case<OP>(secs, 1, mainst, side, sink, stats, first);
        run_case<OP>(secs, 2, mainst, side, sink, stats, first);
    }
    if constexpr (OP + 1 < N_OPS) run_all<OP + 1>(secs, all_neighbours, mainst, side, sink, stats, first);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    const bool all_nb = argc > 2 && !strcmp(argv[2], "all");
    float* sink; uint32_t *stats, *first;
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&stats, 8)); CK(hipMalloc(&first, 64));
    {
        const size_t n = (size_t)2 << 20;
        float* h = (float*)malloc(n * 4);
        srand(3);
        for (size_t i = 0; i < n; i++) h[i] = (float)rand() / ((float)RAND_MAX + 1.0f);
        CK(hipMalloc(&g_mem, n * 4));
        CK(hipMemcpy(g_mem, h, n * 4, hipMemcpyHostToDevice));
        free(h);
    }
    const int first_op = argc > 3 ? atoi(argv[3]) : 0;
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t mainst, side;
    CK(hipStreamCreateWithFlags(&mainst, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    if (first_op >= PAIR_ADD_MULSWAP) run_all<PAIR_ADD_MULSWAP>(secs, all_nb, mainst, side, sink, stats, first);  // (the pairs only)
    else if (first_op >= CHAIN_K3D) run_all<CHAIN_K3D>(secs, all_nb, mainst, side, sink, stats, first);  // (the chains and the pairs)
    else run_all<0>(secs, all_nb, mainst, side, sink, stats, first);
    return 0;
}
