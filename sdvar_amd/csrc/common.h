// Shared device/host helpers for the gfx950 (CDNA4, wave64) kernels of libsdvar_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>

#define SDVAR_OK 0
#define SDVAR_ERR_ARG 1
#define SDVAR_ERR_HIP 2
#define SDVAR_ERR_STATE 3

namespace sdvar {

void set_error(const char* fmt, ...);

#define SDVAR_CHECK_ARG(cond, ...)                        \
    do {                                                  \
        if (!(cond)) {                                    \
            sdvar::set_error(__VA_ARGS__);                \
            return SDVAR_ERR_ARG;                         \
        }                                                 \
    } while (0)

#define SDVAR_HIP(call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            sdvar::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return SDVAR_ERR_HIP;                                                               \
        }                                                                                       \
    } while (0)

#define SDVAR_LAUNCH_CHECK() SDVAR_HIP(hipGetLastError())

// Opt-in to more than 64 KiB of dynamic LDS, once per (launch site, device): one bit per device in an atomic mask, so a second
// GPU driven from the same process gets the attribute too and concurrent host threads race only on an idempotent call.
struct LdsOptIn {
    std::atomic<uint64_t> done[4];
    bool pending(int* dev) {
        if (hipGetDevice(dev) != hipSuccess) *dev = 0;
        return !((done[(*dev >> 6) & 3].load(std::memory_order_acquire) >> (*dev & 63)) & 1ull);
    }
    void mark(int dev) { done[(dev >> 6) & 3].fetch_or(1ull << (dev & 63), std::memory_order_release); }
};
#define SDVAR_LDS_OPT_IN(flag, bytes, ...)                                                                                       \
    do {                                                                                                                         \
        int dev_;                                                                                                                \
        if ((flag).pending(&dev_)) {                                                                                             \
            const void* fns_[] = {__VA_ARGS__};                                                                                  \
            for (const void* f_ : fns_) SDVAR_HIP(hipFuncSetAttribute(f_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            (flag).mark(dev_);                                                                                                   \
        }                                                                                                                        \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- wave64 reductions (DPP/shuffle based; every lane gets the result) -------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// XCD-aware remap of a 1-D block id: blocks that share an XCD (bid % 8 equal) get a contiguous range of logical
// ids, so neighbouring tiles hit the same per-XCD L2 (bijective for any grid size; cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// ---- fp32 -> three bf16 planes holding 8 significand bits each (truncation split): x == p0 + p1 + p2 exactly for
// finite normal x (each residual is exact in fp32 and the last one has at most 8 significant bits).
__device__ __forceinline__ void split3(float x, uint16_t& p0, uint16_t& p1, uint16_t& p2) {
    const uint32_t u0 = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(u0);
    const uint32_t u1 = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(u1);
    p0 = (uint16_t)(u0 >> 16); p1 = (uint16_t)(u1 >> 16); p2 = (uint16_t)(__float_as_uint(r2) >> 16);
}

// ---- fp32 -> two fp16 planes (f16x2 operands, gemm_f16x2.hip): h = fp16(x) round-to-nearest, l = fp16(x - h); a finite x saturates at the
// fp16 range instead of turning into inf - inf = NaN, and a NaN stays a NaN in both planes (v_max / v_min return their non-NaN operand: a bare
// clamp would turn an upstream divergence into a finite -65504).  Saturated / NaN elements are counted by the f16x2 guard (api.hip).
__device__ __forceinline__ float clamp_f16_range(float x) {
    const float c = fminf(fmaxf(x, -65504.0f), 65504.0f);
    return x != x ? x : c;
}
__device__ __forceinline__ void split2h(float x, uint16_t& h, uint16_t& l) {
    x = clamp_f16_range(x);
    const _Float16 hh = (_Float16)x;
    const _Float16 ll = (_Float16)(x - (float)hh);
    h = __builtin_bit_cast(uint16_t, hh); l = __builtin_bit_cast(uint16_t, ll);
}
// ---- the same split two / four values at a time, in packed instructions (v_cvt_pk_f16_f32, v_pk_add_f32): 2.5 vector instructions per value instead of ~11.
// split2h_pk_raw does NO range handling: |x| <= 65504 or NaN only (a NaN flows through both conversions by itself; an inf would turn into inf - inf).
typedef _Float16 f16x2p __attribute__((ext_vector_type(2)));
typedef float f32x2p __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2h_pk_raw(float x0, float x1, uint32_t& h, uint32_t& l) {
    const f32x2p v = {x0, x1};
    const f16x2p hh = __builtin_convertvector(v, f16x2p);               // v_cvt_pk_f16_f32: round to nearest even, as (_Float16)x
    const f32x2p r = v - __builtin_convertvector(hh, f32x2p);
    const f16x2p ll = __builtin_convertvector(r, f16x2p);
    h = __builtin_bit_cast(uint32_t, hh); l = __builtin_bit_cast(uint32_t, ll);
    // (Tried in round 3: l by v_fma_mixlo_f16 / v_fma_mixhi_f16 in inline asm - h as an fp16 source, x as an fp32 source, one rounding: 3 instead of 5 instructions
    //  per pair, bit-identical on 1 M random pairs incl. subnormal low planes, tools/micro/fma_mix_probe.hip.  Not used: hipcc pads no hazards around an asm
    //  statement (cdna_hip_programming.md 5.7) and gfx950 has a transcendental-result and a partial-register-write forwarding hazard, both of which these two
    //  instructions can meet wherever the scheduler puts them; the saving is ~2 issue slots per pair.)
}
// max(|a|, |b|, |c|) ignoring NaNs (v_max3_f32 with source modifiers: one instruction, no canonicalisation)
__device__ __forceinline__ float absmax3(float a, float b, float c) {
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(a), "v"(b), "v"(c));
    return m;
}
// Range handling for a whole wave at once: `m` = this lane's max |x| over the values it is about to split (NaNs ignored).  Almost always every lane is inside
// the fp16 range and the clamp (4 instructions per value) is skipped by a wave-uniform branch; if ANY lane is outside, every lane clamps (finite values
// saturate at +-65504, NaN stays NaN: clamp_f16_range).
__device__ __forceinline__ bool wave_needs_clamp(float m) { return __builtin_amdgcn_ballot_w64(m > 65504.0f) != 0ull; }
// four consecutive values -> the packed words of the two planes
__device__ __forceinline__ void split4h_pk(const float* v, uint2& h, uint2& l) {
    float c[4] = {v[0], v[1], v[2], v[3]};
    if (wave_needs_clamp(fmaxf(absmax3(v[0], v[1], v[2]), fabsf(v[3])))) {
#pragma unroll
        for (int e = 0; e < 4; ++e) c[e] = clamp_f16_range(v[e]);
    }
    split2h_pk_raw(c[0], c[1], h.x, l.x); split2h_pk_raw(c[2], c[3], h.y, l.y);
}

// Output-plane format of the producers of GEMM operands (ln_modulate, attention, GELU epilogues): none, three bf16 planes, two fp16 planes
enum { PLANES_NONE = 0, PLANES_F16X2 = 2, PLANES_BF16X3 = 3 };
// four consecutive values of one row -> the packed 8-byte words of each plane, written at the K-blocked position
__device__ __forceinline__ void store_planes4(uint16_t* outp, size_t ops, size_t o, const float* v, int fmt) {
    if (fmt == PLANES_F16X2) {
        uint2 h, l;
        split4h_pk(v, h, l);
        *reinterpret_cast<uint2*>(outp + o) = h;
        *reinterpret_cast<uint2*>(outp + ops + o) = l;
    } else {
        uint16_t q[3][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) split3(v[e], q[0][e], q[1][e], q[2][e]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            uint2 w;
            w.x = (uint32_t)q[k][0] | ((uint32_t)q[k][1] << 16); w.y = (uint32_t)q[k][2] | ((uint32_t)q[k][3] << 16);
            *reinterpret_cast<uint2*>(outp + k * ops + o) = w;
        }
    }
}

// eight consecutive values -> three packed bf16x8 words (same split); the top halves of two values are packed with one v_perm_b32
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split8_packed(const float* v, u32x4& a, u32x4& b, u32x4& c) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float x0 = v[2 * e], x1 = v[2 * e + 1];
        const float r0 = x0 - __uint_as_float(__float_as_uint(x0) & 0xFFFF0000u), r1 = x1 - __uint_as_float(__float_as_uint(x1) & 0xFFFF0000u);
        const float s0 = r0 - __uint_as_float(__float_as_uint(r0) & 0xFFFF0000u), s1 = r1 - __uint_as_float(__float_as_uint(r1) & 0xFFFF0000u);
        a[e] = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
        b[e] = __builtin_amdgcn_perm(__float_as_uint(r1), __float_as_uint(r0), 0x07060302u);
        c[e] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
}

// LDS-DMA of 16 bytes per lane with the address as "wave-uniform 64-bit base (SGPRs) + per-lane 32-bit byte offset (one VGPR)":
// the __builtin_amdgcn_global_load_lds lowering always builds a 64-bit vector address (v_lshl_add_u64 + moves per instruction)
// and reads the LDS address back through v_readfirstlane; vector ALU work does not hide under MFMAs on gfx950, so the K-loops
// keep all of it in scalar registers.  lds_byte_addr must be wave-uniform (M0 = LDS base of the wave's 1 KB slot).
#define SDVAR_DMA16(voff_u32, sbase_ptr, lds_byte_addr)                                                                     \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((uint32_t)(voff_u32)), "s"(sbase_ptr), \
                 "s"((uint32_t)(lds_byte_addr)) : "memory")
#define SDVAR_LDS_ADDR(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)(p))

// Planar bf16x3 tensors are K-BLOCKED: element (row, k) of plane p of a (rows x K) matrix lives at
//     plane_base[p] + ((k / 32) * rows + row) * 32 + (k % 32)
// i.e. for every block of 32 k the rows are contiguous 64-byte segments, so the (tile rows) x (32 k) slab a GEMM workgroup
// loads per K-step is ONE contiguous run of full 128-byte lines (a [rows][K] row-major plane would give half-line, 64-byte
// fragments at stride 2K: twice the L1/TA requests per byte).
__device__ __forceinline__ size_t kb_index(int row, int k, int rows) { return ((size_t)(k >> 5) * rows + row) * 32 + (k & 31); }

// ---- Philox4x32-10 (must match sdvar_amd/noise.py bit for bit) ----------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

}  // namespace sdvar
