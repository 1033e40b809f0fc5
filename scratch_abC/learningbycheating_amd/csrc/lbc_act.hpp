// Typed activation I/O.  Activations and activation gradients live in HBM either as f32 (precision 0/1) or as bf16
// (precision 2); all arithmetic on them is f32.  ld4/st4 move 4 consecutive elements (16 or 8 bytes), ld8/st8 move 8.
#pragma once
#include "lbc_common.hpp"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

template <typename T> struct Act;
template <> struct Act<float> {
    static constexpr bool kBf16 = false;
    static constexpr int kVec = 4;                 // elements per 16-byte access
    using vec = f32x4;
    using raw = f32x4;                             // 16 bytes as loaded: batched loads keep these and convert at use
    static __device__ __forceinline__ raw ldr(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ vec cvt(raw r) { return r; }
    static __device__ __forceinline__ vec ldv(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void stv(float* p, vec v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Act<__bf16> {
    static constexpr bool kBf16 = true;
    static constexpr int kVec = 8;
    using vec = f32x8;
    using raw = bf16x8;
    static __device__ __forceinline__ raw ldr(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
    static __device__ __forceinline__ vec cvt(raw r) { return __builtin_convertvector(r, f32x8); }
    static __device__ __forceinline__ vec ldv(const __bf16* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x8*>(p), f32x8); }
    static __device__ __forceinline__ void stv(__bf16* p, vec v) { *reinterpret_cast<bf16x8*>(p) = __builtin_convertvector(v, bf16x8); }
    static __device__ __forceinline__ f32x4 ld4(const __bf16* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4); }
    static __device__ __forceinline__ void st4(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
};

// LDS transpose read (ds_read_b64_tr_b16): each lane passes the address of 4 contiguous bf16; within a 16-lane group lane
// t's chunk is row t>>2, columns (t&3)*4.. of a 4 x 16 block and lane t receives column t (4 rows).  It turns a
// [pixel][channel] LDS image into the "8 consecutive pixels of one channel" fragments of a pixel-contracting MFMA.
__device__ __forceinline__ bf16x4 lds_read_tr16(const __bf16* p)
{
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)p);
}

// V consecutive per-channel f32 parameters (V = 4 or 8) as a vector
template <int V> struct ParamVec;
template <> struct ParamVec<4> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ f32x4 splat(float v) { return f32x4{v, v, v, v}; }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct ParamVec<8> {
    static __device__ __forceinline__ f32x8 ld(const float* p)
    {
        const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
        return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    static __device__ __forceinline__ f32x8 splat(float v) { return f32x8{v, v, v, v, v, v, v, v}; }
    static __device__ __forceinline__ void st(float* p, f32x8 v)
    {
        *reinterpret_cast<f32x4*>(p) = __builtin_shufflevector(v, v, 0, 1, 2, 3);
        *reinterpret_cast<f32x4*>(p + 4) = __builtin_shufflevector(v, v, 4, 5, 6, 7);
    }
};

// Micro-tile index -> (channel group, pixel group) of the transposing weight-gradient staging (conv_wgrad.hip explains
// the bank-conflict reasoning): KB channel-group bits and 4-KB pixel-group bits in the low four lane bits.
template <int CG, int KB>
__device__ __forceinline__ void wgrad_tile_coord(int u, int& cg, int& pg)
{
    constexpr int LCG = CG == 32 ? 5 : CG == 16 ? 4 : CG == 8 ? 3 : 2;
    static_assert(KB <= LCG && KB <= 4, "wgrad: lane mapping");
    cg = (u & ((1 << KB) - 1)) | (((u >> 4) & ((1 << (LCG - KB)) - 1)) << KB);
    pg = ((u >> KB) & ((1 << (4 - KB)) - 1)) | (((u >> (4 + LCG - KB)) & ((1 << KB) - 1)) << (4 - KB));
}

// launch helper: pick the instantiation from a runtime flag
#define LBC_DISPATCH_ACT(flag, KERNEL, ...)                      \
    do {                                                         \
        if (flag) { KERNEL(__bf16, __VA_ARGS__); }               \
        else      { KERNEL(float, __VA_ARGS__); }                \
    } while (0)
