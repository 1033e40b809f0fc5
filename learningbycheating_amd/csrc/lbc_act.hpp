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
    static __device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    static __device__ __forceinline__ float ld1(const float* p) { return *p; }
    static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};
template <> struct Act<__bf16> {
    static constexpr bool kBf16 = true;
    static __device__ __forceinline__ f32x4 ld4(const __bf16* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4); }
    static __device__ __forceinline__ void st4(__bf16* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }
    static __device__ __forceinline__ float ld1(const __bf16* p) { return (float)*p; }
    static __device__ __forceinline__ void st1(__bf16* p, float v) { *p = (__bf16)v; }
};

// launch helper: pick the instantiation from a runtime flag
#define LBC_DISPATCH_ACT(flag, KERNEL, ...)                      \
    do {                                                         \
        if (flag) { KERNEL(__bf16, __VA_ARGS__); }               \
        else      { KERNEL(float, __VA_ARGS__); }                \
    } while (0)
