// TEST INFRASTRUCTURE ONLY -- never part of the product path.
//
// A CPU emulation of the handful of HIP device-side constructs that
// learningbycheating_amd/csrc/*.hip uses, so that the *unmodified* kernel
// sources can be compiled with the host clang++ (this directory shadows
// <hip/hip_runtime.h>) and their indexing / masking / MFMA-fragment logic can
// be exercised on a machine with no GPU.  Every workgroup is run as a set of
// cooperatively scheduled fibers (one per HIP thread); __syncthreads(), wave
// shuffles and MFMA are rendezvous points.  The MFMA emulation implements the
// gfx950 lane<->element maps documented in the CDNA4 guide (A[i=l&31][k=l>>5],
// B[k=l>>5][j=l&31], D row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31 for 32x32x2f32)
// with the exact k-ordered fmaf chain of the hardware.
//
// The product loader (learningbycheating_amd/_lib.py) only ever loads the real
// gfx950 library; the emulated library is built and loaded by tests/emu only.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define LBC_HIP_EMULATED_FOR_TESTS 1

// ---- qualifiers -----------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local   /* a workgroup runs on one host thread; workgroups run on several (emu_runtime.cpp) */
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__ __restrict
#endif

// ---- basic types ----------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };
extern thread_local emu_uint3 threadIdx, blockIdx;
extern thread_local dim3 blockDim, gridDim;

struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(4) uchar4 { unsigned char x, y, z, w; };
struct alignas(8) ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline ushort4 make_ushort4(unsigned short x, unsigned short y, unsigned short z, unsigned short w) { return {x, y, z, w}; }

typedef int hipError_t;
typedef struct emu_stream_t* hipStream_t;
typedef struct emu_event_t* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipError(emu)"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
struct hipPointerAttribute_t { int device; };
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->device = 0; return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
#define hipEventDisableTiming 0x2
#define hipStreamNonBlocking 0x1
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

// ---- fiber runtime (emu_runtime.cpp) ---------------------------------------
namespace emu {
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_block();
// Wave rendezvous: returns pointer to the 64-slot exchange area valid until the
// lane's next rendezvous.  Each lane first writes its own slot of `mine`.
struct alignas(16) Slot { unsigned char b[64]; };
Slot* wave_slots_begin();   // slot array (64) this lane must write its slot into
void wave_rendezvous();     // blocks until all 64 lanes of the wave arrived
int lane_id();
}  // namespace emu

template <typename K, typename... Args>
static inline void emu_launch_kernel(K kernel, dim3 grid, dim3 block, Args... args) {
    ::emu::launch(grid, block, [&]() { kernel(args...); });
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    emu_launch_kernel(kernel, dim3(grid), dim3(block), __VA_ARGS__)

static inline void __syncthreads() { emu::sync_block(); }

// ---- wave collectives -------------------------------------------------------
template <typename T>
static inline T emu_wave_read(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "slot too small");
    emu::Slot* s = emu::wave_slots_begin();
    memcpy(s[emu::lane_id()].b, &v, sizeof(T));
    emu::wave_rendezvous();
    T r;
    memcpy(&r, s[src_lane & 63].b, sizeof(T));
    return r;
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
    int l = emu::lane_id();
    int src = l ^ mask;
    if ((src / width) != (l / width)) src = l;
    return emu_wave_read(v, src);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int src = l + (int)d;
    if ((src / width) != (l / width)) src = l;
    return emu_wave_read(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
    int l = emu::lane_id();
    int src = l - (int)d;
    if (src < 0 || (src / width) != (l / width)) src = l;
    return emu_wave_read(v, src);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
    int l = emu::lane_id();
    return emu_wave_read(v, (l / width) * width + (src % width));
}

// ---- MFMA -------------------------------------------------------------------
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));

static inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c, int, int, int) {
    struct AB { float a, b; };
    emu::Slot* s = emu::wave_slots_begin();
    int l = emu::lane_id();
    AB me{a, b};
    memcpy(s[l].b, &me, sizeof(me));
    emu::wave_rendezvous();
    emu_f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            AB A, B;
            memcpy(&A, s[row + 32 * k].b, sizeof(A));   // A[i=row][k] lives in lane row+32k (.a)
            memcpy(&B, s[col + 32 * k].b, sizeof(B));   // B[k][j=col] lives in lane col+32k (.b)
            acc = fmaf(A.a, B.b, acc);
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu_mfma_f32_32x32x2f32

static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
    struct AB { float a, b; };
    emu::Slot* s = emu::wave_slots_begin();
    int l = emu::lane_id();
    AB me{a, b};
    memcpy(s[l].b, &me, sizeof(me));
    emu::wave_rendezvous();
    emu_f32x4 d = c;
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            AB A, B;
            memcpy(&A, s[row + 16 * k].b, sizeof(A));
            memcpy(&B, s[col + 16 * k].b, sizeof(B));
            acc = fmaf(A.a, B.b, acc);
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu_mfma_f32_16x16x4f32

typedef __bf16 emu_bf16x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_bf16: lane l holds A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31], j=0..7; products are exact in
// f32, accumulation in f32 (the hardware's internal summation order is not specified: tests use a tolerance).
static inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c, int, int, int) {
    struct AB { float a[8], b[8]; };
    static_assert(sizeof(AB) <= 64, "slot");
    emu::Slot* s = emu::wave_slots_begin();
    int l = emu::lane_id();
    AB me;
    for (int j = 0; j < 8; ++j) { me.a[j] = (float)a[j]; me.b[j] = (float)b[j]; }
    memcpy(s[l].b, &me, sizeof(me));
    emu::wave_rendezvous();
    emu_f32x16 d = c;
    int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int kh = 0; kh < 2; ++kh) {
            AB A, B;
            memcpy(&A, s[row + 32 * kh].b, sizeof(A));
            memcpy(&B, s[col + 32 * kh].b, sizeof(B));
            for (int j = 0; j < 8; ++j) acc += A.a[j] * B.b[j];
        }
        d[r] = acc;
    }
    return d;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu_mfma_f32_32x32x16_bf16

// ds_read_b64_tr_b16 (gfx950): every lane supplies the address of 4 contiguous 16-bit elements; within a 16-lane group
// lane t's chunk is B[t>>2][(t&3)*4 .. +3] of a 4 x 16 matrix and lane t receives column t (checked against the
// hardware by scripts/probe/tr16_probe.hip).
typedef __bf16 emu_bf16x4 __attribute__((ext_vector_type(4)));
static inline emu_bf16x4 emu_ds_read_tr16_b64(const void* p) {
    emu::Slot* s = emu::wave_slots_begin();
    const int l = emu::lane_id();
    memcpy(s[l].b, &p, sizeof(p));
    emu::wave_rendezvous();
    emu_bf16x4 r;
    for (int j = 0; j < 4; ++j) {
        const __bf16* q;
        memcpy(&q, s[(l & ~15) + j * 4 + ((l & 15) >> 2)].b, sizeof(q));
        r[j] = q[(l & 15) & 3];
    }
    return r;
}
#define __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p) emu_ds_read_tr16_b64((const void*)(p))

// ---- misc device builtins ----------------------------------------------------
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
#define __expf(x) expf(x)
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __fdividef(float a, float b) { return a / b; }
#define __builtin_amdgcn_s_setprio(x) ((void)0)
// counted waits have nothing to wait for here (every emulated load completes at once); the raw workgroup barrier is the
// fiber rendezvous.  NB: the emulator therefore checks indexing and arithmetic of a software pipeline, never its races.
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_s_barrier() emu::sync_block()
// global_load_lds_dwordx4 (LDS-DMA): lane l's 16 bytes land at the wave-uniform LDS base + 16 * l
static inline void emu_global_load_lds(const void* g, void* lds_wave_base, unsigned size) {
    memcpy(static_cast<char*>(lds_wave_base) + (size_t)emu::lane_id() * size, g, size);
}
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) emu_global_load_lds((const void*)(g), (void*)(l), (size))
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
// wave-private LDS hand-offs (a wave's lanes are fibers here: make them meet where hardware lanes run in lock-step)
#define __builtin_amdgcn_wave_barrier() emu::wave_rendezvous()
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __builtin_amdgcn_readfirstlane(v) emu_wave_read((v), 0)
