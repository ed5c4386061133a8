// Device-side helpers shared by the kernel translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include <type_traits>
#include <utility>

namespace vfem {

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0..N-1 (indices stay constant expressions however large the body)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// opaque copy of a wave-uniform pointer: loads through the result cannot be hoisted above this point; readfirstlane
// restores the uniformity that an asm output loses (otherwise the loads become per-lane vector loads)
template <class T>
__device__ __forceinline__ const T *launder_uniform(const T *p) {
    unsigned long long v = reinterpret_cast<unsigned long long>(p);
    asm volatile("" : "+s"(v));
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) (v & 0xffffffffull));
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return reinterpret_cast<const T *>(((unsigned long long) hi << 32) | lo);
}


typedef double d8_t __attribute__((ext_vector_type(8)));

// Wave-uniform coefficient tables are read with explicit scalar loads issued exactly where they are consumed (the
// compiler would otherwise hoist all 576 coefficient loads to the top of the kernel and spill them through
// v_writelane/v_readlane).
typedef double d4_t __attribute__((ext_vector_type(4)));
// 12 consecutive doubles, requested and awaited inside ONE asm statement: the compiler treats asm outputs as complete
// when the statement ends, so a load left in flight between two statements could land in SGPRs it has already spilled
// and reassigned (observed as a wild address once the allocator was under pressure)
__device__ __forceinline__ void sload12(const double *p, int byte_off, d8_t &a, d4_t &b) {
    asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx8 %1, %2, %4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b) : "s"(p), "s"(byte_off), "s"(byte_off + 64));
}


// Split form of sload12 for software pipelining: the request leaves the load in flight, so the destination registers must
// stay allocated (and untouched) until sload12_wait, which takes them as read-write operands.  That holds only while the
// allocator does not spill them: a kernel using the pair must show no SGPR spill of these values (tools/check_sload_pipeline.py
// scans the ISA for any instruction that touches the destination registers between a request and its wait).
// `order` is a value the arithmetic that should overlap the load starts from: tying it to the request keeps the
// scheduler from sinking the request below that arithmetic.
__device__ __forceinline__ void sload12_issue(const double *p, int byte_off, d8_t &a, d4_t &b, double &order) {
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx8 %1, %3, %5"
                 : "=&s"(a), "=&s"(b), "+v"(order) : "s"(p), "s"(byte_off), "s"(byte_off + 64));
}
__device__ __forceinline__ void sload12_wait(d8_t &a, d4_t &b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
}


// one row (24 doubles) of a wave-uniform table, requested and awaited inside one asm statement (see sload12)
__device__ __forceinline__ void sload24(const double *p, int byte_off, d8_t &a, d8_t &b, d8_t &c) {
    asm volatile("s_load_dwordx16 %0, %3, %4\n\ts_load_dwordx16 %1, %3, %5\n\ts_load_dwordx16 %2, %3, %6\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(a), "=&s"(b), "=&s"(c) : "s"(p), "s"(byte_off), "s"(byte_off + 64), "s"(byte_off + 128));
}


// value held by the previous lane of the wave (lane 0 receives 0): DPP wave shift, two 32-bit moves per double, no LDS
__device__ __forceinline__ double lane_below(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned) (b & 0xffffffffull), 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned) (b >> 32), 0x138, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long) hi << 32) | lo);
}


// value held by the next lane of the wave (lane 63 receives 0): DPP wave shift the other way
__device__ __forceinline__ double lane_above(double v) {
    const unsigned long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned) (b & 0xffffffffull), 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned) (b >> 32), 0x130, 0xf, 0xf, true);
    return __longlong_as_double(((unsigned long long) hi << 32) | lo);
}

// component-sequential 3x3 solve of m_smoothNode (MG.hh:254-264)
__device__ __forceinline__ void gs_solve(const double bms[3], const double M[9], uint8_t mask, bool forward,
                                         double ud[3]) {
    ud[0] = ud[1] = ud[2] = 0.0;
    if (forward) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double t = bms[i] - (M[i * 3 + 0] * ud[0] + M[i * 3 + 1] * ud[1] + M[i * 3 + 2] * ud[2]);
            ud[i] = t * (((mask >> i) & 1) ? 0.0 : 1.0 / M[i * 3 + i]);
        }
    } else {
#pragma unroll
        for (int i = 2; i >= 0; --i) {
            const double t = bms[i] - (M[i * 3 + 0] * ud[0] + M[i * 3 + 1] * ud[1] + M[i * 3 + 2] * ud[2]);
            ud[i] = t * (((mask >> i) & 1) ? 0.0 : 1.0 / M[i * 3 + i]);
        }
    }
}


}  // namespace vfem
