// lnr_wave.h -- wave64 helpers shared by the kernel translation units (lnr_kernels.hip, lnr_gap_kernels.hip): ordering points inside one
// wave, DPP reductions / scans.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include "lnr_hd.h"

namespace lnr {

#define WAVE 64
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
// Ordering point inside ONE wave: the leader's LDS / global stores become visible to the other lanes' later loads.
// Lanes of a wave run in lockstep, so no s_barrier is needed -- only the memory waits; this also lets single-wave
// code run inside a multi-wave workgroup (heavy path) without involving the other waves.
__device__ __forceinline__ void WSYNC() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// Ordering point for LDS traffic inside one wave: LDS instructions of a wave execute in issue order, so only the compiler
// has to be kept from moving accesses across it -- unlike WSYNC it does not wait for outstanding global stores.
__device__ __forceinline__ void WLDS() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ u64 lanemask_lt() { return (1ULL << lane_id()) - 1ULL; }
// Cross-lane reductions and scans on the DPP path of the VALU (row operations inside 16 lanes, row broadcasts across the
// four rows of a wave64): a dozen VALU instructions instead of six dependent ds_bpermute round trips through the LDS
// crossbar.  All 64 lanes must be active.  Reductions leave the result in lane 63 and hand it out through v_readlane, so the
// value the callers get is wave-uniform (scalar register).
#define DPP_QUAD_XOR1 0xB1     /* quad_perm:[1,0,3,2] */
#define DPP_QUAD_XOR2 0x4E     /* quad_perm:[2,3,0,1] */
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_MIRROR 0x140
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_BCAST15 0x142
#define DPP_ROW_BCAST31 0x143
#define DPP_MOV(ident, v, ctrl, rows) ((u32)__builtin_amdgcn_update_dpp((int)(ident), (int)(v), (ctrl), (rows), 0xf, false))
__device__ __forceinline__ u32 wave_incl_scan(u32 v) {
    v += DPP_MOV(0, v, DPP_ROW_SHR(1), 0xf);
    v += DPP_MOV(0, v, DPP_ROW_SHR(2), 0xf);
    v += DPP_MOV(0, v, DPP_ROW_SHR(4), 0xf);
    v += DPP_MOV(0, v, DPP_ROW_SHR(8), 0xf);
    v += DPP_MOV(0, v, DPP_ROW_BCAST15, 0xa);      // rows 1 and 3 take the total of the row before them
    v += DPP_MOV(0, v, DPP_ROW_BCAST31, 0xc);      // rows 2 and 3 take the total of rows 0 + 1
    return v;
}
#define WAVE_REDUCE_U32(v, ident, OP)                                   \
    { u32 t_;                                                           \
      t_ = DPP_MOV(ident, v, DPP_QUAD_XOR1, 0xf); v = OP(v, t_);        \
      t_ = DPP_MOV(ident, v, DPP_QUAD_XOR2, 0xf); v = OP(v, t_);        \
      t_ = DPP_MOV(ident, v, DPP_ROW_HALF_MIRROR, 0xf); v = OP(v, t_);  \
      t_ = DPP_MOV(ident, v, DPP_ROW_MIRROR, 0xf); v = OP(v, t_);       \
      t_ = DPP_MOV(ident, v, DPP_ROW_BCAST15, 0xa); v = OP(v, t_);      \
      t_ = DPP_MOV(ident, v, DPP_ROW_BCAST31, 0xc); v = OP(v, t_); }
#define OP_ADD_(a, b) ((a) + (b))
#define OP_MIN_(a, b) ((b) < (a) ? (b) : (a))
#define OP_MAX_(a, b) ((b) > (a) ? (b) : (a))
__device__ __forceinline__ u32 wave_sum(u32 v) {
    WAVE_REDUCE_U32(v, 0u, OP_ADD_);
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_min_u32(u32 v) {
    WAVE_REDUCE_U32(v, 0xffffffffu, OP_MIN_);
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ u32 wave_max_u32(u32 v) {
    WAVE_REDUCE_U32(v, 0u, OP_MAX_);
    return (u32)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ i64 wave_max_i64(i64 v) {
    const i64 ident = (i64)0x8000000000000000LL;
#define STEP64_(ctrl, rows)                                                                                         \
    { u32 lo_ = DPP_MOV((u32)ident, (u32)v, ctrl, rows), hi_ = DPP_MOV((u32)((u64)ident >> 32), (u32)((u64)v >> 32), ctrl, rows); \
      i64 t_ = (i64)(((u64)hi_ << 32) | lo_); v = t_ > v ? t_ : v; }
    STEP64_(DPP_QUAD_XOR1, 0xf) STEP64_(DPP_QUAD_XOR2, 0xf) STEP64_(DPP_ROW_HALF_MIRROR, 0xf) STEP64_(DPP_ROW_MIRROR, 0xf)
    STEP64_(DPP_ROW_BCAST15, 0xa) STEP64_(DPP_ROW_BCAST31, 0xc)
#undef STEP64_
    return (i64)(((u64)(u32)__builtin_amdgcn_readlane((int)(u32)((u64)v >> 32), 63) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)v, 63));
}


}  // namespace lnr
