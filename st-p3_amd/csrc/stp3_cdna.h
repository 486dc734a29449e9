// stp3_cdna.h -- the CDNA-specific instructions the kernels issue by hand (gfx90a+ / gfx950).
// (tests/hipcpu provides a header of the same name that models them for the CPU stand-in.)
#pragma once

// fmac_row_bcast<J>(acc, v, f):  acc += v[(lane & ~15) | J] * f   in ONE VALU instruction: v_fmac_f32 with the DPP
// row_newbcast control (lane J of every 16-lane row is broadcast to the row).  Written with
// __builtin_amdgcn_update_dpp the compiler emits the broadcast as a separate v_mov_b32_dpp (row_newbcast is not folded
// into the multiply-add), which doubles the instruction count of the lift kernel's inner loop.
template <int J>
__device__ __forceinline__ void fmac_row_bcast(float& acc, float v, float f) {
    static_assert(J >= 0 && J < 16, "row_newbcast lane");
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(v), "v"(f), "n"(J));
}

// lds_dma16(src, lds_base): global -> LDS without passing through registers (global_load_lds_dwordx4).  Every ACTIVE
// lane moves the 16 bytes at its own `src` to  lds_base + 16 * lane  (the destination is a wave-uniform base plus the
// lane's slot; inactive lanes leave their slot untouched).  Completion is counted by vmcnt: lds_dma_wait() before the
// wave reads the data.
__device__ __forceinline__ void lds_dma16(const float* src, float* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// the same for ONE dword per lane (global_load_lds_dword): lane l's word lands at lds_base + 4 * l
__device__ __forceinline__ void lds_dma4(const void* src, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_base, 4, 0, 0);
}
__device__ __forceinline__ void lds_dma_wait() {
    __builtin_amdgcn_s_waitcnt(0);          // vmcnt(0) expcnt(0) lgkmcnt(0)
}

// lds_read_tr16(p): ds_read_b64_tr_b16, the gfx950 LDS transpose read.  Within every 16-lane group the 16 lanes x 4
// halfwords named by the lanes' addresses form a [4][16] block -- lane i of the group supplies row i >> 2, columns
// 4 * (i & 3) .. + 3, as four contiguous halfwords at its own 8-byte aligned address -- and lane i RECEIVES column i
// (its element j = row j).  An MFMA operand whose k index runs along the SLOW dimension of a row-major LDS image is
// read with it as it lies: no transposition in registers.  (An address off 8-byte alignment silently returns the
// aligned address's data.)
typedef short stp3_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ stp3_s16x4 lds_read_tr16(const void* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) stp3_s16x4*)p);
}

// All-reduce over the 16 lanes of a DPP row with four rotations (row_ror 8, 4, 2, 1): one VALU instruction per step,
// no LDS round trip (what __shfl_xor costs).
template <int N>
__device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, row_ror<8>(v));
    v = fmaxf(v, row_ror<4>(v));
    v = fmaxf(v, row_ror<2>(v));
    return fmaxf(v, row_ror<1>(v));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    return v + row_ror<1>(v);
}

// Two float32 -> one packed bf16 pair (lo in bits 0..15), round to nearest even: v_cvt_pk_bf16_f32, ONE instruction for
// two elements on gfx950 (the integer round-and-shift it replaces costs ~7 per element -- the memory-bound kernels of
// this library were VALU-bound on it).
typedef __bf16 stp3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float stp3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const stp3_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, stp3_bf16x2));
}
// 1 / x as v_rcp_f32 (1 ulp) -- for activation functions, where the IEEE division sequence (~10 instructions) buys
// nothing: the results are rounded to bf16 or compared at 1e-4
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
// The same for two values in a register pair: gfx950 issues float32 multiply / add / fma for TWO lanes' worth of data in one
// instruction (v_pk_mul_f32, v_pk_add_f32, v_pk_fma_f32); the exponential and the reciprocal stay one instruction per value.
// Bit-identical to fast_sigmoid: x * (-log2 e) [0xbfb8aa3b] -> v_exp_f32 -> + 1 -> v_rcp_f32.
__device__ __forceinline__ stp3_f32x2 pk_fma(stp3_f32x2 a, stp3_f32x2 b, stp3_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ stp3_f32x2 pk_sigmoid(stp3_f32x2 x) {
    const stp3_f32x2 t = x * __builtin_bit_cast(float, 0xbfb8aa3bu);
    stp3_f32x2 e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    e = e + 1.0f;
    return stp3_f32x2{fast_rcp(e.x), fast_rcp(e.y)};
}

// ---- raw buffer loads -----------------------------------------------------------------------------------------------
// A buffer resource (V#) over [base, base + bytes): loads through it take a SCALAR base and a 32-bit byte offset per lane
// (no 64-bit address arithmetic in vector registers) and return ZERO for any offset at or beyond `bytes` -- the hardware's
// range check.  A staging loop marks a padding tap / a row beyond the matrix with the offset kBufOob instead of selecting
// between two 64-bit addresses.  Requires bytes < 2^31 (so that kBufOob is out of range and offset + 16 cannot wrap).
typedef unsigned stp3_u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t stp3_buffer;
constexpr uint32_t kBufOob = 0x80000000u;
__device__ __forceinline__ stp3_buffer make_buffer(const void* base, uint32_t bytes) {
    // base and size are uniform (kernel arguments); saying so keeps the resource in scalar registers -- otherwise every load is
    // wrapped in a readfirstlane loop over the lanes' (identical) resources
    const uint64_t a = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), (short)0,
                                             (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ stp3_u32x4 buffer_load16(stp3_buffer rsrc, uint32_t byte_offset) {
    return __builtin_bit_cast(stp3_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_offset, 0, 0));
}
