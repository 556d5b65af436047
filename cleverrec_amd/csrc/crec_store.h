// Stores of what a kernel hands to the NEXT kernel of the stream (contribution rows, tile partials, updated table rows).
//
// A plain store leaves its line dirty in the writing XCD's L2; the release at the end of the kernel then writes every such line
// back before the next dispatch may start -- measured on ncf_tile (14.7 MB of contribution rows and partials, most of them written
// by the last two phases): 19.1 us per launch with plain stores, 15.1 with the same stores written THROUGH the L2 as they are
// issued (scope bits sc0 sc1; the lines travel while the tile still computes, the kernel's end has nothing left to flush).
// Non-temporal stores end the kernel as early (15.3) but cost the consumer 1.8 us: its reads then come from HBM.
// (profiles/round4_store_modes.txt)
#pragma once
#ifdef __HIPCC__
#ifndef CREC_OUT_WT
#define CREC_OUT_WT 1 // 0: plain stores (timing experiments)
#endif

// one float, written through (a relaxed system-scope atomic store IS global_store_dword ... sc0 sc1)
__device__ __forceinline__ void st_out(float *p, float v)
{
#if CREC_OUT_WT
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#else
    *p = v;
#endif
}
// 16 bytes, written through: a buffer store (the compiler's own builtin, so its hazard and wait-count bookkeeping covers it -- a
// global_store_dwordx4 in inline assembly lost store data to the VALU instruction behind it: on gfx940 and later a store of more
// than 8 bytes needs TWO wait states before its data registers are written again, and the compiler does not look inside an
// assembly statement).  base: wave-uniform buffer pointer; off: float index, off * 4 < 2^31 (contribution rows: at most
// 3 x 65 536 slots x 256 floats = 201 MB).
typedef unsigned crec_st_u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4_out(float *base, int64_t off, float4 v)
{
#if CREC_OUT_WT
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const crec_st_u4v x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)(off * 4), 0, 17 /* sc0 sc1 */);
#else
    *reinterpret_cast<float4 *>(base + off) = v;
#endif
}
#endif
