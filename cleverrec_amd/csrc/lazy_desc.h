// The descriptor the lazy step (bpr_triples_lazy) reads per batch position and role (user, positive item, negative item),
// written by the two index builders: bpr_index_batches (a batch's pairs sorted in LDS, batches of at most 8192 triples) and
// index_big.hip (any batch size up to LAZY_MAX_B, sorted in global memory).  One int4:
//   x = row | single << 31        the row occurs once in the batch
//   y = k | hmod << 20            k: the slot's position in the side's sorted order (user side: of B slots, item side: of 2B);
//                                 hmod (front rows): where in its workgroup of the triple pass the row's first positive slot sits
//   z = start | (ci & 0xfff) << 20   the row's segment [start, start + cnt) in the sorted order
//   w = cnt | (ci >> 12) << 21    ci (front rows: two or more positive slots): the number of the row's positive slots, else 0
// k, start < 2^20, cnt <= 2^20, ci <= 2^19, hmod < 64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int LAZY_MAX_B = 1 << 19; // triples per batch the descriptor's fields hold

__host__ __device__ __forceinline__ int4 lazy_desc_pack(uint32_t row, bool single, int k, int start, int cnt, int ci, int hmod)
{
    return make_int4((int)(row | (single ? 0x80000000u : 0u)), (int)((uint32_t)k | ((uint32_t)hmod << 20)),
                     (int)((uint32_t)start | (((uint32_t)ci & 0xfffu) << 20)), (int)((uint32_t)cnt | (((uint32_t)ci >> 12) << 21)));
}
__host__ __device__ __forceinline__ int lazy_desc_k(int4 d) { return d.y & 0xfffff; }
__host__ __device__ __forceinline__ int lazy_desc_hmod(int4 d) { return (int)((uint32_t)d.y >> 20); }
__host__ __device__ __forceinline__ int lazy_desc_start(int4 d) { return d.z & 0xfffff; }
__host__ __device__ __forceinline__ int lazy_desc_cnt(int4 d) { return d.w & 0x1fffff; }
__host__ __device__ __forceinline__ int lazy_desc_ci(int4 d) { return (int)(((uint32_t)d.z >> 20) | (((uint32_t)d.w >> 21) << 12)); }

// index_big.hip: the lazy step's index of a chunk of batches of ANY size (sorted in global memory); fills the same arrays
// bpr_index_batches fills in lazy mode, bit for bit the same values.
struct crec_ctx;
struct LazyIndexOut {
    int4 *u_desc, *i_desc, *j_desc; // [nbc][B]: u_desc by batch position, i_desc / j_desc by processing slot
    int4 *i_tmp, *j_tmp;            // [nbc][B] scratch: the item descriptors by batch position
    int2 *slot_pu;                  // [nbc][B] by processing slot {batch position, user row}
    int32_t *order;                 // [nbc][B] batch position per processing slot
    int32_t *nslots;                // [nbc][2] {triples of the batch, of which in the item-sorted front region}
};
int crec_lazy_index_big(crec_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n_chunk, int32_t B, int64_t nbc,
                        int bits_u, int bits_q, int spw, const LazyIndexOut &out, hipStream_t st, int64_t reserve_nbc = 0);

// index_big.hip: the row-sharded step's plan and owner-side index (shard_kernels.h: shard_plan_batches, shard_owner_index) for
// batches beyond what a workgroup sorts in LDS: the same outputs, bit for bit.  All pointers device.
int crec_shard_plan_big(crec_ctx *ctx, const int32_t *i, const int32_t *j, int64_t n_chunk, int32_t B, int64_t nbc, int32_t G, int64_t S, int key_bits,
                        int32_t *req_keys, int32_t *n_req, int32_t *req_off, int32_t *ci, int32_t *cj, hipStream_t st);
int crec_shard_owner_index_big_reserve(crec_ctx *ctx, int64_t nbc, int64_t stride, hipStream_t st);
int crec_shard_owner_index_big(crec_ctx *ctx, const int32_t *ids_recv, const int64_t *srv_base, const int32_t *srv_cnt, int32_t G, int64_t nbc,
                               int64_t stride, int key_bits, int32_t *srv_rows, int32_t *order, int32_t *urows, int32_t *useg, int32_t *n_urows,
                               hipStream_t st);
