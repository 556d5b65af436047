// The index arithmetic of the row-sharded step, shared by the kernels (shard_kernels.h) and the host planner (shard_host.cpp):
// item g lives on rank g % G as local row g / G of that rank's shard (S = ceil(I / G) rows per shard).
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define CREC_HD __host__ __device__
#else
#define CREC_HD
#endif

// owner-major key: sorts a batch's items by owner, then by local row
CREC_HD inline uint32_t shard_owner_key(int64_t g, int32_t G, int64_t S) { return (uint32_t)((g % G) * S + g / G); }
CREC_HD inline int32_t shard_key_owner(int64_t key, int64_t S) { return (int32_t)(key / S); }
CREC_HD inline int32_t shard_key_row(int64_t key, int64_t S) { return (int32_t)(key - (key / S) * S); }
