// Device side of the peer transport (crec_peer.hip), for kernels that take part in an exchange themselves: the generic exchange
// kernel, and the row-sharded step's fused forms (shard_kernels.h: serve + exchange, exchange + owner apply).
// Protocol and memory-ordering notes: crec_peer.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int PEER_MAX = 16;     // ranks of a peer communicator
constexpr int PEER_THREADS = 512;
constexpr int PEER_MAX_BLOCKS = 128; // half a wave slot per SIMD of half the CUs: always resident together, beside whatever else runs

// What ONE kernel launch needs to take part in exchange number `seq` of a peer communicator (crec_xch::begin hands it out and
// advances the sequence: every rank must launch exactly one such kernel per sequence number, in the same order).
struct crec_peer_view {
    const char *win[PEER_MAX];         // rank p's window as this process sees it
    unsigned long long *sig[PEER_MAX]; // rank p's signal block: [2 parities][2 phases][PEER_MAX] words
    char *my_half;                     // this rank's window half of this exchange (= win[rank] + half_off)
    int64_t half_off;
    uint32_t seq;
    int rank, world, fences;
    unsigned long long timeout_ticks;
    int32_t *ctr, *err; // device: ctr[0], ctr[2] workgroup counters of phases 0 / 1 (zero between launches), err the error word
};

#if defined(__HIPCC__)
typedef unsigned peer_u4v __attribute__((ext_vector_type(4)));
constexpr int PEER_AUX_SYS = 17; // sc0 | sc1: system scope
__device__ __forceinline__ __amdgpu_buffer_rsrc_t peer_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7fffffff, 0x00020000);
}
// 16 bytes from / to a window (base: wave-uniform, byte offset < 2 GiB): system-scope accesses, or plain ones in the fence form
__device__ __forceinline__ peer_u4v peer_ld16(const char *base, int64_t off, bool fences)
{
    return fences ? *(const peer_u4v *)(base + off) : __builtin_amdgcn_raw_buffer_load_b128(peer_rsrc(base), (int)off, 0, PEER_AUX_SYS);
}
__device__ __forceinline__ void peer_st16(char *base, int64_t off, peer_u4v v, bool fences)
{
    if (fences)
        *(peer_u4v *)(base + off) = v;
    else
        __builtin_amdgcn_raw_buffer_store_b128(v, peer_rsrc(base), (int)off, 0, PEER_AUX_SYS);
}

// Every thread of the workgroup calls, after its last store into this rank's window half.  When the LAST workgroup of the launch
// arrives (counter ctr[2 * phase]), it raises this rank's flag {seq, off_of(p) / 16} in every peer's signal block.  lds: 1 int.
// counted == false: nothing was written by this launch (a previous kernel filled the half): workgroup 0 publishes at once.
template <typename F>
__device__ __forceinline__ void peer_publish(const crec_peer_view &v, int phase, bool counted, int *lds, F off_of)
{
    const int tid = threadIdx.x, h = (int)(v.seq & 1u);
    bool last;
    if (counted) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's window stores are at memory (write-through form)
        __syncthreads();
        if (tid == 0) {
            if (v.fences)
                __threadfence_system(); // release: what the workgroup left dirty in L2
            *lds = atomicAdd(v.ctr + 2 * phase, 1) == (int)gridDim.x - 1;
        }
        __syncthreads();
        last = *lds != 0;
        if (last) {
            if (tid == 0) {
                if (v.fences)
                    __threadfence_system();
                v.ctr[2 * phase] = 0; // the next launch on this stream starts from zero
            }
            __syncthreads();
        }
    } else {
        last = blockIdx.x == 0;
    }
    if (last && tid < v.world)
        __hip_atomic_store(v.sig[tid] + (h * 2 + phase) * PEER_MAX + v.rank,
                           ((unsigned long long)v.seq << 32) | (unsigned long long)(uint32_t)(off_of(tid) >> 4), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

// Every thread calls.  Waits until every rank's flag of this exchange (phase) stands in this rank's own signal block; flag_s[p] (LDS,
// PEER_MAX words) receives them (low 32 bits: the segment's offset / 16 inside the source's window half).  false: a peer did not
// come within the time-out (the error word is set; the caller returns).  bad: 1 LDS int.
__device__ __forceinline__ bool peer_wait_all(const crec_peer_view &v, int phase, unsigned long long *flag_s, int *bad)
{
    const int tid = threadIdx.x, h = (int)(v.seq & 1u);
    if (tid == 0)
        *bad = 0;
    __syncthreads();
    if (tid < v.world) {
        const unsigned long long *mine = v.sig[v.rank] + (h * 2 + phase) * PEER_MAX + tid;
        const unsigned long long t0 = wall_clock64();
        unsigned long long f;
        for (;;) {
            f = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((uint32_t)(f >> 32) == v.seq)
                break;
            if (wall_clock64() - t0 > v.timeout_ticks) {
                *bad = 1;
                atomicExch(v.err, 1 + tid);
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        flag_s[tid] = f;
    }
    __syncthreads();
    if (*bad)
        return false;
    if (v.fences) {
        if (tid == 0)
            __threadfence_system(); // acquire: nothing stale of the peers' windows in this CU's L1 / the L2
        __syncthreads();
    }
    return true;
}
// Eight 16-byte system-scope loads from per-lane addresses (different peers' windows per lane group: no wave-uniform base for the
// buffer form), issued together, then waited for.  Lanes / entries without a row pass any mapped address (e.g. their own window).
__device__ __forceinline__ void peer_ld16x8(const char *const (&a)[8], peer_u4v (&o)[8], bool fences)
{
    if (fences) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
            o[e] = *(const peer_u4v *)a[e];
        return;
    }
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\t"
                 "global_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc0 sc1\n\t"
                 "global_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc0 sc1\n\t"
                 "global_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc0 sc1\n\t"
                 "global_load_dwordx4 %7, %15, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7])
                 : "memory");
}
__device__ __forceinline__ peer_u4v peer_ld16_at(const char *a, bool fences) // one such load (per-lane address), waited for
{
    if (fences)
        return *(const peer_u4v *)a;
    peer_u4v o;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(o) : "v"(a) : "memory");
    return o;
}
// where rank p's segment for this rank starts, given its flag
__device__ __forceinline__ const char *peer_seg(const crec_peer_view &v, int p, unsigned long long flag)
{
    return v.win[p] + v.half_off + ((int64_t)(uint32_t)flag << 4);
}
#endif
