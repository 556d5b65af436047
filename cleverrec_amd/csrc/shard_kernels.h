// Kernels of the row-sharded BPR step (included by bpr_train.hip inside its anonymous namespace, after the shared
// device pieces: ld4 / st4 / adam4 / Hyper).  Items are dealt to ranks round-robin: item g lives on rank g % G as
// local row g / G of that rank's Q shard (S = ceil(I / G) rows per shard).  A rank runs the lazy BPR step of
// bpr_train.hip on its own triples against a COMPACT item table Qc holding just the rows its batch references:
//
//   shard_plan_batches   (per chunk of batches) the batch's 2B item ids -> owner-major keys (g % G) * S + g / G,
//                        stable radix sort in LDS (block_sort.h) -> the unique keys in ascending order = the batch's
//                        request list, grouped by owner; every triple's i and j are replaced by their rank in that
//                        list (the compact id).  The per-owner counts go to the host once per chunk.
//   shard_pack_ids       request lists of all batches of the chunk -> one owner-major buffer (local row ids), which one
//                        all-to-all-v per CHUNK delivers to the owners: the index exchange is off the per-step path.
//   shard_owner_index    owner side, per chunk: for every batch the rows the peers ask for, peer after peer, sorted
//                        by row -> unique rows, their slot lists in peer order (one fixed summation order).
//   shard_serve_rows     per step: the requested rows of the shard, staged peer by peer for the forward all-to-all-v.
//   shard_owner_apply    per step: a row's gradient rows, as they came back from the peers, are added in peer order
//                        and the row takes its lazy Adam update in place.
#pragma once
#include "shard_index.h"

constexpr int SHARD_THREADS = 1024;

inline size_t shard_lds_bytes(int cap)
{
    return sizeof(uint32_t) * cap + sizeof(uint16_t) * (2 * (size_t)cap + 64) +
           sizeof(int) * (crec_sort::hist_ints<SHARD_THREADS>() + crec_sort::scan_ints<SHARD_THREADS>() + 80);
}

// grid = batches of the chunk.  req_keys [nbc][2B], n_req [nbc], req_off [nbc][G+1], ci / cj [n_total]
__global__ __launch_bounds__(SHARD_THREADS) void shard_plan_batches(const int32_t *__restrict__ i,
                                                                    const int32_t *__restrict__ j, int64_t n_total,
                                                                    int32_t B, int32_t cap, int32_t G, int64_t S,
                                                                    int32_t key_bits, int32_t *__restrict__ req_keys,
                                                                    int32_t *__restrict__ n_req,
                                                                    int32_t *__restrict__ req_off,
                                                                    int32_t *__restrict__ ci, int32_t *__restrict__ cj)
{
    extern __shared__ uint32_t rows[]; // [cap]
    uint16_t *tag = (uint16_t *)(rows + cap);
    uint16_t *seg = tag + cap; // unused here (same layout as the index kernels)
    int *hist = (int *)(seg + cap + 64);
    int *scan = hist + crec_sort::hist_ints<SHARD_THREADS>();
    int *per_owner = scan + crec_sort::scan_ints<SHARD_THREADS>(); // [<= 64] + 16
    const int batch = blockIdx.x, tid = threadIdx.x;
    const int64_t t0 = (int64_t)batch * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B);
    const int n = 2 * nb;
    for (int k = tid; k < n; k += SHARD_THREADS) {
        const int64_t g = k < nb ? i[t0 + k] : j[t0 + k - nb];
        rows[k] = shard_owner_key(g, G, S);
        tag[k] = (uint16_t)k;
    }
    if (tid < 80)
        per_owner[tid] = 0;
    crec_sort::block_sort_rows<SHARD_THREADS>(rows, tag, n, key_bits, hist, scan);

    const int per = (n + SHARD_THREADS - 1) / SHARD_THREADS;
    const int k0 = tid * per, k1 = (k0 + per) < n ? (k0 + per) : n;
    int heads = 0;
    for (int k = k0; k < k1; ++k)
        heads += (k == 0 || rows[k] != rows[k - 1]) ? 1 : 0;
    int total = 0;
    int r = crec_sort::block_excl_scan<SHARD_THREADS>(heads, scan, &total);
    int32_t *keys_out = req_keys + (int64_t)batch * 2 * B;
    for (int k = k0; k < k1; ++k) {
        if (k == 0 || rows[k] != rows[k - 1]) {
            keys_out[r] = (int32_t)rows[k];
            atomicAdd(&per_owner[shard_key_owner(rows[k], S)], 1); // counts only: the order of the adds does not matter
            ++r;
        }
        const int slot = tag[k]; // every slot learns the rank of its item among the batch's unique items
        if (slot < nb)
            ci[t0 + slot] = r - 1;
        else
            cj[t0 + slot - nb] = r - 1;
    }
    __syncthreads();
    if (tid == 0) {
        int at = 0;
        int32_t *off = req_off + (int64_t)batch * (G + 1);
        for (int o = 0; o < G; ++o) {
            off[o] = at;
            at += per_owner[o];
        }
        off[G] = at;
        n_req[batch] = total;
    }
}

// grid = batches.  ids_send[pack_off[b][o] + k] = local row of the k-th item this batch requests from owner o
__global__ __launch_bounds__(256) void shard_pack_ids(const int32_t *__restrict__ req_keys, const int32_t *__restrict__ req_off,
                                                      const int64_t *__restrict__ pack_off, int32_t B, int32_t G, int64_t S,
                                                      int32_t *__restrict__ ids_send)
{
    const int b = blockIdx.x;
    const int32_t *keys = req_keys + (int64_t)b * 2 * B, *off = req_off + (int64_t)b * (G + 1);
    const int n = off[G];
    for (int q = threadIdx.x; q < n; q += 256) {
        const int64_t key = keys[q];
        const int o = shard_key_owner(key, S);
        ids_send[pack_off[(int64_t)b * G + o] + (q - off[o])] = shard_key_row(key, S);
    }
}

// grid = batches.  The peers' request lists for batch b (peer p: ids_recv[srv_base[b][p] ..) of srv_cnt[b][p] rows)
// -> srv_rows[b][slot] (peer-major, the order rows are served and gradient rows come back in), order[b][k] = slot of
// the k-th (row, slot) pair sorted by row, urows[b][q] / useg[b][q] = unique rows and where their slots start.
__global__ __launch_bounds__(SHARD_THREADS) void shard_owner_index(const int32_t *__restrict__ ids_recv,
                                                                   const int64_t *__restrict__ srv_base,
                                                                   const int32_t *__restrict__ srv_cnt, int32_t G,
                                                                   int32_t cap, int32_t stride, int32_t key_bits,
                                                                   int32_t *__restrict__ srv_rows,
                                                                   int32_t *__restrict__ order, int32_t *__restrict__ urows,
                                                                   int32_t *__restrict__ useg, int32_t *__restrict__ n_urows)
{
    extern __shared__ uint32_t rows[]; // [cap]
    uint16_t *tag = (uint16_t *)(rows + cap);
    uint16_t *seg = tag + cap;
    int *hist = (int *)(seg + cap + 64);
    int *scan = hist + crec_sort::hist_ints<SHARD_THREADS>();
    const int b = blockIdx.x, tid = threadIdx.x;
    int n = 0;
    for (int p = 0; p < G; ++p) {
        const int c = srv_cnt[(int64_t)b * G + p];
        const int32_t *src = ids_recv + srv_base[(int64_t)b * G + p];
        for (int k = tid; k < c; k += SHARD_THREADS) {
            const int32_t row = src[k];
            rows[n + k] = (uint32_t)row;
            tag[n + k] = (uint16_t)(n + k);
            srv_rows[(int64_t)b * stride + n + k] = row;
        }
        n += c;
    }
    crec_sort::block_sort_rows<SHARD_THREADS>(rows, tag, n, key_bits, hist, scan);
    const int per = (n + SHARD_THREADS - 1) / SHARD_THREADS;
    const int k0 = tid * per, k1 = (k0 + per) < n ? (k0 + per) : n;
    int heads = 0;
    for (int k = k0; k < k1; ++k)
        heads += (k == 0 || rows[k] != rows[k - 1]) ? 1 : 0;
    int total = 0;
    int r = crec_sort::block_excl_scan<SHARD_THREADS>(heads, scan, &total);
    for (int k = k0; k < k1; ++k) {
        if (k == 0 || rows[k] != rows[k - 1]) {
            urows[(int64_t)b * stride + r] = (int32_t)rows[k];
            useg[(int64_t)b * (stride + 1) + r] = k;
            ++r;
        }
        order[(int64_t)b * stride + k] = tag[k];
    }
    if (tid == 0) {
        useg[(int64_t)b * (stride + 1) + total] = n;
        n_urows[b] = total;
    }
}

// one group of lpr lanes per served slot: out[slot] = Q[srv_rows[slot]]
__global__ __launch_bounds__(256) void shard_serve_rows(const float *__restrict__ Q, const int32_t *__restrict__ srv_rows, int n,
                                                        int d, int lpr, float *__restrict__ out, int sys /* out is a peer window */)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t slot = t / lpr;
    const int l = (int)(t % lpr);
    if (slot >= n || 4 * l >= d)
        return;
    st4_dq(out, slot * d + 4 * l, ld4(Q + (int64_t)srv_rows[slot] * d + 4 * l), sys != 0, true);
}

// one group of lpr lanes per unique row of the batch: the row's gradient rows (one per requesting peer, each already
// the peer's sum over its own occurrences incl. the l2 term) are added in peer order, then lazy Adam in place
__global__ __launch_bounds__(256) void shard_owner_apply(float *__restrict__ Q, float *__restrict__ mQ, float *__restrict__ vQ,
                                                         const int32_t *__restrict__ urows, const int32_t *__restrict__ useg,
                                                         const int32_t *__restrict__ order, const int32_t *__restrict__ n_urows_p,
                                                         const float *__restrict__ grads, int d, int lpr, Hyper h)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t q = t / lpr;
    const int l = (int)(t % lpr);
    if (q >= *n_urows_p || 4 * l >= d)
        return;
    const int64_t ro = (int64_t)urows[q] * d + 4 * l;
    const int k0 = useg[q], k1 = useg[q + 1];
    float4 var = ld4(Q + ro), m = ld4(mQ + ro), v = ld4(vQ + ro);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = k0; k < k1; k += 8) { // up to eight peers' rows in flight, added in slot (= peer) order
        float4 c[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
            c[e] = k + e < k1 ? ld4(grads + (int64_t)order[k + e] * d + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (k + e < k1)
                add4(g, c[e]);
    }
    adam4(var, m, v, g, true, h);
    st4(Q + ro, var);
    st4(mQ + ro, m);
    st4(vQ + ro, v);
}

// ---- over the peer transport (crec_peer.hip) the two exchanges of a step fuse with the kernels around them ----------------------------
//   forward   shard_serve_rows stages the requested rows straight in this rank's window (crec_xch::next_half) and the exchange
//             (crec_xch::alltoallv_inplace) starts from there: no copy-in, every rank pulls its requests from the owners' windows;
//   backward  the step kernel leaves the gradient rows of its compact table in this rank's window (crec_xch::next_half: the compact
//             table's layout, owner o's segment at row req_off[o]); shard_exchange_apply publishes them, waits for the peers, and every
//             lane group adds its row's gradient rows STRAIGHT FROM THE REQUESTERS' WINDOWS in peer order, then lazy Adam in place.
// Four launches per step instead of five, and no second copy of rows or gradient rows on either side.
struct ShardPeerOffs {
    int64_t seg_off[PEER_MAX];   // bytes: where owner o's segment starts in my gradient rows (= req_off[o] * d * 4)
    int32_t slot_base[PEER_MAX + 1]; // my serve layout: peer p's slots are [slot_base[p], slot_base[p + 1])
};
__global__ __launch_bounds__(PEER_THREADS) void shard_exchange_apply(crec_peer_view pv, ShardPeerOffs so, float *__restrict__ Q, float *__restrict__ mQ,
                                                                     float *__restrict__ vQ, const int32_t *__restrict__ urows,
                                                                     const int32_t *__restrict__ useg, const int32_t *__restrict__ order,
                                                                     const int32_t *__restrict__ n_urows_p, int d, int lpr, Hyper h)
{
    __shared__ unsigned long long flag_s[PEER_MAX];
    __shared__ int last_s, bad_s, base_s[PEER_MAX + 1];
    if (threadIdx.x <= PEER_MAX)
        base_s[threadIdx.x] = so.slot_base[threadIdx.x < pv.world ? threadIdx.x : pv.world];
    peer_publish(pv, 0, false, &last_s, [&](int p) { return so.seg_off[p]; });
    const int64_t t = (int64_t)blockIdx.x * PEER_THREADS + threadIdx.x, groups = (int64_t)gridDim.x * PEER_THREADS / lpr;
    const int l = (int)(t % lpr), nq = *n_urows_p;
    const bool act = 4 * l < d;
    const bool fences = pv.fences != 0;
    // the row, its slots and its slot list do not depend on the peers: asked for before the wait (the usual launch has a lane group
    // per row, so this is all of the owner's own traffic)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t q = t / lpr;
    int64_t ro = q < nq ? (int64_t)urows[q] * d + 4 * l : 0;
    int k0 = q < nq ? useg[q] : 0, k1 = q < nq ? useg[q + 1] : 0;
    float4 var = (q < nq && act) ? ld4(Q + ro) : zero4, m = (q < nq && act) ? ld4(mQ + ro) : zero4, v = (q < nq && act) ? ld4(vQ + ro) : zero4;
    if (!peer_wait_all(pv, 0, flag_s, &bad_s))
        return;
    for (; q < nq; q += groups) {
        if (q != t / lpr) {
            ro = (int64_t)urows[q] * d + 4 * l;
            k0 = useg[q];
            k1 = useg[q + 1];
            var = act ? ld4(Q + ro) : zero4;
            m = act ? ld4(mQ + ro) : zero4;
            v = act ? ld4(vQ + ro) : zero4;
        }
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        const auto where = [&](int slot) {
            int p = 0;
            while (p + 1 < pv.world && slot >= base_s[p + 1])
                ++p;
            // requester p keeps the rows it asked of me in the order it asked: the order I serve them in
            return peer_seg(pv, p, flag_s[p]) + ((int64_t)(slot - base_s[p]) * d + 4 * l) * 4;
        };
        if (k1 - k0 == 1) { // the usual row: one requester
            if (act) {
                const peer_u4v c = peer_ld16_at(where(order[k0]), fences);
                g = make_float4(0.f + __uint_as_float(c.x), 0.f + __uint_as_float(c.y), 0.f + __uint_as_float(c.z), 0.f + __uint_as_float(c.w));
            }
            k0 = k1;
        }
        for (int k = k0; k < k1; k += 8) { // up to eight requesters' rows in flight, added in slot (= peer) order
            const char *addr[8];
            peer_u4v c[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                addr[e] = pv.my_half;
                if (k + e < k1 && act)
                    addr[e] = where(order[k + e]);
            }
            peer_ld16x8(addr, c, fences);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k + e < k1)
                    add4(g, make_float4(__uint_as_float(c[e].x), __uint_as_float(c[e].y), __uint_as_float(c[e].z), __uint_as_float(c[e].w)));
        }
        if (act) {
            adam4(var, m, v, g, true, h);
            st4(Q + ro, var);
            st4(mQ + ro, m);
            st4(vQ + ro, v);
        }
    }
}
