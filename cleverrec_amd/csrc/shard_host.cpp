// Host form of the row-sharded step's index work (include/crec.h "row-sharded training"): what shard_plan_batches +
// shard_pack_ids and shard_owner_index compute on the device, for ONE batch, with the same arithmetic (shard_index.h) and the
// same orders (stable by key, slots in batch order: i-slices before j-slices; peers in rank order).  Two users: the
// multi-process tests run the exchange protocol over gloo with THIS as every rank's planner (tests/test_parallel_cpu.py), and
// the GPU tests hold the kernels to it (crec_debug_shard_plan / crec_debug_shard_owner_index).
#include <algorithm>
#include <numeric>
#include <vector>

#include "crec_internal.h"
#include "shard_index.h"

extern "C" {

int crec_shard_plan_host(const int32_t *i, const int32_t *j, int32_t nb, int32_t G, int64_t I_global, int32_t *req_keys, int32_t *n_req,
                         int32_t *req_off, int32_t *ci, int32_t *cj, int32_t *ids_send)
{
    if (!i || !j || !req_keys || !n_req || !req_off || !ci || !cj || !ids_send || nb < 0 || G < 1 || I_global < 1)
        return CREC_EINVAL;
    const int64_t S = (I_global + G - 1) / G;
    const int n = 2 * nb;
    std::vector<uint32_t> key((size_t)n);
    std::vector<int32_t> slot((size_t)n);
    for (int k = 0; k < n; ++k) {
        const int64_t g = k < nb ? i[k] : j[k - nb];
        if (g < 0 || g >= I_global)
            return CREC_EINVAL;
        key[(size_t)k] = shard_owner_key(g, G, S);
    }
    std::iota(slot.begin(), slot.end(), 0);
    std::stable_sort(slot.begin(), slot.end(), [&](int32_t a, int32_t b) { return key[(size_t)a] < key[(size_t)b]; });
    int r = 0;
    for (int o = 0; o <= G; ++o)
        req_off[o] = 0;
    for (int k = 0; k < n; ++k) {
        const uint32_t kk = key[(size_t)slot[(size_t)k]];
        if (k == 0 || kk != key[(size_t)slot[(size_t)k - 1]]) {
            req_keys[r] = (int32_t)kk;
            ++req_off[shard_key_owner(kk, S) + 1];
            ++r;
        }
        const int32_t s = slot[(size_t)k]; // every slot learns the rank of its item among the batch's unique items
        if (s < nb)
            ci[s] = r - 1;
        else
            cj[s - nb] = r - 1;
    }
    for (int o = 0; o < G; ++o)
        req_off[o + 1] += req_off[o];
    *n_req = r;
    for (int q = 0; q < r; ++q) // owner-major already: the request list of owner o is ids_send[req_off[o] .. req_off[o+1])
        ids_send[q] = shard_key_row(req_keys[q], S);
    return CREC_OK;
}

int crec_shard_owner_index_host(const int32_t *ids_recv, const int32_t *srv_cnt, int32_t G, int32_t *order, int32_t *urows, int32_t *useg,
                                int32_t *n_urows)
{
    if (!ids_recv || !srv_cnt || !order || !urows || !useg || !n_urows || G < 1)
        return CREC_EINVAL;
    int n = 0;
    for (int p = 0; p < G; ++p) {
        if (srv_cnt[p] < 0)
            return CREC_EINVAL;
        n += srv_cnt[p];
    }
    std::vector<int32_t> slot((size_t)n);
    std::iota(slot.begin(), slot.end(), 0);
    std::stable_sort(slot.begin(), slot.end(), [&](int32_t a, int32_t b) { return (uint32_t)ids_recv[a] < (uint32_t)ids_recv[b]; });
    int r = 0;
    for (int k = 0; k < n; ++k) {
        if (k == 0 || ids_recv[slot[(size_t)k]] != ids_recv[slot[(size_t)k - 1]]) {
            urows[r] = ids_recv[slot[(size_t)k]];
            useg[r] = k;
            ++r;
        }
        order[k] = slot[(size_t)k];
    }
    useg[r] = n;
    *n_urows = r;
    return CREC_OK;
}

} // extern "C"
