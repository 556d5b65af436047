// Workgroup-wide primitives of the index kernels (bpr_index_batches, rows_index_batches): an exclusive scan of one
// int per thread and a STABLE sort of up to 16*NTH (row, tag) pairs held in LDS by row.
//
// The sort is a least-significant-digit radix sort, at most 8 bits per pass, over the key_bits low bits of the row (the
// caller knows the table size, so ml-1m takes two passes of 6 bits and a 10 M-row table three of 8).  In a pass wave w owns the
// contiguous range [w*chunk, (w+1)*chunk) of the current order and walks it 64 elements at a time; the lanes that
// hold the same digit find each other with one ballot per digit bit, which gives every element its
// position among its wave's elements of that digit without atomics and in the current order -- that is what makes
// the pass stable.  Per-wave digit counts are then scanned in (digit, wave) order and every element is written to
// its final place of the pass.  All elements sit in registers between the read and the write, so one LDS copy of
// the arrays is enough.  Against the bitonic network this replaces (105 compare-exchange stages over 64-bit keys
// for 16 384 elements, a barrier after each) a pass costs four barriers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace crec_sort {

constexpr int HIST_STRIDE = 257; // per-wave digit counters, padded: the (digit, wave) scan reads a column

template <int NTH> constexpr int hist_ints() { return (NTH / 64) * HIST_STRIDE; }
template <int NTH> constexpr int scan_ints() { return NTH / 64 + 2; }

__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off, 64);
        if (lane >= off)
            v += t;
    }
    return v;
}

// exclusive scan over the NTH threads of the workgroup (thread order); *total = sum.  Two barriers; the scratch
// may be reused right after the call by anything that starts with a barrier of its own (block_excl_scan does)
template <int NTH> __device__ __forceinline__ int block_excl_scan(int v, int *wave_sums /* [NTH/64 + 2] LDS */, int *total)
{
    constexpr int WAVES = NTH / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int incl = wave_incl_scan(v);
    __syncthreads(); // the previous use of wave_sums is over
    if (lane == 63)
        wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const int s = lane < WAVES ? wave_sums[lane] : 0;
        const int si = wave_incl_scan(s);
        if (lane < WAVES)
            wave_sums[lane] = si - s;
        if (lane == WAVES - 1)
            wave_sums[WAVES] = si;
    }
    __syncthreads();
    *total = wave_sums[WAVES];
    return wave_sums[wave] + incl - v;
}

// rows[0..n), tag[0..n) in LDS -> sorted by (rows[k] & mask(key_bits)), equal rows in their incoming order.
// n <= 16*NTH.  Every thread of the workgroup must call; ends with a barrier.
template <int NTH>
__device__ __forceinline__ void block_sort_rows(uint32_t *rows, uint16_t *tag, int n, int key_bits, int *hist, int *scan)
{
    constexpr int WAVES = NTH / 64, MAXI = 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (((n + WAVES - 1) / WAVES) + 63) & ~63; // elements per wave
    const int ns = chunk >> 6;                                 // <= MAXI rounds of 64
    const int base = wave * chunk;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    int *my_hist = hist + wave * HIST_STRIDE;
    __syncthreads(); // the caller's fill of rows / tag
    // the key's bits are dealt evenly to the passes (12 bits: 6 + 6, not 8 + 4): a pass costs a ballot per digit bit and a
    // scan over 2^bits counters per wave, so 6 + 6 is twelve ballots and two 64-digit scans where 8 + 4 was sixteen
    // ballots (the second pass went through all eight) and two 256-digit scans
    const int passes = (key_bits + 7) / 8, bits = (key_bits + passes - 1) / passes, D = 1 << bits;
    const uint32_t dmask = (uint32_t)D - 1u;
    for (int pass = 0, shift = 0; pass < passes; ++pass, shift += bits) {
        uint32_t r[MAXI];
        uint16_t p[MAXI];
        int loc[MAXI];
#pragma unroll
        for (int s = 0; s < MAXI; ++s) {
            const int k = base + s * 64 + lane;
            r[s] = 0;
            p[s] = 0;
            if (s < ns && k < n) {
                r[s] = rows[k];
                p[s] = tag[k];
            }
        }
        for (int c = lane; c < D; c += 64)
            my_hist[c] = 0;
        // rank inside the wave: LDS operations of one wave execute in order, so the running counter written in
        // round s is the one read in round s+1
#pragma unroll
        for (int s = 0; s < MAXI; ++s) {
            if (s < ns) { // wave-uniform
                const bool valid = base + s * 64 + lane < n;
                const uint32_t digit = (r[s] >> shift) & dmask;
                uint64_t peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    if (b < bits) { // wave-uniform
                        const bool bit = (digit >> b) & 1u;
                        const uint64_t bal = __ballot(bit);
                        peers &= bit ? bal : ~bal;
                    }
                }
                const int before = __popcll(peers & lt_mask), cnt = __popcll(peers);
                const int run = my_hist[digit];
                loc[s] = run + before;
                if (valid && before == 0)
                    my_hist[digit] = run + cnt;
            }
        }
        __syncthreads(); // every element is in registers, every wave's counts are final
        // exclusive scan of the D*WAVES counts in (digit, wave) order: up to four consecutive counts per thread
        int v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * tid + q;
            v[q] = c < D * WAVES ? hist[(c % WAVES) * HIST_STRIDE + c / WAVES] : 0;
            sum += v[q];
        }
        int total;
        int run = block_excl_scan<NTH>(sum, scan, &total);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 4 * tid + q;
            if (c < D * WAVES)
                hist[(c % WAVES) * HIST_STRIDE + c / WAVES] = run;
            run += v[q];
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < MAXI; ++s) {
            if (s < ns && base + s * 64 + lane < n) {
                const int dst = my_hist[(r[s] >> shift) & dmask] + loc[s];
                rows[dst] = r[s];
                tag[dst] = p[s];
            }
        }
        __syncthreads();
    }
}

// {first sorted position, count} of every table row from the rank-ordered run starts, written by the thread that
// owns rank q: its own row, and before it the absent rows between the previous present row and this one, which get
// {this run's start, 0} (= the lower bound of the row in the sorted order: consumers that walk several consecutive
// rows as one range rely on it).  q == total is the sentinel that closes the table: rows above the last present one
// get {n, 0}.
template <typename F>
__device__ __forceinline__ void rowseg_fill(int2 *rs, int64_t n_rows, int q, int total, int n, const uint32_t *rows,
                                            const uint16_t *seg, int pos_base, F count_of)
{
    const int k = q < total ? seg[q] : n;
    const int64_t row = q < total ? (int64_t)rows[k] : n_rows;
    const int64_t prev = q == 0 ? -1 : (int64_t)rows[seg[q - 1]];
    for (int64_t r = prev + 1; r < row && r < n_rows; ++r)
        rs[r] = make_int2(pos_base + k, 0);
    if (q < total && row < n_rows)
        rs[row] = make_int2(pos_base + k, count_of((int)seg[q + 1] - k));
}

} // namespace crec_sort
