// gs_sharded_rccl.cpp -- host side of the bucket-sharded MSB sort in C++ (include/gpusort_rccl.h).  The same steps as
// gpu-sort_amd/sharded.py (ShardedSorter, "msb" pipeline, one exchange group), with RCCL called directly.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "gpusort.h"
#include "gpusort_rccl.h"

namespace {
constexpr int RADIX = 256;
constexpr uint64_t MAX_MSG = 3ull << 26;   // elements per send/recv (768 MiB): see the header
inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
inline int nccl_err(ncclResult_t r) { return r == ncclSuccess ? 0 : 1000 + (int)r; }
}  // namespace

extern "C" {

void gs_sharded_compute_splits(const uint64_t *counts, int world, uint8_t *dest, uint64_t *per_rank)
{
    uint64_t tot[RADIX], n = 0;
    for (int b = 0; b < RADIX; ++b) {
        tot[b] = 0;
        for (int r = 0; r < world; ++r) tot[b] += counts[(size_t)r * RADIX + b];
        n += tot[b];
    }
    for (int r = 0; r < world; ++r) per_rank[r] = 0;
    uint64_t before = 0;
    int prev = 0;
    for (int b = 0; b < RADIX; ++b) {
        int d = 0;
        if (n) {
            // floor(world * before / n) in 128-bit arithmetic (sharded.py uses a double: same result wherever the
            // quotient is not within 2^-52 of an integer from below; both sides of an exchange run THIS code)
            d = (int)(((unsigned __int128)before * (unsigned)world) / n);
            if (d > world - 1) d = world - 1;
        }
        if (d < prev) d = prev;
        prev = d;
        dest[b] = (uint8_t)d;
        per_rank[d] += tot[b];
        before += tot[b];
    }
}

size_t gs_msb_sharded_temp_bytes(uint64_t num_items, uint64_t capacity, int has_values, int world)
{
    const size_t a = gs_lsb_temp_bytes(num_items, has_values), b = gs_msb_finish_temp_bytes(capacity, has_values, world);
    return align256(a > b ? a : b) + align256((size_t)RADIX * sizeof(uint64_t)) + align256((size_t)world * RADIX * sizeof(uint64_t));
}

int gs_msb_sort_u32_sharded(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, const uint32_t *d_vals_in,
                            uint64_t num_items, uint32_t *d_grouped_keys, uint32_t *d_grouped_vals, uint32_t *d_recv_keys,
                            uint32_t *d_recv_vals, uint32_t *d_keys_out, uint32_t *d_vals_out, uint64_t capacity,
                            uint64_t *num_out, void *nccl_comm, int rank, int world, int key_type, void *stream)
{
    if (!nccl_comm || world < 1 || world > RADIX || rank < 0 || rank >= world || !num_out) return hipErrorInvalidValue;
    const bool pairs = d_vals_in != nullptr;
    if (!d_temp || temp_bytes < gs_msb_sharded_temp_bytes(num_items, capacity, pairs, world)) return hipErrorInvalidValue;
    if (num_items >= (1ull << 32) || capacity >= (1ull << 32)) return hipErrorInvalidValue;
    if (pairs && (!d_grouped_vals || !d_recv_vals || !d_vals_out)) return hipErrorInvalidValue;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t s = (hipStream_t)stream;
    const size_t sort_ws = align256(std::max(gs_lsb_temp_bytes(num_items, pairs), gs_msb_finish_temp_bytes(capacity, pairs, world)));
    uint64_t *d_counts = (uint64_t *)((char *)d_temp + sort_ws);
    uint64_t *d_all = (uint64_t *)((char *)d_counts + align256((size_t)RADIX * sizeof(uint64_t)));

    // 1. first digit pass: the shard grouped by top byte + the 256 bucket sizes
    int e = gs_msb_first_pass_u32(d_temp, sort_ws, d_keys_in, d_grouped_keys, d_vals_in, d_grouped_vals, num_items, key_type, d_counts, s);
    if (e) return e;
    // 2. every rank learns every rank's bucket sizes (2 KiB per rank)
    if ((e = nccl_err(ncclAllGather(d_counts, d_all, RADIX, ncclUint64, comm, s)))) return e;
    std::vector<uint64_t> all((size_t)world * RADIX);
    if ((e = (int)hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s))) return e;
    if ((e = (int)hipStreamSynchronize(s))) return e;      // the one host synchronisation of the sort
    // 3. the same bucket -> rank map on every rank
    std::vector<uint8_t> dest(RADIX);
    std::vector<uint64_t> per_rank(world);
    gs_sharded_compute_splits(all.data(), world, dest.data(), per_rank.data());
    for (int r = 0; r < world; ++r)
        if (per_rank[r] > capacity) return GS_SHARDED_IMBALANCED;      // decided alike on every rank
    // my grouped shard: rank r's share is one contiguous slice (bucket order = key order)
    std::vector<uint64_t> send_off(world + 1, 0), recv_cnt(world, 0);
    for (int b = 0; b < RADIX; ++b) send_off[dest[b] + 1] += all[(size_t)rank * RADIX + b];
    for (int r = 0; r < world; ++r) send_off[r + 1] += send_off[r];
    std::vector<uint64_t> pieces((size_t)world * RADIX, 0);            // what every source sends me, per top byte
    for (int src = 0; src < world; ++src)
        for (int b = 0; b < RADIX; ++b)
            if (dest[b] == rank) { pieces[(size_t)src * RADIX + b] = all[(size_t)src * RADIX + b]; recv_cnt[src] += all[(size_t)src * RADIX + b]; }
    std::vector<uint64_t> recv_off(world + 1, 0);
    for (int r = 0; r < world; ++r) recv_off[r + 1] = recv_off[r] + recv_cnt[r];
    const uint64_t m = recv_off[world];
    *num_out = m;
    // 4. ONE exchange: a send and a receive per peer inside one group; in rounds where a message exceeds MAX_MSG
    uint64_t biggest = 0;
    for (int src = 0; src < world; ++src) {
        std::vector<uint64_t> to(world, 0);
        for (int b = 0; b < RADIX; ++b) to[dest[b]] += all[(size_t)src * RADIX + b];
        for (int r = 0; r < world; ++r) biggest = std::max(biggest, to[r]);
    }
    const uint64_t rounds = biggest ? (biggest + MAX_MSG - 1) / MAX_MSG : 1;
    for (uint64_t q = 0; q < rounds; ++q) {
        if ((e = nccl_err(ncclGroupStart()))) return e;
        for (int r = 0; r < world; ++r) {
            const uint64_t sc = send_off[r + 1] - send_off[r], rc = recv_cnt[r];
            const uint64_t s0 = std::min(q * MAX_MSG, sc), s1 = std::min((q + 1) * MAX_MSG, sc);
            const uint64_t r0 = std::min(q * MAX_MSG, rc), r1 = std::min((q + 1) * MAX_MSG, rc);
            if (s1 > s0) {
                if ((e = nccl_err(ncclSend(d_grouped_keys + send_off[r] + s0, s1 - s0, ncclUint32, r, comm, s)))) return e;
                if (pairs && (e = nccl_err(ncclSend(d_grouped_vals + send_off[r] + s0, s1 - s0, ncclUint32, r, comm, s)))) return e;
            }
            if (r1 > r0) {
                if ((e = nccl_err(ncclRecv(d_recv_keys + recv_off[r] + r0, r1 - r0, ncclUint32, r, comm, s)))) return e;
                if (pairs && (e = nccl_err(ncclRecv(d_recv_vals + recv_off[r] + r0, r1 - r0, ncclUint32, r, comm, s)))) return e;
            }
        }
        if ((e = nccl_err(ncclGroupEnd()))) return e;
    }
    // 5. the rest of the MSB sort on what arrived (pieces picked up where they lie)
    if (m == 0) return 0;
    return gs_msb_finish_u32(d_temp, sort_ws, d_recv_keys, pairs ? d_recv_vals : nullptr, d_keys_out, pairs ? d_vals_out : nullptr, m,
                             pieces.data(), world, key_type, s, 0);
}

}  // extern "C"
