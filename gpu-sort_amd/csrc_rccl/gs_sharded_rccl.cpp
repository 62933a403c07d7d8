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

void gs_sharded_exchange_plan(const uint64_t *counts, const uint8_t *dest, int rank, int world, uint64_t *send_off,
                              uint64_t *recv_off, uint64_t *pieces, uint64_t *rounds)
{
    // my grouped shard: rank r's share is one contiguous slice (bucket order = key order, dest is monotone)
    for (int r = 0; r <= world; ++r) send_off[r] = recv_off[r] = 0;
    for (int b = 0; b < RADIX; ++b) send_off[dest[b] + 1] += counts[(size_t)rank * RADIX + b];
    for (int r = 0; r < world; ++r) send_off[r + 1] += send_off[r];
    // what every source sends me, per top byte: the receive buffer is source-major, and inside a source's piece the
    // buckets keep their order -- the layout gs_msb_finish_u32 takes its piece table for
    for (int src = 0; src < world; ++src) {
        uint64_t c = 0;
        for (int b = 0; b < RADIX; ++b) {
            const uint64_t x = dest[b] == rank ? counts[(size_t)src * RADIX + b] : 0;
            pieces[(size_t)src * RADIX + b] = x;
            c += x;
        }
        recv_off[src + 1] = recv_off[src] + c;
    }
    // every rank derives the same number of rounds from the gathered sizes: the largest (source, destination) message
    uint64_t biggest = 0;
    for (int src = 0; src < world; ++src) {
        std::vector<uint64_t> to(world, 0);
        for (int b = 0; b < RADIX; ++b) to[dest[b]] += counts[(size_t)src * RADIX + b];
        for (int r = 0; r < world; ++r) biggest = std::max(biggest, to[r]);
    }
    *rounds = biggest ? (biggest + MAX_MSG - 1) / MAX_MSG : 1;
}

size_t gs_msb_sharded_temp_bytes(uint64_t num_items, uint64_t capacity, int has_values, int world)
{
    const size_t a = gs_lsb_temp_bytes(num_items, has_values), b = gs_msb_finish_temp_bytes(capacity, has_values, world);
    return align256(a > b ? a : b) + align256((size_t)RADIX * sizeof(uint64_t)) + align256((size_t)world * RADIX * sizeof(uint64_t));
}

// One grouped exchange round.  Whatever happens between ncclGroupStart and ncclGroupEnd, the group is CLOSED before
// this returns: an early return would leave the communicator inside an open group (every later call on it would be
// queued into that group and never run).
static int exchange_round(ncclComm_t comm, hipStream_t s, int world, bool pairs, uint64_t q, const uint64_t *send_off,
                          const uint64_t *recv_off, const uint32_t *gk, const uint32_t *gv, uint32_t *rk, uint32_t *rv)
{
    int e = nccl_err(ncclGroupStart());
    if (e) return e;
    for (int r = 0; r < world && !e; ++r) {
        const uint64_t sc = send_off[r + 1] - send_off[r], rc = recv_off[r + 1] - recv_off[r];
        const uint64_t s0 = std::min(q * MAX_MSG, sc), s1 = std::min((q + 1) * MAX_MSG, sc);
        const uint64_t r0 = std::min(q * MAX_MSG, rc), r1 = std::min((q + 1) * MAX_MSG, rc);
        if (s1 > s0) {
            e = nccl_err(ncclSend(gk + send_off[r] + s0, s1 - s0, ncclUint32, r, comm, s));
            if (!e && pairs) e = nccl_err(ncclSend(gv + send_off[r] + s0, s1 - s0, ncclUint32, r, comm, s));
        }
        if (!e && r1 > r0) {
            e = nccl_err(ncclRecv(rk + recv_off[r] + r0, r1 - r0, ncclUint32, r, comm, s));
            if (!e && pairs) e = nccl_err(ncclRecv(rv + recv_off[r] + r0, r1 - r0, ncclUint32, r, comm, s));
        }
    }
    const int e2 = nccl_err(ncclGroupEnd());
    return e ? e : e2;
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

    // 1. first digit pass: the shard grouped by top byte + the 256 bucket sizes.  A LOCAL failure up to the size
    // exchange still takes part in it (with zero sizes and a flag), so that the peers are not left waiting in the
    // all-gather: every rank learns that a rank failed and all return.
    int e = gs_msb_first_pass_u32(d_temp, sort_ws, d_keys_in, d_grouped_keys, d_vals_in, d_grouped_vals, num_items, key_type, d_counts, s);
    const int local_fail = e;
    if (local_fail) {
        std::vector<uint64_t> poison(RADIX, ~0ull);      // no real bucket holds 2^64 - 1 keys
        if (hipMemcpyAsync(d_counts, poison.data(), RADIX * sizeof(uint64_t), hipMemcpyHostToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess)
            return local_fail;                           // the device itself is gone: nothing more can be done for the peers
    }
    // 2. every rank learns every rank's bucket sizes (2 KiB per rank)
    if ((e = nccl_err(ncclAllGather(d_counts, d_all, RADIX, ncclUint64, comm, s)))) return local_fail ? local_fail : e;
    std::vector<uint64_t> all((size_t)world * RADIX);
    if ((e = (int)hipMemcpyAsync(all.data(), d_all, all.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, s))) return e;
    if ((e = (int)hipStreamSynchronize(s))) return e;      // the one host synchronisation of the sort
    if (local_fail) return local_fail;
    for (int r = 0; r < world; ++r)
        if (all[(size_t)r * RADIX] == ~0ull) return GS_SHARDED_PEER_FAILED;      // seen alike by every rank
    // 3. the same bucket -> rank map on every rank
    std::vector<uint8_t> dest(RADIX);
    std::vector<uint64_t> per_rank(world);
    gs_sharded_compute_splits(all.data(), world, dest.data(), per_rank.data());
    for (int r = 0; r < world; ++r)
        if (per_rank[r] > capacity) return GS_SHARDED_IMBALANCED;      // decided alike on every rank
    std::vector<uint64_t> send_off(world + 1), recv_off(world + 1), pieces((size_t)world * RADIX);
    uint64_t rounds = 1;
    gs_sharded_exchange_plan(all.data(), dest.data(), rank, world, send_off.data(), recv_off.data(), pieces.data(), &rounds);
    const uint64_t m = recv_off[world];
    *num_out = m;
    // 4. ONE exchange: a send and a receive per peer inside one group; in rounds where a message exceeds MAX_MSG.
    // A failure inside the exchange leaves no way to tell the peers (they may already wait in it): the communicator is
    // aborted, so that their pending operations end with an error instead of waiting for this rank for ever.
    for (uint64_t q = 0; q < rounds; ++q) {
        e = exchange_round(comm, s, world, pairs, q, send_off.data(), recv_off.data(), d_grouped_keys, d_grouped_vals, d_recv_keys, d_recv_vals);
        if (e) {
            ncclCommAbort(comm);
            return e;
        }
    }
    // 5. the rest of the MSB sort on what arrived (pieces picked up where they lie)
    if (m == 0) return 0;
    return gs_msb_finish_u32(d_temp, sort_ws, d_recv_keys, pairs ? d_recv_vals : nullptr, d_keys_out, pairs ? d_vals_out : nullptr, m,
                             pieces.data(), world, key_type, s, 0);
}

int gs_sharded_selftest(void *nccl_comm, int rank, int world, uint64_t elements, void *stream)
{
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || elements == 0 || elements >= (1ull << 32)) return hipErrorInvalidValue;
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t s = (hipStream_t)stream;
    const int nxt = (rank + 1) % world, prv = (rank + world - 1) % world;
    uint32_t *d_src = nullptr, *d_dst = nullptr;
    int e = (int)hipMalloc(&d_src, elements * sizeof(uint32_t));
    if (!e) e = (int)hipMalloc(&d_dst, elements * sizeof(uint32_t));
    constexpr uint64_t CH = 1ull << 24;                 // staged through 64 MiB of host memory
    std::vector<uint32_t> h(std::min(CH, elements));
    for (uint64_t at = 0; at < elements && !e; at += CH) {
        const uint64_t c = std::min(CH, elements - at);
        for (uint64_t i = 0; i < c; ++i) h[i] = (uint32_t)(at + i) * 747796405u + (uint32_t)(rank + 1);
        e = (int)hipMemcpyAsync(d_src + at, h.data(), c * sizeof(uint32_t), hipMemcpyHostToDevice, s);
        if (!e) e = (int)hipStreamSynchronize(s);
    }
    if (!e) e = (int)hipMemsetAsync(d_dst, 0, elements * sizeof(uint32_t), s);
    if (!e) {
        e = nccl_err(ncclGroupStart());
        if (!e) {
            int e1 = nccl_err(ncclSend(d_src, elements, ncclUint32, nxt, comm, s));
            if (!e1) e1 = nccl_err(ncclRecv(d_dst, elements, ncclUint32, prv, comm, s));
            const int e2 = nccl_err(ncclGroupEnd());
            e = e1 ? e1 : e2;
        }
    }
    if (!e) e = (int)hipStreamSynchronize(s);
    uint64_t bad = 0;
    for (uint64_t at = 0; at < elements && !e; at += CH) {
        const uint64_t c = std::min(CH, elements - at);
        e = (int)hipMemcpyAsync(h.data(), d_dst + at, c * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (!e) e = (int)hipStreamSynchronize(s);
        for (uint64_t i = 0; i < c && !e; ++i) bad += h[i] != (uint32_t)(at + i) * 747796405u + (uint32_t)(prv + 1);
    }
    if (d_src) (void)hipFree(d_src);
    if (d_dst) (void)hipFree(d_dst);
    if (e) return e;
    return bad ? GS_SHARDED_TRUNCATED : 0;
}

}  // extern "C"
