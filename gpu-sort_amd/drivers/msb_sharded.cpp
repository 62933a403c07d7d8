// msb_sharded -- BASELINE configs[4] with a C++ host: one process per GPU, the MSB sort cut at its first digit, ONE
// grouped RCCL exchange (include/gpusort_rccl.h).  The reference has no multi-GPU driver (msb/tests/main.cu:30-31 picks
// one device); this one follows msb/src/test.cu's shape (generate, sort, time, check) for N ranks.
//   msb_sharded [--log2n 24] [--pairs] [--reps 3] [--spawn W]          W processes, GPU r % device_count each
//   msb_sharded ... --world W --rank R --id-file PATH                  one rank of a job started by something else
// Rank 0 prints one JSON line; exit code 0 only if every rank verified its slice (sorted, in range order across ranks,
// the global multiset unchanged).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <signal.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "gpusort.h"
#include "gpusort_rccl.h"

#define CK(x) do { int e_ = (int)(x); if (e_) { std::fprintf(stderr, "rank %d: %s failed: %d (line %d)\n", rank, #x, e_, __LINE__); return 3; } } while (0)

static int run_rank(int rank, int world, const std::string &id_file, int log2n, bool pairs, int reps)
{
    int ndev = 0;
    CK(hipGetDeviceCount(&ndev));
    CK(hipSetDevice(rank % ndev));
    ncclUniqueId id;
    if (rank == 0) {
        CK(ncclGetUniqueId(&id));
        if (world > 1) {
            FILE *f = std::fopen((id_file + ".tmp").c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) return 3;
            std::fclose(f);
            std::rename((id_file + ".tmp").c_str(), id_file.c_str());
        }
    } else {
        FILE *f = nullptr;
        for (int i = 0; i < 6000 && !(f = std::fopen(id_file.c_str(), "rb")); ++i) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || std::fread(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "rank %d: no id file\n", rank); return 3; }
        std::fclose(f);
    }
    ncclComm_t comm;
    CK(ncclCommInitRank(&comm, world, id, rank));
    hipStream_t s;
    CK(hipStreamCreate(&s));

    // start-up check: the communicator delivers the sorter's largest message (768 MiB) whole (skipped for small runs)
    if (log2n >= 28) CK(gs_sharded_selftest(comm, rank, world, 3ull << 26, s));

    const uint64_t n = 1ull << log2n, cap = n + n / 4 + 65536;
    uint32_t *kin, *grouped, *recv, *out, *vin = nullptr, *vgrouped = nullptr, *vrecv = nullptr, *vout = nullptr;
    CK(hipMalloc(&kin, n * 4)); CK(hipMalloc(&grouped, n * 4)); CK(hipMalloc(&recv, cap * 4)); CK(hipMalloc(&out, cap * 4));
    if (pairs) { CK(hipMalloc(&vin, n * 4)); CK(hipMalloc(&vgrouped, n * 4)); CK(hipMalloc(&vrecv, cap * 4)); CK(hipMalloc(&vout, cap * 4)); }
    const size_t tb = gs_msb_sharded_temp_bytes(n, cap, pairs, world);
    void *temp;
    CK(hipMalloc(&temp, tb));
    uint64_t *d_chk, *d_gather;
    CK(hipMalloc(&d_chk, 8 * sizeof(uint64_t))); CK(hipMalloc(&d_gather, (size_t)world * 8 * sizeof(uint64_t)));

    CK(gs_generate_u32(kin, n, 0 /*uniform*/, 0, (uint64_t)rank * n, 0, s));
    if (pairs) CK(gs_generate_u32(vin, n, 3 /*enumerated*/, 0, (uint64_t)rank * n, 0, s));
    uint64_t h_in[3];
    CK(gs_check_sorted_u32(kin, n, 0, d_chk, s));
    CK(hipMemcpyAsync(h_in, d_chk, sizeof(h_in), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));

    uint64_t m = 0;
    double best_ms = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipStreamSynchronize(s));
        const auto t0 = std::chrono::steady_clock::now();
        CK(gs_msb_sort_u32_sharded(temp, tb, kin, vin, n, grouped, vgrouped, recv, vrecv, out, vout, cap, &m, comm, rank, world, 0 /*u32*/, s));
        CK(hipStreamSynchronize(s));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms < best_ms) best_ms = ms;
    }
    // verification: my slice is sorted; slices are in rank order; the global multiset (sum and xor of a key hash) is the input's
    uint64_t h_out[3] = {0, 0, 0};
    uint32_t first = 0, last = 0;
    if (m) {
        CK(gs_check_sorted_u32(out, m, 0, d_chk, s));
        CK(hipMemcpyAsync(h_out, d_chk, sizeof(h_out), hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(&first, out, 4, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync(&last, out + m - 1, 4, hipMemcpyDeviceToHost, s));
        CK(hipStreamSynchronize(s));
    }
    const uint64_t mine[8] = {m, first, last, h_out[0], h_out[1], h_out[2], h_in[1], h_in[2]};
    CK(hipMemcpyAsync(d_chk, mine, sizeof(mine), hipMemcpyHostToDevice, s));
    CK(ncclAllGather(d_chk, d_gather, 8, ncclUint64, comm, s));
    std::vector<uint64_t> all((size_t)world * 8);
    CK(hipMemcpyAsync(all.data(), d_gather, all.size() * 8, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    bool ok = true;
    uint64_t total = 0, so = 0, xo = 0, si = 0, xi = 0, prev_last = 0;
    bool have_prev = false;
    for (int r = 0; r < world; ++r) {
        const uint64_t *a = &all[(size_t)r * 8];
        total += a[0]; ok = ok && a[3] == 0; so += a[4]; xo ^= a[5]; si += a[6]; xi ^= a[7];
        if (a[0]) { if (have_prev) ok = ok && prev_last <= a[1]; prev_last = a[2]; have_prev = true; }
    }
    ok = ok && total == (uint64_t)world * n && so == si && xo == xi;
    if (rank == 0)
        std::printf("{\"driver\": \"msb_sharded\", \"n_gpus\": %d, \"keys_per_gpu\": %llu, \"has_values\": %s, \"ms_per_sort_best_of_%d\": %.3f, "
                    "\"Gkeys_per_s\": %.2f, \"rank0_received\": %llu, \"verified\": %s}\n",
                    world, (unsigned long long)n, pairs ? "true" : "false", reps, best_ms, (double)world * n / best_ms / 1e6,
                    (unsigned long long)m, ok ? "true" : "false");
    ncclCommDestroy(comm);
    return ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    int log2n = 24, reps = 3, world = 1, rank = 0, spawn = 0;
    bool pairs = false;
    std::string id_file = "/tmp/gs_msb_sharded_id_" + std::to_string((long)getpid());
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() { return (i + 1 < argc) ? argv[++i] : (char *)"0"; };
        if (a == "--log2n") log2n = std::atoi(next());
        else if (a == "--reps") reps = std::atoi(next());
        else if (a == "--world") world = std::atoi(next());
        else if (a == "--rank") rank = std::atoi(next());
        else if (a == "--spawn") spawn = std::atoi(next());
        else if (a == "--id-file") id_file = next();
        else if (a == "--pairs") pairs = true;
        else { std::fprintf(stderr, "usage: msb_sharded [--log2n N] [--pairs] [--reps R] [--spawn W | --world W --rank R --id-file F]\n"); return 2; }
    }
    if (log2n < 10 || log2n > 31 || reps < 1) return 2;
    if (spawn > 1) {
        // children are started BEFORE this process touches the GPU; the parent only waits
        std::remove(id_file.c_str());
        std::vector<pid_t> kids;
        for (int r = 0; r < spawn; ++r) {
            const pid_t p = fork();
            if (p == 0) _exit(run_rank(r, spawn, id_file, log2n, pairs, reps));
            kids.push_back(p);
        }
        // Wait for ANY child.  As soon as one rank fails (or dies), its peers may be blocked in a collective that will
        // never complete: give them a bounded time to return on their own (the host reports a peer's LOCAL failure through
        // the size exchange, GS_SHARDED_PEER_FAILED), then end them, and exit non-zero -- the parent never waits for ever.
        int rc = 0;
        size_t left = kids.size();
        auto reap = [&](pid_t p, int st) {
            for (pid_t &k : kids)
                if (k == p) { k = -1; --left; }
            if (!WIFEXITED(st) || WEXITSTATUS(st)) rc = 1;
        };
        while (left && rc == 0) {
            int st = 0;
            const pid_t p = waitpid(-1, &st, 0);
            if (p < 0) { rc = 1; break; }
            reap(p, st);
        }
        if (left) {
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(20);
            while (left && std::chrono::steady_clock::now() < deadline) {
                int st = 0;
                const pid_t p = waitpid(-1, &st, WNOHANG);
                if (p > 0) reap(p, st);
                else std::this_thread::sleep_for(std::chrono::milliseconds(50));
            }
            for (pid_t k : kids)
                if (k > 0) { std::fprintf(stderr, "msb_sharded: ending rank process %ld after a peer's failure\n", (long)k); kill(k, SIGKILL); }
            for (pid_t k : kids)
                if (k > 0) { int st = 0; waitpid(k, &st, 0); }
        }
        std::remove(id_file.c_str());
        return rc;
    }
    return run_rank(rank, world, id_file, log2n, pairs, reps);
}
