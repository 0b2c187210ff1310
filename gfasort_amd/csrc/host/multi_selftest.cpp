// multi_rank_selftest — the multi-device C ABI driven the way a Rust host with ONE THREAD PER GPU would drive it
// (INTEGRATION.md §6), in C++: R ranks (host threads), each a gfs_rank, gfs_rank_run with a caller-supplied all-reduce.
// The collective here is a host-staged sum behind a barrier (all ranks share device 0 on a one-GPU box; RCCL refuses two
// ranks on one device): what is under test is gfs_rank_create / gfs_rank_run / the exchange of the shared slots / the
// final completion through the callback path — not a collective library.
//   usage: multi_rank_selftest [ranks = 2] [nodes = 60000] [paths = 24] [window = 6000] [dims = 0] [merge_every = 1] [unvisited = 0]
// unvisited: that many extra nodes no path steps on (dense indices nodes .. nodes + unvisited - 1): no rank's span covers them, and
// they must end where they started on every replica (sgd.rs:286-294), not at the 0 a sum of "what I own" with no owner gives.
// Prints "ok ..." and exits 0 when every rank ends with the same positions and (dims = 0) the chain sorts exactly.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <thread>
#include <vector>

#include "../../../include/gfasort_hip.h"

namespace {

struct Barrier {
    std::mutex m; std::condition_variable cv; int n, waiting = 0; uint64_t gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
    }
};

struct Shared {
    int world;
    Barrier bar;
    std::vector<std::vector<unsigned char>> stage;     // per rank
    std::vector<unsigned char> sum;
    uint64_t collectives = 0, bytes = 0;
    explicit Shared(int w) : world(w), bar(w), stage(w) {}
};
struct RankUser { Shared *sh; int rank; };

// sum device_buf[0..count) over the ranks, in place, ordered on hip_stream
int host_allreduce(void *user, void *device_buf, uint64_t count, int is_f64, void *hip_stream) {
    RankUser *u = static_cast<RankUser *>(user);
    Shared &sh = *u->sh;
    const size_t bytes = (size_t)count * (is_f64 ? 8 : 4);
    if (hipStreamSynchronize((hipStream_t)hip_stream) != hipSuccess) return 1;
    sh.stage[u->rank].resize(bytes);
    if (hipMemcpy(sh.stage[u->rank].data(), device_buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    sh.bar.wait();
    if (u->rank == 0) {
        sh.sum.assign(bytes, 0);
        for (int r = 0; r < sh.world; ++r) {
            if (sh.stage[r].size() != bytes) { std::fprintf(stderr, "rank %d brought %zu bytes, rank 0 %zu\n", r, sh.stage[r].size(), bytes); std::abort(); }
            if (is_f64) { double *d = (double *)sh.sum.data(); const double *s = (const double *)sh.stage[r].data(); for (uint64_t k = 0; k < count; ++k) d[k] += s[k]; }
            else { float *d = (float *)sh.sum.data(); const float *s = (const float *)sh.stage[r].data(); for (uint64_t k = 0; k < count; ++k) d[k] += s[k]; }
        }
        sh.collectives++; sh.bytes += bytes;
    }
    sh.bar.wait();
    if (hipMemcpy(device_buf, sh.sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    sh.bar.wait();                                     // nobody overwrites `sum` before everybody has read it
    return 0;
}

uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace

int main(int argc, char **argv) {
    const int R = argc > 1 ? std::atoi(argv[1]) : 2;
    const uint64_t N = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 60000, P = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 24,
                   W = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 6000, D = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;
    const uint32_t merge_every = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 1;
    const uint64_t U = argc > 7 ? std::strtoull(argv[7], nullptr, 10) : 0, NT = N + U;
    if (gfs_device_count() < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    // windows(N, P, W): a chain of N nodes, path p covers nodes o_p .. o_p + W - 1; nodes in block-shuffled input order
    std::vector<uint32_t> order(N), inv(N), node_len(NT), step_node;
    std::iota(order.begin(), order.end(), 0u);
    uint64_t s1 = 11, s2 = 12;
    for (uint64_t b = 0; b < N; b += 64) {
        const uint64_t e = std::min(N, b + 64);
        for (uint64_t k = e - 1; k > b; --k) std::swap(order[k], order[b + splitmix(s1) % (k - b + 1)]);
    }
    for (uint64_t k = 0; k < N; ++k) inv[order[k]] = (uint32_t)k;              // chain node c sits at dense index inv[c]
    for (uint64_t k = 0; k < NT; ++k) node_len[k] = 1 + (uint32_t)(splitmix(s2) % 16);
    std::vector<uint64_t> first(1, 0);
    for (uint64_t p = 0; p < P; ++p) {
        const uint64_t o = P > 1 ? p * (N - W) / (P - 1) : 0;
        for (uint64_t k = 0; k < W; ++k) step_node.push_back(inv[o + k]);
        first.push_back(step_node.size());
    }
    std::vector<uint8_t> rev(step_node.size(), 0);
    gfs_graph_view g{NT, step_node.size(), P, node_len.data(), step_node.data(), rev.data(), first.data()};
    gfs_sgd_params prm{};
    prm.iter_max = D ? 30 : 100; prm.min_term_updates = (D ? 10 : 1) * step_node.size(); prm.eps = 0.01; prm.eta_max = (double)W * (double)W;
    prm.theta = 0.99; prm.space = D ? W : W * 17; prm.space_max = D ? 1000 : 100; prm.space_quantization_step = 100; prm.cooling_start = 0.5;
    prm.seed = 9399220;

    Shared sh(R);
    std::vector<std::vector<double>> result(R);
    std::vector<gfs_rank_info> info(R);
    std::vector<int> rc(R, 0);
    std::vector<std::thread> th;
    for (int r = 0; r < R; ++r) th.emplace_back([&, r] {
        gfs_rank_config cfg{};
        cfg.rank = (uint32_t)r; cfg.world = (uint32_t)R; cfg.device = 0; cfg.merge_every = merge_every;
        gfs_rank *rk = nullptr;
        RankUser user{&sh, r};
        int e = gfs_rank_create(&g, &prm, D, &cfg, &rk);
        if (e < 0) { std::fprintf(stderr, "rank %d create: %s\n", r, gfs_last_error()); rc[r] = e; return; }
        const uint64_t len = D ? NT * 2 * D : NT;
        std::vector<double> x(len, 0.0);
        if (D) {
            gfs_init_layout(&g, D, prm.seed, x.data());
            e = gfs_rank_set_positions(rk, x.data(), len);
        } else e = gfs_rank_set_positions(rk, nullptr, 0);
        if (e >= 0) e = gfs_rank_run(rk, R > 1 ? host_allreduce : nullptr, &user, nullptr);
        if (e < 0) { std::fprintf(stderr, "rank %d run: %s\n", r, gfs_last_error()); rc[r] = e; gfs_rank_destroy(rk); return; }
        gfs_rank_get_info(rk, &info[r]);
        gfs_rank_get_positions(rk, x.data(), len);
        result[r] = std::move(x);
        gfs_rank_destroy(rk);
    });
    for (auto &t : th) t.join();
    for (int r = 0; r < R; ++r) if (rc[r] < 0) return 1;
    for (int r = 1; r < R; ++r)
        if (std::memcmp(result[0].data(), result[r].data(), result[0].size() * 8) != 0) { std::fprintf(stderr, "FAIL: rank %d ends with other positions than rank 0\n", r); return 1; }
    uint64_t quota = 0;
    for (int r = 0; r < R; ++r) quota += info[r].quota;
    if (quota != prm.min_term_updates) { std::fprintf(stderr, "FAIL: quotas sum to %llu, not %llu\n", (unsigned long long)quota, (unsigned long long)prm.min_term_updates); return 1; }
    if (U) {
        std::vector<double> x0(D ? NT * 2 * D : NT, 0.0);
        if (D) gfs_init_layout(&g, D, prm.seed, x0.data()); else gfs_init_positions(&g, x0.data());
        const uint64_t w = D ? 2 * D : 1;
        for (uint64_t k = N * w; k < NT * w; ++k)
            if (result[0][k] != x0[k]) { std::fprintf(stderr, "FAIL: unvisited node %llu moved from %g to %g\n", (unsigned long long)(k / w), x0[k], result[0][k]); return 1; }
    }
    uint64_t inversions = 0;
    if (D == 0) {
        std::vector<uint64_t> ord(N);
        gfs_sort_order(result[0].data(), N, ord.data());                       // (the visited nodes: the chain)
        uint64_t up = 0, down = 0;
        for (uint64_t k = 1; k < N; ++k) { const int64_t d = (int64_t)order[ord[k]] - (int64_t)order[ord[k - 1]]; up += d != 1; down += d != -1; }
        inversions = std::min(up, down);
        if (inversions) { std::fprintf(stderr, "FAIL: %llu inversions against chain order\n", (unsigned long long)inversions); return 1; }
    } else {
        for (double v : result[0]) if (!(v == v)) { std::fprintf(stderr, "FAIL: NaN\n"); return 1; }
    }
    std::printf("ok ranks %d nodes %llu paths %llu dims %llu merge_every %u unvisited %llu: replicas identical, %s; shared slots %llu of %llu, "
                "%llu collectives, %.1f MB through the callback\n", R, (unsigned long long)N, (unsigned long long)P, (unsigned long long)D, merge_every, (unsigned long long)U,
                D ? "finite" : "exact chain order", (unsigned long long)info[0].shared_slots, (unsigned long long)N,
                (unsigned long long)sh.collectives, (double)sh.bytes / 1e6);
    return 0;
}
