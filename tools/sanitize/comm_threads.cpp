// Sanitizer harness for the HOST transport of pgicp_allgather_edges (pgicp_comm.cpp compiled WITH the sanitizer and linked
// into this program): `world` threads of one process play the ranks -- each makes its own communicator on the same
// shared-memory file, runs `rounds` collectives with uneven shards and checks the gathered list.  Under
// -fsanitize=thread this checks the generation-counter protocol for races; under address,undefined the packing.
//   comm_threads WORLD ROUNDS SHM_PATH
#include "pgicp.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>

static std::atomic<int> failures{0};

static void rank_main(int world, int rank, int rounds, const char *path, int n_pairs)
{
    pgicp_comm *comm = nullptr;
    if (pgicp_comm_create_host(world, rank, path, n_pairs, &comm) != PGICP_OK) { std::fprintf(stderr, "rank %d: create failed\n", rank); failures++; return; }
    std::vector<int64_t> cost(n_pairs);
    for (int i = 0; i < n_pairs; i++) cost[i] = 1000 + 37 * ((i * 7919) % 101);
    std::vector<int> own(n_pairs), shard(n_pairs, -1);
    int n_own = 0;
    if (pgicp_shard_pairs(n_pairs, cost.data(), world, rank, own.data(), n_pairs, &n_own) != PGICP_OK) { failures++; return; }
    for (int k = 0; k < n_own; k++) shard[own[k]] = rank;
    int slots = 0;
    if (pgicp_shard_slots(n_pairs, cost.data(), world, &slots) != PGICP_OK) { failures++; return; }
    for (int r = 0; r < rounds; r++) {
        std::vector<pgicp_edge> mine; std::vector<int> idx;
        for (int i = 0; i < n_pairs; i++)
            if (shard[i] == rank && !(r == 1 && i % 5 == 0)) {            // round 1: every fifth pair goes unreported
                pgicp_edge e; std::memset(&e, 0, sizeof e);
                e.from_id = i; e.to_id = 1000 + i; e.status = 0; e.accepted = 1; e.iterations = r; e.residual = 0.5 * i + r;
                mine.push_back(e); idx.push_back(i);
            }
        std::vector<pgicp_edge> all(n_pairs);
        if (pgicp_allgather_edges(comm, mine.data(), idx.data(), (int)mine.size(), slots, n_pairs, all.data()) != PGICP_OK) { failures++; break; }
        for (int i = 0; i < n_pairs; i++) {
            const bool skipped = (r == 1 && i % 5 == 0);
            if (skipped ? all[i].from_id != -1 : (all[i].from_id != i || all[i].to_id != 1000 + i || all[i].residual != 0.5 * i + r || all[i].reserved[0] != 0.0)) {
                std::fprintf(stderr, "rank %d round %d pair %d: wrong edge (from %lld)\n", rank, r, i, (long long)all[i].from_id); failures++; break;
            }
        }
    }
    pgicp_comm_destroy(comm);
}

int main(int argc, char **argv)
{
    const int world = argc > 1 ? std::atoi(argv[1]) : 4, rounds = argc > 2 ? std::atoi(argv[2]) : 3;
    const char *path = argc > 3 ? argv[3] : "/dev/shm/pgicp_sanitize_comm";
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, world, r, rounds, path, 37);
    for (auto &t : th) t.join();
    std::printf("comm_threads world %d rounds %d: %s\n", world, rounds, failures.load() ? "FAILED" : "ok");
    return failures.load() ? 1 : 0;
}
