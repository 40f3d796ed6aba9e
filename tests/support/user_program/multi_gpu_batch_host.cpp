// TEST FIXTURE / EXAMPLE — a C++ HOST program of the C-ABI (include/slpx.h), not part of the product.
//
// INTEGRATION.md §5 as a program: one process per GPU, a batch of independent problems split in
// contiguous blocks (slpx_shard_range — the rule of bench.py --workload batch512 and
// sleipnir_amd/dist.py, multistart.hpp:52-62), every rank stepping its block on its own device
// with NO collective inside the timed region, and RCCL only at its ends: a barrier, the MAX of the
// elapsed times, and an all-gather of the per-problem {index, info, delta, gamma} rows.
//
//   RANK, LOCAL_RANK, WORLD_SIZE      as set by any launcher (defaults 0, 0, 1)
//   SLPX_NCCL_ID_FILE                 where rank 0 leaves the ncclUniqueId for the others
//   multi_gpu_batch_host <problems> <N> <steps> [--no-comm]
// --no-comm skips RCCL and prints this rank's rows only: with RANK / WORLD_SIZE set by hand it
// shows on ONE device that a problem's step does not depend on the shard it is in.
//
//   hipcc -O2 -std=c++17 multi_gpu_batch_host.cpp -I include -L sleipnir_amd -lslpx -L tests/support -lslpx_models -lrccl
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <slpx.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

// the model is the host's own program: here the benchmark fixture (tests/support/models/bench_models.cpp)
extern "C" slpx_problem* bench_models_cart_pole(int32_t N, double dt);

namespace {
#define CHECK_HIP(x)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (x);                                                                 \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)
#define CHECK_NCCL(x)                                                                    \
  do {                                                                                   \
    ncclResult_t r_ = (x);                                                               \
    if (r_ != ncclSuccess) {                                                             \
      std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_));                      \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)
#define CHECK_SLPX(x)                                                                    \
  do {                                                                                   \
    if ((x) != 0) {                                                                      \
      std::fprintf(stderr, "%s: %s\n", #x, slpx_last_error());                           \
      std::exit(2);                                                                      \
    }                                                                                    \
  } while (0)

int env_int(const char* name, int fallback) {
  const char* v = std::getenv(name);
  return v ? std::atoi(v) : fallback;
}

// rank 0 writes the id (atomically: temporary name, then rename), the others wait for the file
ncclUniqueId exchange_id(int rank, const std::string& path) {
  ncclUniqueId id;
  if (rank == 0) {
    CHECK_NCCL(ncclGetUniqueId(&id));
    const std::string tmp = path + ".tmp";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) std::exit(2);
    std::fclose(f);
    std::rename(tmp.c_str(), path.c_str());
    return id;
  }
  for (int tries = 0; tries < 6000; ++tries) {
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
      const bool ok = std::fread(&id, sizeof(id), 1, f) == 1;
      std::fclose(f);
      if (ok) return id;
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  std::fprintf(stderr, "no ncclUniqueId at %s after 60 s\n", path.c_str());
  std::exit(2);
}

uint64_t fnv(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = static_cast<const unsigned char*>(p);
  for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 1099511628211ull;
  return h;
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <problems> <N> <steps> [--no-comm]\n", argv[0]);
    return 2;
  }
  const int total = std::atoi(argv[1]), N = std::atoi(argv[2]), steps = std::atoi(argv[3]);
  const bool no_comm = argc > 4 && std::strcmp(argv[4], "--no-comm") == 0;
  const int rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", 0), world = env_int("WORLD_SIZE", 1);
  if (slpx_device_count() <= 0) {
    std::printf("no HIP device (there is no CPU fallback)\n");
    return 3;
  }
  CHECK_HIP(hipSetDevice(no_comm ? 0 : local));

  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  CHECK_HIP(hipStreamCreate(&stream));
  if (!no_comm) {
    const char* id_file = std::getenv("SLPX_NCCL_ID_FILE");
    const ncclUniqueId id = exchange_id(rank, id_file ? id_file : "/tmp/slpx_nccl_id");
    CHECK_NCCL(ncclCommInitRank(&comm, world, id, rank));
  }

  // this rank's block of the batch
  int64_t lo = 0, hi = 0;
  slpx_shard_range(total, rank, world, &lo, &hi);
  const int mine = static_cast<int>(hi - lo);
  slpx_problem* xp = bench_models_cart_pole(N, 5.0 / N);  // the same structure on every rank
  int32_t n = 0, m_e = 0, m_i = 0;
  slpx_problem_dims(xp, &n, &m_e, &m_i);
  std::vector<double> x0(n);
  slpx_problem_get_x(xp, x0.data());
  slpx_system* sys = slpx_system_create(xp, mine, no_comm ? 0 : local, nullptr, 0);
  if (!sys) {
    std::fprintf(stderr, "slpx_system_create: %s\n", slpx_last_error());
    return 2;
  }

  // value sets: a function of the GLOBAL problem index only
  std::vector<double> x(size_t(mine) * n), s(size_t(mine) * m_i), y(size_t(mine) * m_e), z(size_t(mine) * m_i), mu(mine, 0.1);
  for (int b = 0; b < mine; ++b) {
    std::mt19937_64 gen(20260928ull + static_cast<uint64_t>(lo + b));
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; ++i) x[size_t(b) * n + i] = x0[i] + 0.05 * (u(gen) - 0.5);
    for (int i = 0; i < m_i; ++i) s[size_t(b) * m_i + i] = 0.5 + u(gen);
    for (int i = 0; i < m_e; ++i) y[size_t(b) * m_e + i] = 0.2 * (u(gen) - 0.5);
    for (int i = 0; i < m_i; ++i) z[size_t(b) * m_i + i] = 0.5 + u(gen);
  }
  CHECK_SLPX(slpx_system_set_state(sys, x.data(), s.data(), y.data(), z.data(), mu.data()));

  std::vector<int32_t> info(mine);
  CHECK_SLPX(slpx_newton_steps(sys, 2, 1, 1, info.data()));  // warm-up
  int* d_flag = nullptr;
  double* d_time = nullptr;
  CHECK_HIP(hipMalloc(&d_flag, sizeof(int)));
  CHECK_HIP(hipMalloc(&d_time, 2 * sizeof(double)));
  CHECK_HIP(hipMemset(d_flag, 0, sizeof(int)));
  auto barrier = [&] {
    if (comm) CHECK_NCCL(ncclAllReduce(d_flag, d_flag, 1, ncclInt, ncclSum, comm, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipDeviceSynchronize());
  };
  barrier();
  const auto t0 = std::chrono::steady_clock::now();
  CHECK_SLPX(slpx_newton_steps(sys, steps, 1, 1, info.data()));  // no collective in here
  CHECK_SLPX(slpx_system_sync(sys));
  double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  barrier();

  // rows {global index, info, delta, gamma} and a hash of the step itself
  std::vector<double> reg(2 * size_t(mine));
  CHECK_SLPX(slpx_system_regularization(sys, reg.data()));
  int64_t sizes[SLPX_INFO_COUNT];
  slpx_system_info(sys, sizes);
  const int64_t dim = sizes[SLPX_INFO_N] + sizes[SLPX_INFO_ME];
  std::vector<double> p(size_t(mine) * dim);
  slpx_system_get(sys, 3, p.data());
  const int cap = (total + world - 1) / world;
  std::vector<double> rows(size_t(cap) * 4, -1.0);
  for (int b = 0; b < mine; ++b) {
    rows[4 * size_t(b) + 0] = static_cast<double>(lo + b);
    rows[4 * size_t(b) + 1] = info[b];
    rows[4 * size_t(b) + 2] = reg[2 * size_t(b)];
    rows[4 * size_t(b) + 3] = reg[2 * size_t(b) + 1];
    if (no_comm)
      std::printf("row %lld info %d delta %a gamma %a step %016llx\n", static_cast<long long>(lo + b), info[b],
                  reg[2 * size_t(b)], reg[2 * size_t(b) + 1],
                  static_cast<unsigned long long>(fnv(1469598103934665603ull, p.data() + size_t(b) * dim, 8 * dim)));
  }
  if (no_comm) {
    // the hash rank 0 of an RCCL run prints, over this process's rows (RANK=0 WORLD_SIZE=1: all of them)
    uint64_t h = 1469598103934665603ull;
    for (int b = 0; b < mine; ++b) h = fnv(h, &rows[4 * size_t(b) + 2], 16);
    std::printf("{\"ranks\": 1, \"problems\": %d, \"rows\": %d, \"regularization_hash\": \"%016llx\"}\n", total, mine,
                static_cast<unsigned long long>(h));
  }
  if (comm) {
    double* d_rows = nullptr;
    double* d_table = nullptr;
    CHECK_HIP(hipMalloc(&d_rows, rows.size() * sizeof(double)));
    CHECK_HIP(hipMalloc(&d_table, rows.size() * sizeof(double) * world));
    CHECK_HIP(hipMemcpy(d_rows, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(d_time, &elapsed, sizeof(double), hipMemcpyHostToDevice));
    CHECK_NCCL(ncclAllReduce(d_time, d_time + 1, 1, ncclDouble, ncclMax, comm, stream));
    CHECK_NCCL(ncclAllGather(d_rows, d_table, rows.size(), ncclDouble, comm, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    std::vector<double> table(rows.size() * world);
    double t_max = 0.0;
    CHECK_HIP(hipMemcpy(table.data(), d_table, table.size() * sizeof(double), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&t_max, d_time + 1, sizeof(double), hipMemcpyDeviceToHost));
    if (rank == 0) {
      int seen = 0, failed = 0;
      uint64_t h = 1469598103934665603ull;
      for (size_t r = 0; r < table.size() / 4; ++r) {
        if (table[4 * r] < 0) continue;  // padding of a short block
        failed += table[4 * r] != static_cast<double>(seen) || table[4 * r + 1] != 0.0;  // in problem order, all Success
        h = fnv(h, &table[4 * r + 2], 16);
        ++seen;
      }
      std::printf("{\"ranks\": %d, \"problems\": %d, \"N\": %d, \"steps\": %d, \"seconds_max_over_ranks\": %.6f, "
                  "\"newton_steps_per_s\": %.1f, \"rows\": %d, \"rows_out_of_order_or_failed\": %d, "
                  "\"regularization_hash\": \"%016llx\"}\n",
                  world, total, N, steps, t_max, double(total) * steps / t_max, seen, failed,
                  static_cast<unsigned long long>(h));
      if (seen != total || failed) return 1;
    }
    CHECK_HIP(hipFree(d_rows));
    CHECK_HIP(hipFree(d_table));
    CHECK_NCCL(ncclCommDestroy(comm));
  }
  slpx_system_destroy(sys);
  slpx_problem_destroy(xp);
  return 0;
}
