// Sanitizer driver for csrc/comm.cpp (socket transport): `world` ranks as threads of one process, each with its own cg1_comm --
// rendezvous (with a stranger knocking on the hub's port first), byte all-gathers of several sizes, the G1 all-reduce against a
// locally computed sum, barriers, a collective the ranks disagree on (must fail, not hang), destruction.  Built by
// tests/test_sanitizers.py with -fsanitize=address,undefined and with -fsanitize=thread.  The HIP / RCCL half of comm.cpp is not
// exercised here (no GPU in the sanitizer build): the HIP entry points it references are stubbed to fail.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>
#include <atomic>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "../../curdleproofs_pie_amd/csrc/host_g1.h"
#include "../../include/curdle_g1.h"

extern "C" {
// stand-ins for the symbols comm.cpp takes from the HIP runtime and from msm_gpu.hip (never reached on the socket transport)
hipError_t hipSetDevice(int) { return hipErrorNoDevice; }
hipError_t hipMalloc(void**, size_t) { return hipErrorNoDevice; }
hipError_t hipFree(void*) { return hipSuccess; }
hipError_t hipHostMalloc(void**, size_t, unsigned int) { return hipErrorNoDevice; }
hipError_t hipHostFree(void*) { return hipSuccess; }
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipErrorNoDevice; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipErrorNoDevice; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
int cg1_ctx_device(const cg1_ctx*) { return -1; }
void* cg1_ctx_stream(cg1_ctx*) { return nullptr; }
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); _exit(1); } } while (0)

int main() {
  const int world = 4;
  const uint64_t nonce = 0x1234567890abcdefull;
  cg1_comm* hub = cg1_comm_create(0, world);
  CHECK(hub && cg1_comm_port(hub) > 0);
  const int port = cg1_comm_port(hub);
  std::atomic<int> bad{0};
  // a stranger connects first and sends garbage: the hub must drop it and keep waiting for the real ranks
  std::thread stranger([&] {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{}; a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port); a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    if (::connect(fd, (sockaddr*)&a, sizeof a) == 0) { const char junk[24] = "not a curdle_g1 rank!!!"; (void)!::write(fd, junk, sizeof junk); }
    usleep(200000);
    ::close(fd);
  });
  auto rank_main = [&](int r) {
    cg1_comm* c = r == 0 ? hub : cg1_comm_create(r, world);
    if (!c) { ++bad; return; }
    int rc = CG1_ERR_COMM;
    for (int attempt = 0; attempt < 50 && rc != CG1_OK; ++attempt) rc = cg1_comm_connect(c, "127.0.0.1", port, nonce, 20000);
    if (rc != CG1_OK) { fprintf(stderr, "rank %d: %s\n", r, cg1_comm_error(c)); ++bad; return; }
    if (cg1_comm_world_seen(c) != world || strcmp(cg1_comm_transport(c), "socket") != 0) ++bad;
    for (size_t bytes : {size_t(1), size_t(144), size_t(100000)}) {
      std::vector<uint8_t> mine(bytes, (uint8_t)(r + 1)), all(bytes * world);
      if (cg1_comm_allgather(c, mine.data(), bytes, all.data()) != CG1_OK) { ++bad; return; }
      for (int q = 0; q < world; ++q) if (all[q * bytes] != q + 1 || all[q * bytes + bytes - 1] != q + 1) ++bad;
    }
    // G1 all-reduce: rank r contributes (r + 2) * G; everyone must end with (2 + 3 + 4 + 5) * G
    uint8_t k[32] = {0}; k[0] = (uint8_t)(r + 2);
    cg1h::jac part = cg1h::jac_mul(cg1h::jac_generator(), k), sum;
    uint8_t blob[CG1_POINT_BYTES], out[CG1_POINT_BYTES], allb[CG1_POINT_BYTES * 4];
    memcpy(blob, &part, sizeof part);
    if (cg1_comm_allreduce_g1(c, blob, out, allb) != CG1_OK) { ++bad; return; }
    memcpy(&sum, out, sizeof sum);
    uint8_t k14[32] = {0}; k14[0] = 14;
    if (!cg1h::jac_eq(sum, cg1h::jac_mul(cg1h::jac_generator(), k14))) ++bad;
    if (cg1_comm_barrier(c) != CG1_OK) ++bad;
    // ranks disagree on the size of a collective: an error on the hub and on at least the odd one out, never a hang
    cg1_comm_set_timeout(c, 3000);
    std::vector<uint8_t> mine(8 + (r == 2 ? 1 : 0), 7), all(9 * world);
    const int mrc = cg1_comm_allgather_host(c, mine.data(), mine.size(), all.data());
    if (r == 0 && mrc == CG1_OK) ++bad;
    cg1_comm_destroy(c);
  };
  std::vector<std::thread> th;
  for (int r = 0; r < world; ++r) th.emplace_back(rank_main, r);
  for (auto& t : th) t.join();
  stranger.join();
  CHECK(bad.load() == 0);
  CHECK(cg1_comm_create(3, 2) == nullptr && cg1_comm_create(-1, 2) == nullptr);
  cg1_comm* solo = cg1_comm_create(0, 1);
  uint8_t x = 9, y = 0;
  CHECK(solo && cg1_comm_allgather(solo, &x, 1, &y) == CG1_OK && y == 9 && cg1_comm_barrier(solo) == CG1_OK);
  cg1_comm_destroy(solo);
  printf("sanitize ok\n");
  return 0;
}
