// The N>1 exchange of libcurdle_g1.so: one process per GPU, ONE tiny collective per MSM.
//
// The reference has no multi-device code at all; the contract is BASELINE.json's north_star: "a single very
// large MSM shards its window buckets across the 8 GPUs of one node with a final RCCL all-reduce of partial G1 sums over
// xGMI".  RCCL has no elliptic-curve reduction operator, so that all-reduce is an all-gather of one 144-byte point blob per
// rank followed by world-1 host additions on every rank (G1 addition is commutative and associative and the final encoding
// is canonical, so every rank ends bit-identical).
//
// Two transports behind one handle, no PyTorch anywhere:
//   * a CONTROL channel over TCP on the loopback interface (rank 0 is the hub of a star): rendezvous, barriers, the
//     max-over-ranks clock of bench.py, the per-proof verdict gather of proof-per-GPU sharding -- and, alone, the whole
//     exchange when several ranks rehearse on ONE GPU or on a CPU-only box (RCCL refuses two ranks on one device);
//   * RCCL (cg1_comm_attach_rccl): ncclGetUniqueId on rank 0, the id broadcast over the control channel,
//     ncclCommInitRank on the context's device; from then on cg1_comm_allgather / cg1_comm_allreduce_g1 are one
//     ncclAllGather on the context's compute stream (payload staged through pinned memory: 144 B per rank).
// librccl.so (573 MB) is dlopen'ed the first time a communicator attaches it, so processes that never go multi-GPU do not
// map it.
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>          // types and enums only; the entry points are resolved with dlsym
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
#include "host_g1.h"
#include "../../include/curdle_g1.h"

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

bool load_rccl(char* err, size_t errlen) {
  if (g_rccl.lib) return true;
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  for (const char* nm : names) { h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { snprintf(err, errlen, "dlopen(librccl.so) failed: %s", dlerror()); return false; }
  Rccl r;
  r.lib = h;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name); if (!r.field) { snprintf(err, errlen, "librccl.so has no %s", name); return false; }
  SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
  SYM(CommCount, "ncclCommCount") SYM(CommUserRank, "ncclCommUserRank") SYM(AllGather, "ncclAllGather")
  SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
  g_rccl = r;
  return true;
}

constexpr uint32_t MAGIC = 0x43473143u;      // "CG1C"
struct Hello { uint32_t magic; uint32_t rank; uint32_t world; uint32_t pad; uint64_t nonce; };
struct OpHeader { uint32_t magic; uint32_t seq; uint64_t bytes; };

}  // namespace

struct cg1_comm {
  int rank = 0, world = 1;
  int listen_fd = -1, port = 0;
  std::vector<int> peer;               // rank 0: fd of every other rank (index = rank); others: peer[0] = hub
  bool connected = false;
  uint32_t seq = 0;
  int timeout_ms = 120000;
  char err[256] = {0};
  // RCCL half
  cg1_ctx* ctx = nullptr;
  ncclComm_t nccl = nullptr;
  int device = -1;
  hipStream_t stream = nullptr;
  void *d_send = nullptr, *d_recv = nullptr; uint8_t* h_stage = nullptr; size_t cap = 0;
  std::vector<uint8_t> scratch;
};

namespace {

int fail(cg1_comm* c, const char* what, int e = 0) {
  if (e) snprintf(c->err, sizeof c->err, "%s: %s", what, strerror(e));
  else snprintf(c->err, sizeof c->err, "%s", what);
  return CG1_ERR_COMM;
}

int64_t now_ms() { return std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// whole-buffer send / receive with a deadline (sockets are blocking; poll() bounds each wait)
int send_all(cg1_comm* c, int fd, const void* buf, size_t n) {
  const uint8_t* p = static_cast<const uint8_t*>(buf);
  const int64_t deadline = now_ms() + c->timeout_ms;
  while (n) {
    // the sockets are blocking: wait for room with poll() (bounded by the deadline), then send without blocking -- a stalled peer
    // can no longer hold the hub inside ::send for ever
    pollfd pf{fd, POLLOUT, 0};
    const int pr = poll(&pf, 1, 1000);
    if (pr < 0 && errno == EINTR) continue;
    if (pr < 0) return fail(c, "poll", errno);
    if (pr == 0) { if (now_ms() > deadline) return fail(c, "send timed out (a peer rank stopped reading)"); continue; }
    ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL | MSG_DONTWAIT);
    if (k > 0) { p += k; n -= (size_t)k; continue; }
    if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
    return fail(c, "send", errno);
  }
  return CG1_OK;
}
int recv_all(cg1_comm* c, int fd, void* buf, size_t n) {
  uint8_t* p = static_cast<uint8_t*>(buf);
  const int64_t deadline = now_ms() + c->timeout_ms;
  while (n) {
    pollfd pf{fd, POLLIN, 0};
    int pr = poll(&pf, 1, 1000);
    if (pr < 0 && errno == EINTR) continue;
    if (pr < 0) return fail(c, "poll", errno);
    if (pr == 0) { if (now_ms() > deadline) return fail(c, "receive timed out (a peer rank died or never arrived)"); continue; }
    ssize_t k = ::recv(fd, p, n, 0);
    if (k > 0) { p += k; n -= (size_t)k; continue; }
    if (k == 0) return fail(c, "peer closed the connection");
    if (errno == EINTR || errno == EAGAIN) continue;
    return fail(c, "recv", errno);
  }
  return CG1_OK;
}

void tune(int fd) {
  int one = 1;
  (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
}

// all-gather over the star: every rank's `bytes` to the hub, the concatenation back to every rank
int socket_allgather(cg1_comm* c, const void* send, size_t bytes, void* recv) {
  if (c->world == 1) { if (bytes) memcpy(recv, send, bytes); return CG1_OK; }
  if (!c->connected) return fail(c, "communicator is not connected");
  const OpHeader h{MAGIC, ++c->seq, (uint64_t)bytes};
  uint8_t* out = static_cast<uint8_t*>(recv);
  if (c->rank == 0) {
    if (bytes) memcpy(out, send, bytes);
    for (int r = 1; r < c->world; ++r) {
      OpHeader g;
      int rc = recv_all(c, c->peer[r], &g, sizeof g);
      if (rc) return rc;
      if (g.magic != MAGIC || g.seq != h.seq || g.bytes != h.bytes) return fail(c, "ranks disagree on the collective (sequence / size mismatch)");
      if (bytes && (rc = recv_all(c, c->peer[r], out + (size_t)r * bytes, bytes))) return rc;
    }
    for (int r = 1; r < c->world; ++r) {
      int rc = send_all(c, c->peer[r], &h, sizeof h);
      if (rc) return rc;
      if (bytes && (rc = send_all(c, c->peer[r], out, bytes * (size_t)c->world))) return rc;
    }
    return CG1_OK;
  }
  int rc = send_all(c, c->peer[0], &h, sizeof h);
  if (rc) return rc;
  if (bytes && (rc = send_all(c, c->peer[0], send, bytes))) return rc;
  OpHeader g;
  if ((rc = recv_all(c, c->peer[0], &g, sizeof g))) return rc;
  if (g.magic != MAGIC || g.seq != h.seq || g.bytes != h.bytes) return fail(c, "ranks disagree on the collective (sequence / size mismatch)");
  if (bytes && (rc = recv_all(c, c->peer[0], out, bytes * (size_t)c->world))) return rc;
  return CG1_OK;
}

#define COMM_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(c->err, sizeof c->err, "%s failed: %s", #x, hipGetErrorString(e_)); return CG1_ERR_HIP; } } while (0)
#define COMM_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { snprintf(c->err, sizeof c->err, "%s failed: %s", #x, g_rccl.GetErrorString(r_)); return CG1_ERR_COMM; } } while (0)

int rccl_reserve(cg1_comm* c, size_t bytes) {
  if (bytes <= c->cap) return CG1_OK;
  size_t cap = 256;
  while (cap < bytes) cap *= 2;
  COMM_HIP(hipSetDevice(c->device));
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  c->d_send = c->d_recv = nullptr; c->h_stage = nullptr; c->cap = 0;
  COMM_HIP(hipMalloc(&c->d_send, cap));
  COMM_HIP(hipMalloc(&c->d_recv, cap * (size_t)c->world));
  COMM_HIP(hipHostMalloc((void**)&c->h_stage, cap * (size_t)(c->world + 1)));
  c->cap = cap;
  return CG1_OK;
}

// one ncclAllGather on the context's compute stream; payload staged through pinned memory
int rccl_allgather(cg1_comm* c, const void* send, size_t bytes, void* recv) {
  if (bytes == 0) return CG1_OK;
  int rc = rccl_reserve(c, bytes);
  if (rc) return rc;
  COMM_HIP(hipSetDevice(c->device));
  memcpy(c->h_stage, send, bytes);
  COMM_HIP(hipMemcpyAsync(c->d_send, c->h_stage, bytes, hipMemcpyHostToDevice, c->stream));
  COMM_NCCL(g_rccl.AllGather(c->d_send, c->d_recv, bytes, ncclUint8, c->nccl, c->stream));
  uint8_t* back = c->h_stage + c->cap;
  COMM_HIP(hipMemcpyAsync(back, c->d_recv, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream));
  COMM_HIP(hipStreamSynchronize(c->stream));
  memcpy(recv, back, bytes * (size_t)c->world);
  return CG1_OK;
}

}  // namespace

extern "C" {

cg1_comm* cg1_comm_create(int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) return nullptr;
  cg1_comm* c = new cg1_comm();
  c->rank = rank; c->world = world;
  c->peer.assign(world, -1);
  if (world == 1) { c->connected = true; return c; }
  if (rank == 0) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) { delete c; return nullptr; }
    int one = 1;
    (void)setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in a{};
    a.sin_family = AF_INET; a.sin_addr.s_addr = htonl(INADDR_LOOPBACK); a.sin_port = 0;       // one node: loopback only, ephemeral port
    socklen_t al = sizeof a;
    if (bind(fd, (sockaddr*)&a, sizeof a) < 0 || listen(fd, world + 8) < 0 || getsockname(fd, (sockaddr*)&a, &al) < 0) { ::close(fd); delete c; return nullptr; }
    c->listen_fd = fd;
    c->port = ntohs(a.sin_port);
  }
  return c;
}

int cg1_comm_port(const cg1_comm* c) { return c ? c->port : 0; }
int cg1_comm_rank(const cg1_comm* c) { return c ? c->rank : -1; }
const char* cg1_comm_error(const cg1_comm* c) { return c ? c->err : "null communicator"; }
const char* cg1_comm_transport(const cg1_comm* c) { return (c && c->nccl) ? "rccl" : "socket"; }

int cg1_comm_set_timeout(cg1_comm* c, int timeout_ms) {
  if (!c || timeout_ms < 1) return CG1_ERR_ARG;
  c->timeout_ms = timeout_ms;
  return CG1_OK;
}

int cg1_comm_connect(cg1_comm* c, const char* host, int port, uint64_t nonce, int timeout_ms) {
  if (!c) return CG1_ERR_ARG;
  if (c->connected) return CG1_OK;
  const int64_t deadline = now_ms() + (timeout_ms > 0 ? timeout_ms : c->timeout_ms);
  if (c->rank == 0) {
    int have = 0;
    while (have < c->world - 1) {
      pollfd pf{c->listen_fd, POLLIN, 0};
      int pr = poll(&pf, 1, 500);
      if (pr < 0 && errno != EINTR) return fail(c, "poll(listen)", errno);
      if (pr <= 0) { if (now_ms() > deadline) { snprintf(c->err, sizeof c->err, "rendezvous timed out: %d of %d ranks arrived", have + 1, c->world); return CG1_ERR_COMM; } continue; }
      int fd = ::accept(c->listen_fd, nullptr, nullptr);
      if (fd < 0) continue;
      tune(fd);
      Hello h{};
      const int keep = c->timeout_ms; c->timeout_ms = 5000;
      int rc = recv_all(c, fd, &h, sizeof h);
      c->timeout_ms = keep;
      // a stranger on the port (stale rendezvous file of another run, a port scanner): drop it and keep waiting
      if (rc || h.magic != MAGIC || h.nonce != nonce || h.world != (uint32_t)c->world || h.rank == 0 || h.rank >= (uint32_t)c->world || c->peer[h.rank] >= 0) { ::close(fd); continue; }
      c->peer[h.rank] = fd;
      ++have;
    }
    const Hello ok{MAGIC, 0, (uint32_t)c->world, 0, nonce};
    for (int r = 1; r < c->world; ++r) { int rc = send_all(c, c->peer[r], &ok, sizeof ok); if (rc) return rc; }
    ::close(c->listen_fd); c->listen_fd = -1;
    c->err[0] = 0;
    c->connected = true;
    return CG1_OK;
  }
  sockaddr_in a{};
  a.sin_family = AF_INET; a.sin_port = htons((uint16_t)port);
  if (inet_pton(AF_INET, host && *host ? host : "127.0.0.1", &a.sin_addr) != 1) return fail(c, "rendezvous host must be an IPv4 literal (one node: 127.0.0.1)");
  // Knock for at most 2 s: a refused connection usually means the rendezvous data is stale (the caller re-reads it and calls
  // again).  Once the hub has taken the hello, wait for its acknowledgement -- sent when ALL ranks are in -- up to timeout_ms.
  const int64_t knock_until = std::min<int64_t>(deadline, now_ms() + 2000);
  for (;;) {
    int fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return fail(c, "socket", errno);
    if (::connect(fd, (sockaddr*)&a, sizeof a) == 0) {
      tune(fd);
      const Hello h{MAGIC, (uint32_t)c->rank, (uint32_t)c->world, 0, nonce};
      Hello ok{};
      const int keep = c->timeout_ms;
      c->timeout_ms = (int)std::max<int64_t>(1000, deadline - now_ms());
      int rc = send_all(c, fd, &h, sizeof h);
      if (!rc) rc = recv_all(c, fd, &ok, sizeof ok);
      c->timeout_ms = keep;
      if (!rc && ok.magic == MAGIC && ok.nonce == nonce && ok.world == (uint32_t)c->world) { c->peer[0] = fd; c->connected = true; c->err[0] = 0; return CG1_OK; }
      ::close(fd);
      return fail(c, "rendezvous: the listener on that port did not acknowledge this rank (stale rendezvous data?)");
    }
    ::close(fd);
    if (now_ms() > knock_until) return fail(c, "rendezvous: cannot connect to rank 0 (not listening yet, or stale rendezvous data)");
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

int cg1_comm_attach_rccl(cg1_comm* c, cg1_ctx* ctx) {
  if (!c || !ctx) return CG1_ERR_ARG;
  if (c->nccl) return CG1_OK;
  if (!c->connected) return fail(c, "attach_rccl needs the control channel connected first");
  // every rank reports whether it could load the library BEFORE anyone enters ncclCommInitRank (which would hang on a missing rank)
  uint8_t mine = load_rccl(c->err, sizeof c->err) ? 1 : 0;
  std::vector<uint8_t> all(c->world);
  int rc = socket_allgather(c, &mine, 1, all.data());
  if (rc) return rc;
  for (int r = 0; r < c->world; ++r) if (!all[r]) { if (mine) snprintf(c->err, sizeof c->err, "rank %d could not load librccl.so", r); return CG1_ERR_COMM; }
  // rank 0's unique id travels over the control channel with a validity byte, so a failed ncclGetUniqueId is an error on every
  // rank at once (nobody is left waiting inside ncclCommInitRank for a rank that never comes)
  struct IdMsg { uint8_t ok; uint8_t pad[7]; ncclUniqueId id; } msg;
  memset(&msg, 0, sizeof msg);
  if (c->rank == 0) {
    ncclResult_t r = g_rccl.GetUniqueId(&msg.id);
    msg.ok = r == ncclSuccess ? 1 : 0;
    if (!msg.ok) snprintf(c->err, sizeof c->err, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
  }
  std::vector<uint8_t> ids(sizeof msg * (size_t)c->world);
  if ((rc = socket_allgather(c, &msg, sizeof msg, ids.data()))) return rc;
  memcpy(&msg, ids.data(), sizeof msg);                                // rank 0's
  if (!msg.ok) { if (c->rank != 0) snprintf(c->err, sizeof c->err, "rank 0 could not create the RCCL unique id"); return CG1_ERR_COMM; }
  ncclUniqueId id = msg.id;
  c->ctx = ctx;
  c->device = cg1_ctx_device(ctx);
  c->stream = static_cast<hipStream_t>(cg1_ctx_stream(ctx));
  COMM_HIP(hipSetDevice(c->device));
  COMM_NCCL(g_rccl.CommInitRank(&c->nccl, c->world, id, c->rank));
  return rccl_reserve(c, 256);
}

int cg1_comm_world_seen(const cg1_comm* c) {
  if (!c) return 0;
  if (c->nccl) { int n = 0; if (g_rccl.CommCount(c->nccl, &n) == ncclSuccess) return n; return -1; }
  if (!c->connected) return 0;
  if (c->world == 1) return 1;
  int n = 1;
  if (c->rank == 0) { for (int r = 1; r < c->world; ++r) if (c->peer[r] >= 0) ++n; return n; }
  return c->world;       // the hub's acknowledgement only arrives when every rank is in
}

int cg1_comm_allgather(cg1_comm* c, const void* send, size_t bytes, void* recv) {
  if (!c || (bytes && (!send || !recv))) return CG1_ERR_ARG;
  if (c->nccl) return rccl_allgather(c, send, bytes, recv);
  return socket_allgather(c, send, bytes, recv);
}

// host-side control collectives always use the TCP channel (no device work: barriers, clocks, verdict gathers)
int cg1_comm_allgather_host(cg1_comm* c, const void* send, size_t bytes, void* recv) {
  if (!c || (bytes && (!send || !recv))) return CG1_ERR_ARG;
  return socket_allgather(c, send, bytes, recv);
}

int cg1_comm_barrier(cg1_comm* c) {
  if (!c) return CG1_ERR_ARG;
  uint8_t b = 1;
  c->scratch.resize((size_t)c->world);
  return socket_allgather(c, &b, 1, c->scratch.data());
}

// the "all-reduce of partial G1 sums": all-gather of one point blob per rank, then world-1 host additions in rank order
int cg1_comm_allreduce_g1(cg1_comm* c, const uint8_t* partial, uint8_t* sum, uint8_t* all_blobs) {
  if (!c || !partial || !sum) return CG1_ERR_ARG;
  c->scratch.resize((size_t)c->world * CG1_POINT_BYTES);
  int rc = cg1_comm_allgather(c, partial, CG1_POINT_BYTES, c->scratch.data());
  if (rc) return rc;
  cg1h::jac acc = cg1h::jac_identity();
  for (int r = 0; r < c->world; ++r) {
    cg1h::jac p;
    memcpy(&p, c->scratch.data() + (size_t)r * CG1_POINT_BYTES, sizeof p);
    // a peer's blob is data from another process: canonical field elements on the curve (or the identity), or the call fails
    if (!cg1h::fe_is_canonical(p.X) || !cg1h::fe_is_canonical(p.Y) || !cg1h::fe_is_canonical(p.Z) || !cg1h::jac_on_curve(p)) {
      snprintf(c->err, sizeof c->err, "rank %d sent a partial sum that is not a point of the curve", r);
      return CG1_ERR_COMM;
    }
    acc = cg1h::jac_add(acc, p);
  }
  memcpy(sum, &acc, sizeof acc);
  if (all_blobs) memcpy(all_blobs, c->scratch.data(), c->scratch.size());
  return CG1_OK;
}

void cg1_comm_destroy(cg1_comm* c) {
  if (!c) return;
  if (c->nccl) {
    (void)hipSetDevice(c->device);
    (void)g_rccl.CommDestroy(c->nccl);
  }
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (int fd : c->peer) if (fd >= 0) ::close(fd);
  if (c->listen_fd >= 0) ::close(c->listen_fd);
  delete c;
}

}  // extern "C"
