// mock_rccl.cpp -- TEST INFRASTRUCTURE ONLY: an in-process stand-in for the handful of RCCL entry points
// csrc/rt_multi.cpp calls, built as librccl.so.1 and LD_PRELOADed by tests/test_parity_gpu.py::test_render_multi_rccl_call_sequence.
//
// Why: the builder's GPU boxes have ONE GPU, and real RCCL refuses a communicator with two ranks on one device, so the
// RCCL branch of rt_render_multi (ncclCommInitAll; ncclGroupStart / ncclRecv x (n-1) / ncclSend / ncclGroupEnd) can
// never run there.  This mock checks what can be checked without a second GPU: that every receive posted by the root
// is matched by exactly one send of the same size and type from the rank it names, that nothing is sent twice or left
// over, and that the bytes land where the scatter kernel expects them (the frame must equal the single-GPU frame).
// It is never linked into or shipped with the product.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

struct ncclComm {
  int rank, n, dev;
};
namespace {
struct Op {
  bool send;
  void* buf;
  size_t count;
  ncclDataType_t type;
  int peer;
  ncclComm* comm;
  hipStream_t stream;
  bool used = false;
  hipEvent_t ready = nullptr;  // deferred mode: recorded on the sender's stream when the send was posted
};
std::mutex g_mu;
int g_depth = 0;
std::vector<Op> g_ops;
char g_msg[256] = "mock RCCL: ok";
ncclResult_t bad(const char* m) {
  snprintf(g_msg, sizeof(g_msg), "mock RCCL: %s", m);
  fprintf(stderr, "%s\n", g_msg);
  return ncclInvalidUsage;
}
// MOCK_RCCL_DEFER=1: one process plays several process-per-GPU ranks one after the other (rt_render_gather_device is
// called rank by rank, each call its own ncclGroup): an operation whose partner has not been posted yet stays pending
// instead of failing the group, as a real send / receive would wait on its stream for the peer.
bool deferred() {
  static const bool d = getenv("MOCK_RCCL_DEFER") != nullptr;
  return d;
}
ncclResult_t flush() {
  // match every receive with its send and move the bytes (stream ordered: the receive waits for the sender's stream)
  for (Op& r : g_ops) {
    if (r.send || r.used) continue;
    Op* s = nullptr;
    for (Op& c : g_ops)
      if (c.send && !c.used && c.comm->rank == r.peer && c.peer == r.comm->rank) {
        s = &c;
        break;
      }
    if (!s && deferred()) continue;  // the sender has not called yet
    if (!s) return bad("a receive has no matching send");
    if (s->count != r.count || s->type != r.type) return bad("send / receive sizes differ");
    if (r.type != ncclUint32) return bad("unexpected data type");
    hipEvent_t ev = s->ready;
    if (!ev && (hipSetDevice(s->comm->dev) != hipSuccess || hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess ||
                hipEventRecord(ev, s->stream) != hipSuccess))
      return bad("event on the sender's stream failed");
    if (hipSetDevice(r.comm->dev) != hipSuccess || hipStreamWaitEvent(r.stream, ev, 0) != hipSuccess ||
        hipMemcpyAsync(r.buf, s->buf, r.count * 4, hipMemcpyDeviceToDevice, r.stream) != hipSuccess)
      return bad("copy failed");
    (void)hipEventDestroy(ev);
    s->used = r.used = true;
  }
  size_t pending = 0;
  for (Op& c : g_ops)
    if (!c.used) {
      if (!deferred()) return bad("a send has no matching receive");
      pending++;
    }
  if (deferred()) {
    std::vector<Op> keep;
    for (Op& c : g_ops)
      if (!c.used) keep.push_back(c);
    fprintf(stderr, "mock RCCL: %zu operations matched, %zu pending\n", g_ops.size() - pending, pending);
    g_ops.swap(keep);
    return ncclSuccess;
  }
  fprintf(stderr, "mock RCCL: group of %zu operations matched\n", g_ops.size());
  g_ops.clear();
  return ncclSuccess;
}
}  // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 7, sizeof(*id));
  return ncclSuccess;
}
ncclResult_t ncclCommInitAll(ncclComm_t* comms, int n, const int* devs) {
  for (int i = 0; i < n; i++) comms[i] = new ncclComm{i, n, devs ? devs[i] : i};
  fprintf(stderr, "mock RCCL: ncclCommInitAll(%d ranks)\n", n);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n, ncclUniqueId, int rank) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  *comm = new ncclComm{rank, n, dev};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t c) {
  fprintf(stderr, "mock RCCL: ncclCommAbort(rank %d)\n", c ? c->rank : -1);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) {
  *n = c->n;
  return ncclSuccess;
}
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) {
  *r = c->rank;
  return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t) { return g_msg; }
ncclResult_t ncclGroupStart() {
  std::lock_guard<std::mutex> l(g_mu);
  g_depth++;
  return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return bad("ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  return flush();
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return bad("ncclSend outside a group");
  if (peer < 0 || peer >= comm->n || peer == comm->rank) return bad("ncclSend: bad peer");
  Op op{true, const_cast<void*>(buf), count, type, peer, comm, stream};
  if (deferred()) {  // the data is complete at this point of the sender's stream
    if (hipSetDevice(comm->dev) != hipSuccess || hipEventCreateWithFlags(&op.ready, hipEventDisableTiming) != hipSuccess ||
        hipEventRecord(op.ready, stream) != hipSuccess)
      return bad("event on the sender's stream failed");
  }
  g_ops.push_back(op);
  return ncclSuccess;
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_depth <= 0) return bad("ncclRecv outside a group");
  if (peer < 0 || peer >= comm->n || peer == comm->rank) return bad("ncclRecv: bad peer");
  g_ops.push_back(Op{false, buf, count, type, peer, comm, stream});
  return ncclSuccess;
}
}
