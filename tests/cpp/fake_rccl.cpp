// fake_rccl.cpp - a TEST DOUBLE for librccl, for ONE purpose: to take the library's multi-rank code (uh_rccl_attach, the reservoir
// bands' all-gather, uh_rccl_gather_tiles: pack -> grouped ncclSend / ncclRecv -> compose) through a real job of several rank
// PROCESSES on a box that has one GPU, where RCCL itself refuses ("duplicate GPU"). Built by tests/test_gpu_rehearsal.py as
// librccl.so.1 into a scratch directory that the test puts first on the ranks' LD_LIBRARY_PATH: the library opens librccl by name at run
// time, so nothing in the product knows. Never shipped, never on a product path; a real multi-GPU job runs RCCL.
// What it implements (the ten entry points the library resolves): a communicator = one Unix-domain socket per pair of ranks (paths
// from the ncclUniqueId); Send / Recv / AllGather move DEVICE buffers through host staging (hipStreamSynchronize, hipMemcpy) - the
// semantics of a collective enqueued on a stream, made synchronous: everything enqueued before it has run when it starts, everything
// after it sees its result. Groups run their sends on a helper thread beside the receives, so two ranks that send to each other do
// not wait for each other.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Op {
   bool send;
   void* buf;
   size_t bytes;
   int peer;
   hipStream_t stream;
};
struct Comm {
   int rank = 0, world = 1;
   std::vector<int> sock;  // per peer (-1: self)
   int listener = -1;
   std::string path;
};
thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm*, Op>> g_ops;

size_t type_size(ncclDataType_t t) {
   switch ((int)t) {
      case 0: case 1: return 1;   // int8 / char, uint8
      case 2: case 3: return 4;   // int32, uint32
      case 4: case 5: return 8;   // int64, uint64
      case 6: return 2;           // half
      case 7: return 4;           // float
      case 8: return 8;           // double
      case 9: return 2;           // bfloat16
      default: return 1;
   }
}
std::string path_of(const ncclUniqueId& id, int rank) {
   char hex[33];
   for (int k = 0; k < 16; k++) std::snprintf(hex + 2 * k, 3, "%02x", (unsigned char)id.internal[k]);
   return std::string("/tmp/fake_rccl_") + hex + "_" + std::to_string(rank);
}
bool write_all(int fd, const void* p, size_t n) {
   const char* c = (const char*)p;
   while (n) {
      ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
      if (k <= 0) return false;
      c += k;
      n -= (size_t)k;
   }
   return true;
}
bool read_all(int fd, void* p, size_t n) {
   char* c = (char*)p;
   while (n) {
      ssize_t k = ::recv(fd, c, n, 0);
      if (k <= 0) return false;
      c += k;
      n -= (size_t)k;
   }
   return true;
}
// every op of a finished group, in the order it was posted
ncclResult_t run(std::vector<std::pair<Comm*, Op>>& ops) {
   bool ok = true;
   for (auto& co : ops) (void)hipStreamSynchronize(co.second.stream);  // what was enqueued before the collective has run
   // a rank's sends to itself pair up with its receives from itself, in order: device to device
   std::vector<size_t> self_send, self_recv;
   for (size_t i = 0; i < ops.size(); i++)
      if (ops[i].second.peer == ops[i].first->rank) (ops[i].second.send ? self_send : self_recv).push_back(i);
   for (size_t k = 0; k < self_send.size() && k < self_recv.size(); k++) {
      const Op &s = ops[self_send[k]].second, &r = ops[self_recv[k]].second;
      if (s.buf != r.buf) ok &= hipMemcpy(r.buf, s.buf, s.bytes < r.bytes ? s.bytes : r.bytes, hipMemcpyDeviceToDevice) == hipSuccess;
   }
   std::thread sender([&] {
      std::vector<char> host;
      for (auto& co : ops) {
         const Op& o = co.second;
         if (!o.send || o.peer == co.first->rank) continue;
         host.resize(o.bytes);
         if (hipMemcpy(host.data(), o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) ok = false;
         const uint64_t n = o.bytes;
         if (!write_all(co.first->sock[o.peer], &n, sizeof n) || !write_all(co.first->sock[o.peer], host.data(), o.bytes)) ok = false;
      }
   });
   {
      std::vector<char> host;
      for (auto& co : ops) {
         const Op& o = co.second;
         if (o.send || o.peer == co.first->rank) continue;
         uint64_t n = 0;
         if (!read_all(co.first->sock[o.peer], &n, sizeof n) || n != o.bytes) {
            ok = false;
            continue;
         }
         host.resize(o.bytes);
         if (!read_all(co.first->sock[o.peer], host.data(), o.bytes)) ok = false;
         if (hipMemcpy(o.buf, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false;
      }
   }
   sender.join();
   ops.clear();
   return ok ? ncclSuccess : ncclSystemError;
}
ncclResult_t post(Comm* c, Op o) {
   g_ops.push_back({c, o});
   return g_depth ? ncclSuccess : run(g_ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
   std::memset(id, 0, sizeof *id);
   FILE* f = std::fopen("/dev/urandom", "rb");
   if (!f || std::fread(id->internal, 1, 16, f) != 16) {
      if (f) std::fclose(f);
      return ncclSystemError;
   }
   std::fclose(f);
   return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
   Comm* c = new Comm;
   c->rank = rank;
   c->world = nranks;
   c->sock.assign(nranks, -1);
   c->path = path_of(id, rank);
   if (nranks > 1) {
      c->listener = ::socket(AF_UNIX, SOCK_STREAM, 0);
      sockaddr_un a{};
      a.sun_family = AF_UNIX;
      std::strncpy(a.sun_path, c->path.c_str(), sizeof a.sun_path - 1);
      ::unlink(c->path.c_str());
      if (c->listener < 0 || ::bind(c->listener, (sockaddr*)&a, sizeof a) != 0 || ::listen(c->listener, nranks) != 0) return ncclSystemError;
      // the pair (lo, hi): hi connects to lo
      for (int p = 0; p < rank; p++) {
         const std::string pp = path_of(id, p);
         int fd = -1;
         for (int tries = 0; tries < 3000; tries++) {  // 30 s
            fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
            sockaddr_un b{};
            b.sun_family = AF_UNIX;
            std::strncpy(b.sun_path, pp.c_str(), sizeof b.sun_path - 1);
            if (::connect(fd, (sockaddr*)&b, sizeof b) == 0) break;
            ::close(fd);
            fd = -1;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
         }
         if (fd < 0) return ncclSystemError;
         const int32_t me = rank;
         if (!write_all(fd, &me, sizeof me)) return ncclSystemError;
         c->sock[p] = fd;
      }
      for (int k = rank + 1; k < nranks; k++) {
         const int fd = ::accept(c->listener, nullptr, nullptr);
         int32_t who = -1;
         if (fd < 0 || !read_all(fd, &who, sizeof who) || who <= rank || who >= nranks) return ncclSystemError;
         c->sock[who] = fd;
      }
   }
   *out = (ncclComm_t)c;
   return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
   Comm* c = (Comm*)comm;
   if (!c) return ncclSuccess;
   for (int fd : c->sock)
      if (fd >= 0) ::close(fd);
   if (c->listener >= 0) {
      ::close(c->listener);
      ::unlink(c->path.c_str());
   }
   delete c;
   return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
   *count = ((Comm*)comm)->world;
   return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: a socket or a copy failed"; }

ncclResult_t ncclGroupStart() {
   g_depth++;
   return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
   if (g_depth > 0) g_depth--;
   return g_depth == 0 ? run(g_ops) : ncclSuccess;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
   return post((Comm*)comm, Op{true, const_cast<void*>(buf), count * type_size(t), peer, stream});
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
   return post((Comm*)comm, Op{false, buf, count * type_size(t), peer, stream});
}

ncclResult_t ncclAllGather(const void* sendbuf, void* recvbuf, size_t sendcount, ncclDataType_t t, ncclComm_t comm, hipStream_t stream) {
   Comm* c = (Comm*)comm;
   const size_t bytes = sendcount * type_size(t);
   g_depth++;
   for (int p = 0; p < c->world; p++) {
      (void)post(c, Op{true, const_cast<void*>(sendbuf), bytes, p, stream});
      (void)post(c, Op{false, (char*)recvbuf + (size_t)p * bytes, bytes, p, stream});
   }
   g_depth--;
   return g_depth == 0 ? run(g_ops) : ncclSuccess;
}

}  // extern "C"
