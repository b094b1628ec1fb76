// vtx_comm_test.hip — TEST TRANSPORT for vtx_comm_init / vtx_gather_coo / vtx_gather_abort (never used in production).
//
// The multi-GPU row gather (include/vtx.h; the concatenation of the chunks' results in the reference's merge loop,
// src/main.rs:320-348, when the loci are sharded over processes, :284-291) talks to RCCL through a table of nine entry points
// (vtx_api.hip: Rccl).  This project's GPU boxes have ONE device, and RCCL refuses two ranks on one device — so the exchange's
// own logic (the status rounds, the plan, a Recv from a second rank landing at its final offset, a failing rank, an empty rank)
// had never executed with world > 1.  With VTX_COMM_TEST_TRANSPORT=<directory> in the environment the table is filled with the
// functions below instead of librccl's: same signatures, ranks = PROCESSES THAT MAY SHARE ONE DEVICE, payloads staged through
// host memory and moved over Unix-domain sockets in <directory>.  Everything above the table — vtx_gather_coo as shipped — runs
// unchanged; what is NOT exercised is RCCL itself and xGMI.  tests/test_gpu_shard.py drives it with world 2 and 4.
//
// Semantics kept: calls are issued in program order on the caller's stream (the transport synchronises the stream before it
// reads a send buffer and writes receive buffers with blocking copies); AllGather = everybody to rank 0, rank 0 to everybody;
// Send / Recv are matched by (source rank, order); GroupStart / GroupEnd are no-ops (every operation completes eagerly — the
// gather's senders only send and its destination only receives, so eager completion cannot deadlock); every blocking step has
// a 60 s timeout, so a peer that died turns into an error, not a hang.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
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

struct TestComm {
    int rank = 0, world = 1;
    std::string base;                    // <dir>/<id hex>
    int listen_fd = -1;
    std::vector<int> out_fd, in_fd;      // per peer
};

const int kTimeoutMs = 60000;

std::string sock_path(const TestComm* c, int r) { return c->base + "." + std::to_string(r); }

bool write_all(int fd, const void* p, size_t n) {
    const char* b = (const char*)p;
    while (n) {
        pollfd pf{fd, POLLOUT, 0};
        if (poll(&pf, 1, kTimeoutMs) <= 0) return false;
        const ssize_t w = send(fd, b, n, MSG_NOSIGNAL);
        if (w < 0) { if (errno == EINTR || errno == EAGAIN) continue; return false; }
        b += w; n -= (size_t)w;
    }
    return true;
}
bool read_all(int fd, void* p, size_t n) {
    char* b = (char*)p;
    while (n) {
        pollfd pf{fd, POLLIN, 0};
        if (poll(&pf, 1, kTimeoutMs) <= 0) return false;
        const ssize_t r = recv(fd, b, n, 0);
        if (r == 0) return false;                                  // the peer left
        if (r < 0) { if (errno == EINTR || errno == EAGAIN) continue; return false; }
        b += r; n -= (size_t)r;
    }
    return true;
}

int out_to(TestComm* c, int peer) {
    if (c->out_fd[(size_t)peer] >= 0) return c->out_fd[(size_t)peer];
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const int fd = socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) return -1;
        sockaddr_un a{};
        a.sun_family = AF_UNIX;
        snprintf(a.sun_path, sizeof a.sun_path, "%s", sock_path(c, peer).c_str());
        if (connect(fd, (sockaddr*)&a, sizeof a) == 0) {
            const int32_t me = c->rank;
            if (!write_all(fd, &me, sizeof me)) { close(fd); return -1; }
            return c->out_fd[(size_t)peer] = fd;
        }
        close(fd);
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(kTimeoutMs)) return -1;   // the peer never listened
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
}
int in_from(TestComm* c, int peer) {
    while (c->in_fd[(size_t)peer] < 0) {
        pollfd pf{c->listen_fd, POLLIN, 0};
        if (poll(&pf, 1, kTimeoutMs) <= 0) return -1;
        const int fd = accept(c->listen_fd, nullptr, nullptr);
        if (fd < 0) { if (errno == EINTR) continue; return -1; }
        int32_t who = -1;
        if (!read_all(fd, &who, sizeof who) || who < 0 || who >= c->world || c->in_fd[(size_t)who] >= 0) { close(fd); return -1; }
        c->in_fd[(size_t)who] = fd;
    }
    return c->in_fd[(size_t)peer];
}

size_t dtype_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclUint64: case ncclInt64: case ncclFloat64: return 8;
        case ncclUint32: case ncclInt32: case ncclFloat32: return 4;
        case ncclUint8: case ncclInt8: return 1;
        default: return 0;
    }
}

ncclResult_t t_get_unique_id(ncclUniqueId* id) {
    memset(id, 0, sizeof *id);
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, id->internal, 16) != 16) { if (fd >= 0) close(fd); return ncclSystemError; }
    close(fd);
    return ncclSuccess;
}

ncclResult_t t_comm_init_rank(ncclComm_t* comm, int world, ncclUniqueId id, int rank) {
    const char* dir = getenv("VTX_COMM_TEST_TRANSPORT");
    if (!dir || world < 1 || rank < 0 || rank >= world) return ncclInvalidArgument;
    TestComm* c = new TestComm;
    c->rank = rank; c->world = world;
    char hex[33];
    for (int i = 0; i < 16; ++i) snprintf(hex + 2 * i, 3, "%02x", (unsigned)(uint8_t)id.internal[i]);
    c->base = std::string(dir) + "/vtxcomm-" + hex;
    c->out_fd.assign((size_t)world, -1); c->in_fd.assign((size_t)world, -1);
    c->listen_fd = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a{};
    a.sun_family = AF_UNIX;
    snprintf(a.sun_path, sizeof a.sun_path, "%s", sock_path(c, rank).c_str());
    unlink(a.sun_path);
    if (c->listen_fd < 0 || bind(c->listen_fd, (sockaddr*)&a, sizeof a) != 0 || listen(c->listen_fd, world + 4) != 0) {
        if (c->listen_fd >= 0) close(c->listen_fd);
        delete c;
        return ncclSystemError;
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t t_comm_destroy(ncclComm_t comm) {
    TestComm* c = (TestComm*)comm;
    if (!c) return ncclSuccess;
    for (int fd : c->out_fd) if (fd >= 0) close(fd);
    for (int fd : c->in_fd) if (fd >= 0) close(fd);
    if (c->listen_fd >= 0) close(c->listen_fd);
    unlink(sock_path(c, c->rank).c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t t_send(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    TestComm* c = (TestComm*)comm;
    const size_t bytes = count * dtype_bytes(t);
    std::vector<char> h(bytes);
    if (hipStreamSynchronize(s) != hipSuccess || (bytes && hipMemcpy(h.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess)) return ncclUnhandledCudaError;
    const int fd = out_to(c, peer);
    const uint64_t n = bytes;
    if (fd < 0 || !write_all(fd, &n, sizeof n) || !write_all(fd, h.data(), bytes)) return ncclSystemError;
    return ncclSuccess;
}

ncclResult_t t_recv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
    TestComm* c = (TestComm*)comm;
    const size_t bytes = count * dtype_bytes(t);
    const int fd = in_from(c, peer);
    uint64_t n = 0;
    if (fd < 0 || !read_all(fd, &n, sizeof n) || n != bytes) return ncclSystemError;      // sizes are part of the protocol
    std::vector<char> h(bytes);
    if (!read_all(fd, h.data(), bytes)) return ncclSystemError;
    if (hipStreamSynchronize(s) != hipSuccess || (bytes && hipMemcpy(buf, h.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t t_all_gather(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
    TestComm* c = (TestComm*)comm;
    const size_t chunk = count * dtype_bytes(t);
    std::vector<char> all(chunk * (size_t)c->world);
    if (hipStreamSynchronize(s) != hipSuccess ||
        hipMemcpy(all.data() + chunk * (size_t)c->rank, sendbuf, chunk, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (c->rank == 0) {
        for (int r = 1; r < c->world; ++r) {
            const int fd = in_from(c, r);
            if (fd < 0 || !read_all(fd, all.data() + chunk * (size_t)r, chunk)) return ncclSystemError;
        }
        for (int r = 1; r < c->world; ++r) {
            const int fd = out_to(c, r);
            if (fd < 0 || !write_all(fd, all.data(), all.size())) return ncclSystemError;
        }
    } else {
        const int to0 = out_to(c, 0);
        if (to0 < 0 || !write_all(to0, all.data() + chunk * (size_t)c->rank, chunk)) return ncclSystemError;
        const int from0 = in_from(c, 0);
        if (from0 < 0 || !read_all(from0, all.data(), all.size())) return ncclSystemError;
    }
    if (hipMemcpy(recvbuf, all.data(), all.size(), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t t_group() { return ncclSuccess; }
const char* t_error_string(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "ok";
        case ncclSystemError: return "test transport: a peer did not answer within 60 s, or left";
        case ncclUnhandledCudaError: return "test transport: HIP copy failed";
        case ncclInvalidArgument: return "test transport: invalid argument (VTX_COMM_TEST_TRANSPORT must name a directory)";
        default: return "test transport error";
    }
}

}  // namespace

ncclResult_t t_comm_count(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((const TestComm*)comm)->world;
    return ncclSuccess;
}

// fills the ten entry points (vtx_api.hip: Rccl) with the test transport; returns a non-null cookie for Rccl::lib
extern "C" void* vtxt_comm_test_table(void** fns) {
    fns[0] = (void*)t_get_unique_id; fns[1] = (void*)t_comm_init_rank; fns[2] = (void*)t_comm_destroy; fns[3] = (void*)t_all_gather;
    fns[4] = (void*)t_send; fns[5] = (void*)t_recv; fns[6] = (void*)t_group; fns[7] = (void*)t_group; fns[8] = (void*)t_error_string;
    fns[9] = (void*)t_comm_count;
    static int cookie;
    return &cookie;
}
