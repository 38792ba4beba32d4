// rccl_stub.cpp — TEST INFRASTRUCTURE, never part of the product.
//
// A stand-in for the dozen RCCL entry points libwindtunnel.so calls (ncclGetUniqueId, ncclCommInitRank, ncclCommCount, ncclCommDestroy,
// ncclGroupStart / ncclGroupEnd, ncclSend, ncclRecv, ncclAllReduce, ncclGetErrorString), so that the library's MULTI-RANK code — the slab
// state machine over the TR_RCCL transport, the grouped ghost-column exchange, the schedule-agreement all-reduce of wt_comm_init_rank, the
// macro ghost exchange of the vorticity field, bench.py --gpus N — runs as N real processes on the ONE GPU of a test box, where RCCL itself
// refuses to ("Duplicate GPU detected").  Loaded with LD_PRELOAD in front of the production libwindtunnel.so: the library under test is the
// shipped binary, only the transport underneath it is replaced.
//
// What it is: POSIX shared memory between the ranks of one host, messages staged through the host (hipMemcpy device -> shm -> device),
// everything synchronous (ncclGroupEnd returns when the group's sends are posted and its receives have landed).  What it keeps of NCCL's
// semantics: sends and receives between a pair of ranks match IN ORDER; a group is deadlock-free whatever the order of its calls (all sends
// of a group are posted before any receive is waited for; rings of two slots per ordered pair); a receive whose byte count differs from what
// the peer sent FAILS (ncclInvalidArgument) — the mismatch of slab widths or halo depths RCCL would turn into a hang; every wait has a
// timeout (ncclSystemError).  What it does not keep: performance, overlap with compute, anything about xGMI.
// Built WITHOUT the HIP and RCCL headers and linked against neither library: the few types and enumerators it needs are restated below (values as in
// /opt/rocm/include/rccl/rccl.h and hip/driver_types.h), and hipMemcpy / hipStreamSynchronize are looked up in the process at first use — so the stub
// works on whatever HIP runtime the process has already loaded (PyTorch ships its own), never a second one.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef struct ihipStream_t *hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
typedef struct { char internal[128]; } ncclUniqueId;
struct ncclComm;
typedef struct ncclComm *ncclComm_t;

namespace {
typedef hipError_t (*memcpy_fn)(void *, const void *, size_t, int);
typedef hipError_t (*sync_fn)(hipStream_t);
memcpy_fn p_memcpy = nullptr;
sync_fn p_sync = nullptr;
bool bind_hip()
{
    if (!p_memcpy) p_memcpy = reinterpret_cast<memcpy_fn>(dlsym(RTLD_DEFAULT, "hipMemcpy"));
    if (!p_sync) p_sync = reinterpret_cast<sync_fn>(dlsym(RTLD_DEFAULT, "hipStreamSynchronize"));
    return p_memcpy && p_sync;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, int kind) { return p_memcpy(d, s, n, kind); }
hipError_t hipStreamSynchronize(hipStream_t st) { return p_sync(st); }

constexpr int MAX_RANKS = 8;
constexpr int RING = 2;
constexpr size_t SLOT_BYTES = 12u << 20;        // one group's messages between one ordered pair of ranks
constexpr size_t AR_BYTES = 4096;
constexpr double TIMEOUT_S = 60.0;

struct Slot { size_t bytes; size_t nmsg; size_t msg_bytes[64]; };
struct Ring { std::atomic<unsigned long long> head, tail; Slot slot[RING]; };
struct Shared {
    std::atomic<int> arrived, departed;
    std::atomic<unsigned long long> ar_posted[MAX_RANKS], ar_done[MAX_RANKS];
    char ar_data[MAX_RANKS][AR_BYTES];
    Ring ring[MAX_RANKS][MAX_RANKS];             // [src][dst]
    // slot payloads follow: [src][dst][RING][SLOT_BYTES]
};
inline char *payload(Shared *s, int src, int dst, int k)
{
    return reinterpret_cast<char *>(s) + ((sizeof(Shared) + 4095) / 4096) * 4096 + ((((size_t)src * MAX_RANKS + dst) * RING + k) * SLOT_BYTES);
}
constexpr size_t SHM_BYTES = ((sizeof(Shared) + 4095) / 4096) * 4096 + (size_t)MAX_RANKS * MAX_RANKS * RING * SLOT_BYTES;

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <typename F> bool wait_for(F cond)
{
    const double t0 = now();
    while (!cond()) {
        if (now() - t0 > TIMEOUT_S) return false;
        usleep(20);
    }
    return true;
}
size_t dtype_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
struct Op { bool send; void *buf; size_t bytes; int peer; struct ncclComm *comm; hipStream_t stream; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
}  // namespace

struct ncclComm {
    int rank, nranks;
    Shared *sh;
    char name[136];
    unsigned long long ar_seq;
    std::vector<char> host;
};

extern "C" {

ncclResult_t ncclGetVersion(int *v) { if (v) *v = 22606; return ncclSuccess; }

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "rccl_stub: timeout or system error (a peer is missing or the ranks took different schedules)";
    case ncclInvalidArgument: return "rccl_stub: invalid argument (message sizes of a send / receive pair differ?)";
    case ncclInvalidUsage: return "rccl_stub: invalid usage";
    default: return "rccl_stub: error";
    }
}
const char *ncclGetLastError(ncclComm_t) { return "rccl_stub"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/wt_rccl_stub_%d_%lld", (int)getpid(), (long long)(now() * 1e6));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    if (!bind_hip()) { fprintf(stderr, "[rccl_stub] no HIP runtime in this process\n"); return ncclSystemError; }
    ncclComm *c = new ncclComm();
    c->rank = rank; c->nranks = nranks; c->ar_seq = 0;
    snprintf(c->name, sizeof(c->name), "%s", id.internal);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return ncclSystemError; }
    if (ftruncate(fd, (off_t)SHM_BYTES) != 0) { close(fd); delete c; return ncclSystemError; }      // sparse: pages exist once touched; zero-filled
    void *p = mmap(nullptr, SHM_BYTES, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared *>(p);
    c->sh->arrived.fetch_add(1);
    if (!wait_for([&] { return c->sh->arrived.load() >= nranks; })) { munmap(p, SHM_BYTES); delete c; return ncclSystemError; }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int *n) { if (!c || !n) return ncclInvalidArgument; *n = c->nranks; return ncclSuccess; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int *r) { if (!c || !r) return ncclInvalidArgument; *r = c->rank; return ncclSuccess; }

ncclResult_t ncclCommDestroy(ncclComm_t c)
{
    if (!c) return ncclSuccess;
    const int left = c->sh->departed.fetch_add(1) + 1;
    if (left >= c->nranks) shm_unlink(c->name);       // the last rank out removes the segment's name
    munmap(c->sh, SHM_BYTES);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }

static ncclResult_t run_group()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    if (ops.empty()) return ncclSuccess;
    ncclComm *c = ops[0].comm;
    for (const Op &o : ops) {
        if (o.comm != c || o.peer < 0 || o.peer >= c->nranks) return ncclInvalidArgument;
        if (hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;      // everything the stream was told before the exchange
    }
    Shared *s = c->sh;
    // 1. post, per destination, all the sends of this group as one blob
    for (int p = 0; p < c->nranks; p++) {
        size_t total = 0, n = 0;
        for (const Op &o : ops) if (o.send && o.peer == p) { total += o.bytes; n++; }
        if (n == 0) continue;
        if (total > SLOT_BYTES || n > 64) return ncclInvalidArgument;
        Ring &r = s->ring[c->rank][p];
        if (!wait_for([&] { return r.head.load() - r.tail.load() < RING; })) return ncclSystemError;
        const int k = (int)(r.head.load() % RING);
        char *dst = payload(s, c->rank, p, k);
        size_t off = 0, i = 0;
        for (const Op &o : ops)
            if (o.send && o.peer == p) {
                if (hipMemcpy(dst + off, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
                r.slot[k].msg_bytes[i++] = o.bytes;
                off += o.bytes;
            }
        r.slot[k].bytes = total; r.slot[k].nmsg = n;
        r.head.fetch_add(1);
    }
    // 2. take, per source, the blob that answers this group's receives — message by message, sizes must agree
    for (int p = 0; p < c->nranks; p++) {
        size_t n = 0;
        for (const Op &o : ops) if (!o.send && o.peer == p) n++;
        if (n == 0) continue;
        Ring &r = s->ring[p][c->rank];
        if (!wait_for([&] { return r.head.load() > r.tail.load(); })) return ncclSystemError;
        const int k = (int)(r.tail.load() % RING);
        const char *src = payload(s, p, c->rank, k);
        if (r.slot[k].nmsg != n) { fprintf(stderr, "[rccl_stub] rank %d expects %zu messages from rank %d, which sent %zu\n", c->rank, n, p, r.slot[k].nmsg); return ncclInvalidArgument; }
        size_t off = 0, i = 0;
        for (const Op &o : ops)
            if (!o.send && o.peer == p) {
                if (r.slot[k].msg_bytes[i] != o.bytes) {
                    fprintf(stderr, "[rccl_stub] rank %d: message %zu from rank %d has %zu bytes, the receive expects %zu\n", c->rank, i, p, r.slot[k].msg_bytes[i], o.bytes);
                    return ncclInvalidArgument;
                }
                if (hipMemcpy(o.buf, src + off, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
                off += o.bytes; i++;
            }
        r.tail.fetch_add(1);
    }
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    return run_group();
}

static ncclResult_t enqueue(bool send, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st)
{
    const size_t es = dtype_size(t);
    if (!c || !buf || es == 0) return ncclInvalidArgument;
    g_ops.push_back(Op{send, buf, count * es, peer, c, st});
    if (g_depth == 0) return run_group();
    return ncclSuccess;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) { return enqueue(true, const_cast<void *>(buf), count, t, peer, c, st); }
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) { return enqueue(false, buf, count, t, peer, c, st); }

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st)
{
    if (!c || !send || !recv) return ncclInvalidArgument;
    if (t != ncclInt64 && t != ncclFloat64) return ncclInvalidArgument;
    const size_t bytes = count * 8;
    if (bytes > AR_BYTES) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    Shared *s = c->sh;
    const unsigned long long seq = ++c->ar_seq;
    // nobody may overwrite its slot before every rank has finished reading the previous round
    if (!wait_for([&] { for (int r = 0; r < c->nranks; r++) if (s->ar_done[r].load() < seq - 1) return false; return true; })) return ncclSystemError;
    if (hipMemcpy(s->ar_data[c->rank], send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    s->ar_posted[c->rank].store(seq);
    if (!wait_for([&] { for (int r = 0; r < c->nranks; r++) if (s->ar_posted[r].load() < seq) return false; return true; })) return ncclSystemError;
    std::vector<char> out(bytes);
    for (size_t i = 0; i < count; i++) {
        if (t == ncclInt64) {
            long long acc = reinterpret_cast<long long *>(s->ar_data[0])[i];
            for (int r = 1; r < c->nranks; r++) {
                const long long v = reinterpret_cast<long long *>(s->ar_data[r])[i];
                acc = op == ncclSum ? acc + v : (op == ncclMax ? (v > acc ? v : acc) : (op == ncclMin ? (v < acc ? v : acc) : acc));
            }
            reinterpret_cast<long long *>(out.data())[i] = acc;
        } else {
            double acc = reinterpret_cast<double *>(s->ar_data[0])[i];
            for (int r = 1; r < c->nranks; r++) {
                const double v = reinterpret_cast<double *>(s->ar_data[r])[i];
                acc = op == ncclSum ? acc + v : (op == ncclMax ? (v > acc ? v : acc) : (op == ncclMin ? (v < acc ? v : acc) : acc));
            }
            reinterpret_cast<double *>(out.data())[i] = acc;
        }
    }
    s->ar_done[c->rank].store(seq);
    if (hipMemcpy(recv, out.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
