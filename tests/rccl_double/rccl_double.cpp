// TEST DOUBLE of librccl.so (tests only; never part of the product): the nccl* symbols holoagent_amd/csrc/hmsg_comm.hip
// resolves -- ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllGather, ncclAllReduce, ncclGetErrorString -- over
// POSIX shared memory between the processes of one machine, so that hmsg_allgather_nodes / hmsg_allreduce_feature_sums run with
// world > 1 on the kernel simulator (where "device" buffers are host memory and a stream's work is done when the call returns):
// the padding to the largest table, the per-rank un-padding, room_off shifting and a rank without nodes are library code that no
// GPU-less machine could execute otherwise.  Selected with HMSG_RCCL_LIB=<path of this .so>.
//   rendezvous: the 128-byte id names a shared-memory segment; every rank maps it, rank 0 initialises a process-shared barrier;
//   all-gather: every rank copies its piece into its slot, barrier, everybody copies all slots out, barrier;
//   all-reduce (sum): the same with the slots added in rank order 0 .. world-1 (a fixed order: the result is the same on every
//   rank and in every run; RCCL's ring order is its own).
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
const size_t DATA_BYTES = (size_t)256 << 20;          // payload area (all ranks' pieces of one collective)
const int MAX_RANKS = 16;
const size_t P2P_BYTES = (size_t)8 << 20;             // one sender's mailbox (behind the collectives' payload area); larger messages go in pieces
struct Mail {
    std::atomic<int> full;                            // 1: the sender's mailbox holds `bytes` for rank `dst`
    int dst;
    size_t bytes;
};
struct Shared {
    std::atomic<int> ready;                           // 1 once rank 0 has initialised the barrier
    pthread_barrier_t bar;
    Mail mail[MAX_RANKS];
    unsigned char data[1];
};
struct Comm {
    Shared* sh = nullptr;
    size_t map_bytes = 0;
    int rank = 0, world = 1;
    char name[160];
};
size_t elsize(int type) {                             // rccl.h type codes
    switch (type) {
        case 0: case 1: return 1;                     // int8 / uint8
        case 2: case 3: return 4;                     // int32 / uint32
        case 4: case 5: return 8;                     // int64 / uint64
        case 6: return 2;                             // half
        case 7: return 4;                             // float
        case 8: return 8;                             // double
        default: return 0;
    }
}
}  // namespace

extern "C" {

struct ncclUniqueId {
    char internal[128];
};

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, sizeof id->internal);
    static std::atomic<int> counter{0};
    snprintf(id->internal, sizeof id->internal, "/hmsg_rccl_double_%d_%d_%ld", (int)getpid(), counter++, (long)random());
    return 0;
}

int ncclCommInitRank(void** out, int world, ncclUniqueId id, int rank) {
    Comm* c = new Comm();
    c->rank = rank;
    c->world = world;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    if (world > MAX_RANKS) return 2;
    c->map_bytes = sizeof(Shared) + DATA_BYTES + (size_t)MAX_RANKS * P2P_BYTES;
    int fd = -1;
    for (int tries = 0; tries < 20000 && fd < 0; ++tries) {
        fd = shm_open(c->name, rank == 0 ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
        if (fd < 0) usleep(1000);
    }
    if (fd < 0) return 2;
    if (rank == 0 && ftruncate(fd, (off_t)c->map_bytes) != 0) return 2;
    if (rank != 0) {                                   // wait until rank 0 has sized the segment
        struct stat st;
        for (int tries = 0; tries < 20000; ++tries) {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
            usleep(1000);
        }
    }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 2;
    c->sh = (Shared*)p;
    if (rank == 0) {
        pthread_barrierattr_t a;
        pthread_barrierattr_init(&a);
        pthread_barrierattr_setpshared(&a, PTHREAD_PROCESS_SHARED);
        pthread_barrier_init(&c->sh->bar, &a, (unsigned)world);
        for (int r = 0; r < MAX_RANKS; ++r) c->sh->mail[r].full.store(0);
        c->sh->ready.store(1);
    } else {
        while (c->sh->ready.load() != 1) usleep(200);
    }
    pthread_barrier_wait(&c->sh->bar);
    *out = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    pthread_barrier_wait(&c->sh->bar);
    munmap(c->sh, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}

int ncclAllGather(const void* send, void* recv, size_t count, int type, void* comm, void* /*stream*/) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * elsize(type);
    if (!elsize(type) || bytes * (size_t)c->world > DATA_BYTES) return 3;
    memcpy(c->sh->data + (size_t)c->rank * bytes, send, bytes);
    pthread_barrier_wait(&c->sh->bar);
    memcpy(recv, c->sh->data, bytes * (size_t)c->world);
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int type, int op, void* comm, void* /*stream*/) {
    Comm* c = (Comm*)comm;
    const size_t bytes = count * elsize(type);
    if (op != 0 || (type != 7 && type != 3 && type != 2) || bytes * (size_t)c->world > DATA_BYTES) return 3;
    memcpy(c->sh->data + (size_t)c->rank * bytes, send, bytes);
    pthread_barrier_wait(&c->sh->bar);
    for (size_t i = 0; i < count; ++i) {
        if (type == 7) {
            float s = 0.f;
            for (int r = 0; r < c->world; ++r) s += ((const float*)(c->sh->data + (size_t)r * bytes))[i];
            ((float*)recv)[i] = s;
        } else {
            uint32_t s = 0;
            for (int r = 0; r < c->world; ++r) s += ((const uint32_t*)(c->sh->data + (size_t)r * bytes))[i];
            ((uint32_t*)recv)[i] = s;
        }
    }
    pthread_barrier_wait(&c->sh->bar);
    return 0;
}

// point to point: the sender's mailbox, a piece at a time (blocking: the call returns when the receiver has taken the last piece --
// a stream's work is done when the call returns on the simulator)
int ncclSend(const void* buf, size_t count, int type, int peer, void* comm, void* /*stream*/) {
    Comm* c = (Comm*)comm;
    size_t left = count * elsize(type);
    if (!elsize(type) || peer < 0 || peer >= c->world || peer == c->rank) return 3;
    Mail& m = c->sh->mail[c->rank];
    unsigned char* box = c->sh->data + DATA_BYTES + (size_t)c->rank * P2P_BYTES;
    const unsigned char* src = (const unsigned char*)buf;
    while (left) {
        const size_t n = left < P2P_BYTES ? left : P2P_BYTES;
        while (m.full.load(std::memory_order_acquire) != 0) usleep(50);
        memcpy(box, src, n);
        m.dst = peer;
        m.bytes = n;
        m.full.store(1, std::memory_order_release);
        src += n;
        left -= n;
    }
    while (m.full.load(std::memory_order_acquire) != 0) usleep(50);
    return 0;
}
int ncclRecv(void* buf, size_t count, int type, int peer, void* comm, void* /*stream*/) {
    Comm* c = (Comm*)comm;
    size_t left = count * elsize(type);
    if (!elsize(type) || peer < 0 || peer >= c->world || peer == c->rank) return 3;
    Mail& m = c->sh->mail[peer];
    const unsigned char* box = c->sh->data + DATA_BYTES + (size_t)peer * P2P_BYTES;
    unsigned char* dst = (unsigned char*)buf;
    while (left) {
        while (!(m.full.load(std::memory_order_acquire) == 1 && m.dst == c->rank)) usleep(50);
        const size_t n = m.bytes;
        if (n > left) return 3;
        memcpy(dst, box, n);
        m.full.store(0, std::memory_order_release);
        dst += n;
        left -= n;
    }
    return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "success" : (rc == 3 ? "rccl double: unsupported type / op / size" : "rccl double: rendezvous failed"); }

}  // extern "C"
