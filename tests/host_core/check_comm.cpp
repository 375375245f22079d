// check_comm.cpp - TEST ONLY.  The RCCL branch of necat_amd/csrc/comm.h (allgatherv_inplace, gatherv: the exchanges of a sharded index build
// and of the record gather) run on a machine WITHOUT a GPU and without a second rank: the SOURCE of comm.h compiled with g++, `nranks` ranks as
// threads of this process, ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd replaced by an in-process mailbox that moves the bytes and logs every
// call, the few HIP calls of that branch by host stand-ins ("device" memory = host memory).  What it checks at world 2, 3 and 8:
//   * the bytes arrive: every rank ends with the complete buffer (all-gather-v) / the root with every rank's records (gather-v);
//   * the call pattern: one group per exchange, in step d a rank sends its own part to rank + d and receives part rank - d from rank - d, byte
//     counts = the parts', nothing is posted for empty parts (a rank with nothing to give, a zero-length slice), root != 0 works;
//   * a failing call on ONE rank: the group is still closed, that rank returns the error, comm::agree gives every rank a non-zero verdict and
//     nobody blocks;
//   * comm::first_contact (the ring of small messages necat_comm_create sends over a new communicator): every rank receives its left neighbour's
//     pattern and returns 0; with one rank's send failing - or one message corrupted on the way - EVERY rank returns non-zero.
//
//   check_comm   (no arguments; exit code 0 = all good)
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "../../necat_amd/csrc/comm.h"

// ---------------------------------------------------------------- stand-ins for the HIP calls the RCCL branch makes (host memory is "device" memory)
extern "C" {
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { memcpy(dst, src, n); return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stand-in"; }
hipError_t hipMemGetAddressRange(hipDeviceptr_t*, size_t*, hipDeviceptr_t) { return hipErrorNotSupported; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorNotSupported; }
hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned int) { return hipErrorNotSupported; }
hipError_t hipIpcCloseMemHandle(void*) { return hipErrorNotSupported; }
}

// ---------------------------------------------------------------- the fake RCCL: ranks are threads, messages go through a mailbox
namespace fake {
struct Call { char kind; int peer; size_t bytes; };      // 'S'end / 'R'ecv / '(' group start / ')' group end
struct Rank { int id; std::vector<Call> log; std::vector<Call> pending; int group_depth = 0; };
std::mutex mu;
std::condition_variable cv;
std::map<std::pair<int, int>, std::vector<std::vector<char>>> box;      // (from, to) -> queue of messages
int fail_send_on_rank = -1;
int corrupt_from_rank = -1;       // messages of this rank arrive with one byte flipped
thread_local Rank* me = nullptr;
struct Buf { const void* src; void* dst; };
thread_local std::vector<Buf> bufs;

ncclResult_t GroupStart() { me->log.push_back({'(', -1, 0}); ++me->group_depth; return ncclSuccess; }
ncclResult_t Send(const void* p, size_t n, ncclDataType_t, int peer, ncclComm_t, hipStream_t)
{
    if (me->id == fail_send_on_rank) return ncclInternalError;
    if (me->group_depth != 1) return ncclInvalidUsage;
    me->log.push_back({'S', peer, n}); me->pending.push_back({'S', peer, n}); bufs.push_back({p, nullptr});
    return ncclSuccess;
}
ncclResult_t Recv(void* p, size_t n, ncclDataType_t, int peer, ncclComm_t, hipStream_t)
{
    if (me->group_depth != 1) return ncclInvalidUsage;
    me->log.push_back({'R', peer, n}); me->pending.push_back({'R', peer, n}); bufs.push_back({nullptr, p});
    return ncclSuccess;
}
ncclResult_t GroupEnd()
{
    me->log.push_back({')', -1, 0});
    if (--me->group_depth != 0) return ncclInvalidUsage;
    // sends first (never block), then receives: the all-pairs pattern cannot deadlock
    for (size_t i = 0; i < me->pending.size(); ++i) if (me->pending[i].kind == 'S') {
        std::lock_guard<std::mutex> lk(mu);
        box[{me->id, me->pending[i].peer}].emplace_back((const char*)bufs[i].src, (const char*)bufs[i].src + me->pending[i].bytes);
        if (me->id == corrupt_from_rank && me->pending[i].bytes) box[{me->id, me->pending[i].peer}].back()[me->pending[i].bytes / 2] ^= 0x40;
        cv.notify_all();
    }
    for (size_t i = 0; i < me->pending.size(); ++i) if (me->pending[i].kind == 'R') {
        std::unique_lock<std::mutex> lk(mu);
        auto& q = box[{me->pending[i].peer, me->id}];
        if (!cv.wait_for(lk, std::chrono::seconds(3), [&] { return !q.empty(); })) return ncclSystemError;     // a peer never sent: fail, do not hang
        if (q.front().size() != me->pending[i].bytes) return ncclInvalidArgument;
        memcpy(bufs[i].dst, q.front().data(), q.front().size());
        q.erase(q.begin());
    }
    me->pending.clear(); bufs.clear();
    return ncclSuccess;
}
const char* ErrStr(ncclResult_t) { return "fake rccl error"; }
}  // namespace fake

// host all-gather among the threads (the launcher's callback)
struct Gather {
    int n; std::mutex mu; std::condition_variable cv; std::vector<char> buf; int arrived = 0, left = 0; size_t bytes = 0; unsigned long gen = 0;
    int run(int rank, const void* send, void* recv, size_t b)
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return left == 0 || arrived > 0; });             // the previous exchange has been read by everybody
        if (arrived == 0) { buf.assign(b * n, 0); bytes = b; }
        memcpy(buf.data() + (size_t)rank * b, send, b);
        const unsigned long g = gen;
        if (++arrived == n) { left = n; arrived = 0; ++gen; cv.notify_all(); }
        else cv.wait(lk, [&] { return gen != g; });
        memcpy(recv, buf.data(), bytes * n);
        if (--left == 0) cv.notify_all();
        return 0;
    }
};
struct GUser { Gather* g; int rank; };
static int gather_cb(void* user, const void* send, void* recv, size_t bytes) { GUser* u = (GUser*)user; return u->g->run(u->rank, send, recv, bytes); }

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static void setup(necat_comm& c, int r, int G, GUser* u)
{
    c.rank = r; c.nranks = G; c.gather = gather_cb; c.user = u; c.transport = 0; c.nccl = (ncclComm_t)(uintptr_t)(r + 1);
    c.p_Send = fake::Send; c.p_Recv = fake::Recv; c.p_GroupStart = fake::GroupStart; c.p_GroupEnd = fake::GroupEnd; c.p_GetErrorString = fake::ErrStr;
}

static void run_world(int G, int root, int fail_rank)
{
    using namespace necat;
    // parts of the all-gather-v: rank g owns (g * 37 + 5) % 11 * 100 bytes - zero for some ranks - laid end to end
    std::vector<comm::Part> parts(G);
    size_t total = 0;
    for (int g = 0; g < G; ++g) { parts[g].off = total; parts[g].bytes = (size_t)(((g * 37 + 5) % 11) * 100) * (g % 4 == 3 ? 0 : 1); total += parts[g].bytes; }
    std::vector<size_t> counts(G);
    for (int g = 0; g < G; ++g) counts[g] = (size_t)((g * 13 + 2) % 7) * 96;           // record bytes per rank, some zero
    Gather gather; gather.n = G;
    std::vector<fake::Rank> ranks(G);
    std::vector<std::vector<char>> bufs(G), recvs(G);
    std::vector<int> rc_ag(G, -99), rc_gv(G, -99), verdict(G, -99);
    fake::box.clear();
    fake::fail_send_on_rank = fail_rank;
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r) th.emplace_back([&, r]() {
        fake::me = &ranks[r]; ranks[r].id = r;
        necat_ctx ctx;
        GUser u{&gather, r};
        necat_comm c; setup(c, r, G, &u);
        bufs[r].assign(total + 16, (char)0xEE);
        for (size_t i = 0; i < parts[r].bytes; ++i) bufs[r][parts[r].off + i] = (char)(r * 31 + i);
        rc_ag[r] = comm::allgatherv_inplace(&ctx, &c, bufs[r].data(), parts, nullptr);
        verdict[r] = comm::agree(&ctx, &c, rc_ag[r]);
        if (fail_rank >= 0) return;                     // the failure scenario ends here: everybody has a verdict, nobody hangs
        std::vector<char> mine(counts[r] + 1);
        for (size_t i = 0; i < counts[r]; ++i) mine[i] = (char)(r * 7 + i * 3);
        size_t all = 0; for (size_t x : counts) all += x;
        recvs[r].assign(all + 16, (char)0xDD);
        rc_gv[r] = comm::gatherv(&ctx, &c, mine.data(), counts, root, recvs[r].data(), nullptr);
    });
    for (auto& t : th) t.join();
    if (fail_rank >= 0) {
        for (int r = 0; r < G; ++r) {
            EXPECT((rc_ag[r] != 0) == (r == fail_rank) || rc_ag[r] != 0, "world %d: rank %d returned %d", G, r, rc_ag[r]);
            EXPECT(verdict[r] != 0, "world %d: rank %d left the agreement with verdict 0 although rank %d failed", G, r, fail_rank);
            int opens = 0, closes = 0;
            for (auto& cl : ranks[r].log) { opens += cl.kind == '('; closes += cl.kind == ')'; }
            EXPECT(opens == closes, "world %d: rank %d left a group open (%d starts, %d ends)", G, r, opens, closes);
        }
        EXPECT(rc_ag[fail_rank] != 0, "the failing rank reported success");
        return;
    }
    for (int r = 0; r < G; ++r) {
        EXPECT(rc_ag[r] == 0 && verdict[r] == 0 && rc_gv[r] == 0, "world %d rank %d: rc %d / %d / %d", G, r, rc_ag[r], verdict[r], rc_gv[r]);
        for (int g = 0; g < G; ++g) for (size_t i = 0; i < parts[g].bytes; ++i)
            if (bufs[r][parts[g].off + i] != (char)(g * 31 + i)) { EXPECT(false, "world %d: rank %d holds a wrong byte of part %d", G, r, g); break; }
        EXPECT((unsigned char)bufs[r][total] == 0xEE, "world %d: rank %d wrote past the buffer", G, r);
        // the call pattern of the all-gather-v: ( S/R per step d = 1 .. G - 1, empty parts skipped )
        size_t k = 0;
        const auto& L = ranks[r].log;
        if (G == 1) { EXPECT(L.empty(), "world 1: a lone rank posted %zu RCCL calls", L.size()); continue; }
        EXPECT(L.size() > k && L[k].kind == '(', "world %d rank %d: no group start", G, r); ++k;
        for (int d = 1; d < G; ++d) {
            const int to = (r + d) % G, from = (r - d + G) % G;
            if (parts[r].bytes) { EXPECT(k < L.size() && L[k].kind == 'S' && L[k].peer == to && L[k].bytes == parts[r].bytes, "world %d rank %d step %d: send", G, r, d); ++k; }
            if (parts[from].bytes) { EXPECT(k < L.size() && L[k].kind == 'R' && L[k].peer == from && L[k].bytes == parts[from].bytes, "world %d rank %d step %d: recv", G, r, d); ++k; }
        }
        EXPECT(k < L.size() && L[k].kind == ')', "world %d rank %d: the group is not closed after the last step", G, r); ++k;
        // gather-v: the root receives from every rank with records, the others send theirs (or nothing)
        EXPECT(k < L.size() && L[k].kind == '(', "world %d rank %d: gather-v group", G, r); ++k;
        if (r == root) { for (int g = 0; g < G; ++g) if (g != root && counts[g]) { EXPECT(k < L.size() && L[k].kind == 'R' && L[k].peer == g && L[k].bytes == counts[g], "world %d root: recv from %d", G, g); ++k; } }
        else if (counts[r]) { EXPECT(k < L.size() && L[k].kind == 'S' && L[k].peer == root && L[k].bytes == counts[r], "world %d rank %d: send to the root", G, r); ++k; }
        EXPECT(k < L.size() && L[k].kind == ')' && k + 1 == L.size(), "world %d rank %d: %zu calls logged, %zu expected", G, r, L.size(), k + 1);
    }
    size_t at = 0;
    for (int g = 0; g < G; ++g) {
        for (size_t i = 0; i < counts[g]; ++i) if (recvs[root][at + i] != (char)(g * 7 + i * 3)) { EXPECT(false, "world %d: the root holds a wrong byte of rank %d's records", G, g); break; }
        at += counts[g];
    }
    EXPECT((unsigned char)recvs[root][at] == 0xDD, "world %d: the root wrote past its records", G);
}

static void run_contact(int G, int fail_rank, int corrupt_rank)
{
    using namespace necat;
    Gather gather; gather.n = G;
    std::vector<fake::Rank> ranks(G);
    std::vector<int> rc(G, -99);
    fake::box.clear();
    fake::fail_send_on_rank = fail_rank; fake::corrupt_from_rank = corrupt_rank;
    std::vector<std::thread> th;
    for (int r = 0; r < G; ++r) th.emplace_back([&, r]() {
        fake::me = &ranks[r]; ranks[r].id = r;
        necat_ctx ctx;
        GUser u{&gather, r};
        necat_comm c; setup(c, r, G, &u);
        std::vector<unsigned char> dbuf(2 * 4096);
        rc[r] = comm::first_contact(&ctx, &c, dbuf.data(), 4096, nullptr);
    });
    for (auto& t : th) t.join();
    fake::fail_send_on_rank = -1; fake::corrupt_from_rank = -1;
    for (int r = 0; r < G; ++r) {
        if (fail_rank < 0 && corrupt_rank < 0) EXPECT(rc[r] == 0, "first contact, world %d: rank %d returned %d", G, r, rc[r]);
        else EXPECT(rc[r] != 0, "first contact, world %d: rank %d returned 0 although rank %d %s", G, r, fail_rank >= 0 ? fail_rank : corrupt_rank, fail_rank >= 0 ? "could not send" : "sent a corrupted message");
        int opens = 0, closes = 0;
        for (auto& cl : ranks[r].log) { opens += cl.kind == '('; closes += cl.kind == ')'; }
        EXPECT(opens == closes && (G == 1 || opens == 1), "first contact, world %d: rank %d: %d group starts, %d ends", G, r, opens, closes);
    }
}

int main()
{
    for (int G : {1, 2, 3, 8}) run_contact(G, -1, -1);
    for (int G : {2, 3, 8}) { run_contact(G, G / 2, -1); run_contact(G, -1, G - 1); }
    for (int G : {1, 2, 3, 8}) for (int root : {0, G - 1}) run_world(G, root, -1);
    for (int G : {2, 3, 8}) run_world(G, 0, G / 2);            // one rank's ncclSend fails
    if (failures) { fprintf(stderr, "%d check(s) failed\n", failures); return 1; }
    printf("check_comm: all-gather-v and gather-v through the RCCL branch at world 1, 2, 3, 8 (roots 0 and last), and a failing rank; first contact (clean, a failing send, a corrupted message): ok\n");
    return 0;
}
