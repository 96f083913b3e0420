// Kernels whose workgroups WAIT for each other inside a launch (the register-resident fp64 Sinkhorn: the row slabs of a pair; the
// clustered fp64 layer tail: the four workgroups of a 16-row block) are admitted ONE LAUNCH AT A TIME per device, whatever stream
// they are on: a waiting workgroup holds its CU, and partly resident groups of several launches can fill the chip with waiters
// whose partners find no slot (measured with the Sinkhorn: four streams, every spin ran into its bound - sinkhorn_f64.hip).  A
// launch records an event behind itself and the next one waits for it on its own stream - unless that is the same stream
// (consecutive launches on ONE stream are ordered already).  The record alone costs ~3 us of gap per launch in the kernel trace
// (one pair per call: 18 clustered layer launches per forward), so a forward enqueues its launches as a GROUP (CoopGroup: the
// device's chain stays locked - only the ENQUEUE of exact-mode forwards is serialised per device, a millisecond of host time - and ONE
// event is recorded behind the group's last waiting launch).  Streams under graph capture are left alone (an event wait on foreign
// work cannot be captured): the Sinkhorn launches unchained there, the clustered layer tail is not used at all (its flags count
// launches: a replay would meet them already set).
// The chain is per PROCESS (one process per GPU is the model): two processes that run such launches on one device are
// not ordered against each other - a starved launch then runs into its spin bound (seconds) and the call is refused through the range
// status, never answered wrongly; MDGAT_F64_LAYER_FUSION=2 keeps the layer tails out of it.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

struct CoopChain {
    std::recursive_mutex m;
    hipEvent_t ev = nullptr;
    hipStream_t last_stream = nullptr;         // (only compared, never used: it may be gone)
    bool has_last = false;
    int group_depth = 0;                       // > 0: inside a CoopGroup (the mutex is held by its thread)
    bool group_dirty = false;                  // a waiting launch of the group has no event behind it yet
    unsigned long long epoch = 0;              // launches of the clustered layer tail so far (its flags carry it)
    unsigned long long* cluster_flags = nullptr;
};
inline CoopChain& coop_chain_of(int dev) {
    static CoopChain chains[16];
    return chains[dev >= 0 && dev < 16 ? dev : 0];
}
inline bool coop_stream_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &st);
    return st != hipStreamCaptureStatusNone;
}
inline hipError_t coop_chain_event(CoopChain& c, hipStream_t s) {
    if (!c.ev && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) c.ev = nullptr;
    return c.ev ? hipEventRecord(c.ev, s) : hipSuccess;
}
// (the caller holds c.m from before coop_chain_wait until after coop_chain_record, the launch in between)
inline hipError_t coop_chain_wait(CoopChain& c, hipStream_t s) {
    if (!c.has_last || c.last_stream == s || !c.ev) return hipSuccess;
    return hipStreamWaitEvent(s, c.ev, 0);      // (a group that changed streams recorded behind its launches on the old one: CoopGroup::flush)
}
inline hipError_t coop_chain_record(CoopChain& c, hipStream_t s) {
    c.last_stream = s;
    c.has_last = true;
    if (c.group_depth > 0) { c.group_dirty = true; return hipSuccess; }
    return coop_chain_event(c, s);
}
// A forward's launches on ONE stream as a group: one event behind the last waiting launch instead of one behind each.
struct CoopGroup {
    CoopChain* c = nullptr;
    hipStream_t s = nullptr;
    CoopGroup(int dev, hipStream_t stream, bool on) : s(stream) {
        if (!on) return;
        c = &coop_chain_of(dev);
        c->m.lock();
        ++c->group_depth;
    }
    ~CoopGroup() {
        if (!c) return;
        if (--c->group_depth == 0 && c->group_dirty) {
            c->group_dirty = false;
            if (!coop_stream_capturing(s)) (void)coop_chain_event(*c, s);
        }
        c->m.unlock();
    }
    CoopGroup(const CoopGroup&) = delete;
    CoopGroup& operator=(const CoopGroup&) = delete;
};
