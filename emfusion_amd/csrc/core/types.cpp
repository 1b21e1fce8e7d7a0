// types.cpp -- RAII wrappers over HIP streams / device memory used by the host classes.
#include "types.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace emf {

void hipCheck(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw HipError(std::string(what) + ": " + hipGetErrorString(e), static_cast<int>(e));
}

void emfCheck(int rc, const char* what) {
    if (rc != EMF_OK)
        throw HipError(std::string(what) + ": " + emf_hip_last_error_string(), rc);
}

// ---- live streams + device memory pool ------------------------------------------------------------
// hipFree synchronises the whole device and hipMalloc of tens of MB takes a fraction of a millisecond;
// objects are created, resized and deleted INSIDE frames (reference EMFusion.cpp:495-560, 827-863,
// 922-980).  So a released DeviceBuffer is not freed: it goes to a pool together with one event per live
// stream of its HOME thread -- the host thread that allocated it, which is the thread that drives its
// emf::EMFusion instance and created every emf::Stream that can touch it -- recorded at the streams' tails
// no earlier than the release, and is handed out again (same size, same home) once all of them have
// completed (hipEventQuery: no wait).
// Registry and pool are process-wide under one mutex; what stays per thread is WHO fences WHAT:
//   * events are only ever recorded by the thread that created the stream (several instances on several
//     threads -- the multi-rank rehearsal -- must not record on each other's streams);
//   * a buffer released on a FOREIGN thread (a handle closed or finalised elsewhere) is parked unfenced and
//     fenced by its home thread at that thread's next pool operation -- later than the release, so still
//     behind everything that was enqueued on the buffer; if the home thread has ended it is hipFree'd;
//   * a stream destroyed on a foreign thread is drained first and its fences are retired wherever they
//     are (an event must not outlive the stream it was recorded on: hipEventQuery then throws from inside
//     the runtime), and the owner's parked buffers are fenced by that drain as well.
// A thread's buffers are really freed when the thread ends, by trimPool(), or when hipMalloc fails.
namespace {
struct Fence {
    hipEvent_t event;
    hipStream_t stream;  // where it was recorded
};
struct Pooled {
    void* p;
    size_t bytes;
    uint64_t home;
    bool parked;  // released on a foreign thread: not fenced yet
    std::vector<Fence> fences;
};
struct LiveStream {
    hipStream_t s;
    uint64_t owner;
};
struct Registry {
    std::mutex m;
    std::vector<LiveStream> streams;  // every live emf::Stream of the process
    std::vector<Pooled> pool;
    std::vector<uint64_t> threads;    // live threads that have used the pool
    size_t pooledBytes = 0;
    uint64_t nextId = 1;
};
Registry& reg() {
    static Registry* r = new Registry;  // never destroyed: thread_local destructors may run after statics
    return *r;
}
void free_entries_locked(Registry& r, std::vector<Pooled>& out, uint64_t home, bool all) {
    for (size_t i = 0; i < r.pool.size();) {
        if (all || r.pool[i].home == home) {
            r.pooledBytes -= r.pool[i].bytes;
            out.push_back(std::move(r.pool[i]));
            r.pool[i] = std::move(r.pool.back());
            r.pool.pop_back();
        } else {
            ++i;
        }
    }
}
void really_free(std::vector<Pooled>& v) {
    for (Pooled& b : v) (void)hipFree(b.p);  // synchronises the device: every fence has passed afterwards
    for (Pooled& b : v)
        for (const Fence& f : b.fences) (void)hipEventDestroy(f.event);
    v.clear();
}
struct ThreadTag {
    uint64_t id;
    ThreadTag() {
        Registry& r = reg();
        std::lock_guard<std::mutex> lock(r.m);
        id = r.nextId++;
        r.threads.push_back(id);
    }
    ~ThreadTag() {
        Registry& r = reg();
        std::vector<Pooled> mine;
        {
            std::lock_guard<std::mutex> lock(r.m);
            r.threads.erase(std::remove(r.threads.begin(), r.threads.end(), id), r.threads.end());
            free_entries_locked(r, mine, id, false);
        }
        really_free(mine);
    }
};
uint64_t this_thread() {
    static thread_local ThreadTag t;
    return t.id;
}

size_t pool_cap() {  // bytes the pool may hold before a release really frees (EMF_POOL_MIB, default 16 GiB of 288)
    static const size_t cap = [] {
        const char* e = std::getenv("EMF_POOL_MIB");
        return (e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : size_t(16384)) << 20;
    }();
    return cap;
}
// fence `b` on every live stream of thread `me` and on the null stream (clears and uploads of constructors run
// there); called by `me` only, registry locked.  false: could not fence it.
bool fence_locked(Registry& r, Pooled& b, uint64_t me) {
    std::vector<hipStream_t> streams;
    for (const LiveStream& ls : r.streams)
        if (ls.owner == me) streams.push_back(ls.s);
    streams.push_back(nullptr);
    for (hipStream_t st : streams) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
        if (!ev || hipEventRecord(ev, st) != hipSuccess) {
            (void)hipGetLastError();
            if (ev) (void)hipEventDestroy(ev);
            return false;
        }
        b.fences.push_back(Fence{ev, st});
    }
    b.parked = false;
    return true;
}
bool fences_passed(Pooled& b) {
    while (!b.fences.empty()) {
        if (hipEventQuery(b.fences.back().event) != hipSuccess) {
            (void)hipGetLastError();  // hipErrorNotReady is not an error
            return false;
        }
        (void)hipEventDestroy(b.fences.back().event);  // (not re-used: an event remembers its last stream)
        b.fences.pop_back();
    }
    return true;
}
// buffers of `me` that a foreign thread released: fence them now (registry locked, called by `me`)
void adopt_parked_locked(Registry& r, uint64_t me, std::vector<Pooled>& unfenceable) {
    for (size_t i = 0; i < r.pool.size();) {
        Pooled& b = r.pool[i];
        if (b.home == me && b.parked && !fence_locked(r, b, me)) {
            r.pooledBytes -= b.bytes;
            unfenceable.push_back(std::move(b));
            r.pool[i] = std::move(r.pool.back());
            r.pool.pop_back();
        } else {
            ++i;
        }
    }
}
void* pool_acquire(size_t bytes) {
    Registry& r = reg();
    const uint64_t me = this_thread();
    std::vector<Pooled> slow;
    void* p = nullptr;
    {
        std::lock_guard<std::mutex> lock(r.m);
        adopt_parked_locked(r, me, slow);
        for (size_t i = 0; i < r.pool.size(); ++i)
            if (r.pool[i].home == me && r.pool[i].bytes == bytes && fences_passed(r.pool[i])) {
                p = r.pool[i].p;
                r.pooledBytes -= bytes;
                r.pool[i] = std::move(r.pool.back());
                r.pool.pop_back();
                break;
            }
    }
    really_free(slow);
    return p;
}
void pool_release(void* p, size_t bytes, uint64_t home) {
    Registry& r = reg();
    const uint64_t me = this_thread();
    std::vector<Pooled> slow;
    {
        std::lock_guard<std::mutex> lock(r.m);
        Pooled b{p, bytes, home, me != home, {}};
        const bool homeAlive = std::find(r.threads.begin(), r.threads.end(), home) != r.threads.end();
        if (r.pooledBytes + bytes > pool_cap() || !homeAlive) {
            slow.push_back(std::move(b));  // over the cap, or nobody left to fence it: a real free
        } else {
            if (me == home) {
                adopt_parked_locked(r, me, slow);
                if (!fence_locked(r, b, me)) {
                    slow.push_back(std::move(b));
                    b.p = nullptr;
                }
            }
            if (b.p) {
                r.pooledBytes += bytes;
                r.pool.push_back(std::move(b));
            }
        }
    }
    really_free(slow);
}
void register_stream(hipStream_t s) {
    Registry& r = reg();
    const uint64_t me = this_thread();
    std::lock_guard<std::mutex> lock(r.m);
    r.streams.push_back(LiveStream{s, me});
}
// A stream is about to be destroyed (by any thread).  It is drained if a fence was recorded on it or if its
// owner has parked buffers (work on them may sit on this stream and nothing else would order a later fence
// behind it); that completes every fence recorded on it, and those are destroyed before the stream goes.
void unregister_stream(hipStream_t s) {
    Registry& r = reg();
    bool drain = false;
    {
        std::lock_guard<std::mutex> lock(r.m);
        uint64_t owner = 0;
        for (const LiveStream& ls : r.streams)
            if (ls.s == s) owner = ls.owner;
        for (const Pooled& b : r.pool) {
            drain = drain || (b.parked && b.home == owner);
            for (const Fence& f : b.fences) drain = drain || f.stream == s;
        }
    }
    // The wait happens OUTSIDE the registry lock: work queued on `s` may be spinning on a peer's flag (the
    // rehearsal runs several ranks as threads of one process), and the rank that would raise the flag needs
    // pool_acquire / pool_release -- the same mutex -- to enqueue the exchange that does.
    if (drain) (void)hipStreamSynchronize(s);
    std::vector<hipEvent_t> dead;
    {
        std::lock_guard<std::mutex> lock(r.m);
        if (drain)
            for (Pooled& b : r.pool) {
                for (Fence& f : b.fences)
                    if (f.stream == s) dead.push_back(f.event);
                b.fences.erase(std::remove_if(b.fences.begin(), b.fences.end(), [s](const Fence& f) { return f.stream == s; }),
                               b.fences.end());
            }
        r.streams.erase(std::remove_if(r.streams.begin(), r.streams.end(), [s](const LiveStream& ls) { return ls.s == s; }),
                        r.streams.end());
    }
    for (hipEvent_t e : dead) (void)hipEventDestroy(e);
}
}  // namespace

size_t DeviceBuffer::pooledBytes() {
    Registry& r = reg();
    std::lock_guard<std::mutex> lock(r.m);
    return r.pooledBytes;
}
void DeviceBuffer::trimPool() {
    Registry& r = reg();
    std::vector<Pooled> all;
    {
        std::lock_guard<std::mutex> lock(r.m);
        free_entries_locked(r, all, 0, true);
    }
    really_free(all);
}

Stream::Stream() : owned_(true) {
    hipCheck(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking), "hipStreamCreate");
    register_stream(s_);
}
Stream::Stream(int priority) : owned_(true) {
    int least = 0, greatest = 0;  // numerically lower = higher priority
    hipCheck(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    const int p = priority > 0 ? greatest : (priority < 0 ? least : (least + greatest) / 2);
    hipCheck(hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, p), "hipStreamCreateWithPriority");
    register_stream(s_);
}
Stream::Stream(hipStream_t s) : s_(s), owned_(false) {}
Stream::~Stream() {
    if (ev_) (void)hipEventDestroy(ev_);
    if (owned_ && s_) {
        unregister_stream(s_);
        (void)hipStreamDestroy(s_);
    }
}
Stream::Stream(Stream&& o) noexcept : s_(o.s_), owned_(o.owned_), ev_(o.ev_) {
    o.s_ = nullptr;
    o.owned_ = false;
    o.ev_ = nullptr;
}
Stream& Stream::operator=(Stream&& o) noexcept {
    if (this != &o) {
        if (ev_) (void)hipEventDestroy(ev_);
        if (owned_ && s_) {
            unregister_stream(s_);
            (void)hipStreamDestroy(s_);
        }
        s_ = o.s_;
        owned_ = o.owned_;
        ev_ = o.ev_;
        o.s_ = nullptr;
        o.owned_ = false;
        o.ev_ = nullptr;
    }
    return *this;
}
Stream& Stream::Null() {
    static Stream null(nullptr);
    return null;
}
void Stream::waitForCompletion() const { hipCheck(hipStreamSynchronize(s_), "hipStreamSynchronize"); }
void Stream::record() {
    if (!ev_) hipCheck(hipEventCreateWithFlags(&ev_, hipEventDisableTiming), "hipEventCreate");
    hipCheck(hipEventRecord(ev_, s_), "hipEventRecord");
}
void Stream::waitOn(const Stream& other) {
    if (other.ev_) hipCheck(hipStreamWaitEvent(s_, other.ev_, 0), "hipStreamWaitEvent");
}
void Stream::waitFor(Stream& other) {
    other.record();
    waitOn(other);
}

DeviceBuffer::DeviceBuffer(size_t bytes) : n_(bytes), home_(this_thread()) {
    if (!bytes) return;
    p_ = pool_acquire(bytes);
    if (p_) return;
    hipError_t e = hipMalloc(&p_, bytes);
    if (e == hipErrorOutOfMemory) {  // give the pool back (waits for the device) and try once more
        (void)hipGetLastError();
        trimPool();
        e = hipMalloc(&p_, bytes);
    }
    hipCheck(e, "hipMalloc");
}
DeviceBuffer::~DeviceBuffer() {
    if (p_) pool_release(p_, n_, home_);
}
DeviceBuffer::DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_), home_(o.home_) {
    o.p_ = nullptr;
    o.n_ = 0;
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
        if (p_) pool_release(p_, n_, home_);
        p_ = o.p_;
        n_ = o.n_;
        home_ = o.home_;
        o.p_ = nullptr;
        o.n_ = 0;
    }
    return *this;
}
void DeviceBuffer::setZero(const Stream& s) const {
    if (n_) hipCheck(hipMemsetAsync(p_, 0, n_, s.get()), "hipMemsetAsync");
}
void DeviceBuffer::fill32(uint32_t pattern, const Stream& s) const {
    if (n_) hipCheck(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(p_),
                                       static_cast<int>(pattern), n_ / 4, s.get()),
                     "hipMemsetD32Async");
}
void DeviceBuffer::download(void* host, const Stream& s) const {
    if (!n_) return;
    hipCheck(hipMemcpyAsync(host, p_, n_, hipMemcpyDeviceToHost, s.get()), "hipMemcpyAsync D2H");
    s.waitForCompletion();
}
void DeviceBuffer::upload(const void* host, const Stream& s) const {
    if (n_) hipCheck(hipMemcpyAsync(p_, host, n_, hipMemcpyHostToDevice, s.get()),
                     "hipMemcpyAsync H2D");
}

// A switch that only -DEMF_DEBUG_SWITCHES builds read (types.hpp debugEnv) is set in the environment of a product build:
// one line on stderr per variable and process, then ignored.
const char* demotedSwitchSet(const char* name) {
    if (!std::getenv(name)) return nullptr;
    static std::mutex m;
    static std::vector<std::string> warned;
    std::lock_guard<std::mutex> lock(m);
    for (const auto& w : warned)
        if (w == name) return nullptr;
    warned.emplace_back(name);
    std::fprintf(stderr, "emfusion_amd: %s is set, but this build ignores it (a switch whose A/B is on record as lost; "
                         "`make -C emfusion_amd/csrc dbg` builds libemf_fusion_dbg.so, which reads it)\n", name);
    return nullptr;
}

}  // namespace emf
