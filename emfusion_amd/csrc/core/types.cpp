// types.cpp -- RAII wrappers over HIP streams / device memory used by the host classes.
#include "types.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
// stream of the releasing host thread, recorded at the stream's tail at the moment of the release --
// everything that could still touch the memory was enqueued before -- and is handed out again (same size)
// once all of them have completed (hipEventQuery: no wait).
// Pool, stream registry and events are PER HOST THREAD: an emf::EMFusion instance is driven by one thread
// (its streams are created there, its buffers released there), and several instances on several threads
// (the multi-rank rehearsal) must not record events on each other's streams -- HIP's event bookkeeping
// throws from inside the runtime when they do.  A thread's pool is really freed when the thread ends.
namespace {
struct Fence {
    hipEvent_t event;
    hipStream_t stream;  // where it was recorded: an event must not outlive its stream (see retire_fences_of)
};
struct Pooled {
    void* p;
    size_t bytes;
    std::vector<Fence> fences;
};
struct ThreadPool {
    std::vector<hipStream_t> streams;  // streams created by emf::Stream on this thread
    std::vector<Pooled> pool;
    size_t pooledBytes = 0;
    ~ThreadPool() {
        for (Pooled& b : pool) (void)hipFree(b.p);
        for (Pooled& b : pool)
            for (const Fence& f : b.fences) (void)hipEventDestroy(f.event);
    }
};
ThreadPool& tp() {
    static thread_local ThreadPool t;
    return t;
}
// every live emf::Stream of the process: a thread's list may name a stream that another thread destroyed
std::mutex g_liveMutex;
std::vector<hipStream_t> g_live;
bool stream_is_live(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_liveMutex);
    return std::find(g_live.begin(), g_live.end(), s) != g_live.end();
}

size_t pool_cap() {  // bytes a thread's pool may hold before it really frees (EMF_POOL_MIB, default 16 GiB of 288)
    static const size_t cap = [] {
        const char* e = std::getenv("EMF_POOL_MIB");
        return (e ? static_cast<size_t>(std::strtoull(e, nullptr, 10)) : size_t(16384)) << 20;
    }();
    return cap;
}
bool fences_passed(ThreadPool& t, Pooled& b) {
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
void* pool_acquire(size_t bytes) {
    ThreadPool& t = tp();
    for (size_t i = 0; i < t.pool.size(); ++i)
        if (t.pool[i].bytes == bytes && fences_passed(t, t.pool[i])) {
            void* p = t.pool[i].p;
            t.pooledBytes -= bytes;
            t.pool[i] = std::move(t.pool.back());
            t.pool.pop_back();
            return p;
        }
    return nullptr;
}
void pool_release(void* p, size_t bytes) {
    ThreadPool& t = tp();
    if (t.pooledBytes + bytes > pool_cap()) {  // over the cap: a real free (synchronises the device)
        (void)hipFree(p);
        return;
    }
    Pooled b{p, bytes, {}};
    t.streams.erase(std::remove_if(t.streams.begin(), t.streams.end(), [](hipStream_t s) { return !stream_is_live(s); }),
                    t.streams.end());
    std::vector<hipStream_t> streams = t.streams;
    streams.push_back(nullptr);  // the null stream: clears and uploads of constructors run there
    for (hipStream_t st : streams) {
        hipEvent_t ev = nullptr;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
        if (!ev || hipEventRecord(ev, st) != hipSuccess) {  // cannot fence it: free it the slow, safe way
            (void)hipGetLastError();
            if (ev) (void)hipEventDestroy(ev);
            for (const Fence& f : b.fences) (void)hipEventDestroy(f.event);
            (void)hipFree(p);
            return;
        }
        b.fences.push_back(Fence{ev, st});
    }
    t.pooledBytes += bytes;
    t.pool.push_back(std::move(b));
}
void register_stream(hipStream_t s) {
    tp().streams.push_back(s);
    std::lock_guard<std::mutex> lock(g_liveMutex);
    g_live.push_back(s);
}
// A stream is about to be destroyed: HIP's event bookkeeping keeps a reference to the stream an event was last
// recorded on, and querying such an event after the stream is gone throws from inside the runtime
// ("std::get: wrong index for variant").  So the stream is drained here, which completes every fence recorded
// on it, and those fences are destroyed -- not re-used -- before the stream goes.
void retire_fences_of(hipStream_t s) {
    ThreadPool& t = tp();
    bool any = false;
    for (const Pooled& b : t.pool)
        for (const Fence& f : b.fences) any = any || f.stream == s;
    if (!any) return;
    (void)hipStreamSynchronize(s);
    for (Pooled& b : t.pool) {
        for (Fence& f : b.fences)
            if (f.stream == s) (void)hipEventDestroy(f.event);
        b.fences.erase(std::remove_if(b.fences.begin(), b.fences.end(), [s](const Fence& f) { return f.stream == s; }),
                       b.fences.end());
    }
}
void unregister_stream(hipStream_t s) {
    retire_fences_of(s);
    auto& v = tp().streams;
    v.erase(std::remove(v.begin(), v.end(), s), v.end());
    std::lock_guard<std::mutex> lock(g_liveMutex);
    g_live.erase(std::remove(g_live.begin(), g_live.end(), s), g_live.end());
}
}  // namespace

size_t DeviceBuffer::pooledBytes() { return tp().pooledBytes; }
void DeviceBuffer::trimPool() {
    ThreadPool& t = tp();
    std::vector<Pooled> all;
    all.swap(t.pool);
    t.pooledBytes = 0;
    for (Pooled& b : all) {
        (void)hipFree(b.p);  // synchronises the device: the fences have passed afterwards
        for (const Fence& f : b.fences) (void)hipEventDestroy(f.event);
    }
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

DeviceBuffer::DeviceBuffer(size_t bytes) : n_(bytes) {
    if (!bytes) return;
    p_ = pool_acquire(bytes);
    if (!p_) hipCheck(hipMalloc(&p_, bytes), "hipMalloc");
}
DeviceBuffer::~DeviceBuffer() {
    if (p_) pool_release(p_, n_);
}
DeviceBuffer::DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) {
    o.p_ = nullptr;
    o.n_ = 0;
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
        if (p_) pool_release(p_, n_);
        p_ = o.p_;
        n_ = o.n_;
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

}  // namespace emf
