// types.cpp -- RAII wrappers over HIP streams / device memory used by the host classes.
#include "types.hpp"

#include <cstring>

namespace emf {

void hipCheck(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw HipError(std::string(what) + ": " + hipGetErrorString(e), static_cast<int>(e));
}

void emfCheck(int rc, const char* what) {
    if (rc != EMF_OK)
        throw HipError(std::string(what) + ": " + emf_hip_last_error_string(), rc);
}

Stream::Stream() : owned_(true) {
    hipCheck(hipStreamCreateWithFlags(&s_, hipStreamNonBlocking), "hipStreamCreate");
}
Stream::Stream(int priority) : owned_(true) {
    int least = 0, greatest = 0;  // numerically lower = higher priority
    hipCheck(hipDeviceGetStreamPriorityRange(&least, &greatest), "hipDeviceGetStreamPriorityRange");
    const int p = priority > 0 ? greatest : (priority < 0 ? least : (least + greatest) / 2);
    hipCheck(hipStreamCreateWithPriority(&s_, hipStreamNonBlocking, p), "hipStreamCreateWithPriority");
}
Stream::Stream(hipStream_t s) : s_(s), owned_(false) {}
Stream::~Stream() {
    if (ev_) (void)hipEventDestroy(ev_);
    if (owned_ && s_) (void)hipStreamDestroy(s_);
}
Stream::Stream(Stream&& o) noexcept : s_(o.s_), owned_(o.owned_), ev_(o.ev_) {
    o.s_ = nullptr;
    o.owned_ = false;
    o.ev_ = nullptr;
}
Stream& Stream::operator=(Stream&& o) noexcept {
    if (this != &o) {
        if (ev_) (void)hipEventDestroy(ev_);
        if (owned_ && s_) (void)hipStreamDestroy(s_);
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
    if (bytes) hipCheck(hipMalloc(&p_, bytes), "hipMalloc");
}
DeviceBuffer::~DeviceBuffer() {
    if (p_) (void)hipFree(p_);
}
DeviceBuffer::DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), n_(o.n_) {
    o.p_ = nullptr;
    o.n_ = 0;
}
DeviceBuffer& DeviceBuffer::operator=(DeviceBuffer&& o) noexcept {
    if (this != &o) {
        if (p_) (void)hipFree(p_);
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
