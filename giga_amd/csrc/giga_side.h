// The library's second stream of a device (round 5).  A training step is made of sequences of latency-bound launches, each waiting
// for its predecessor (the data-gradient chain of the U-Net: 13 convolutions + 2 pooling steps) and of launches that wait for ONE link
// of such a chain only (the 13 weight gradients, the decoders' weight-gradient reduces): on one stream they take the sum of their
// times, at 2-3 TB/s and a few per cent of the MFMA rate.  The second kind therefore goes to a library-owned stream: forked from the
// caller's stream with an event behind the launch each one needs, joined back with one event, so the caller sees ordinary stream
// semantics (a capturing stream captures both branches).  One side stream per device; a SideScope holds that device's mutex
// for the enqueue of a whole backward pass because the events are shared by the device's callers.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

namespace giga {

struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork[16] = {};
    hipEvent_t join[4] = {};
    bool ok = false;
};

class SideScope {
  public:
    // enable = false (or no stream could be made): stream() is the caller's stream and fork / join do nothing
    SideScope(hipStream_t main, bool enable);
    ~SideScope();
    SideScope(const SideScope&) = delete;
    SideScope& operator=(const SideScope&) = delete;
    bool active() const { return side_ != nullptr; }
    hipStream_t stream() const { return side_ ? side_->stream : main_; }
    hipStream_t main() const { return main_; }
    int fork();          // what the caller's stream holds so far precedes what is enqueued on stream() from here on
    int join();          // what stream() holds so far precedes what is enqueued on the caller's stream from here on
  private:
    hipStream_t main_;
    SideStream* side_ = nullptr;
    int nfork_ = 0, njoin_ = 0, dev_ = 0;
};

void side_streams_forget();       // giga_forget_device_state(): the handles died with the device's context

}  // namespace giga
