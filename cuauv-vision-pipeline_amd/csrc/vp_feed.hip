// Frame feeder: a thread of the library's own that keeps the newest frame of a shared-memory block in HBM.
//
// The module runtime's loop (reference core/base.py:711-844) reads a frame, processes it, reads the next.  With the frame moved out of
// the ring slot by a DMA (vp_host_register + cmf_peek_frame, round 3) the loop still waited for that copy - 0.12 ms of a 0.43 ms
// iteration at 1080p.  The feeder takes the wait off the loop: it waits on the block's condition variable, and whenever a newer frame
// exists it copies it from the slot into one of a few device buffers on a stream of its own, checks the slot's sequence number AFTER
// the copy (a copy the writer overtook is dropped) and publishes the buffer as "newest".  The loop's read is then a pointer hand-over.
//
// The block library is not linked: its entry points arrive as function pointers (the Python binding passes the addresses of
// cmf_wait_for_frame / cmf_peek_frame / cmf_peek_validate / create_frame / delete_frame), so libvp keeps no dependency on it.
// A buffer the consumer gives back may still be read by kernels queued on the consumer's stream: the release records an event there
// and the next copy into that buffer waits for it on the feeder's stream.
#include "vp_internal.h"
#include <atomic>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>

namespace {
constexpr int kSlots = 4;
constexpr size_t kFrameBytes = 360;          // sizeof(Frame) of include/camera_message_framework_c.h; the two offsets below likewise: asserted
constexpr size_t kFrameUidOff = 40;          // at compile time in csrc/cmf.cpp and by the binding before it starts a feeder
constexpr size_t kFrameTotalOff = 56;        // (Frame: width, height, depth, type_size (4 x 8), acquisition_time @32, uid @40, data @48, total_size @56)
typedef int (*fn_wait_t)(void*, uint64_t, uint32_t);
typedef int (*fn_peek_t)(void*, void*, const void**, uint64_t*);
typedef int (*fn_validate_t)(void*, uint64_t, uint64_t);
typedef void* (*fn_create_frame_t)(void);
typedef void (*fn_delete_frame_t)(void*);

struct Slot {
    void* dev = nullptr;
    int state = 0;                           // 0 free, 1 ready (holds a complete frame nobody has taken), 2 held by the consumer
    bool wait_release = false;               // an event recorded at release has to pass before the buffer is written again
    hipEvent_t released = nullptr;
    uint64_t uid = 0;
    unsigned char meta[kFrameBytes];
};
}  // namespace

struct vp_feeder {
    int device = 0;
    hipStream_t stream = nullptr;
    void* block = nullptr;
    void* frame = nullptr;
    size_t entry_bytes = 0;
    fn_wait_t wait = nullptr;
    fn_peek_t peek = nullptr;
    fn_validate_t validate = nullptr;
    fn_create_frame_t create_frame = nullptr;
    fn_delete_frame_t delete_frame = nullptr;
    Slot slots[kSlots];
    std::mutex mu;
    int newest = -1;
    uint64_t taken_uid = 0;
    std::atomic<bool> stop{false};
    std::atomic<int> deleted{0};
    std::atomic<unsigned long long> fetched{0}, torn{0};
    std::thread th;
    char err[160] = "";
};

static void feeder_loop(vp_feeder* f)
{
    (void)hipSetDevice(f->device);
    uint64_t have = 0;
    while (!f->stop.load(std::memory_order_acquire)) {
        const auto t_wait = std::chrono::steady_clock::now();
        if (f->wait(f->block, have, 2000) != 1) {
            // a wait that comes back empty-handed well before its bound could not wait (the block's mutex is unusable): do not spin
            if (std::chrono::steady_clock::now() - t_wait < std::chrono::microseconds(500)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
            continue;
        }
        const void* payload = nullptr;
        uint64_t ticket = 0;
        const int rc = f->peek(f->block, f->frame, &payload, &ticket);
        if (rc == 2) { f->deleted.store(1, std::memory_order_release); std::this_thread::sleep_for(std::chrono::milliseconds(2)); continue; }
        if (rc != 0) continue;
        unsigned char meta[kFrameBytes];
        memcpy(meta, f->frame, kFrameBytes);
        uint64_t uid, total;
        memcpy(&uid, meta + kFrameUidOff, 8);
        memcpy(&total, meta + kFrameTotalOff, 8);
        have = uid;
        if (total == 0 || total > f->entry_bytes) continue;
        int pick = -1;
        {
            std::lock_guard<std::mutex> g(f->mu);
            for (int i = 0; i < kSlots; i++)
                if (f->slots[i].state == 0) { pick = i; break; }
            if (pick < 0)                     // every buffer is ready-or-held: the ready one that is not the newest is stale by definition
                for (int i = 0; i < kSlots; i++)
                    if (f->slots[i].state == 1 && i != f->newest) { pick = i; break; }
            if (pick >= 0) f->slots[pick].state = 3;   // being filled
        }
        if (pick < 0) {
            // The consumer holds every buffer (the binding keeps that from happening: beyond two held buffers it hands out private
            // copies).  Wait a little and look at the SAME frame again: the peek answers "nothing new" while the Frame still carries
            // this uid, so the Frame is put back to the uid before it as well.
            std::this_thread::sleep_for(std::chrono::microseconds(200));
            have = uid - 1;
            memcpy((unsigned char*)f->frame + kFrameUidOff, &have, 8);
            continue;
        }
        Slot& s = f->slots[pick];
        bool ok = true;
        if (s.wait_release) { ok = hipStreamWaitEvent(f->stream, s.released, 0) == hipSuccess; s.wait_release = false; }
        ok = ok && hipMemcpyAsync(s.dev, payload, total, hipMemcpyHostToDevice, f->stream) == hipSuccess;
        ok = ok && hipStreamSynchronize(f->stream) == hipSuccess;
        const bool intact = ok && f->validate(f->block, uid, ticket) == 1;
        std::lock_guard<std::mutex> g(f->mu);
        if (!intact) {
            s.state = 0;
            if (ok) f->torn.fetch_add(1); else snprintf(f->err, sizeof f->err, "feeder: copy out of the ring slot failed: %s", hipGetErrorString(hipGetLastError()));
            continue;
        }
        if (f->newest >= 0 && f->slots[f->newest].state == 1) f->slots[f->newest].state = 0;   // superseded before anybody took it
        memcpy(s.meta, meta, kFrameBytes);
        s.uid = uid;
        s.state = 1;
        f->newest = pick;
        f->fetched.fetch_add(1);
    }
}

extern "C" {

vp_feeder* vp_feeder_start(int device, void* block, size_t entry_bytes, void* fn_wait, void* fn_peek, void* fn_validate, void* fn_create_frame,
                           void* fn_delete_frame)
{
    if (!block || !entry_bytes || !fn_wait || !fn_peek || !fn_validate || !fn_create_frame || !fn_delete_frame) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    vp_feeder* f = new vp_feeder();
    f->device = device;
    f->block = block;
    f->entry_bytes = entry_bytes;
    f->wait = (fn_wait_t)fn_wait; f->peek = (fn_peek_t)fn_peek; f->validate = (fn_validate_t)fn_validate;
    f->create_frame = (fn_create_frame_t)fn_create_frame; f->delete_frame = (fn_delete_frame_t)fn_delete_frame;
    bool ok = hipStreamCreateWithFlags(&f->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i < kSlots; i++)
        ok = hipMalloc(&f->slots[i].dev, entry_bytes) == hipSuccess && hipEventCreateWithFlags(&f->slots[i].released, hipEventDisableTiming) == hipSuccess;
    f->frame = ok ? f->create_frame() : nullptr;
    if (!ok || !f->frame) {
        for (int i = 0; i < kSlots; i++) { if (f->slots[i].dev) hipFree(f->slots[i].dev); if (f->slots[i].released) hipEventDestroy(f->slots[i].released); }
        if (f->stream) hipStreamDestroy(f->stream);
        if (f->frame) f->delete_frame(f->frame);
        delete f;
        return nullptr;
    }
    f->th = std::thread(feeder_loop, f);
    return f;
}

// -> 0: meta_out (sizeof(Frame) bytes) describes the frame in *dev_out, which is the caller's until vp_feeder_release; 1: nothing newer
// than the last frame taken; 2: the block was deleted by its creator; negative: invalid arguments
int vp_feeder_take(vp_feeder* f, void* meta_out, void** dev_out)
{
    if (!f || !meta_out || !dev_out) return VP_ERR_INVALID;
    std::lock_guard<std::mutex> g(f->mu);
    if (f->newest >= 0 && f->slots[f->newest].state == 1 && f->slots[f->newest].uid > f->taken_uid) {
        Slot& s = f->slots[f->newest];
        s.state = 2;
        f->taken_uid = s.uid;
        memcpy(meta_out, s.meta, kFrameBytes);
        *dev_out = s.dev;
        return 0;
    }
    return f->deleted.load(std::memory_order_acquire) ? 2 : 1;
}

// the consumer is done with a buffer (work that still reads it may be queued on `consumer`'s stream: the next copy waits for it)
int vp_feeder_release(vp_feeder* f, vp_ctx* consumer, void* dev)
{
    if (!f || !dev) return VP_ERR_INVALID;
    std::lock_guard<std::mutex> g(f->mu);
    for (int i = 0; i < kSlots; i++) {
        Slot& s = f->slots[i];
        if (s.dev != dev || s.state != 2) continue;
        if (consumer && hipEventRecord(s.released, consumer->stream) == hipSuccess) s.wait_release = true;
        else {
            // no stream to order the next copy behind (context gone, or the record failed): whatever still reads the buffer has to
            // have finished before the buffer is written again - wait for the device here, once, instead of racing
            (void)hipSetDevice(f->device);
            if (hipDeviceSynchronize() != hipSuccess) snprintf(f->err, sizeof f->err, "feeder: could not order a released buffer behind its readers");
            (void)hipGetLastError();
        }
        s.state = 0;
        return VP_OK;
    }
    return VP_ERR_INVALID;
}

int vp_feeder_counts(vp_feeder* f, unsigned long long* fetched, unsigned long long* torn)
{
    if (!f) return VP_ERR_INVALID;
    if (fetched) *fetched = f->fetched.load();
    if (torn) *torn = f->torn.load();
    return VP_OK;
}

// stops the thread; the device buffers live on until vp_feeder_destroy (images handed out earlier may still be in use)
int vp_feeder_stop(vp_feeder* f)
{
    if (!f) return VP_ERR_INVALID;
    f->stop.store(true, std::memory_order_release);
    if (f->th.joinable()) f->th.join();
    if (f->frame) { f->delete_frame(f->frame); f->frame = nullptr; }
    return VP_OK;
}

int vp_feeder_destroy(vp_feeder* f)
{
    if (!f) return VP_ERR_INVALID;
    vp_feeder_stop(f);
    (void)hipSetDevice(f->device);
    (void)hipStreamSynchronize(f->stream);
    for (int i = 0; i < kSlots; i++) { hipFree(f->slots[i].dev); hipEventDestroy(f->slots[i].released); }
    hipStreamDestroy(f->stream);
    delete f;
    return VP_OK;
}

}  // extern "C"
