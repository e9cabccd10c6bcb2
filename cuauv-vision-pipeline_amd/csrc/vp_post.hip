// Posts by DMA: a device image goes into the ring slot of its post block without a host pass over the pixels.
//
// The module runtime publishes debug images for the GUI by default (reference core/base.py:846-876 queues a uint8 copy per post(),
// :832-839 flushes the queue through write_frame = a memcpy into the slot, lib/camera_message_framework.cpp:306-374;
// --enable-performance is opt-in).  With images living in HBM that was a synchronous download into a fresh host array plus the
// library's memcpy - two host passes over 10 MB per red_buoy frame.  Here the copy engine writes the slot itself:
//
//   post():   cmf_write_begin (slot opened: first sequence word bumped)  ->  vp_post_d2h: an event on the context's stream marks
//             "the image as it is now", a post stream (one of four lanes; a block keeps its lane) waits for it and copies device -> slot (the block's mapping is page-locked
//             once, vp_host_register), a second event marks the end of the copy
//   flush:    once that event has passed: cmf_write_commit (metadata, second sequence word, uid, wake-up)
//
// Snapshot rule (post() hands over the image as it is at the call): work queued LATER on the context's stream may overwrite the
// image while the copy still reads it - vp_post_fence makes the context's stream wait for the copy first; the host mirror calls
// it from the one place every in-place writer passes (DeviceMat._before_write).  Kernels that only read the image run beside the copy.
#include "vp_internal.h"

namespace {
constexpr int kPostEvents = (int)(sizeof(((vp_ctx*)nullptr)->post_free) / sizeof(hipEvent_t));

struct PostLock {
    int* w;
    explicit PostLock(vp_ctx* ctx) : w(&ctx->post_lock) { while (__atomic_exchange_n(w, 1, __ATOMIC_ACQUIRE)) __builtin_ia32_pause(); }
    ~PostLock() { __atomic_store_n(w, 0, __ATOMIC_RELEASE); }
};

constexpr int kLanes = (int)(sizeof(((vp_ctx*)nullptr)->post_stream) / sizeof(hipStream_t));

int post_setup(vp_ctx* ctx)
{
    if (ctx->post_stream[0]) return VP_OK;
    (void)hipSetDevice(ctx->device);
    for (int i = kLanes - 1; i >= 0; i--) VP_HIP(ctx, hipStreamCreateWithFlags(&ctx->post_stream[i], hipStreamNonBlocking));
    VP_HIP(ctx, hipEventCreateWithFlags(&ctx->post_fork, hipEventDisableTiming));
    return VP_OK;
}
}  // namespace

extern "C" {

int vp_post_d2h(vp_ctx* ctx, int lane, void* host_dst, const void* dev_src, size_t bytes, void** done)
{
    if (!ctx || !host_dst || !dev_src || !bytes || !done || lane < 0) return VP_ERR_INVALID;
    hipEvent_t ev = nullptr;
    {
        PostLock g(ctx);
        const int rc = post_setup(ctx);
        if (rc != VP_OK) return rc;
        if (ctx->post_nfree > 0) ev = ctx->post_free[--ctx->post_nfree];
    }
    if (!ev) VP_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    // "the image as it is now": everything queued so far on the context's stream, nothing queued later.  Copies of one lane run in
    // the order they were queued (a block keeps its lane, so a slot is never written by two copies at once); different lanes run
    // side by side on the copy engines - a copy has ~30 us of fixed latency, and a module's posts of one frame go to different blocks.
    hipStream_t ps = ctx->post_stream[lane % kLanes];
    hipError_t e = hipEventRecord(ctx->post_fork, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ps, ctx->post_fork, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ps);
    if (e == hipSuccess) e = hipEventRecord(ev, ps);
    if (e != hipSuccess) {
        // whatever part was queued must not be writing into the slot when the caller aborts the write
        (void)hipStreamSynchronize(ps);
        (void)hipEventDestroy(ev);
        return vp_fail(ctx, VP_ERR_HIP, "post copy (device image -> ring slot)", e);
    }
    *done = (void*)ev;
    return VP_OK;
}

int vp_post_done(vp_ctx* ctx, void* done)
{
    if (!ctx || !done) return VP_ERR_INVALID;
    const hipError_t e = hipEventQuery((hipEvent_t)done);
    if (e == hipSuccess) return 1;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 0; }
    return vp_fail(ctx, VP_ERR_HIP, "hipEventQuery (post copy)", e);
}

int vp_post_wait(vp_ctx* ctx, void* done)
{
    if (!ctx || !done) return VP_ERR_INVALID;
    VP_HIP(ctx, hipEventSynchronize((hipEvent_t)done));
    return VP_OK;
}

int vp_post_fence(vp_ctx* ctx, void* done)
{
    if (!ctx || !done) return VP_ERR_INVALID;
    VP_HIP(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)done, 0));
    return VP_OK;
}

int vp_post_free(vp_ctx* ctx, void* done)
{
    if (!done) return VP_ERR_INVALID;
    if (ctx) {
        PostLock g(ctx);
        if (ctx->post_nfree < kPostEvents) { ctx->post_free[ctx->post_nfree++] = (hipEvent_t)done; return VP_OK; }
    }
    (void)hipEventDestroy((hipEvent_t)done);
    return VP_OK;
}

}  // extern "C"

void vp_post_teardown(vp_ctx* ctx)
{
    if (!ctx->post_stream[0]) return;
    for (int i = 0; i < kLanes; i++) (void)hipStreamSynchronize(ctx->post_stream[i]);
    for (int i = 0; i < ctx->post_nfree; i++) (void)hipEventDestroy(ctx->post_free[i]);
    ctx->post_nfree = 0;
    (void)hipEventDestroy(ctx->post_fork);
    for (int i = 0; i < kLanes; i++) { (void)hipStreamDestroy(ctx->post_stream[i]); ctx->post_stream[i] = nullptr; }
}
