// C ABI of libvp.so (see include/vp.h): context, workspace, per-operator host entry points and the
// batched device-resident chain.  No CPU arithmetic path exists here: every operator stages its
// operands into HBM and launches the HIP kernels of vp_color / vp_morph / vp_ccl (+ contours) / vp_balance / vp_filter / vp_yolo.
#include "vp_internal.h"
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

static char g_err[256] = "";

int vp_fail(vp_ctx* ctx, int code, const char* what, hipError_t e)
{
    char* dst = ctx ? ctx->err : g_err;
    if (e != hipSuccess) snprintf(dst, 256, "%s: %s", what, hipGetErrorString(e));
    else snprintf(dst, 256, "%s", what);
    if (ctx) snprintf(g_err, 256, "%s", dst);
    return code;
}

extern "C" {

int vp_version(void) { return 100; }

const char* vp_strerror(int code)
{
    switch (code) {
        case VP_OK: return "ok";
        case VP_ERR_INVALID: return "invalid argument";
        case VP_ERR_HIP: return "HIP runtime error";
        case VP_ERR_NOMEM: return "out of memory";
        case VP_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown error";
    }
}

const char* vp_last_error(const vp_ctx* ctx) { return ctx ? ctx->err : g_err; }

int vp_get_tables(uint16_t* gamma, uint16_t* cbrt_tab, int32_t* sdiv, int32_t* hdiv180, int32_t* lab_coeffs)
{
    vp_host_tables(gamma, cbrt_tab, sdiv, hdiv180, lab_coeffs);
    return VP_OK;
}

int vp_device_count(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

// "0000:05:00.0"-style PCI address of a device: /sys/bus/pci/devices/<address>/numa_node tells which host memory and cores sit next
// to it (the multi-device dispatcher binds each feeder thread there)
int vp_device_pci_bus_id(int device, char* out, int len)
{
    if (!out || len < 16) return VP_ERR_INVALID;
    out[0] = 0;
    hipError_t e = hipDeviceGetPCIBusId(out, len, device);
    if (e != hipSuccess) return vp_fail(nullptr, VP_ERR_HIP, "hipDeviceGetPCIBusId", e);
    for (char* c = out; *c; c++) *c = (char)tolower(*c);
    return VP_OK;
}

vp_ctx* vp_create(int device)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { vp_fail(nullptr, VP_ERR_HIP, "no HIP device (libvp has no CPU path)", e); return nullptr; }
    if (device < 0 || device >= ndev) { vp_fail(nullptr, VP_ERR_INVALID, "device index out of range"); return nullptr; }
    if ((e = hipSetDevice(device)) != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "hipSetDevice", e); return nullptr; }
    vp_ctx* ctx = new (std::nothrow) vp_ctx();
    if (!ctx) return nullptr;
    memset(ctx, 0, sizeof *ctx);
    ctx->device = device;
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "hipGetDeviceProperties", e); delete ctx; return nullptr; }
    ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "hipStreamCreate", e); delete ctx; return nullptr; }
    ctx->stream = ctx->own_stream;
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    ctx->chain_streams = 1;
    if (const char* env = getenv("VP_CHAIN_STREAMS")) { const int v = atoi(env); if (v >= 1 && v <= 4) ctx->chain_streams = v; }
    for (int i = 0; i < 4; i++) {
        if (hipStreamCreateWithFlags(&ctx->aux[i], hipStreamNonBlocking) != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "aux stream"); delete ctx; return nullptr; }
        hipEventCreateWithFlags(&ctx->ev_join[i], hipEventDisableTiming);
    }
    hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming);
    if (hipStreamCreateWithFlags(&ctx->fb_stream, hipStreamNonBlocking) != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "side stream"); delete ctx; return nullptr; }
    hipEventCreateWithFlags(&ctx->ev_fb_fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_fb_join, hipEventDisableTiming);
    hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming);
    ctx->ccl_levels = 2;
    ctx->ccl_mcap = -1;
    ctx->flat_ops = 1;
    for (size_t& v : ctx->c3_lds_set) v = 0;
    if (const char* env = getenv("VP_CCL_LEVELS")) { const int v = atoi(env); if (v == 1 || v == 2) ctx->ccl_levels = v; }
    // tables: gamma u16[256] | cbrt u16[2048] | sdiv i32[256] | hdiv i32[256]
    std::vector<uint16_t> gamma(256), cbrt(3072);
    std::vector<int32_t> sdiv(256), hdiv(256);
    int32_t labC[9];
    vp_host_tables(gamma.data(), cbrt.data(), sdiv.data(), hdiv.data(), labC);
    static const int32_t expectC[9] = {1777, 1541, 778, 871, 2929, 296, 73, 448, 3575};
    if (memcmp(labC, expectC, sizeof labC) != 0) { vp_fail(nullptr, VP_ERR_INVALID, "Lab coefficient table mismatch"); delete ctx; return nullptr; }
    const size_t bytes = 512 + 4096 + 1024 + 1024;
    if ((e = hipMalloc(&ctx->d_tables, bytes + 256)) != hipSuccess) { vp_fail(nullptr, VP_ERR_NOMEM, "hipMalloc tables", e); delete ctx; return nullptr; }
    uint8_t* base = (uint8_t*)ctx->d_tables;
    hipMemcpy(base, gamma.data(), 512, hipMemcpyHostToDevice);
    hipMemcpy(base + 512, cbrt.data(), 4096, hipMemcpyHostToDevice);
    hipMemcpy(base + 512 + 4096, sdiv.data(), 1024, hipMemcpyHostToDevice);
    e = hipMemcpy(base + 512 + 4096 + 1024, hdiv.data(), 1024, hipMemcpyHostToDevice);
    if (e != hipSuccess) { vp_fail(nullptr, VP_ERR_HIP, "table upload", e); hipFree(ctx->d_tables); delete ctx; return nullptr; }
    ctx->tab.gamma = (const uint16_t*)base;
    ctx->tab.cbrt = (const uint16_t*)(base + 512);
    ctx->tab.sdiv = (const int32_t*)(base + 512 + 4096);
    ctx->tab.hdiv = (const int32_t*)(base + 512 + 4096 + 1024);
    ctx->cb_folds_own = (u32*)(base + bytes);       // a word of the context's own: workspace pointers do not survive a later call
    hipMemset(ctx->cb_folds_own, 0, 4);
    return ctx;
}

int vp_destroy(vp_ctx* ctx)
{
    if (!ctx) return VP_ERR_INVALID;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    vp_post_teardown(ctx);
    if (ctx->ws) hipFree(ctx->ws);
    if (ctx->c3_acc) hipFree(ctx->c3_acc);
    if (ctx->ct_hint_host) hipHostFree(ctx->ct_hint_host);
    for (int i = 0; i < 4; i++) {
        if (ctx->ring_buf[i]) hipHostFree(ctx->ring_buf[i]);
        if (ctx->ring_ev[i]) hipEventDestroy(ctx->ring_ev[i]);
    }
    if (ctx->hstage) hipHostFree(ctx->hstage);
    if (ctx->d_tables) hipFree(ctx->d_tables);
    hipEventDestroy(ctx->ev0);
    hipEventDestroy(ctx->ev1);
    for (int i = 0; i < 4; i++) { hipStreamDestroy(ctx->aux[i]); hipEventDestroy(ctx->ev_join[i]); }
    hipEventDestroy(ctx->ev_fork);
    hipStreamDestroy(ctx->fb_stream);
    hipEventDestroy(ctx->ev_fb_fork);
    hipEventDestroy(ctx->ev_fb_join);
    hipEventDestroy(ctx->ev_upload);
    hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return VP_OK;
}

int vp_set_stream(vp_ctx* ctx, void* hip_stream)
{
    if (!ctx) return VP_ERR_INVALID;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return VP_OK;
}
void* vp_get_stream(vp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int vp_set_option(vp_ctx* ctx, int option, int value)
{
    if (!ctx) return VP_ERR_INVALID;
    if (option == VP_OPT_CHAIN_STREAMS && value >= 1 && value <= 4) { ctx->chain_streams = value; return VP_OK; }
    if (option == VP_OPT_CCL_LEVELS && (value == 1 || value == 2)) { ctx->ccl_levels = value; return VP_OK; }
    if (option == VP_OPT_CCL_MERGE_CAP && value >= -1) { ctx->ccl_mcap = value; return VP_OK; }
    if (option == VP_OPT_FLAT_OPS && (value == 0 || value == 1)) { ctx->flat_ops = value; return VP_OK; }
    return vp_fail(ctx, VP_ERR_INVALID, "vp_set_option");
}

int vp_synchronize(vp_ctx* ctx)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VP_OK;
}

int vp_timer_start(vp_ctx* ctx)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return VP_OK;
}
int vp_timer_stop(vp_ctx* ctx, float* ms)
{
    if (!ctx || !ms) return VP_ERR_INVALID;
    VP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    VP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    VP_HIP(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return VP_OK;
}

int vp_profile_begin(vp_ctx* ctx, int max_records)
{
    if (!ctx || max_records <= 0) return VP_ERR_INVALID;
    vp_prof& P = ctx->prof;
    if (P.cap < max_records) {
        for (int i = 0; i < 2 * P.cap; i++) (void)hipEventDestroy(P.ev[i]);
        free(P.ev);
        free(P.ids);
        P.ev = (hipEvent_t*)malloc(sizeof(hipEvent_t) * 2 * max_records);
        P.ids = (int*)malloc(sizeof(int) * max_records);
        if (!P.ev || !P.ids) { P.cap = 0; return vp_fail(ctx, VP_ERR_NOMEM, "profile records"); }
        for (int i = 0; i < 2 * max_records; i++) VP_HIP(ctx, hipEventCreate(&P.ev[i]));
        P.cap = max_records;
    }
    P.used = 0;
    P.on = true;
    return VP_OK;
}

int vp_profile_end(vp_ctx* ctx, double* total_ms, int32_t* launches)
{
    if (!ctx || !total_ms || !launches) return VP_ERR_INVALID;
    vp_prof& P = ctx->prof;
    P.on = false;
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < VP_PROF_KERNELS; k++) { total_ms[k] = 0; launches[k] = 0; }
    for (int r = 0; r < P.used; r++) {
        float ms = 0;
        VP_HIP(ctx, hipEventElapsedTime(&ms, P.ev[2 * r], P.ev[2 * r + 1]));
        total_ms[P.ids[r]] += ms;
        launches[P.ids[r]]++;
    }
    return VP_OK;
}

const char* vp_profile_kernel_name(int id)
{
    static const char* names[VP_PROF_KERNELS] = {"k_color_thresh", "k_morph_bits", "k_ccl_local", "k_ccl_boundary", "k_ccl_flatten", "k_ccl_rank",
                                                  "k_ccl_bg", "k_ccl_stats", "k_ccl_final", "k_ccl_write", "memset", "other",
                                                  "k_ccl2_local", "k_ccl2_merge", "k_ccl2_write"};
    return (id >= 0 && id < VP_PROF_KERNELS) ? names[id] : "?";
}

int vp_dev_alloc(vp_ctx* ctx, size_t bytes, void** p)
{
    if (!ctx || !p) return VP_ERR_INVALID;
    hipSetDevice(ctx->device);
    hipError_t e = hipMalloc(p, bytes ? bytes : 1);
    if (e != hipSuccess) return vp_fail(ctx, VP_ERR_NOMEM, "hipMalloc", e);
    return VP_OK;
}
int vp_dev_free(vp_ctx* ctx, void* p)
{
    if (!ctx) return hipFree(p) == hipSuccess ? VP_OK : VP_ERR_HIP;   // device memory outlives the context it was allocated through
    VP_HIP(ctx, hipFree(p));
    return VP_OK;
}
int vp_host_alloc(vp_ctx* ctx, size_t bytes, void** p)
{
    if (!ctx || !p) return VP_ERR_INVALID;
    hipSetDevice(ctx->device);
    hipError_t e = hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return vp_fail(ctx, VP_ERR_NOMEM, "hipHostMalloc", e);
    return VP_OK;
}
int vp_host_free(vp_ctx* ctx, void* p)
{
    if (!ctx) return hipHostFree(p) == hipSuccess ? VP_OK : VP_ERR_HIP;   // page-locked memory is not tied to a context
    VP_HIP(ctx, hipHostFree(p));
    return VP_OK;
}
int vp_host_register(vp_ctx* ctx, void* p, size_t bytes)
{
    if (!ctx || !p || !bytes) return VP_ERR_INVALID;
    hipSetDevice(ctx->device);
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();                       // a refusal is an answer, not a fault of the context
        snprintf(ctx->err, sizeof ctx->err, "hipHostRegister refused %zu bytes: %s", bytes, hipGetErrorString(e));
        return VP_ERR_UNSUPPORTED;
    }
    return VP_OK;
}
int vp_host_unregister(vp_ctx* ctx, void* p)
{
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { (void)hipGetLastError(); return ctx ? vp_fail(ctx, VP_ERR_HIP, "hipHostUnregister", e) : VP_ERR_HIP; }
    return VP_OK;
}
int vp_memcpy_d2d_async(vp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx || !dst || !src) return VP_ERR_INVALID;
    if (bytes) VP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return VP_OK;
}
int vp_memcpy_h2d(vp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VP_OK;
}
// Enqueues the copy and marks its end on the stream; vp_wait_uploads returns once every copy enqueued so far has read its source
// (kernels enqueued behind the copies are not waited for).
int vp_memcpy_h2d_async(vp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    VP_HIP(ctx, hipEventRecord(ctx->ev_upload, ctx->stream));
    return VP_OK;
}
int vp_wait_uploads(vp_ctx* ctx)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipEventSynchronize(ctx->ev_upload));
    return VP_OK;
}
int vp_memcpy_d2h(vp_ctx* ctx, void* dst, const void* src, size_t bytes)
{
    if (!ctx) return VP_ERR_INVALID;
    VP_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return VP_OK;
}

}  // extern "C"

// ---- workspace ------------------------------------------------------------------------------------

int vp_ws_reserve(vp_ctx* ctx, size_t bytes)
{
    ctx->ws_off = 0;
    if (bytes <= ctx->ws_cap) return VP_OK;
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) { hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_cap = 0; }
    const size_t want = vp_align(bytes + bytes / 8, 1 << 20);
    hipError_t e = hipMalloc((void**)&ctx->ws, want);
    if (e != hipSuccess) return vp_fail(ctx, VP_ERR_NOMEM, "workspace hipMalloc", e);
    ctx->ws_cap = want;
    return VP_OK;
}

void* vp_ws_take(vp_ctx* ctx, size_t bytes)
{
    const size_t off = vp_align(ctx->ws_off);
    if (off + bytes > ctx->ws_cap) return nullptr;
    ctx->ws_off = off + bytes;
    return ctx->ws + off;
}

void* vp_hstage(vp_ctx* ctx, size_t bytes)
{
    if (bytes <= ctx->hstage_cap) return ctx->hstage;
    if (ctx->hstage) { (void)hipStreamSynchronize(ctx->stream); (void)hipHostFree(ctx->hstage); ctx->hstage = nullptr; ctx->hstage_cap = 0; }
    const size_t cap = (bytes + (1u << 20) - 1) >> 20 << 20;
    void* p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    ctx->hstage = (uint8_t*)p;
    ctx->hstage_cap = cap;
    return p;
}

#define TAKE(var, type, bytes)                                                   \
    type var = (type)vp_ws_take(ctx, (bytes));                                   \
    if (!var) return vp_fail(ctx, VP_ERR_NOMEM, "workspace exhausted: " #var)

static int h2d(vp_ctx* ctx, void* d, const void* h, size_t n)
{
    VP_HIP(ctx, hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, ctx->stream));
    return VP_OK;
}
static int d2h(vp_ctx* ctx, void* h, const void* d, size_t n)
{
    VP_HIP(ctx, hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, ctx->stream));
    return VP_OK;
}
static int h2d_rows(vp_ctx* ctx, void* d, size_t dpitch, const void* h, size_t spitch, size_t rowbytes, size_t rows)
{
    if (spitch == rowbytes && dpitch == rowbytes) return h2d(ctx, d, h, rowbytes * rows);
    VP_HIP(ctx, hipMemcpy2DAsync(d, dpitch, h, spitch, rowbytes, rows, hipMemcpyHostToDevice, ctx->stream));
    return VP_OK;
}
#define VP_TRY(x) do { int rc__ = (x); if (rc__ != VP_OK) return rc__; } while (0)

static int check_ctx(vp_ctx* ctx)
{
    if (!ctx) return VP_ERR_INVALID;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return vp_fail(ctx, VP_ERR_HIP, "hipSetDevice", e);
    return VP_OK;
}

// cv2.inRange bound normalisation (arithm.cpp): empty when lo > hi, lo > 255 or hi < 0
static void norm_range(int cn, const int32_t* lo, const int32_t* hi, vp_range3* q)
{
    for (int c = 0; c < 3; c++) {
        if (c >= cn) { q->lo[c] = 0; q->hi[c] = 255; continue; }
        int l = lo[c], u = hi[c];
        if (l > u || l > 255 || u < 0) { l = 1; u = 0; }
        else { if (l < 0) l = 0; if (u > 255) u = 255; }
        q->lo[c] = l;
        q->hi[c] = u;
    }
}

// ---- morphology planning ---------------------------------------------------------------------------

struct rect_se { int kw, kh, ax, ay; };

// Adds one rect erode/dilate to a stage list, split so that every stage has extents <= 31 (the kernels' funnel shifts).
static void push_rect_stage(std::vector<vp_bitstage>& v, int dilate, const rect_se& k)
{
    int l = k.ax, r = k.kw - 1 - k.ax, u = k.ay, d = k.kh - 1 - k.ay;
    // merge with the previous stage of the same kind (erode∘erode / dilate∘dilate with cv2's border
    // rule equal one pass with summed extents: the image is a box, clamping an intermediate sample
    // into it never increases a coordinate distance)
    if (!v.empty() && v.back().dilate == dilate) {
        l += v.back().l; r += v.back().r; u += v.back().u; d += v.back().d;
        v.pop_back();
    }
    do {
        vp_bitstage s;
        s.dilate = dilate;
        s.l = l > 31 ? 31 : l; s.r = r > 31 ? 31 : r; s.u = u > 31 ? 31 : u; s.d = d > 31 ? 31 : d;
        l -= s.l; r -= s.r; u -= s.u; d -= s.d;
        v.push_back(s);
    } while (l | r | u | d);
}

// Runs a stage list over bit images, grouping stages into launches whose halo fits LDS.
// bits_a holds the input; bits_b is scratch of equal size.  The final launch writes out_bits /
// out_mask (either may be NULL).  With an empty list the input is forwarded.
static int run_bit_stages(vp_ctx* ctx, const std::vector<vp_bitstage>& st, u64* bits_a, u64* bits_b, int w, int h, int n,
                          u64* out_bits, uint8_t* out_mask)
{
    const size_t words = (size_t)n * h * vp_ww(w);
    if (st.empty()) {
        if (out_bits && out_bits != bits_a) VP_HIP(ctx, hipMemcpyAsync(out_bits, bits_a, words * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (out_mask) VP_TRY(vpk_unpack_bits(ctx, bits_a, w, h, n, out_mask));
        return VP_OK;
    }
    const size_t lds_limit = 150 * 1024;
    const int ww = vp_ww(w);
    size_t i = 0;
    u64* cur = bits_a;
    u64* other = bits_b;
    while (i < st.size()) {
        vp_bitplan plan;
        plan.n = 0;
        int halo = 0;
        while (i < st.size() && plan.n < VP_MAX_STAGES) {
            const int nh = halo + st[i].u + st[i].d;
            const size_t lds = (size_t)2 * (32 + nh) * ww * 8;
            if (plan.n > 0 && lds > lds_limit) break;
            if (plan.n == 0 && lds > lds_limit) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "image too wide for the LDS bit-morphology strip");
            plan.s[plan.n++] = st[i++];
            halo = nh;
        }
        const bool last = i == st.size();
        u64* dst_bits = last ? out_bits : other;
        VP_TRY(vpk_morph_bits(ctx, plan, cur, w, h, n, dst_bits, last ? out_mask : nullptr));
        if (!last) { u64* t = cur; cur = other; other = t; }
    }
    return VP_OK;
}

static int stages_for_op(std::vector<vp_bitstage>& v, int op, const rect_se& k)
{
    switch (op) {
        case VP_MORPH_ERODE: push_rect_stage(v, 0, k); break;
        case VP_MORPH_DILATE: push_rect_stage(v, 1, k); break;
        case VP_MORPH_OPEN: push_rect_stage(v, 0, k); push_rect_stage(v, 1, k); break;
        case VP_MORPH_CLOSE: push_rect_stage(v, 1, k); push_rect_stage(v, 0, k); break;
        default: return VP_ERR_INVALID;
    }
    return VP_OK;
}

// cv2 morphOp() normalisation of (kernel, anchor, iterations).  Returns 1 when the op degenerates to a copy.
struct norm_se { std::vector<uint8_t> k; int kw, kh, ax, ay, iterations; bool allones; };
static int normalise_se(const uint8_t* kernel, int kw, int kh, int ax, int ay, int iterations, norm_se* o)
{
    if (iterations < 0) return VP_ERR_INVALID;
    if (!kernel || kw * kh == 0) {
        kw = kh = 1 + iterations * 2;
        ax = ay = iterations;
        iterations = 1;
        o->k.assign((size_t)kw * kh, 1);
    } else {
        if (kw <= 0 || kh <= 0) return VP_ERR_INVALID;
        o->k.assign(kernel, kernel + (size_t)kw * kh);
    }
    if (ax < 0) ax = kw / 2;
    if (ay < 0) ay = kh / 2;
    if (ax >= kw || ay >= kh) return VP_ERR_INVALID;
    bool allones = true;
    for (uint8_t b : o->k) allones = allones && b != 0;
    if (iterations > 1 && allones) {
        ax *= iterations;
        ay *= iterations;
        kw = kw + (iterations - 1) * (kw - 1);
        kh = kh + (iterations - 1) * (kh - 1);
        iterations = 1;
        o->k.assign((size_t)kw * kh, 1);
    }
    o->kw = kw; o->kh = kh; o->ax = ax; o->ay = ay; o->iterations = iterations; o->allones = allones;
    return VP_OK;
}

extern "C" {

int vp_structuring_element(int shape, int kw, int kh, uint8_t* out)
{
    // imgproc getStructuringElement(): integer geometry, anchor at the centre
    if (!out || kw <= 0 || kh <= 0 || shape < 0 || shape > 2) return VP_ERR_INVALID;
    if (kw == 1 && kh == 1) shape = VP_SHAPE_RECT;
    const int r = kh / 2, c = kw / 2;
    const double inv_r2 = (shape == VP_SHAPE_ELLIPSE && r) ? 1.0 / ((double)r * r) : 0.0;
    for (int i = 0; i < kh; i++) {
        int j1 = 0, j2 = 0;
        if (shape == VP_SHAPE_RECT || (shape == VP_SHAPE_CROSS && i == r)) j2 = kw;
        else if (shape == VP_SHAPE_CROSS) { j1 = c; j2 = c + 1; }
        else {
            const int dy = i - r;
            if (abs(dy) <= r) {
                const int dx = (int)__builtin_nearbyint(c * __builtin_sqrt((r * r - dy * dy) * inv_r2));
                j1 = c - dx > 0 ? c - dx : 0;
                j2 = c + dx + 1 < kw ? c + dx + 1 : kw;
            }
        }
        for (int j = 0; j < kw; j++) out[i * kw + j] = (j >= j1 && j < j2) ? 1 : 0;
    }
    return VP_OK;
}

int vp_cvt_color_u8(vp_ctx* ctx, int code, const uint8_t* src, size_t src_stride, int w, int h, uint8_t* dst_i,
                    uint8_t* const* planes)
{
    VP_TRY(check_ctx(ctx));
    if (!src || w <= 0 || h <= 0 || h > 65535) return vp_fail(ctx, VP_ERR_INVALID, "vp_cvt_color_u8 arguments");
    if (code < VP_BGR2LAB || code > VP_BGR2HLS) return vp_fail(ctx, VP_ERR_INVALID, "conversion code");
    const int scn = code == VP_GRAY2BGR ? 1 : 3, dcn = code == VP_BGR2GRAY ? 1 : 3;
    if (src_stride < (size_t)w * scn) return vp_fail(ctx, VP_ERR_INVALID, "src_stride");
    const size_t npx = (size_t)w * h;
    uint8_t* hp[3] = {nullptr, nullptr, nullptr};
    if (planes)
        for (int c = 0; c < dcn; c++) hp[c] = planes[c];
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx * scn) + vp_align(npx * dcn) + 3 * vp_align(npx) + 4096));
    TAKE(d_src, uint8_t*, npx * scn);
    TAKE(d_dst, uint8_t*, npx * dcn);
    uint8_t* dp[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < 3; c++)
        if (hp[c]) { dp[c] = (uint8_t*)vp_ws_take(ctx, npx); if (!dp[c]) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
    VP_TRY(h2d_rows(ctx, d_src, (size_t)w * scn, src, src_stride, (size_t)w * scn, h));
    if (code == VP_HSV2BGR) {
        if (planes) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "HSV2BGR: split planes");
        VP_TRY(vpk_hsv2bgr(ctx, d_src, npx, d_dst));
    } else {
        VP_TRY(vpk_cvt_color(ctx, code, d_src, (size_t)w * scn, w, h, dst_i ? d_dst : nullptr, dp[0], dp[1], dp[2]));
    }
    if (dst_i) VP_TRY(d2h(ctx, dst_i, d_dst, npx * dcn));
    for (int c = 0; c < 3; c++)
        if (hp[c]) VP_TRY(d2h(ctx, hp[c], dp[c], npx));
    return vp_synchronize(ctx);
}

int vp_color_balance_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, int flags, int hblocks, int vblocks, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || hblocks <= 0 || vblocks <= 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_color_balance_u8 arguments");
    const size_t npx = (size_t)w * h;
    const size_t tiles = (flags & VP_CB_EQUALIZE_RGB) ? (size_t)hblocks * vblocks : 1;
    if (tiles > 1024) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "colour balance: too many tiles");
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx * 3) + vp_balance_ws_bytes(1, (int)tiles) + 4096));
    TAKE(d_img, uint8_t*, npx * 3);
    VP_TRY(h2d(ctx, d_img, src, npx * 3));
    VP_TRY(vpk_color_balance(ctx, d_img, d_img, w, h, 1, flags, hblocks, vblocks));
    VP_TRY(d2h(ctx, dst, d_img, npx * 3));
    return vp_synchronize(ctx);
}

int vp_color_balance_last_folds(vp_ctx* ctx, int32_t* tiles_folded)
{
    VP_TRY(check_ctx(ctx));
    if (!tiles_folded) return vp_fail(ctx, VP_ERR_INVALID, "vp_color_balance_last_folds arguments");
    *tiles_folded = 0;
    if (!ctx->cb_folds_dev) return VP_OK;
    uint32_t v = 0;
    VP_HIP(ctx, hipMemcpyAsync(&v, ctx->cb_folds_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *tiles_folded = (int32_t)v;
    return VP_OK;
}

int vp_color_balance_dev(vp_ctx* ctx, const uint8_t* src, uint8_t* dst, int w, int h, int n, int flags, int hblocks, int vblocks)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || n <= 0 || hblocks <= 0 || vblocks <= 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_color_balance_dev arguments");
    const size_t tiles = (flags & VP_CB_EQUALIZE_RGB) ? (size_t)hblocks * vblocks : 1;
    if (tiles > 1024) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "colour balance: too many tiles");
    VP_TRY(vp_ws_reserve(ctx, vp_balance_ws_bytes(n, (int)tiles) + 4096));
    return vpk_color_balance(ctx, src, dst, w, h, n, flags, hblocks, vblocks);
}

int vp_cvt_bgr2lab_f32(vp_ctx* ctx, const float* src, int w, int h, float* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_cvt_bgr2lab_f32 arguments");
    const size_t npx = (size_t)w * h;
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(npx * 12) + 1024));
    TAKE(d_src, float*, npx * 12);
    TAKE(d_dst, float*, npx * 12);
    VP_TRY(h2d(ctx, d_src, src, npx * 12));
    VP_TRY(vpk_bgr2lab_f32(ctx, d_src, npx, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, npx * 12));
    return vp_synchronize(ctx);
}

int vp_order_stats_f32(vp_ctx* ctx, const float* src, size_t n, size_t k, float* v_k, float* v_k1)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !v_k || n == 0 || k >= n) return vp_fail(ctx, VP_ERR_INVALID, "vp_order_stats_f32 arguments");
    VP_TRY(vp_ws_reserve(ctx, vp_align(n * 4) + 4096));
    TAKE(d_src, float*, n * 4);
    TAKE(d_hist, u32*, 1024);
    VP_TRY(h2d(ctx, d_src, src, n * 4));
    VP_TRY(vpk_kth_f32(ctx, d_src, n, k, d_hist, v_k));
    if (v_k1) VP_TRY(vpk_kth_f32(ctx, d_src, n, k + 1 < n ? k + 1 : n - 1, d_hist, v_k1));
    return VP_OK;
}

int vp_inrange_u8(vp_ctx* ctx, const uint8_t* src, size_t src_stride, int w, int h, int cn, const int32_t* lo, const int32_t* hi,
                  uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || !lo || !hi || w <= 0 || h <= 0 || h > 65535 || (cn != 1 && cn != 3) || src_stride < (size_t)w * cn)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_inrange_u8 arguments");
    vp_range3 q;
    norm_range(cn, lo, hi, &q);
    const size_t npx = (size_t)w * h;
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx * cn) + vp_align(npx) + 1024));
    TAKE(d_src, uint8_t*, npx * cn);
    TAKE(d_dst, uint8_t*, npx);
    VP_TRY(h2d_rows(ctx, d_src, (size_t)w * cn, src, src_stride, (size_t)w * cn, h));
    VP_TRY(vpk_inrange_u8(ctx, d_src, (size_t)w * cn, w, h, cn, q, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, npx));
    return vp_synchronize(ctx);
}

int vp_inrange_f32(vp_ctx* ctx, const float* src, size_t src_stride_bytes, int w, int h, float lo, float hi, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || h > 65535 || src_stride_bytes < (size_t)w * 4)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_inrange_f32 arguments");
    const size_t npx = (size_t)w * h;
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx * 4) + vp_align(npx) + 1024));
    TAKE(d_src, float*, npx * 4);
    TAKE(d_dst, uint8_t*, npx);
    VP_TRY(h2d_rows(ctx, d_src, (size_t)w * 4, src, src_stride_bytes, (size_t)w * 4, h));
    VP_TRY(vpk_inrange_f32(ctx, d_src, (size_t)w * 4, w, h, lo, hi, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, npx));
    return vp_synchronize(ctx);
}

int vp_color_distance_u8(vp_ctx* ctx, const uint8_t* const* planes, int w, int h, const float* color, const float* wts, int skipmask,
                         float* dist2_out, uint8_t* sqrt_out)
{
    VP_TRY(check_ctx(ctx));
    if (!planes || !color || !wts || w <= 0 || h <= 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_color_distance_u8 arguments");
    for (int c = 0; c < 3; c++)
        if (!(skipmask & (1 << c)) && !planes[c]) return vp_fail(ctx, VP_ERR_INVALID, "missing plane");
    const size_t npx = (size_t)w * h;
    VP_TRY(vp_ws_reserve(ctx, 3 * vp_align(npx) + vp_align(npx * 4) + vp_align(npx) + 2048));
    uint8_t* dp[3] = {nullptr, nullptr, nullptr};
    for (int c = 0; c < 3; c++) {
        if (skipmask & (1 << c)) continue;
        dp[c] = (uint8_t*)vp_ws_take(ctx, npx);
        if (!dp[c]) return vp_fail(ctx, VP_ERR_NOMEM, "workspace");
        VP_TRY(h2d(ctx, dp[c], planes[c], npx));
    }
    TAKE(d_d2, float*, npx * 4);
    TAKE(d_sq, uint8_t*, npx);
    VP_TRY(vpk_color_distance(ctx, dp[0], dp[1], dp[2], npx, color, wts, skipmask, d_d2, d_sq));
    if (dist2_out) VP_TRY(d2h(ctx, dist2_out, d_d2, npx * 4));
    if (sqrt_out) VP_TRY(d2h(ctx, sqrt_out, d_sq, npx));
    return vp_synchronize(ctx);
}

// one erode or dilate (after cv2 normalisation) on a device image; result in d_out
static int morph_basic_dev(vp_ctx* ctx, int dilate, const norm_se& se, const uint8_t* d_in, int w, int h, int cn, bool binary,
                           uint8_t* d_out, uint8_t* d_tmp, u64* bits_a, u64* bits_b, int16_t* d_offs, uint8_t* d_tab = nullptr)
{
    const size_t nbytes = (size_t)w * h * cn;
    if (se.iterations == 0 || se.kw * se.kh == 1) {
        VP_HIP(ctx, hipMemcpyAsync(d_out, d_in, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
        return VP_OK;
    }
    if (se.allones && cn == 1 && binary) {
        std::vector<vp_bitstage> st;
        rect_se k = {se.kw, se.kh, se.ax, se.ay};
        push_rect_stage(st, dilate, k);
        VP_TRY(vpk_pack_bits(ctx, d_in, (size_t)w, w, h, 1, bits_a, nullptr));
        return run_bit_stages(ctx, st, bits_a, bits_b, w, h, 1, nullptr, d_out);
    }
    // generic: offsets of the structuring element (all-ones kernels are applied separably)
    std::vector<int16_t> offs;
    int passes_first = 0;
    if (se.allones) {
        for (int j = 0; j < se.kw; j++) { offs.push_back((int16_t)(j - se.ax)); offs.push_back(0); }
        passes_first = se.kw;
        for (int i = 0; i < se.kh; i++) { offs.push_back(0); offs.push_back((int16_t)(i - se.ay)); }
    } else {
        for (int i = 0; i < se.kh; i++)
            for (int j = 0; j < se.kw; j++)
                if (se.k[(size_t)i * se.kw + j]) { offs.push_back((int16_t)(j - se.ax)); offs.push_back((int16_t)(i - se.ay)); }
    }
    VP_HIP(ctx, hipMemcpyAsync(d_offs, offs.data(), offs.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // offs is a local vector
    if (se.allones) {
        VP_TRY(vpk_morph_generic(ctx, dilate, d_in, w, h, cn, d_offs, passes_first, d_tmp));
        VP_TRY(vpk_morph_generic(ctx, dilate, d_tmp, w, h, cn, d_offs + 2 * passes_first, se.kh, d_out));
        return VP_OK;
    }
    const int noffs = (int)(offs.size() / 2);
    // span form when it saves reads: one (dy, x0, x1) triple per run of members in a row of the element
    std::vector<int16_t> spans;
    int max_len = 1;
    for (int i = 0; i < se.kh; i++)
        for (int j = 0; j < se.kw;) {
            if (!se.k[(size_t)i * se.kw + j]) { j++; continue; }
            int e = j;
            while (e + 1 < se.kw && se.k[(size_t)i * se.kw + e + 1]) e++;
            spans.push_back((int16_t)(i - se.ay)); spans.push_back((int16_t)(j - se.ax)); spans.push_back((int16_t)(e - se.ax));
            max_len = std::max(max_len, e - j + 1);
            j = e + 1;
        }
    const int nspans = (int)(spans.size() / 3);
    const bool use_spans = d_tab && 2 * nspans < noffs && max_len <= 255 && nspans <= 2048 && (size_t)w * cn <= 16384;
    if (use_spans) {
        VP_HIP(ctx, hipMemcpyAsync(d_offs, spans.data(), spans.size() * 2, hipMemcpyHostToDevice, ctx->stream));
        VP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // spans is a local vector
    }
    const uint8_t* cur = d_in;
    uint8_t* bufs[2] = {d_out, d_tmp};
    // arrange so that the last pass lands in d_out
    int which = (se.iterations % 2 == 1) ? 0 : 1;
    for (int it = 0; it < se.iterations; it++) {
        if (use_spans) VP_TRY(vpk_morph_spans(ctx, dilate, cur, w, h, cn, d_offs, nspans, max_len, d_tab, bufs[which]));
        else VP_TRY(vpk_morph_generic(ctx, dilate, cur, w, h, cn, d_offs, noffs, bufs[which]));
        cur = bufs[which];
        which ^= 1;
    }
    return VP_OK;
}

// workspace a morphology call needs besides its source / result images
static size_t morph_ws_bytes(const norm_se& se, int w, int h, int cn)
{
    const size_t nbytes = (size_t)w * h * cn;
    const size_t bitbytes = (size_t)h * vp_ww(w) * 8;
    const size_t offbytes = ((size_t)se.kw * se.kh + se.kw + se.kh) * 4 + 64;
    const size_t tabbytes = se.allones ? 0 : 7 * nbytes;   // running min/max tables of the span form
    return 4 * vp_align(nbytes) + 2 * vp_align(bitbytes) + vp_align(offbytes) + vp_align(tabbytes) + 4096;
}

// One morphology operation between device images (d_dst may equal neither d_src nor overlap it); temporaries are carved from the
// workspace, which the caller has reserved (morph_ws_bytes).  binary_hint: 1 = the image is known to hold only 0 / 255 (a mask this
// library produced), 0 = unknown: one flag comes back from the device to decide between the bit-plane and the grey-level path.
static int morph_core(vp_ctx* ctx, int op, const norm_se& se, const uint8_t* d_src, int w, int h, int cn, int binary_hint, uint8_t* d_dst)
{
    const size_t nbytes = (size_t)w * h * cn;
    const size_t bitbytes = (size_t)h * vp_ww(w) * 8;
    const size_t offbytes = ((size_t)se.kw * se.kh + se.kw + se.kh) * 4 + 64;
    const size_t tabbytes = se.allones ? 0 : 7 * nbytes;
    uint8_t* d_tab = tabbytes ? (uint8_t*)vp_ws_take(ctx, tabbytes) : nullptr;
    if (tabbytes && !d_tab) return vp_fail(ctx, VP_ERR_NOMEM, "workspace");
    TAKE(d_b, uint8_t*, nbytes);
    TAKE(d_c, uint8_t*, nbytes);
    TAKE(d_tmp, uint8_t*, nbytes);
    TAKE(bits_a, u64*, bitbytes);
    TAKE(bits_b, u64*, bitbytes);
    TAKE(d_offs, int16_t*, offbytes);
    TAKE(d_flag, int*, 4);
    bool binary = false;
    if (cn == 1 && se.allones) {
        // a 0/255 mask can take the bit-plane path; anything else is grey-level
        if (binary_hint == 1) {
            VP_TRY(vpk_pack_bits(ctx, d_src, (size_t)w, w, h, 1, bits_a, nullptr));
            binary = true;
        } else {
            VP_HIP(ctx, hipMemsetAsync(d_flag, 0, 4, ctx->stream));
            VP_TRY(vpk_pack_bits(ctx, d_src, (size_t)w, w, h, 1, bits_a, d_flag));
            int flag = 1;
            VP_TRY(d2h(ctx, &flag, d_flag, 4));
            VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            binary = flag == 0;
        }
    }
    if (binary && op != VP_MORPH_GRADIENT) {
        // whole op (incl. OPEN/CLOSE) as one fused bit-plane launch
        std::vector<vp_bitstage> st;
        rect_se k = {se.kw, se.kh, se.ax, se.ay};
        if (!(se.iterations == 0 || se.kw * se.kh == 1)) stages_for_op(st, op, k);
        VP_TRY(run_bit_stages(ctx, st, bits_a, bits_b, w, h, 1, nullptr, d_dst));
    } else if (op == VP_MORPH_ERODE || op == VP_MORPH_DILATE) {
        VP_TRY(morph_basic_dev(ctx, op == VP_MORPH_DILATE, se, d_src, w, h, cn, binary, d_dst, d_tmp, bits_a, bits_b, d_offs, d_tab));
    } else if (op == VP_MORPH_OPEN || op == VP_MORPH_CLOSE) {
        const int first = op == VP_MORPH_CLOSE;
        VP_TRY(morph_basic_dev(ctx, first, se, d_src, w, h, cn, binary, d_b, d_tmp, bits_a, bits_b, d_offs, d_tab));
        VP_TRY(morph_basic_dev(ctx, !first, se, d_b, w, h, cn, binary, d_dst, d_tmp, bits_a, bits_b, d_offs, d_tab));
    } else {  // GRADIENT = dilate - erode
        VP_TRY(morph_basic_dev(ctx, 1, se, d_src, w, h, cn, binary, d_b, d_tmp, bits_a, bits_b, d_offs, d_tab));
        VP_TRY(morph_basic_dev(ctx, 0, se, d_src, w, h, cn, binary, d_c, d_tmp, bits_a, bits_b, d_offs, d_tab));
        VP_TRY(vpk_absdiff_sub_u8(ctx, d_b, d_c, nbytes, d_dst));
    }
    return VP_OK;
}

int vp_morph_u8(vp_ctx* ctx, int op, const uint8_t* src, int w, int h, int cn, const uint8_t* kernel, int kw, int kh, int ax, int ay,
                int iterations, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || h > 65535 || cn < 1 || cn > 4 || op < VP_MORPH_ERODE || op > VP_MORPH_GRADIENT)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_morph_u8 arguments");
    norm_se se;
    if (normalise_se(kernel, kw, kh, ax, ay, iterations, &se) != VP_OK) return vp_fail(ctx, VP_ERR_INVALID, "structuring element");
    const size_t nbytes = (size_t)w * h * cn;
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(nbytes) + morph_ws_bytes(se, w, h, cn)));
    TAKE(d_src, uint8_t*, nbytes);
    TAKE(d_a, uint8_t*, nbytes);
    VP_TRY(h2d(ctx, d_src, src, nbytes));
    VP_TRY(morph_core(ctx, op, se, d_src, w, h, cn, 0, d_a));
    VP_TRY(d2h(ctx, dst, d_a, nbytes));
    return vp_synchronize(ctx);
}

// ---- debug overlays (host only, no device work) -----------------------------------------------------------------------------
// utils/draw.py:283-327 draw_contours / draw_polylines modify the caller's host image in place on every frame
// (modules/red_buoy.py:39).  The Python mirror's rasteriser (Bresenham steps, square brush of the requested thickness) is the
// same statement sequence here in C, because at 1080p a dozen contours of a few hundred points are 10^4 brush stamps per frame.
// utils/feature.py:240-265 contour_centroid / contour_area (cv2.moments, cv2.contourArea on an integer contour): the three Green sums
// a00 = sum(x[i-1] y[i] - x[i] y[i-1]), a10 = sum(d (x[i-1] + x[i])), a01 = sum(d (y[i-1] + y[i])) as exact integers (host code).
int vp_polygon_sums_i32(const int32_t* pts, int npts, int64_t* out3)
{
    if (!pts || !out3 || npts < 0) return VP_ERR_INVALID;
    long long a00 = 0, a10 = 0, a01 = 0;
    if (npts > 0) {
        long long xp = pts[2 * (npts - 1)], yp = pts[2 * (npts - 1) + 1];
        for (int i = 0; i < npts; i++) {
            const long long x = pts[2 * i], y = pts[2 * i + 1];
            const long long d = xp * y - x * yp;
            a00 += d; a10 += d * (xp + x); a01 += d * (yp + y);
            xp = x; yp = y;
        }
    }
    out3[0] = a00; out3[1] = a10; out3[2] = a01;
    return VP_OK;
}

// Convex hull of integer points (host code; cv2.minAreaRect of the stand-in, modules/bins.py:62): Andrew's monotone chain over the
// sorted distinct points, collinear points dropped, counter-clockwise from the lexicographically smallest point - exact in 64-bit
// integers.  out must hold npts points; returns the number of hull vertices in *nout.
int vp_convex_hull_i32(const int32_t* pts, int npts, int32_t* out, int* nout)
{
    if (!pts || !out || !nout || npts < 0) return VP_ERR_INVALID;
    std::vector<std::pair<int32_t, int32_t>> p((size_t)npts);
    for (int i = 0; i < npts; i++) p[(size_t)i] = {pts[2 * i], pts[2 * i + 1]};
    std::sort(p.begin(), p.end());
    p.erase(std::unique(p.begin(), p.end()), p.end());
    const int n = (int)p.size();
    if (n <= 2) {
        for (int i = 0; i < n; i++) { out[2 * i] = p[(size_t)i].first; out[2 * i + 1] = p[(size_t)i].second; }
        *nout = n;
        return VP_OK;
    }
    auto cross = [](const std::pair<int32_t, int32_t>& o, const std::pair<int32_t, int32_t>& a, const std::pair<int32_t, int32_t>& b) {
        return ((long long)a.first - o.first) * ((long long)b.second - o.second) - ((long long)a.second - o.second) * ((long long)b.first - o.first);
    };
    std::vector<std::pair<int32_t, int32_t>> lower, upper;
    for (int i = 0; i < n; i++) {
        while (lower.size() >= 2 && cross(lower[lower.size() - 2], lower.back(), p[(size_t)i]) <= 0) lower.pop_back();
        lower.push_back(p[(size_t)i]);
    }
    for (int i = n - 1; i >= 0; i--) {
        while (upper.size() >= 2 && cross(upper[upper.size() - 2], upper.back(), p[(size_t)i]) <= 0) upper.pop_back();
        upper.push_back(p[(size_t)i]);
    }
    int k = 0;
    for (size_t i = 0; i + 1 < lower.size(); i++, k++) { out[2 * k] = lower[i].first; out[2 * k + 1] = lower[i].second; }
    for (size_t i = 0; i + 1 < upper.size(); i++, k++) { out[2 * k] = upper[i].first; out[2 * k + 1] = upper[i].second; }
    *nout = k;
    return VP_OK;
}

// cv2.minAreaRect of the stand-in for integer points (host code; modules/bins.py:62 calls it for every contour): rotating calipers
// over the hull above - for every hull edge the extent of the hull along and across it, the first edge of smallest area wins -
// in the doubles of the Python statements (vision/cv2_facade.py _min_area_rect_loop; edge lengths as sqrt of an exact integer).
// out5 = cx, cy, width, height, angle in degrees (OpenCV >= 4.5.1 convention: angle in (0, 90]), already rounded to float.
int vp_min_area_rect_i32(const int32_t* pts, int npts, float* out5)
{
    if (!pts || !out5 || npts < 0) return VP_ERR_INVALID;
    std::vector<int32_t> hull((size_t)std::max(npts, 1) * 2);
    int n = 0;
    if (vp_convex_hull_i32(pts, npts, hull.data(), &n) != VP_OK) return VP_ERR_INVALID;
    if (n == 0) { for (int i = 0; i < 5; i++) out5[i] = 0.f; return VP_OK; }
    if (n == 1) { out5[0] = (float)hull[0]; out5[1] = (float)hull[1]; out5[2] = out5[3] = 0.f; out5[4] = 90.f; return VP_OK; }
    bool have = false;
    double best_area = 0, bcx = 0, bcy = 0, bwd = 0, bht = 0, bang = 0;
    const int edges = n > 2 ? n : 1;
    for (int i = 0; i < edges; i++) {
        const int i1 = (i + 1) % n;
        const double ex = (double)hull[2 * i1] - (double)hull[2 * i], ey = (double)hull[2 * i1 + 1] - (double)hull[2 * i + 1];
        const double ln = sqrt(ex * ex + ey * ey);
        if (ln == 0) continue;
        const double ux = ex / ln, uy = ey / ln;
        double amax = 0, amin = 0, bmax = 0, bmin = 0;
        for (int k = 0; k < n; k++) {
            const double hx = (double)hull[2 * k], hy = (double)hull[2 * k + 1];
            const double a = hx * ux + hy * uy, b = -hx * uy + hy * ux;
            if (k == 0) { amax = amin = a; bmax = bmin = b; }
            else { amax = std::max(amax, a); amin = std::min(amin, a); bmax = std::max(bmax, b); bmin = std::min(bmin, b); }
        }
        const double wd = amax - amin, ht = bmax - bmin;
        if (!have || wd * ht < best_area) {
            const double ca = (amax + amin) / 2, cb = (bmax + bmin) / 2;
            have = true;
            best_area = wd * ht;
            bcx = ca * ux - cb * uy; bcy = ca * uy + cb * ux; bwd = wd; bht = ht;
            bang = atan2(uy, ux) * (180.0 / 3.141592653589793);
        }
    }
    if (!have) { out5[0] = (float)hull[0]; out5[1] = (float)hull[1]; out5[2] = out5[3] = 0.f; out5[4] = 90.f; return VP_OK; }
    while (bang <= 0) { bang += 90; std::swap(bwd, bht); }
    while (bang > 90) { bang -= 90; std::swap(bwd, bht); }
    out5[0] = (float)bcx; out5[1] = (float)bcy; out5[2] = (float)bwd; out5[3] = (float)bht; out5[4] = (float)bang;
    return VP_OK;
}

// counts[k] points per polyline, back to back in pts; one call draws them all (a frame's contours).
// All stamps carry one colour, so the image is "colour wherever some stamp covers": the stamps are collected in a coverage bit plane
// (one bit per pixel, 259 KB at 1080p, per thread, left zeroed) and the image is written once, row by row, run by run.
int vp_draw_polylines_u8(uint8_t* img, size_t stride, int w, int h, int cn, const int32_t* pts, const int32_t* counts, int npolys, int closed,
                         const uint8_t* color, int thickness)
{
    if (!img || !pts || !counts || !color || w <= 0 || h <= 0 || cn < 1 || cn > 4 || npolys < 0 || stride < (size_t)w * cn) return VP_ERR_INVALID;
    if (thickness < 1) thickness = 1;
    const int r0 = (thickness - 1) / 2, r1 = thickness / 2;
    const int ww = (w + 63) >> 6;
    static thread_local std::vector<uint64_t> cover;
    if (cover.size() < (size_t)ww * h) cover.assign((size_t)ww * h, 0);
    uint64_t* cv = cover.data();
    int ylo = h, yhi = -1, wlo = ww, whi = -1;              // rows / words touched
    auto span = [&](int y, int xa, int xb) {               // bits [xa, xb) of row y; the caller has clipped y
        xa = std::max(xa, 0); xb = std::min(xb, w);
        if (xa >= xb) return;
        uint64_t* row = cv + (size_t)y * ww;
        const int wa = xa >> 6, wb = (xb - 1) >> 6;
        const uint64_t ma = ~0ull << (xa & 63), mb = ~0ull >> (63 - ((xb - 1) & 63));
        if (wa == wb) row[wa] |= ma & mb;
        else { row[wa] |= ma; for (int k = wa + 1; k < wb; k++) row[k] = ~0ull; row[wb] |= mb; }
        wlo = std::min(wlo, wa); whi = std::max(whi, wb);
    };
    auto fill = [&](int xa, int xb, int ya, int yb) {      // [xa, xb) x [ya, yb), clipped
        ya = std::max(ya, 0); yb = std::min(yb, h);
        if (ya >= yb || xb <= 0 || xa >= w) return;
        ylo = std::min(ylo, ya); yhi = std::max(yhi, yb - 1);
        for (int yy = ya; yy < yb; yy++) span(yy, xa, xb);
    };
    auto column = [&](int x, int ya, int yb) {             // one pixel wide: the strip a horizontal step adds
        ya = std::max(ya, 0); yb = std::min(yb, h);
        if (ya >= yb || x < 0 || x >= w) return;
        ylo = std::min(ylo, ya); yhi = std::max(yhi, yb - 1);
        const int k = x >> 6;
        wlo = std::min(wlo, k); whi = std::max(whi, k);
        const uint64_t bit = 1ull << (x & 63);
        uint64_t* q = cv + (size_t)ya * ww + k;
        for (int yy = ya; yy < yb; yy++, q += ww) *q |= bit;
    };
    // The brush is a square stamped at every Bresenham step.  A step moves by at most one pixel per axis, so the square at the new
    // position adds one column and / or one row to what the previous stamp covered: only that strip is marked.
    bool have = false;
    int lx = 0, ly = 0;
    auto stamp = [&](int x, int y) {
        if (have && x == lx && y == ly) return;
        if (have && abs(x - lx) <= 1 && abs(y - ly) <= 1) {
            if (x != lx) { const int cx = x > lx ? x + r1 : x - r0; column(cx, y - r0, y + r1 + 1); }
            if (y != ly) { const int cy = y > ly ? y + r1 : y - r0; fill(x - r0, x + r1 + 1, cy, cy + 1); }
        } else {
            fill(x - r0, x + r1 + 1, y - r0, y + r1 + 1);
        }
        have = true; lx = x; ly = y;
    };
    auto line = [&](int x0, int y0, int x1, int y1) {
        if (y0 == y1 && abs(x1 - x0) > 2) {                 // a horizontal run (straight stretches of a simplified contour): one box
            fill(std::min(x0, x1) - r0, std::max(x0, x1) + r1 + 1, y0 - r0, y0 + r1 + 1);
            have = true; lx = x1; ly = y1;
            return;
        }
        const int dx = abs(x1 - x0), dy = -abs(y1 - y0);
        const int sx = x0 < x1 ? 1 : -1, sy = y0 < y1 ? 1 : -1;
        long long err = (long long)dx + dy;
        for (;;) {
            stamp(x0, y0);
            if (x0 == x1 && y0 == y1) break;
            const long long e2 = 2 * err;
            if (e2 >= dy) { err += dy; x0 += sx; }
            if (e2 <= dx) { err += dx; y0 += sy; }
        }
    };
    size_t o = 0;
    int rc = VP_OK;
    for (int k = 0; k < npolys; k++) {
        const int npts = counts[k];
        if (npts < 0) { rc = VP_ERR_INVALID; break; }
        const int32_t* p = pts + 2 * o;
        o += (size_t)npts;
        have = false;
        if (npts == 0) continue;
        if (npts == 1) { line(p[0], p[1], p[0], p[1]); continue; }
        const int last = closed ? npts : npts - 1;
        for (int i = 0; i < last; i++) {
            const int j = i + 1 < npts ? i + 1 : 0;
            line(p[2 * i], p[2 * i + 1], p[2 * j], p[2 * j + 1]);
        }
    }
    // write the covered pixels, run by run, and hand the plane back zeroed (also after an error)
    const uint8_t c0 = color[0], c1 = color[cn > 1 ? 1 : 0], c2 = color[cn > 2 ? 2 : 0];
    // The caller's image has usually just been written by a 6 MB copy and is not in the core's cache: every run below would wait for
    // its line.  The plane says which lines those are, so they are requested some rows ahead of the writes (MI355X host, one frame's
    // contours at 1080p, thickness 10: 115 -> 57 us; stamping strips straight into the image: 93 us).
    auto prefetch_row = [&](int y) {
        if (y > yhi) return;
        const uint64_t* row = cv + (size_t)y * ww;
        const uint8_t* out = img + (size_t)y * stride;
        for (int k = wlo; k <= whi; k++) {
            uint64_t m = row[k];
            if (!m) continue;
            const int a = __builtin_ctzll(m), b = 63 - __builtin_clzll(m);
            const uint8_t* q0 = out + ((size_t)k * 64 + a) * cn;
            const uint8_t* q1 = out + ((size_t)k * 64 + b) * cn + cn - 1;
            for (const uint8_t* q = (const uint8_t*)((uintptr_t)q0 & ~(uintptr_t)63); q <= q1; q += 64) __builtin_prefetch(q, 1, 3);
        }
    };
    for (int y = ylo; y < ylo + 16; y++) prefetch_row(y);
    for (int y = ylo; y <= yhi; y++) {
        prefetch_row(y + 16);
        uint64_t* row = cv + (size_t)y * ww;
        uint8_t* out = img + (size_t)y * stride;
        for (int k = wlo; k <= whi; k++) {
            uint64_t m = row[k];
            if (!m) continue;
            row[k] = 0;
            if (rc != VP_OK) continue;
            while (m) {
                const int a = __builtin_ctzll(m);
                const uint64_t rest = ~(m >> a);            // first zero above a = end of the run
                const int len = rest ? __builtin_ctzll(rest) : 64 - a;
                uint8_t* q = out + ((size_t)k * 64 + a) * cn;
                if (cn == 3) for (int i = 0; i < len; i++, q += 3) { q[0] = c0; q[1] = c1; q[2] = c2; }
                else if (cn == 1) memset(q, c0, (size_t)len);
                else for (int i = 0; i < len; i++, q += cn) memcpy(q, color, (size_t)cn);
                if (a + len >= 64) break;
                m &= ~0ull << (a + len);
            }
        }
    }
    return rc;
}



int vp_draw_polyline_u8(uint8_t* img, size_t stride, int w, int h, int cn, const int32_t* pts, int npts, int closed, const uint8_t* color,
                        int thickness)
{
    const int32_t cnt = npts;
    return vp_draw_polylines_u8(img, stride, w, h, cn, pts, &cnt, 1, closed, color, thickness);
}

// ---- device-resident forms of the per-operator entry points ---------------------------------------------------------------------
// Same arithmetic, same argument meaning; images are device pointers (packed rows unless a stride is taken), nothing is copied
// and nothing is synchronised: the call enqueues on the context's stream and returns.  They let the Python mirror keep the
// intermediate images of a module's process() in HBM between operator calls (modules/red_buoy.py:21-38: the Lab image, its
// planes, the threshold mask and both cleaned masks never need to visit the host).

int vp_cvt_color_dev(vp_ctx* ctx, int code, const uint8_t* d_src, size_t src_stride, int w, int h, uint8_t* d_dst, uint8_t* const* d_planes)
{
    VP_TRY(check_ctx(ctx));
    if (!d_src || w <= 0 || h <= 0 || h > 65535) return vp_fail(ctx, VP_ERR_INVALID, "vp_cvt_color_dev arguments");
    if (code < VP_BGR2LAB || code > VP_BGR2HLS) return vp_fail(ctx, VP_ERR_INVALID, "conversion code");
    const int scn = code == VP_GRAY2BGR ? 1 : 3, dcn = code == VP_BGR2GRAY ? 1 : 3;
    if (src_stride < (size_t)w * scn) return vp_fail(ctx, VP_ERR_INVALID, "src_stride");
    uint8_t* dp[3] = {nullptr, nullptr, nullptr};
    if (d_planes)
        for (int c = 0; c < dcn; c++) dp[c] = d_planes[c];
    if (code == VP_HSV2BGR) {
        if (d_planes || !d_dst || src_stride != (size_t)w * 3) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "HSV2BGR: packed rows, no split planes");
        return vpk_hsv2bgr(ctx, d_src, (size_t)w * h, d_dst);
    }
    return vpk_cvt_color(ctx, code, d_src, src_stride, w, h, d_dst, dp[0], dp[1], dp[2]);
}

int vp_inrange_u8_dev(vp_ctx* ctx, const uint8_t* d_src, size_t src_stride, int w, int h, int cn, const int32_t* lo, const int32_t* hi, uint8_t* d_dst)
{
    VP_TRY(check_ctx(ctx));
    if (!d_src || !d_dst || !lo || !hi || w <= 0 || h <= 0 || h > 65535 || (cn != 1 && cn != 3) || src_stride < (size_t)w * cn)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_inrange_u8_dev arguments");
    vp_range3 q;
    norm_range(cn, lo, hi, &q);
    return vpk_inrange_u8(ctx, d_src, src_stride, w, h, cn, q, d_dst);
}

int vp_inrange_u8_bits_dev(vp_ctx* ctx, const uint8_t* d_src, size_t src_stride, int w, int h, int cn, const int32_t* lo, const int32_t* hi, uint8_t* d_dst,
                           unsigned long long* d_bits, int* made_bits)
{
    VP_TRY(check_ctx(ctx));
    if (!d_src || !d_dst || !lo || !hi || w <= 0 || h <= 0 || h > 65535 || (cn != 1 && cn != 3) || src_stride < (size_t)w * cn)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_inrange_u8_bits_dev arguments");
    vp_range3 q;
    norm_range(cn, lo, hi, &q);
    return vpk_inrange_u8(ctx, d_src, src_stride, w, h, cn, q, d_dst, reinterpret_cast<u64*>(d_bits), made_bits);
}

// The polylines of vp_draw_polylines_u8 drawn into a packed device image (bins.py draws its rectangles into an overlay that only ever
// leaves the device when it is posted).  Points and counts are host arrays; the same pixels as the host rasteriser.
// ring of pinned chunks for small host -> device hand-overs (see vp_ctx): the chunk is the caller's until ring_done, and is handed
// out again only after everything queued on the context's stream up to ring_done has run
static uint8_t* ring_take(vp_ctx* ctx, size_t bytes, int* slot)
{
    const int s = ctx->ring_next;
    ctx->ring_next = (s + 1) & 3;
    if (ctx->ring_busy[s]) { (void)hipEventSynchronize(ctx->ring_ev[s]); ctx->ring_busy[s] = 0; }
    if (bytes > ctx->ring_cap[s]) {
        if (ctx->ring_buf[s]) { (void)hipHostFree(ctx->ring_buf[s]); ctx->ring_buf[s] = nullptr; ctx->ring_cap[s] = 0; }
        const size_t cap = (bytes + (1u << 16) - 1) >> 16 << 16;
        void* p = nullptr;
        if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        ctx->ring_buf[s] = (uint8_t*)p;
        ctx->ring_cap[s] = cap;
    }
    if (!ctx->ring_ev[s] && hipEventCreateWithFlags(&ctx->ring_ev[s], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *slot = s;
    return ctx->ring_buf[s];
}
static void ring_done(vp_ctx* ctx, int slot)
{
    if (slot < 0) return;
    if (hipEventRecord(ctx->ring_ev[slot], ctx->stream) == hipSuccess) ctx->ring_busy[slot] = 1;
    else { (void)hipGetLastError(); (void)hipStreamSynchronize(ctx->stream); }
}

int vp_draw_polylines_dev(vp_ctx* ctx, uint8_t* d_img, int w, int h, int cn, const int32_t* pts, const int32_t* counts, int npolys, int closed,
                          const uint8_t* color, int thickness)
{
    VP_TRY(check_ctx(ctx));
    if (!d_img || !pts || !counts || !color || w <= 0 || h <= 0 || cn < 1 || cn > 4 || npolys < 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_draw_polylines_dev arguments");
    if (thickness < 1) thickness = 1;
    if (thickness > 255) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "vp_draw_polylines_dev: thickness");
    // the vertices and, per vertex, the vertex it is joined to go over in one pinned chunk; the device walks the lines (k_draw_segments)
    long long total = 0;
    for (int k = 0; k < npolys; k++) {
        if (counts[k] < 0) return vp_fail(ctx, VP_ERR_INVALID, "vp_draw_polylines_dev: counts");
        total += counts[k];
    }
    if (total == 0) return VP_OK;
    if (total > (1ll << 28)) return vp_fail(ctx, VP_ERR_INVALID, "vp_draw_polylines_dev: too many points");
    const size_t N = (size_t)total;
    if (N <= 48) {                                       // a few vertices travel as kernel arguments (vpk_draw_small)
        int32_t nx[48];
        size_t o = 0;
        for (int k = 0; k < npolys; k++) {
            const size_t npts = (size_t)counts[k];
            for (size_t i = 0; i + 1 < npts; i++) nx[o + i] = (int32_t)(o + i + 1);
            if (npts) nx[o + npts - 1] = (npts == 1 || closed) ? (int32_t)o : -1;
            o += npts;
        }
        return vpk_draw_small(ctx, d_img, w, h, cn, pts, nx, (int)N, thickness, color);
    }
    int slot = -1;
    uint8_t* hp = ring_take(ctx, N * 12, &slot);
    if (!hp) return vp_fail(ctx, VP_ERR_NOMEM, "pinned staging");
    memcpy(hp, pts, N * 8);
    int32_t* nxt = reinterpret_cast<int32_t*>(hp + N * 8);
    size_t o = 0;
    for (int k = 0; k < npolys; k++) {
        const size_t npts = (size_t)counts[k];
        for (size_t i = 0; i + 1 < npts; i++) nxt[o + i] = (int32_t)(o + i + 1);
        if (npts) nxt[o + npts - 1] = (npts == 1 || closed) ? (int32_t)o : -1;
        o += npts;
    }
    int rc = vp_ws_reserve(ctx, vp_align(N * 12) + 4096);
    uint8_t* d_buf = rc == VP_OK ? (uint8_t*)vp_ws_take(ctx, N * 12) : nullptr;
    if (rc == VP_OK && !d_buf) rc = vp_fail(ctx, VP_ERR_NOMEM, "workspace exhausted: overlay vertices");
    if (rc == VP_OK) rc = h2d(ctx, d_buf, hp, N * 12);
    if (rc == VP_OK)
        rc = vpk_draw_segments(ctx, d_img, w, h, cn, reinterpret_cast<const int32_t*>(d_buf), reinterpret_cast<const int32_t*>(d_buf + N * 8), (int)N, thickness, color);
    ring_done(ctx, slot);
    return rc;
}

// cv2.addWeighted on two device images of n bytes each (modules/bins.py:20: the mask overlay); d_dst may be one of the sources
int vp_add_weighted_u8_dev(vp_ctx* ctx, const uint8_t* d_a, double alpha, const uint8_t* d_b, double beta, double gamma, size_t n, uint8_t* d_dst)
{
    VP_TRY(check_ctx(ctx));
    if (!d_a || !d_b || !d_dst || n == 0 || n > ((size_t)1 << 40)) return vp_fail(ctx, VP_ERR_INVALID, "vp_add_weighted_u8_dev arguments");
    return vpk_add_weighted_u8(ctx, d_a, d_b, n, alpha, beta, gamma, d_dst);
}

int vp_morph_u8_dev(vp_ctx* ctx, int op, const uint8_t* d_src, int w, int h, int cn, const uint8_t* kernel, int kw, int kh, int ax, int ay,
                    int iterations, int binary_hint, uint8_t* d_dst)
{
    VP_TRY(check_ctx(ctx));
    if (!d_src || !d_dst || d_src == d_dst || w <= 0 || h <= 0 || h > 65535 || cn < 1 || cn > 4 || op < VP_MORPH_ERODE || op > VP_MORPH_GRADIENT)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_morph_u8_dev arguments");
    norm_se se;
    if (normalise_se(kernel, kw, kh, ax, ay, iterations, &se) != VP_OK) return vp_fail(ctx, VP_ERR_INVALID, "structuring element");
    VP_TRY(vp_ws_reserve(ctx, morph_ws_bytes(se, w, h, cn)));
    return morph_core(ctx, op, se, d_src, w, h, cn, binary_hint, d_dst);
}

int vp_ccl_u8(vp_ctx* ctx, const uint8_t* src, size_t src_stride, int w, int h, int numbering, int32_t* labels, int32_t* stats,
              double* centroids, int max_labels, int32_t* nlabels)
{
    VP_TRY(check_ctx(ctx));
    if (!src || w <= 0 || h <= 0 || src_stride < (size_t)w || max_labels < 1 || !nlabels)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_ccl_u8 arguments");
    if (numbering != VP_CCL_BLOCK2X2 && numbering != VP_CCL_PIXEL) return vp_fail(ctx, VP_ERR_INVALID, "numbering");
    const size_t npx = (size_t)w * h;
    const size_t bitbytes = (size_t)h * vp_ww(w) * 8;
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx) + vp_align(bitbytes) + vp_align(npx * 4) + vp_align((size_t)max_labels * 20) +
                                  vp_align((size_t)max_labels * 16) + vp_ccl_ws_bytes(w, h, 1, max_labels) + 8192));
    TAKE(d_src, uint8_t*, npx);
    TAKE(d_bits, u64*, bitbytes);
    TAKE(d_labels, int32_t*, npx * 4);
    TAKE(d_stats, int32_t*, (size_t)max_labels * 20);
    TAKE(d_cent, double*, (size_t)max_labels * 16);
    TAKE(d_nl, int32_t*, 4);
    vp_ccl_ws ws;
    vp_ccl_ws_carve(ctx, w, h, 1, max_labels, &ws);
    if (!vp_ccl_ws_ok(ws)) return vp_fail(ctx, VP_ERR_NOMEM, "ccl workspace");
    VP_TRY(h2d_rows(ctx, d_src, (size_t)w, src, src_stride, (size_t)w, h));
    VP_TRY(vpk_pack_bits(ctx, d_src, (size_t)w, w, h, 1, d_bits, nullptr));
    VP_TRY(vpk_ccl(ctx, d_bits, w, h, 1, numbering, ws, labels ? d_labels : nullptr, d_stats, d_cent, max_labels, d_nl));
    VP_TRY(d2h(ctx, nlabels, d_nl, 4));
    if (labels) VP_TRY(d2h(ctx, labels, d_labels, npx * 4));
    if (stats) VP_TRY(d2h(ctx, stats, d_stats, (size_t)max_labels * 20));
    if (centroids) VP_TRY(d2h(ctx, centroids, d_cent, (size_t)max_labels * 16));
    return vp_synchronize(ctx);
}

// src: host image (uploaded) or, with src_on_device, a device image read in place; or (bits_in) its bit-packed form, already on the device
static int find_contours_impl(vp_ctx* ctx, const uint8_t* src, bool src_on_device, size_t src_stride, int w, int h, int mode, int method,
                              int32_t* points, int64_t max_points, int32_t* counts, uint8_t* is_hole, int max_contours, int32_t* n_contours,
                              int64_t* n_points, const u64* bits_in = nullptr)
{
    VP_TRY(check_ctx(ctx));
    if ((!src && !bits_in) || w <= 0 || h <= 0 || (!bits_in && src_stride < (size_t)w) || !n_contours || !n_points || max_contours < 0 || max_points < 0)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_find_contours arguments");
    const size_t npx = (size_t)w * h;
    const size_t bitbytes = (size_t)h * vp_ww(w) * 8;
    const int mc = max_contours > 0 ? max_contours : 1;
    const long long mp = max_points > 0 ? max_points : 1;
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx) + vp_align(bitbytes) + vp_contours_ws_bytes(w, h, 1, mc) + vp_align(16 + (size_t)mc * 9) +
                                  vp_align((size_t)mp * 8) + 8192));
    TAKE(d_stage, uint8_t*, npx);
    const uint8_t* d_src = d_stage;
    size_t d_stride = (size_t)w;
    TAKE(d_bits, u64*, bitbytes);
    // result header, one block so that one copy brings it back: info[2] (16 B) | counts[mc] | offsets[mc] | is_hole[mc]
    const size_t hdr_bytes = 16 + (size_t)mc * 9;
    TAKE(d_hdr, uint8_t*, hdr_bytes);
    TAKE(d_points, int32_t*, (size_t)mp * 8);
    int32_t* d_info = reinterpret_cast<int32_t*>(d_hdr);
    int32_t* d_counts = reinterpret_cast<int32_t*>(d_hdr + 16);
    int32_t* d_offsets = d_counts + mc;
    uint8_t* d_hole = reinterpret_cast<uint8_t*>(d_offsets + mc);
    const u64* bits_use = d_bits;
    if (bits_in) {
        bits_use = bits_in;                               // the caller made the bit plane with the mask (vp_inrange_u8_bits_dev): no packing launch
    } else {
        if (src_on_device) { d_src = src; d_stride = src_stride; }
        else VP_TRY(h2d_rows(ctx, d_stage, (size_t)w, src, src_stride, (size_t)w, h));
        VP_TRY(vpk_pack_bits(ctx, d_src, d_stride, w, h, 1, d_bits, nullptr));
    }
    // one block does the bookkeeping between the two follower passes - unless the last pass of this context met a speckled mask
    // (more border segments than the block's LDS tables hold: 8192): then it is launched over the chip (VP_CT_MANY=0 / 1: never / always)
    const char* many_s = getenv("VP_CT_MANY");
    const int many_env = many_s ? atoi(many_s) : -1;
    const bool many = many_env >= 0 ? many_env != 0 : ctx->ct_heads_hint > 8192u;
    // The header and the first points come back without a copy: the kernels that make them write them into the pinned staging
    // buffer as well, and the call only synchronises (longer point lists take a copy afterwards).  VP_CT_MIRROR=0: a copy, as before.
    const size_t spec_pts = points ? (size_t)std::min<long long>(mp, 8192) : 0;
    const size_t hdr_pad = vp_align(hdr_bytes);
    uint8_t* hs = (uint8_t*)vp_hstage(ctx, hdr_pad + spec_pts * 8);
    if (!hs) return vp_fail(ctx, VP_ERR_NOMEM, "pinned staging");
    static const bool mirror_off = getenv("VP_CT_MIRROR") && atoi(getenv("VP_CT_MIRROR")) == 0;
    vp_contour_mirror hm;
    hm.info = reinterpret_cast<int32_t*>(hs);
    hm.counts = reinterpret_cast<int32_t*>(hs + 16);
    hm.offsets = hm.counts + mc;
    hm.is_hole = reinterpret_cast<uint8_t*>(hm.offsets + mc);
    hm.points = spec_pts ? reinterpret_cast<int32_t*>(hs + hdr_pad) : nullptr;
    hm.points_cap = (long long)spec_pts;
    // One block only up to what its LDS tables hold: a mask that turns out to have more heads than that while none was expected says so
    // in place of a result (0.1 ms), and the pass is repeated as launches - instead of one block working through 600 k heads in global
    // memory (9 ms at 10 % noise).  Not when a form is forced.
    bool many_now = many;
    const size_t ws_mark = ctx->ws_off;
    const int32_t* info = nullptr;
    for (;;) {
        const bool defer = !many_now && many_env < 0;
        ctx->ws_off = ws_mark;
        VP_TRY(vpk_find_contours(ctx, bits_use, w, h, 1, mode, method, d_counts, d_hole, d_offsets, d_points, mc, mp, d_info, many_now,
                                 reinterpret_cast<uint32_t*>(d_info + 2), mirror_off ? nullptr : &hm, defer));
        if (mirror_off) {
            if (spec_pts && reinterpret_cast<uint8_t*>(d_points) == d_hdr + hdr_pad) {
                VP_TRY(d2h(ctx, hs, d_hdr, hdr_pad + spec_pts * 8));           // header and points lie back to back in the workspace: one copy
            } else {
                VP_TRY(d2h(ctx, hs, d_hdr, hdr_bytes));
                if (spec_pts) VP_TRY(d2h(ctx, hs + hdr_pad, d_points, spec_pts * 8));
            }
        }
        VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        info = reinterpret_cast<const int32_t*>(hs);
        ctx->ct_heads_hint = (uint32_t)info[2];
        if (info[0] != -1 || many_now) break;
        many_now = true;
    }
    if (info[0] < 0) return vp_fail(ctx, VP_ERR_HIP, "contours: no result");
    const int K = info[0];
    const int64_t P = info[1];
    *n_contours = K;
    if (K > max_contours) {   // the point total is only known for the contours that were traced
        *n_points = P > max_points ? P : max_points;
        return VP_OK;
    }
    *n_points = P;
    if (P > max_points || K == 0) return VP_OK;
    std::vector<int32_t> hc(K), ho(K);
    std::vector<uint8_t> hh(K);
    memcpy(hc.data(), hs + 16, (size_t)K * 4);
    memcpy(ho.data(), hs + 16 + (size_t)mc * 4, (size_t)K * 4);
    memcpy(hh.data(), hs + 16 + (size_t)mc * 8, (size_t)K);
    const int32_t* hp = reinterpret_cast<const int32_t*>(hs + hdr_pad);
    if (points && (size_t)P > spec_pts) {
        hs = (uint8_t*)vp_hstage(ctx, (size_t)P * 8 + 256);
        if (!hs) return vp_fail(ctx, VP_ERR_NOMEM, "pinned staging");
        VP_TRY(d2h(ctx, hs, d_points, (size_t)P * 8));
        VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        hp = reinterpret_cast<const int32_t*>(hs);
    }
    // device order = discovery order; cv2 hands contours back newest first
    size_t o = 0;
    for (int k = K - 1, j = 0; k >= 0; k--, j++) {
        if (points) memcpy(points + 2 * o, hp + 2 * (size_t)ho[k], (size_t)hc[k] * 8);
        o += (size_t)hc[k];
        if (counts) counts[j] = hc[k];
        if (is_hole) is_hole[j] = hh[k];
    }
    return VP_OK;
}

int vp_find_contours_u8(vp_ctx* ctx, const uint8_t* src, size_t src_stride, int w, int h, int mode, int method, int32_t* points,
                        int64_t max_points, int32_t* counts, uint8_t* is_hole, int max_contours, int32_t* n_contours, int64_t* n_points)
{
    return find_contours_impl(ctx, src, false, src_stride, w, h, mode, method, points, max_points, counts, is_hole, max_contours, n_contours, n_points);
}

// the mask is a device image; the contour lists come back to host memory as with vp_find_contours_u8 (synchronised on return)
int vp_find_contours_dev(vp_ctx* ctx, const uint8_t* d_src, size_t src_stride, int w, int h, int mode, int method, int32_t* points,
                         int64_t max_points, int32_t* counts, uint8_t* is_hole, int max_contours, int32_t* n_contours, int64_t* n_points)
{
    return find_contours_impl(ctx, d_src, true, src_stride, w, h, mode, method, points, max_points, counts, is_hole, max_contours, n_contours, n_points);
}

unsigned int vp_contours_last_heads(vp_ctx* ctx)
{
    if (!ctx) return 0;
    const uint32_t b = vp_ct_batch_hint(ctx);
    return b > ctx->ct_heads_hint ? b : ctx->ct_heads_hint;
}

int vp_find_contours_bits_dev(vp_ctx* ctx, const unsigned long long* d_bits, int w, int h, int mode, int method, int32_t* points, int64_t max_points,
                              int32_t* counts, uint8_t* is_hole, int max_contours, int32_t* n_contours, int64_t* n_points)
{
    return find_contours_impl(ctx, nullptr, true, 0, w, h, mode, method, points, max_points, counts, is_hole, max_contours, n_contours, n_points,
                              reinterpret_cast<const u64*>(d_bits));
}

// ---- chain -------------------------------------------------------------------------------------------

static int check_desc(vp_ctx* ctx, const vp_chain_desc* d, int n)
{
    if (!d || n <= 0 || d->width <= 0 || d->height <= 0) return vp_fail(ctx, VP_ERR_INVALID, "chain: size");
    // frames ride on gridDim.y; segment ids and pixel indices are 32-bit
    if (n > 65535 || (unsigned long long)d->width * (unsigned long long)d->height > (1ull << 30)) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "chain: batch or frame too large");
    if (d->color_mode != VP_BGR2LAB && d->color_mode != VP_BGR2HSV && d->color_mode != VP_BGR2GRAY)
        return vp_fail(ctx, VP_ERR_INVALID, "chain: color_mode");
    if (d->n_morph < 0 || d->n_morph > VP_CHAIN_MAX_MORPH) return vp_fail(ctx, VP_ERR_INVALID, "chain: n_morph");
    for (int i = 0; i < d->n_morph; i++)
        if (d->morph_op[i] < VP_MORPH_ERODE || d->morph_op[i] > VP_MORPH_CLOSE || d->morph_kw[i] <= 0 || d->morph_kh[i] <= 0 ||
            d->morph_iter[i] < 0)
            return vp_fail(ctx, VP_ERR_INVALID, "chain: morph op");
    if (d->ccl < 0 || d->ccl > 2) return vp_fail(ctx, VP_ERR_INVALID, "chain: ccl");
    if (d->ccl && d->numbering != VP_CCL_BLOCK2X2 && d->numbering != VP_CCL_PIXEL) return vp_fail(ctx, VP_ERR_INVALID, "chain: numbering");
    if (d->ccl && d->max_labels < 1) return vp_fail(ctx, VP_ERR_INVALID, "chain: max_labels");
    return VP_OK;
}

static size_t chain_ws_bytes(const vp_chain_desc* d, int n)
{
    const size_t bitbytes = (size_t)n * d->height * vp_ww(d->width) * 8;
    size_t need = 3 * vp_align(bitbytes) + vp_align((size_t)n * 4) + 8192 + 4 * 16384;
    if (d->ccl) need += vp_ccl_ws_bytes(d->width, d->height, n, d->max_labels);
    return need;
}

// core: all pointers device; workspace already reserved and not yet carved past `ctx->ws_off`
// frames per contour pass: the contour scratch is about 41 B/px per frame (85 MB at 1080p: sized for the worst case of 1.25 heads per
// pixel); a pass takes as many frames as fit a budget of 16 GiB (VP_CT_SCRATCH_MB overrides) - measured at 1080p, batch 128 (round 3): 16
// frames per pass 2.26 ms, 64 1.57 ms, 128 1.40 ms
static int ct_group_for(int w, int h, int max_contours)
{
    const char* e = getenv("VP_CT_SCRATCH_MB");
    long long budget = (e ? atoll(e) : 16384ll) << 20;
    if (budget < (1ll << 20)) budget = 1ll << 20;
    const long long per = (long long)vp_contours_ws_bytes(w, h, 1, max_contours);
    return (int)std::max<long long>(1, std::min<long long>(budget / per, 1 << 20));
}
static int chain_core(vp_ctx* ctx, const vp_chain_desc* d, const vp_chain_buffers* b, int n, const vp_contour_desc* cd = nullptr,
                      const vp_contour_buffers* cb = nullptr)
{
    const int w = d->width, h = d->height;
    const size_t bitbytes = (size_t)n * h * vp_ww(w) * 8;
    TAKE(bits_t, u64*, bitbytes);   // threshold bits
    TAKE(bits_a, u64*, bitbytes);
    TAKE(bits_b, u64*, bitbytes);
    TAKE(d_nl, int32_t*, (size_t)n * 4);
    vp_range3 q;
    norm_range(d->color_mode == VP_BGR2GRAY ? 1 : 3, d->lo, d->hi, &q);
    VP_TRY(vpk_color_thresh(ctx, d->color_mode, b->bgr, (size_t)w * 3, w, h, n, q, b->threshed, bits_t));

    std::vector<vp_bitstage> st;
    for (int i = 0; i < d->n_morph; i++) {
        norm_se se;
        // rect kernel kw x kh, centre anchor, cv2 iteration collapse for all-ones kernels
        std::vector<uint8_t> ones((size_t)d->morph_kw[i] * d->morph_kh[i], 1);
        if (normalise_se(ones.data(), d->morph_kw[i], d->morph_kh[i], -1, -1, d->morph_iter[i], &se) != VP_OK)
            return vp_fail(ctx, VP_ERR_INVALID, "chain: kernel");
        if (se.iterations == 0 || se.kw * se.kh == 1) continue;
        rect_se k = {se.kw, se.kh, se.ax, se.ay};
        if (stages_for_op(st, d->morph_op[i], k) != VP_OK) return vp_fail(ctx, VP_ERR_INVALID, "chain: op");
    }
    const bool need_clean_bits = d->ccl == 1 || (cd && cd->source == 1);
    const u64* ccl_bits = bits_t;
    const u64* clean_bits = bits_t;   // no morphology: the cleaned mask is the threshold mask
    vp_ccl_ws ws;
    memset(&ws, 0, sizeof ws);
    if (d->ccl) {
        vp_ccl_ws_carve(ctx, w, h, n, d->max_labels, &ws);
        if (!vp_ccl_ws_ok(ws)) return vp_fail(ctx, VP_ERR_NOMEM, "ccl workspace");
    }
    if ((!st.empty() && (need_clean_bits || b->cleaned)) || (st.empty() && b->cleaned)) {
        if (st.empty()) {
            VP_TRY(vpk_unpack_bits(ctx, bits_t, w, h, n, b->cleaned));
        } else {
            // bits_t must survive when CCL labels the threshold mask; run_bit_stages only reads its input
            VP_TRY(run_bit_stages(ctx, st, bits_t, bits_b, w, h, n, need_clean_bits ? bits_a : nullptr, b->cleaned));
            if (need_clean_bits) clean_bits = bits_a;
            if (d->ccl == 1) ccl_bits = bits_a;
        }
    }
    auto contours_of_batch = [&]() -> int {
        const u64* src = cd->source == 1 ? clean_bits : bits_t;
        const size_t fw = (size_t)h * vp_ww(w);
        const size_t mc = (size_t)cd->max_contours;
        const size_t mark = ctx->ws_off;
        const int group = ct_group_for(w, h, cd->max_contours);
        // speckled frames in the last batch (its prefix kernel left the head counts in pinned memory): the bookkeeping as launches over
        // the chip instead of one block per frame (VP_CT_MANY=0 / 1: never / always)
        const char* many_s = getenv("VP_CT_MANY");
        const bool many = many_s ? atoi(many_s) != 0 : vp_ct_batch_hint(ctx) > 8192u;
        for (int f0 = 0; f0 < n; f0 += group) {
            const int g = std::min(group, n - f0);
            ctx->ws_off = mark;   // every group reuses the same scratch (stream order keeps them apart)
            VP_TRY(vpk_find_contours(ctx, src + (size_t)f0 * fw, w, h, g, cd->mode, cd->method, cb->counts + f0 * mc, cb->is_hole + f0 * mc,
                                     cb->offsets + f0 * mc, cb->points + 2 * (size_t)f0 * (size_t)cd->max_points, cd->max_contours,
                                     cd->max_points, cb->info + 2 * (size_t)f0, many));
        }
        if (cb->features) VP_TRY(vpk_contour_features(ctx, cb->info, cb->counts, cb->offsets, cb->points, n, cd->max_contours, cd->max_points, cb->features));
        return VP_OK;
    };
    // The contour pass needs the mask only, not the labelling: when the chain does both it is queued on the context's side stream as
    // soon as the mask exists and runs beside the labelling and its label write (latency-bound launches beside a bandwidth-bound one);
    // the caller's stream joins it at the end.  One pass for the whole batch only (the scratch is carved once).  VP_CT_SIDE=0: in a row.
    static const bool ct_side_off = getenv("VP_CT_SIDE") && atoi(getenv("VP_CT_SIDE")) == 0;
    const bool ct_side = cd && d->ccl && !ct_side_off && ctx->chain_streams == 1 && ct_group_for(w, h, cd->max_contours) >= n;
    int rc_ct = VP_OK;
    bool joined = true;
    hipStream_t s_main = ctx->stream;
    if (ct_side) {
        VP_HIP(ctx, hipEventRecord(ctx->ev_fb_fork, s_main));
        VP_HIP(ctx, hipStreamWaitEvent(ctx->fb_stream, ctx->ev_fb_fork, 0));
        ctx->stream = ctx->fb_stream;
        rc_ct = contours_of_batch();
        const hipError_t ej = hipEventRecord(ctx->ev_fb_join, ctx->fb_stream);
        ctx->stream = s_main;
        joined = false;
        if (ej != hipSuccess) { (void)hipStreamSynchronize(ctx->fb_stream); joined = true; if (rc_ct == VP_OK) rc_ct = vp_fail(ctx, VP_ERR_HIP, "hipEventRecord", ej); }
    }
    int rc_ccl = VP_OK;
    if (d->ccl) rc_ccl = vpk_ccl(ctx, ccl_bits, w, h, n, d->numbering, ws, b->labels, b->stats, b->centroids, d->max_labels, b->nlabels ? b->nlabels : d_nl);
    if (!joined && hipStreamWaitEvent(s_main, ctx->ev_fb_join, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(ctx->fb_stream); }
    if (rc_ccl != VP_OK) return rc_ccl;
    if (rc_ct != VP_OK) return rc_ct;
    if (cd && !ct_side) VP_TRY(contours_of_batch());
    return VP_OK;
}

static int check_cdesc(vp_ctx* ctx, const vp_contour_desc* cd, const vp_contour_buffers* cb)
{
    if (!cd || !cb) return vp_fail(ctx, VP_ERR_INVALID, "contours: descriptor");
    if (cd->source != 1 && cd->source != 2) return vp_fail(ctx, VP_ERR_INVALID, "contours: source");
    if (cd->mode != VP_RETR_EXTERNAL && cd->mode != VP_RETR_LIST) return vp_fail(ctx, VP_ERR_INVALID, "contour mode");
    if (cd->method != VP_CHAIN_APPROX_NONE && cd->method != VP_CHAIN_APPROX_SIMPLE) return vp_fail(ctx, VP_ERR_INVALID, "contour approximation");
    if (cd->max_contours <= 0 || cd->max_points <= 0) return vp_fail(ctx, VP_ERR_INVALID, "contours: capacity");
    if (!cb->info || !cb->counts || !cb->offsets || !cb->is_hole || !cb->points) return vp_fail(ctx, VP_ERR_INVALID, "contours: buffers");
    return VP_OK;
}

// Runs the chain for n frames, split into sub-batches on the context's internal streams (fork/join around the
// caller-visible stream).  Frames are independent, so the split changes scheduling only.
static int chain_split(vp_ctx* ctx, const vp_chain_desc* d, const vp_chain_buffers* b, int n)
{
    int S = ctx->chain_streams;
    if (S > n / 4) S = n / 4;      // keep sub-batches worth a launch
    if (S <= 1) return chain_core(ctx, d, b, n);
    const size_t npx = (size_t)d->width * d->height;
    const size_t ml = (size_t)(d->ccl ? d->max_labels : 0);
    hipStream_t user = ctx->stream;
    VP_HIP(ctx, hipEventRecord(ctx->ev_fork, user));
    int rc = VP_OK;
    int f0 = 0;
    for (int s = 0; s < S && rc == VP_OK; s++) {
        const int cnt = n / S + (s < n % S ? 1 : 0);
        vp_chain_buffers sb = *b;
        sb.bgr = b->bgr + (size_t)f0 * npx * 3;
        if (b->threshed) sb.threshed = b->threshed + (size_t)f0 * npx;
        if (b->cleaned) sb.cleaned = b->cleaned + (size_t)f0 * npx;
        if (b->labels) sb.labels = b->labels + (size_t)f0 * npx;
        if (b->stats) sb.stats = b->stats + (size_t)f0 * ml * 5;
        if (b->centroids) sb.centroids = b->centroids + (size_t)f0 * ml * 2;
        if (b->nlabels) sb.nlabels = b->nlabels + f0;
        hipError_t e = hipStreamWaitEvent(ctx->aux[s], ctx->ev_fork, 0);
        if (e != hipSuccess) { rc = vp_fail(ctx, VP_ERR_HIP, "hipStreamWaitEvent", e); break; }
        ctx->stream = ctx->aux[s];
        rc = chain_core(ctx, d, &sb, cnt);
        ctx->stream = user;
        // join whatever was queued on the side stream, also after an error: it must not still run when the next call reuses the workspace
        const hipError_t e1 = hipEventRecord(ctx->ev_join[s], ctx->aux[s]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(user, ctx->ev_join[s], 0) : e1;
        if (rc != VP_OK) break;
        if (e1 != hipSuccess) { rc = vp_fail(ctx, VP_ERR_HIP, "hipEventRecord", e1); break; }
        if (e2 != hipSuccess) { rc = vp_fail(ctx, VP_ERR_HIP, "hipStreamWaitEvent", e2); break; }
        f0 += cnt;
    }
    ctx->stream = user;
    return rc;
}

int vp_chain_run(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* dev, int n_frames)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_desc(ctx, desc, n_frames));
    if (!dev || !dev->bgr) return vp_fail(ctx, VP_ERR_INVALID, "chain: bgr");
    VP_TRY(vp_ws_reserve(ctx, chain_ws_bytes(desc, n_frames) + 4 * 65536));
    return chain_split(ctx, desc, dev, n_frames);
}

int vp_chain_run_host(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* host, int n)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_desc(ctx, desc, n));
    if (!host || !host->bgr) return vp_fail(ctx, VP_ERR_INVALID, "chain: bgr");
    const size_t npx = (size_t)n * desc->width * desc->height;
    const size_t ml = (size_t)(desc->ccl ? desc->max_labels : 1);
    VP_TRY(vp_ws_reserve(ctx, chain_ws_bytes(desc, n) + vp_align(npx * 3) + 2 * vp_align(npx) + vp_align(npx * 4) +
                                  vp_align(n * ml * 20) + vp_align(n * ml * 16) + vp_align((size_t)n * 4) + 8192));
    vp_chain_buffers d;
    memset(&d, 0, sizeof d);
    TAKE(d_bgr, uint8_t*, npx * 3);
    d.bgr = d_bgr;
    if (host->threshed) { d.threshed = (uint8_t*)vp_ws_take(ctx, npx); if (!d.threshed) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
    if (host->cleaned) { d.cleaned = (uint8_t*)vp_ws_take(ctx, npx); if (!d.cleaned) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
    if (desc->ccl) {
        if (host->labels) { d.labels = (int32_t*)vp_ws_take(ctx, npx * 4); if (!d.labels) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        if (host->stats) { d.stats = (int32_t*)vp_ws_take(ctx, n * ml * 20); if (!d.stats) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        if (host->centroids) { d.centroids = (double*)vp_ws_take(ctx, n * ml * 16); if (!d.centroids) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        d.nlabels = (int32_t*)vp_ws_take(ctx, (size_t)n * 4);
        if (!d.nlabels) return vp_fail(ctx, VP_ERR_NOMEM, "workspace");
    }
    VP_TRY(h2d(ctx, d_bgr, host->bgr, npx * 3));
    VP_TRY(chain_split(ctx, desc, &d, n));
    if (host->threshed) VP_TRY(d2h(ctx, host->threshed, d.threshed, npx));
    if (host->cleaned) VP_TRY(d2h(ctx, host->cleaned, d.cleaned, npx));
    if (desc->ccl) {
        if (host->labels) VP_TRY(d2h(ctx, host->labels, d.labels, npx * 4));
        if (host->stats) VP_TRY(d2h(ctx, host->stats, d.stats, n * ml * 20));
        if (host->centroids) VP_TRY(d2h(ctx, host->centroids, d.centroids, n * ml * 16));
        if (host->nlabels) VP_TRY(d2h(ctx, host->nlabels, d.nlabels, (size_t)n * 4));
    }
    return vp_synchronize(ctx);
}

int vp_chain_run_contours(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* dev, const vp_contour_desc* cdesc,
                          const vp_contour_buffers* cdev, int n_frames)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_desc(ctx, desc, n_frames));
    VP_TRY(check_cdesc(ctx, cdesc, cdev));
    if (!dev || !dev->bgr) return vp_fail(ctx, VP_ERR_INVALID, "chain: bgr");
    VP_TRY(vp_ws_reserve(ctx, chain_ws_bytes(desc, n_frames) + 4 * 65536 +
                                  vp_contours_ws_bytes(desc->width, desc->height, std::min(n_frames, ct_group_for(desc->width, desc->height, cdesc->max_contours)), cdesc->max_contours)));
    return chain_core(ctx, desc, dev, n_frames, cdesc, cdev);
}

int vp_chain_run_contours_host(vp_ctx* ctx, const vp_chain_desc* desc, const vp_chain_buffers* host, const vp_contour_desc* cdesc,
                               const vp_contour_buffers* chost, int n)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_desc(ctx, desc, n));
    VP_TRY(check_cdesc(ctx, cdesc, chost));
    if (!host || !host->bgr) return vp_fail(ctx, VP_ERR_INVALID, "chain: bgr");
    const size_t npx = (size_t)n * desc->width * desc->height;
    const size_t ml = (size_t)(desc->ccl ? desc->max_labels : 1);
    const size_t mc = (size_t)cdesc->max_contours, mp = (size_t)cdesc->max_points;
    VP_TRY(vp_ws_reserve(ctx, chain_ws_bytes(desc, n) + vp_align(npx * 3) + 2 * vp_align(npx) + vp_align(npx * 4) + vp_align(n * ml * 20) +
                                  vp_align(n * ml * 16) + vp_align((size_t)n * 4) + vp_align((size_t)n * 8) + 2 * vp_align(n * mc * 4) +
                                  vp_align(n * mc) + vp_align(n * mp * 8) + vp_align(n * mc * 64) + 16384 +
                                  vp_contours_ws_bytes(desc->width, desc->height, std::min(n, ct_group_for(desc->width, desc->height, cdesc->max_contours)), cdesc->max_contours)));
    vp_chain_buffers d;
    memset(&d, 0, sizeof d);
    TAKE(d_bgr, uint8_t*, npx * 3);
    d.bgr = d_bgr;
    if (host->threshed) { d.threshed = (uint8_t*)vp_ws_take(ctx, npx); if (!d.threshed) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
    if (host->cleaned) { d.cleaned = (uint8_t*)vp_ws_take(ctx, npx); if (!d.cleaned) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
    if (desc->ccl) {
        if (host->labels) { d.labels = (int32_t*)vp_ws_take(ctx, npx * 4); if (!d.labels) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        if (host->stats) { d.stats = (int32_t*)vp_ws_take(ctx, n * ml * 20); if (!d.stats) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        if (host->centroids) { d.centroids = (double*)vp_ws_take(ctx, n * ml * 16); if (!d.centroids) return vp_fail(ctx, VP_ERR_NOMEM, "workspace"); }
        d.nlabels = (int32_t*)vp_ws_take(ctx, (size_t)n * 4);
        if (!d.nlabels) return vp_fail(ctx, VP_ERR_NOMEM, "workspace");
    }
    vp_contour_buffers c;
    c.info = (int32_t*)vp_ws_take(ctx, (size_t)n * 8);
    c.counts = (int32_t*)vp_ws_take(ctx, n * mc * 4);
    c.offsets = (int32_t*)vp_ws_take(ctx, n * mc * 4);
    c.is_hole = (uint8_t*)vp_ws_take(ctx, n * mc);
    c.points = (int32_t*)vp_ws_take(ctx, n * mp * 8);
    c.features = chost->features ? (double*)vp_ws_take(ctx, n * mc * 64) : nullptr;
    if (!c.info || !c.counts || !c.offsets || !c.is_hole || !c.points || (chost->features && !c.features)) return vp_fail(ctx, VP_ERR_NOMEM, "workspace");
    VP_TRY(h2d(ctx, d_bgr, host->bgr, npx * 3));
    VP_TRY(chain_core(ctx, desc, &d, n, cdesc, &c));
    if (host->threshed) VP_TRY(d2h(ctx, host->threshed, d.threshed, npx));
    if (host->cleaned) VP_TRY(d2h(ctx, host->cleaned, d.cleaned, npx));
    if (desc->ccl) {
        if (host->labels) VP_TRY(d2h(ctx, host->labels, d.labels, npx * 4));
        if (host->stats) VP_TRY(d2h(ctx, host->stats, d.stats, n * ml * 20));
        if (host->centroids) VP_TRY(d2h(ctx, host->centroids, d.centroids, n * ml * 16));
        if (host->nlabels) VP_TRY(d2h(ctx, host->nlabels, d.nlabels, (size_t)n * 4));
    }
    VP_TRY(d2h(ctx, chost->info, c.info, (size_t)n * 8));
    VP_TRY(d2h(ctx, chost->counts, c.counts, n * mc * 4));
    VP_TRY(d2h(ctx, chost->offsets, c.offsets, n * mc * 4));
    VP_TRY(d2h(ctx, chost->is_hole, c.is_hole, n * mc));
    VP_TRY(d2h(ctx, chost->points, c.points, n * mp * 8));
    if (chost->features) VP_TRY(d2h(ctx, chost->features, c.features, n * mc * 64));
    return vp_synchronize(ctx);
}

static int check_lb(vp_ctx* ctx, const void* src, const void* dst, int w, int h, int dw, int dh, int pad)
{
    if (!src || !dst || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || dh > 65535 || pad < 0 || pad > 255) return vp_fail(ctx, VP_ERR_INVALID, "letterbox arguments");
    return VP_OK;
}

int vp_threshold_u8(vp_ctx* ctx, const uint8_t* src, size_t n, double thresh, double maxval, int type, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || n == 0 || type < VP_THRESH_BINARY || type > VP_THRESH_TOZERO_INV || thresh != thresh || maxval != maxval)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_threshold_u8 arguments");
    const double ft = floor(thresh);
    const int ithresh = ft < -1 ? -1 : (ft > 256 ? 256 : (int)ft);
    const double rm = nearbyint(maxval);
    const int imaxval = rm < 0 ? 0 : (rm > 255 ? 255 : (int)rm);
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(n) + 1024));
    TAKE(d_src, uint8_t*, n);
    TAKE(d_dst, uint8_t*, n);
    VP_TRY(h2d(ctx, d_src, src, n));
    VP_TRY(vpk_threshold_u8(ctx, d_src, n, ithresh, imaxval, type, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, n));
    return vp_synchronize(ctx);
}

int vp_otsu_threshold_u8(vp_ctx* ctx, const uint8_t* src, size_t n, double maxval, int type, double* thresh_out, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || n == 0 || n > 0xffffffffull || type < VP_THRESH_BINARY || type > VP_THRESH_TOZERO_INV || maxval != maxval)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_otsu_threshold_u8 arguments");
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(n) + 2048));
    TAKE(d_src, uint8_t*, n);
    TAKE(d_dst, uint8_t*, n);
    TAKE(d_hist, u32*, 1024);
    VP_TRY(h2d(ctx, d_src, src, n));
    VP_TRY(vpk_hist_u8(ctx, d_src, n, d_hist));
    u32 h[256];
    VP_TRY(d2h(ctx, h, d_hist, 1024));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // imgproc/src/thresh.cpp getThreshVal_Otsu_8u, statement by statement
    const double scale = 1. / (double)n;
    double mu = 0;
    for (int i = 0; i < 256; i++) mu += i * (double)h[i];
    mu *= scale;
    double mu1 = 0, q1 = 0, max_sigma = 0, max_val = 0;
    for (int i = 0; i < 256; i++) {
        const double p_i = h[i] * scale;
        mu1 *= q1;
        q1 += p_i;
        const double q2 = 1. - q1;
        if (std::min(q1, q2) < 1.1920929e-07 || std::max(q1, q2) > 1. - 1.1920929e-07) continue;
        mu1 = (mu1 + i * p_i) / q1;
        const double mu2 = (mu - q1 * mu1) / q2;
        const double sigma = q1 * q2 * (mu1 - mu2) * (mu1 - mu2);
        if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
    }
    if (thresh_out) *thresh_out = max_val;
    const double rm = nearbyint(maxval);
    VP_TRY(vpk_threshold_u8(ctx, d_src, n, (int)max_val, rm < 0 ? 0 : (rm > 255 ? 255 : (int)rm), type, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, n));
    return vp_synchronize(ctx);
}

int vp_gaussian_blur_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, int cn, int kw, int kh, double sigma1, double sigma2, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || h > 65535 || cn < 1 || cn > 4 || kw <= 0 || kh <= 0 || !(kw & 1) || !(kh & 1) || kw > 511 || kh > 511)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_gaussian_blur_u8 arguments");
    if (sigma1 < 0) sigma1 = 0;
    if (sigma2 <= 0) sigma2 = sigma1;
    const size_t nbytes = (size_t)w * h * cn;
    uint16_t taps[1024];
    vp_gaussian_taps(kw, sigma1, taps);
    vp_gaussian_taps(kh, sigma2, taps + kw);
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(nbytes) + vp_align(nbytes * 2) + 4096));
    TAKE(d_src, uint8_t*, nbytes);
    TAKE(d_dst, uint8_t*, nbytes);
    TAKE(d_tmp, uint16_t*, nbytes * 2);
    TAKE(d_taps, uint16_t*, 2048);
    VP_TRY(h2d(ctx, d_src, src, nbytes));
    VP_TRY(h2d(ctx, d_taps, taps, (size_t)(kw + kh) * 2));
    VP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // taps is a local array
    if (kw == 1 && kh == 1) { VP_TRY(d2h(ctx, dst, d_src, nbytes)); return vp_synchronize(ctx); }
    VP_TRY(vpk_gaussian_blur(ctx, d_src, w, h, cn, d_taps, kw, kh, d_tmp, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, nbytes));
    return vp_synchronize(ctx);
}

int vp_resize_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, int cn, int dw, int dh, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || dh > 65535 || cn < 1 || cn > 4) return vp_fail(ctx, VP_ERR_INVALID, "vp_resize_u8 arguments");
    const size_t sbytes = (size_t)w * h * cn, dbytes = (size_t)dw * dh * cn;
    VP_TRY(vp_ws_reserve(ctx, vp_align(sbytes) + vp_align(dbytes) + 1024));
    TAKE(d_src, uint8_t*, sbytes);
    TAKE(d_dst, uint8_t*, dbytes);
    VP_TRY(h2d(ctx, d_src, src, sbytes));
    VP_TRY(vpk_resize_u8(ctx, d_src, w, h, cn, dw, dh, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, dbytes));
    return vp_synchronize(ctx);
}

int vp_adaptive_threshold_mean_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, double max_value, int type, int block, double c, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || h > 65535 || (type != VP_THRESH_BINARY && type != VP_THRESH_BINARY_INV) || !std::isfinite(max_value) ||
        !std::isfinite(c) || std::fabs(c) > 1e6)
        return vp_fail(ctx, VP_ERR_INVALID, "vp_adaptive_threshold_mean_u8 arguments");
    if (block < 3 || (block & 1) == 0) return vp_fail(ctx, VP_ERR_INVALID, "adaptive threshold: block size must be odd and > 1");
    if (block > 151) return vp_fail(ctx, VP_ERR_UNSUPPORTED, "adaptive threshold: block size above 151");
    const size_t npx = (size_t)w * h;
    if (max_value < 0) { memset(dst, 0, npx); return VP_OK; }
    const int imax = (int)std::min(255.0, std::max(0.0, std::nearbyint(max_value)));
    const int idelta = type == VP_THRESH_BINARY ? (int)std::ceil(c) : (int)std::floor(c);
    VP_TRY(vp_ws_reserve(ctx, 2 * vp_align(npx) + vp_align(npx * 2) + 1024));
    TAKE(d_src, uint8_t*, npx);
    TAKE(d_dst, uint8_t*, npx);
    TAKE(d_tmp, uint16_t*, npx * 2);
    VP_TRY(h2d(ctx, d_src, src, npx));
    VP_TRY(vpk_adaptive_threshold_mean(ctx, d_src, w, h, imax, idelta, type == VP_THRESH_BINARY_INV, block, d_tmp, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, npx));
    return vp_synchronize(ctx);
}

int vp_canny_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, int cn, double t1, double t2, uint8_t* dst)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || w <= 0 || h <= 0 || h > 65535 || (size_t)w * h > ((size_t)1 << 30) || cn < 1 || cn > 4 || !std::isfinite(t1) || !std::isfinite(t2))
        return vp_fail(ctx, VP_ERR_INVALID, "vp_canny_u8 arguments");
    if (t1 > t2) std::swap(t1, t2);
    const int low = (int)std::floor(std::min(std::max(t1, -1.0), 1e9)), high = (int)std::floor(std::min(std::max(t2, -1.0), 1e9));
    const size_t npx = (size_t)w * h;
    VP_TRY(vp_ws_reserve(ctx, vp_align(npx * cn) + vp_align(npx) + vp_canny_ws_bytes(w, h) + 1024));
    TAKE(d_src, uint8_t*, npx * cn);
    TAKE(d_dst, uint8_t*, npx);
    VP_TRY(h2d(ctx, d_src, src, npx * cn));
    VP_TRY(vpk_canny_u8(ctx, d_src, w, h, cn, low, high, d_dst));
    VP_TRY(d2h(ctx, dst, d_dst, npx));
    return vp_synchronize(ctx);
}

int vp_warp_affine_u8(vp_ctx* ctx, const uint8_t* src, int w, int h, int cn, const double* m23, int flags, int border_mode,
                      const uint8_t* border_value, uint8_t* dst, int dw, int dh)
{
    VP_TRY(check_ctx(ctx));
    if (!src || !dst || !m23 || w <= 0 || h <= 0 || dw <= 0 || dh <= 0 || dh > 65535 || cn < 1 || cn > 4 || (flags & ~VP_WARP_INVERSE_MAP) ||
        (border_mode != VP_BORDER_CONSTANT && border_mode != VP_BORDER_REPLICATE))
        return vp_fail(ctx, VP_ERR_INVALID, "vp_warp_affine_u8 arguments");
    for (int i = 0; i < 6; i++)
        if (!std::isfinite(m23[i])) return vp_fail(ctx, VP_ERR_INVALID, "vp_warp_affine_u8: matrix is not finite");
    const size_t sbytes = (size_t)w * h * cn, dbytes = (size_t)dw * dh * cn;
    VP_TRY(vp_ws_reserve(ctx, vp_align(sbytes) + vp_align(dbytes) + 1024));
    TAKE(d_src, uint8_t*, sbytes);
    TAKE(d_dst, uint8_t*, dbytes);
    VP_TRY(h2d(ctx, d_src, src, sbytes));
    VP_TRY(vpk_warp_affine_u8(ctx, d_src, w, h, cn, m23, (flags & VP_WARP_INVERSE_MAP) != 0, border_mode, border_value, d_dst, dw, dh));
    VP_TRY(d2h(ctx, dst, d_dst, dbytes));
    return vp_synchronize(ctx);
}

int vp_letterbox_u8_f32(vp_ctx* ctx, const uint8_t* src, int w, int h, int dw, int dh, int pad, float* dst, float* geom_out)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_lb(ctx, src, dst, w, h, dw, dh, pad));
    const size_t sbytes = (size_t)w * h * 3, dbytes = (size_t)dw * dh * 12;
    VP_TRY(vp_ws_reserve(ctx, vp_align(sbytes) + vp_align(dbytes) + 1024));
    TAKE(d_src, uint8_t*, sbytes);
    TAKE(d_dst, float*, dbytes);
    VP_TRY(h2d(ctx, d_src, src, sbytes));
    VP_TRY(vpk_letterbox(ctx, d_src, w, h, dw, dh, pad, d_dst, geom_out));
    VP_TRY(d2h(ctx, dst, d_dst, dbytes));
    return vp_synchronize(ctx);
}

int vp_letterbox_dev(vp_ctx* ctx, const uint8_t* src, int w, int h, int dw, int dh, int pad, float* dst, float* geom_out)
{
    VP_TRY(check_ctx(ctx));
    VP_TRY(check_lb(ctx, src, dst, w, h, dw, dh, pad));
    return vpk_letterbox(ctx, src, w, h, dw, dh, pad, dst, geom_out);
}

int vp_nms_f32(vp_ctx* ctx, const float* boxes, const float* scores, int n, float thr, int rotated, int max_keep, int32_t* keep_out,
               int32_t* n_keep_out)
{
    VP_TRY(check_ctx(ctx));
    if (n < 0 || max_keep < 0 || !n_keep_out || (n > 0 && (!boxes || !scores)) || (max_keep > 0 && !keep_out))
        return vp_fail(ctx, VP_ERR_INVALID, "vp_nms_f32 arguments");
    *n_keep_out = 0;
    if (n == 0 || max_keep == 0) return VP_OK;
    const int bs = rotated ? 5 : 4;
    VP_TRY(vp_ws_reserve(ctx, vp_align((size_t)n * bs * 4) + vp_align((size_t)n * 4) + vp_align((size_t)max_keep * 4) + vp_nms_ws_bytes(n) + 2048));
    TAKE(d_boxes, float*, (size_t)n * bs * 4);
    TAKE(d_scores, float*, (size_t)n * 4);
    TAKE(d_keep, int*, (size_t)max_keep * 4);
    TAKE(d_nk, int*, 4);
    VP_TRY(h2d(ctx, d_boxes, boxes, (size_t)n * bs * 4));
    VP_TRY(h2d(ctx, d_scores, scores, (size_t)n * 4));
    VP_TRY(vpk_nms(ctx, d_boxes, d_scores, n, thr, rotated ? 1 : 0, max_keep, d_keep, d_nk));
    VP_TRY(d2h(ctx, n_keep_out, d_nk, 4));
    VP_TRY(vp_synchronize(ctx));
    if (*n_keep_out > 0) { VP_TRY(d2h(ctx, keep_out, d_keep, (size_t)*n_keep_out * 4)); VP_TRY(vp_synchronize(ctx)); }
    return VP_OK;
}

int vp_nms_dev(vp_ctx* ctx, const float* boxes, const float* scores, int n, float thr, int rotated, int max_keep, int32_t* keep_out,
               int32_t* n_keep)
{
    VP_TRY(check_ctx(ctx));
    if (n < 0 || max_keep <= 0 || !n_keep || !keep_out || (n > 0 && (!boxes || !scores))) return vp_fail(ctx, VP_ERR_INVALID, "vp_nms_dev arguments");
    VP_TRY(vp_ws_reserve(ctx, vp_nms_ws_bytes(n > 0 ? n : 1) + 2048));
    return vpk_nms(ctx, boxes, scores, n, thr, rotated ? 1 : 0, max_keep, keep_out, n_keep);
}

uint64_t vp_chain_algorithmic_bytes(const vp_chain_desc* desc, const vp_chain_buffers* bufs, int n)
{
    if (!desc || !bufs || n <= 0) return 0;
    const uint64_t npx = (uint64_t)n * desc->width * desc->height;
    uint64_t per = 3;
    if (bufs->threshed) per += 1;
    if (bufs->cleaned) per += 1;
    if (desc->ccl && bufs->labels) per += 4;
    return npx * per;
}

}  // extern "C"
