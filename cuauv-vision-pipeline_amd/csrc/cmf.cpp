// libcamera_message_framework.so — shared-memory latest-wins frame ring (see include/camera_message_framework_c.h).
//
// File layout of /dev/shm/auv_visiond_<direction> (must match the reference so mixed processes interoperate;
// offsets from lib/camera_message_framework.cpp:27-54 compiled for x86-64 / glibc):
//     0  uid                  u64, atomic: number of frames ever written
//     8  max_entry_size_bytes u64
//    16  deleted              u8
//    24  slots[3]             360 B each: seq_begin u64 | seq_end u64 | acquisition_time | total_size | width |
//                             height | depth | type_size | plane_count | planes[4] x {w,h,d,type_size,offset,name[32]}
//  1104  pthread_cond_t       process-shared
//  1152  pthread_mutex_t      process-shared, robust
//  1216  payload              3 x max_entry_size_bytes, planes of a frame back to back
//
// Protocol (single writer per block): slot = (uid + 1) % 3; seq_begin += 1; copy payload + metadata;
// seq_end = seq_begin; uid += 1; broadcast.  A reader takes slot uid % 3 and retries until it saw
// seq_end == seq_begin around its copy.
#include "../../include/camera_message_framework_c.h"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <map>
#include <memory>
#include <mutex>
#include <pthread.h>
#include <string>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

namespace {

constexpr const char* kStub = "/dev/shm/auv_visiond_";
constexpr const char* kLockFile = "/dev/shm/auv_visiond.lock";

struct ShmPlane { uint64_t width, height, depth, type_size, offset; char name[CMF_PLANE_NAME_MAX_LEN]; };
struct ShmSlot {
    uint64_t seq_begin, seq_end;
    uint64_t acquisition_time, total_size, width, height, depth, type_size, plane_count;
    ShmPlane planes[CMF_MAX_PLANE_CNT];
};
struct ShmHeader {
    uint64_t uid;
    uint64_t max_entry_size_bytes;
    uint8_t deleted;
    ShmSlot slots[CMF_BUFFER_CNT];
    pthread_cond_t cond;
    pthread_mutex_t mutex;
    alignas(64) unsigned char payload[1];
};
static_assert(sizeof(ShmPlane) == 72, "plane layout");
static_assert(sizeof(ShmSlot) == 360, "slot layout");
constexpr size_t kSlotMetaOffset = offsetof(ShmSlot, acquisition_time);   // metadata = everything after the two sequence numbers
static_assert(offsetof(ShmHeader, slots) == 24, "slots offset");
static_assert(offsetof(ShmHeader, cond) == 1104, "cond offset");
static_assert(offsetof(ShmHeader, mutex) == 1152, "mutex offset");
static_assert(offsetof(ShmHeader, payload) == 1216, "payload offset");
static_assert(sizeof(FramePlane) == 72 && sizeof(Frame) == 360 && sizeof(FramePlaneWrite) == 48, "ABI structs");
// libvp's feeder (csrc/vp_feed.hip) reads these two fields out of a Frame it only knows as 360 bytes
static_assert(offsetof(Frame, uid) == 40 && offsetof(Frame, total_size) == 56, "Frame field offsets the feeder relies on");

thread_local char t_err[256] = "";
void set_err(const char* what, const std::string& detail = "")
{
    snprintf(t_err, sizeof t_err, "%s%s%s", what, detail.empty() ? "" : ": ", detail.c_str());
}

// RAII exclusive flock on the global lock file (reference: lib/filelock.cpp:11-31)
struct GlobalLock {
    int fd;
    GlobalLock() : fd(open(kLockFile, O_RDWR | O_CREAT, 0666))
    {
        if (fd >= 0) flock(fd, LOCK_EX);
    }
    ~GlobalLock()
    {
        if (fd >= 0) { flock(fd, LOCK_UN); close(fd); }
    }
};

inline uint64_t load_acq(const uint64_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store_rel(uint64_t* p, uint64_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

// Slot contents (metadata and payload) are read while the writer may be overwriting them; the sequence numbers decide afterwards
// whether the copy is kept.  That is the seqlock's design, and to the C++ memory model (and to ThreadSanitizer) it is a data race
// unless the racing accesses are atomic.  Normal builds copy with memcpy (what the hardware does is fine and a 6 MB frame must move
// at memory speed); a build with -fsanitize=thread copies with relaxed atomic accesses instead, so that the sanitizer checks
// everything else - sequence numbers, publication order, mutex / condition variable - without drowning in the intended race
// (tests/native/cmf_tsan_main.cpp, run by tests/test_cmf.py).
#if defined(__SANITIZE_THREAD__)
#define CMF_TSAN 1
#elif defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define CMF_TSAN 1
#endif
#endif
#ifdef CMF_TSAN
inline void racy_read(void* dst, const void* src, size_t n)
{
    unsigned char* d = static_cast<unsigned char*>(dst);
    const unsigned char* s = static_cast<const unsigned char*>(src);
    for (size_t i = 0; i < n; i++) d[i] = __atomic_load_n(s + i, __ATOMIC_RELAXED);
}
inline void racy_write(void* dst, const void* src, size_t n)
{
    unsigned char* d = static_cast<unsigned char*>(dst);
    const unsigned char* s = static_cast<const unsigned char*>(src);
    for (size_t i = 0; i < n; i++) __atomic_store_n(d + i, s[i], __ATOMIC_RELAXED);
}
#else
inline void racy_read(void* dst, const void* src, size_t n) { memcpy(dst, src, n); }
inline void racy_write(void* dst, const void* src, size_t n) { memcpy(dst, src, n); }
#endif

struct PrivFrame {   // what create_frame really allocates: the public Frame first, bookkeeping after
    Frame pub;
    size_t capacity;
    bool foreign;      // pub.data belongs to the caller (cmf_frame_set_buffer): never reallocated, never freed here
};

}  // namespace

struct Block {
    std::string direction, filename;
    bool creator = false;
    ShmHeader* shm = nullptr;
    size_t mapped = 0;
    int refs = 0;   // handles given out by create_block / open_block in this process
    uint64_t open_seq = 0;   // sequence number of a deferred write in progress (cmf_write_begin), 0: none; single writer per block
    size_t open_idx = 0;     // its slot

    size_t shm_size() const { return offsetof(ShmHeader, payload) + shm->max_entry_size_bytes * CMF_BUFFER_CNT; }
    ~Block()
    {
        if (!shm) return;
        if (creator) {
            __atomic_store_n(&shm->deleted, (uint8_t)1, __ATOMIC_RELEASE);
            unlink(filename.c_str());
        }
        munmap(shm, mapped);
    }
};

namespace {

std::mutex g_registry_mutex;
std::map<std::string, std::unique_ptr<Block>> g_registry;

bool valid_direction(const char* direction)
{
    if (!direction || !*direction) { set_err("empty block name"); return false; }
    if (strchr(direction, '/')) { set_err("block name contains '/'", direction); return false; }
    return true;
}

ShmHeader* map_file(int fd, size_t* mapped)
{
    const off_t len = lseek(fd, 0, SEEK_END);
    if (len < (off_t)offsetof(ShmHeader, payload)) { errno = EINVAL; return nullptr; }
    void* mem = mmap(nullptr, (size_t)len, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (mem == MAP_FAILED) return nullptr;
    *mapped = (size_t)len;
    return static_cast<ShmHeader*>(mem);
}

void init_header(ShmHeader* h, size_t max_entry)
{
    memset(h, 0, offsetof(ShmHeader, payload));
    h->max_entry_size_bytes = max_entry;
    pthread_condattr_t ca;
    pthread_condattr_init(&ca);
    pthread_condattr_setpshared(&ca, PTHREAD_PROCESS_SHARED);
    pthread_cond_init(&h->cond, &ca);
    pthread_condattr_destroy(&ca);
    pthread_mutexattr_t ma;
    pthread_mutexattr_init(&ma);
    pthread_mutexattr_setpshared(&ma, PTHREAD_PROCESS_SHARED);
    pthread_mutexattr_setrobust(&ma, PTHREAD_MUTEX_ROBUST);
    pthread_mutex_init(&h->mutex, &ma);
    pthread_mutexattr_destroy(&ma);
}

// creator == true: create the file if missing (the creator later unlinks it); false: attach only.
std::unique_ptr<Block> attach(const std::string& direction, bool creator, size_t max_entry)
{
    GlobalLock lock;
    auto b = std::make_unique<Block>();
    b->direction = direction;
    b->filename = std::string(kStub) + direction;
    b->creator = creator;
    const bool existed = access(b->filename.c_str(), F_OK) == 0;
    if (!creator && !existed) { set_err("block does not exist", b->filename); return nullptr; }
    const int fd = open(b->filename.c_str(), creator ? (O_RDWR | O_CREAT) : O_RDWR, 0700);
    if (fd < 0) { set_err("open failed", b->filename + ": " + strerror(errno)); return nullptr; }
    if (!existed) {
        const size_t bytes = offsetof(ShmHeader, payload) + max_entry * CMF_BUFFER_CNT;
        if (ftruncate(fd, (off_t)bytes) != 0) {
            set_err("ftruncate failed", strerror(errno));
            close(fd);
            unlink(b->filename.c_str());
            return nullptr;
        }
    }
    b->shm = map_file(fd, &b->mapped);
    close(fd);
    if (!b->shm) {
        set_err("mmap failed", strerror(errno));
        if (!existed) unlink(b->filename.c_str());
        b->creator = false;
        return nullptr;
    }
    if (!existed) init_header(b->shm, max_entry);
    if (creator && b->shm->max_entry_size_bytes != max_entry) {
        // reference: removes the file and throws invalid_argument (lib/camera_message_framework.cpp:171-180)
        set_err("size mismatch with the existing block", b->filename);
        unlink(b->filename.c_str());
        b->creator = false;
        return nullptr;
    }
    return b;
}

// Copies the metadata of slot `s` (frame number `uid`) into the caller's Frame; returns the payload size, clamped to the entry size
// (metadata read while the writer is at work can be torn: the sequence numbers decide afterwards whether they are kept).
size_t snapshot_meta(const ShmHeader* h, const ShmSlot& s, uint64_t uid, Frame* frame)
{
    ShmSlot meta;
    racy_read(reinterpret_cast<unsigned char*>(&meta) + kSlotMetaOffset, reinterpret_cast<const unsigned char*>(&s) + kSlotMetaOffset,
              sizeof(ShmSlot) - kSlotMetaOffset);
    frame->width = meta.width; frame->height = meta.height; frame->depth = meta.depth; frame->type_size = meta.type_size;
    frame->acquisition_time = meta.acquisition_time;
    frame->uid = uid;
    size_t total = meta.total_size;
    if (total > h->max_entry_size_bytes) total = h->max_entry_size_bytes;
    frame->total_size = total;
    frame->plane_count = (size_t)meta.plane_count <= CMF_MAX_PLANE_CNT ? (size_t)meta.plane_count : 0;
    for (size_t i = 0; i < CMF_MAX_PLANE_CNT; i++) {
        const ShmPlane& m = meta.planes[i];
        FramePlane& o = frame->planes[i];
        o.width = m.width; o.height = m.height; o.depth = m.depth; o.type_size = m.type_size; o.offset = m.offset;
        memcpy(o.name, m.name, CMF_PLANE_NAME_MAX_LEN);
        o.name[CMF_PLANE_NAME_MAX_LEN - 1] = '\0';
    }
    return total;
}

}  // namespace

extern "C" {

const char* BLOCK_STUB_CSTR = kStub;
const int SUCCESS = 0;
const int NO_NEW_FRAME = 1;
const int FRAMEWORK_DELETED = 2;

const char* cmf_last_error(void) { return t_err; }

Block* create_block(const char* direction, size_t max_entry_size_bytes)
{
    if (!valid_direction(direction)) return nullptr;
    if (max_entry_size_bytes == 0) { set_err("max_entry_size_bytes must be positive"); return nullptr; }
    std::lock_guard<std::mutex> g(g_registry_mutex);
    auto it = g_registry.find(direction);
    if (it != g_registry.end()) {
        if (it->second->shm->max_entry_size_bytes == max_entry_size_bytes) { it->second->refs++; return it->second.get(); }
        set_err("duplicate allocation with a different size", direction);
        return nullptr;
    }
    auto b = attach(direction, true, max_entry_size_bytes);
    if (!b) return nullptr;
    b->refs = 1;
    Block* raw = b.get();
    g_registry.emplace(direction, std::move(b));
    return raw;
}

Block* open_block(const char* direction)
{
    if (!valid_direction(direction)) return nullptr;
    std::lock_guard<std::mutex> g(g_registry_mutex);
    auto it = g_registry.find(direction);
    if (it != g_registry.end()) { it->second->refs++; return it->second.get(); }
    auto b = attach(direction, false, 0);
    if (!b) return nullptr;
    b->refs = 1;
    Block* raw = b.get();
    g_registry.emplace(direction, std::move(b));
    return raw;
}

void delete_block(Block* block)
{
    if (!block) return;
    std::lock_guard<std::mutex> g(g_registry_mutex);
    // look the handle up by value: a stale pointer (already released) must not be dereferenced.  The reference
    // destroys the block on the first delete even when another accessor of the same process still holds it;
    // here every create/open is matched by one delete and the last one destroys.
    for (auto it = g_registry.begin(); it != g_registry.end(); ++it) {
        if (it->second.get() != block) continue;
        if (--it->second->refs <= 0) {
            GlobalLock lock;   // destruction is serialised with attach
            g_registry.erase(it);
        }
        return;
    }
}

// Validates the planes of a write and returns the bytes they take in the slot (negative: a CMF_ERR_* status, text in cmf_last_error).
// need_data: write_frame_planes copies from planes[i].data itself; the deferred form (cmf_write_commit) only describes what was moved.
static long long planes_bytes(const ShmHeader* h, const FramePlaneWrite* planes, size_t plane_count, bool need_data)
{
    if (!planes) { set_err("planes pointer cannot be null"); return CMF_ERR_INVALID; }
    if (plane_count == 0 || plane_count > CMF_MAX_PLANE_CNT) { set_err("invalid plane count"); return CMF_ERR_INVALID; }
    size_t entry = 0;
    for (size_t i = 0; i < plane_count; i++) {
        const FramePlaneWrite& p = planes[i];
        if (need_data && !p.data) { set_err("plane has null data pointer"); return CMF_ERR_INVALID; }
        if (p.type_size != 1 && p.type_size != 4 && p.type_size != 8) { set_err("unsupported type size (expected 1, 4 or 8)"); return CMF_ERR_INVALID; }
        entry += p.width * p.height * p.depth * p.type_size;
    }
    if (entry > h->max_entry_size_bytes) { set_err("frame larger than the block's max_entry_size_bytes"); return CMF_ERR_TOO_LARGE; }
    return (long long)entry;
}

// First half of a write: the slot after the newest one is opened (readers that still copy its old frame now see begin != end and
// retry).  The slot is NOT the one readers are sent to (that is uid % 3) until publish_slot.
static uint64_t open_slot(ShmHeader* h, size_t* idx_out)
{
    const uint64_t uid = load_acq(&h->uid);
    const size_t idx = (size_t)((uid + 1) % CMF_BUFFER_CNT);
    ShmSlot& s = h->slots[idx];
    const uint64_t seq = load_acq(&s.seq_begin) + 1;
    store_rel(&s.seq_begin, seq);                       // readers of this slot now see begin != end
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    *idx_out = idx;
    return seq;
}

// Second half: metadata, closing sequence number, uid, wake-up.  The payload is in the slot by now.
static void publish_slot(ShmHeader* h, size_t idx, uint64_t seq, uint64_t acquisition_time, const FramePlaneWrite* planes, size_t plane_count,
                         size_t entry)
{
    ShmSlot& s = h->slots[idx];
    ShmSlot meta;                                       // the slot's metadata, built here and copied over in one go
    memset(&meta, 0, sizeof meta);
    size_t cursor = 0;
    for (size_t i = 0; i < plane_count; i++) {
        const FramePlaneWrite& p = planes[i];
        ShmPlane& m = meta.planes[i];
        m.width = p.width; m.height = p.height; m.depth = p.depth; m.type_size = p.type_size; m.offset = cursor;
        if (p.name) strncpy(m.name, p.name, CMF_PLANE_NAME_MAX_LEN - 1);
        cursor += p.width * p.height * p.depth * p.type_size;
    }
    meta.acquisition_time = acquisition_time;
    meta.total_size = entry;
    meta.plane_count = plane_count;
    meta.width = planes[0].width; meta.height = planes[0].height; meta.depth = planes[0].depth; meta.type_size = planes[0].type_size;
    // everything after the two sequence numbers (they lead the slot and are written with release stores of their own)
    racy_write(reinterpret_cast<unsigned char*>(&s) + kSlotMetaOffset, reinterpret_cast<const unsigned char*>(&meta) + kSlotMetaOffset,
               sizeof(ShmSlot) - kSlotMetaOffset);
    store_rel(&s.seq_end, seq);                         // frame complete
    __atomic_fetch_add(&h->uid, 1, __ATOMIC_ACQ_REL);   // publish: slot uid % 3 is the newest
    pthread_cond_broadcast(&h->cond);
}

int write_frame_planes(Block* block, uint64_t acquisition_time, const FramePlaneWrite* planes, size_t plane_count)
{
    if (!block || !block->shm) { set_err("null block"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    const long long entry = planes_bytes(h, planes, plane_count, true);
    if (entry < 0) return (int)entry;
    if (block->open_seq) { set_err("a deferred write (cmf_write_begin) is open on this block: commit or abort it first"); return CMF_ERR_INVALID; }
    size_t idx;
    const uint64_t seq = open_slot(h, &idx);
    unsigned char* dst = h->payload + idx * h->max_entry_size_bytes;
    size_t cursor = 0;
    for (size_t i = 0; i < plane_count; i++) {
        const FramePlaneWrite& p = planes[i];
        const size_t bytes = p.width * p.height * p.depth * p.type_size;
        racy_write(dst + cursor, p.data, bytes);
        cursor += bytes;
    }
    publish_slot(h, idx, seq, acquisition_time, planes, plane_count, (size_t)entry);
    return SUCCESS;
}

int cmf_write_begin(Block* block, uint64_t entry_bytes, void** payload, uint64_t* ticket)
{
    if (!block || !block->shm || !payload || !ticket) { set_err("null block, payload or ticket"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    if (entry_bytes == 0) { set_err("empty frame"); return CMF_ERR_INVALID; }
    if (entry_bytes > h->max_entry_size_bytes) { set_err("frame larger than the block's max_entry_size_bytes"); return CMF_ERR_TOO_LARGE; }
    if (block->open_seq) { set_err("a deferred write is already open on this block"); return CMF_ERR_INVALID; }
    size_t idx;
    const uint64_t seq = open_slot(h, &idx);
    block->open_seq = seq;
    block->open_idx = idx;
    *payload = h->payload + idx * h->max_entry_size_bytes;
    *ticket = seq;
    return SUCCESS;
}

int cmf_write_commit(Block* block, uint64_t ticket, uint64_t acquisition_time, const FramePlaneWrite* planes, size_t plane_count)
{
    if (!block || !block->shm) { set_err("null block"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (!block->open_seq || block->open_seq != ticket) { set_err("no deferred write with this ticket is open on the block"); return CMF_ERR_INVALID; }
    const long long entry = planes_bytes(h, planes, plane_count, false);
    if (entry < 0) return (int)entry;
    // the payload was moved by somebody else (a copy engine): order its completion, which the caller has observed, before the metadata
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    publish_slot(h, block->open_idx, ticket, acquisition_time, planes, plane_count, (size_t)entry);
    block->open_seq = 0;
    return __atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE) ? FRAMEWORK_DELETED : SUCCESS;
}

int cmf_write_abort(Block* block, uint64_t ticket)
{
    if (!block || !block->shm) { set_err("null block"); return CMF_ERR_INVALID; }
    if (!block->open_seq || block->open_seq != ticket) { set_err("no deferred write with this ticket is open on the block"); return CMF_ERR_INVALID; }
    // The slot never became the newest one, so no reader is sent to it; a reader that was still copying its OLD frame has seen
    // begin != end since cmf_write_begin and retries on a newer slot.  Closing the sequence pair leaves the slot reusable: the next
    // write opens the same slot again (uid did not move) and bumps the pair once more.
    store_rel(&block->shm->slots[block->open_idx].seq_end, ticket);
    block->open_seq = 0;
    return SUCCESS;
}

int write_frame(Block* block, uint64_t acquisition_time, size_t width, size_t height, size_t depth, size_t type_size,
                const unsigned char* data)
{
    FramePlaneWrite p{width, height, depth, type_size, data, nullptr};
    return write_frame_planes(block, acquisition_time, &p, 1);
}

int read_frame(Block* block, Frame* frame, bool block_thread)
{
    if (!block || !block->shm || !frame) { set_err("null block or frame"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    if (block_thread && frame->uid >= load_acq(&h->uid)) {
        int rc = pthread_mutex_lock(&h->mutex);
        if (rc == EOWNERDEAD) { pthread_mutex_consistent(&h->mutex); rc = 0; }   // a holder died: repair and carry on
        if (rc == 0) {
            if (frame->uid >= load_acq(&h->uid)) {
                struct timeval now;
                gettimeofday(&now, nullptr);
                struct timespec until;
                until.tv_sec = now.tv_sec + 1;
                until.tv_nsec = now.tv_usec * 1000L;
                pthread_cond_timedwait(&h->cond, &h->mutex, &until);
            }
            pthread_mutex_unlock(&h->mutex);
        }
        if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    }
    if (frame->uid >= load_acq(&h->uid)) return NO_NEW_FRAME;

    PrivFrame* pf = reinterpret_cast<PrivFrame*>(frame);
    if (pf->capacity < h->max_entry_size_bytes) {
        if (pf->foreign) { set_err("the buffer given to cmf_frame_set_buffer is smaller than the block's entry size"); return CMF_ERR_INVALID; }
        void* grown = realloc(frame->data, h->max_entry_size_bytes);
        if (!grown) { set_err("out of memory"); return CMF_ERR_INVALID; }
        frame->data = grown;
        pf->capacity = h->max_entry_size_bytes;
    }
    for (;;) {
        const uint64_t uid = load_acq(&h->uid);
        const ShmSlot& s = h->slots[uid % CMF_BUFFER_CNT];
        const uint64_t end = load_acq(&s.seq_end);
        const size_t total = snapshot_meta(h, s, uid, frame);
        racy_read(frame->data, h->payload + (uid % CMF_BUFFER_CNT) * h->max_entry_size_bytes, total);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        const uint64_t begin = load_acq(&s.seq_begin);
        if (begin == end) return SUCCESS;
        if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    }
}

int cmf_peek_frame(Block* block, Frame* frame, const void** payload, uint64_t* ticket)
{
    if (!block || !block->shm || !frame || !payload || !ticket) { set_err("null block, frame, payload or ticket"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    if (frame->uid >= load_acq(&h->uid)) return NO_NEW_FRAME;
    for (;;) {
        const uint64_t uid = load_acq(&h->uid);
        const ShmSlot& s = h->slots[uid % CMF_BUFFER_CNT];
        const uint64_t end = load_acq(&s.seq_end);
        snapshot_meta(h, s, uid, frame);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        if (load_acq(&s.seq_begin) == end) {               // the metadata are those of a complete frame
            *payload = h->payload + (uid % CMF_BUFFER_CNT) * h->max_entry_size_bytes;
            *ticket = end;
            return SUCCESS;
        }
        if (__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return FRAMEWORK_DELETED;
    }
}

int cmf_peek_validate(Block* block, uint64_t uid, uint64_t ticket)
{
    if (!block || !block->shm) { set_err("null block"); return CMF_ERR_INVALID; }
    const ShmSlot& s = block->shm->slots[uid % CMF_BUFFER_CNT];
    __atomic_thread_fence(__ATOMIC_SEQ_CST);               // the caller's copy of the payload comes before this load
    return load_acq(&s.seq_begin) == ticket ? 1 : 0;
}

int cmf_wait_for_frame(Block* block, uint64_t have_uid, uint32_t timeout_us)
{
    if (!block || !block->shm) { set_err("null block"); return CMF_ERR_INVALID; }
    ShmHeader* h = block->shm;
    if (load_acq(&h->uid) > have_uid || __atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) return 1;
    int rc = pthread_mutex_lock(&h->mutex);
    if (rc == EOWNERDEAD) { pthread_mutex_consistent(&h->mutex); rc = 0; }
    if (rc != 0) return 0;
    if (!(load_acq(&h->uid) > have_uid) && !__atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) {
        struct timeval now;
        gettimeofday(&now, nullptr);
        struct timespec until;
        const uint64_t ns = (uint64_t)now.tv_usec * 1000ull + (uint64_t)timeout_us * 1000ull;
        until.tv_sec = now.tv_sec + (time_t)(ns / 1000000000ull);
        until.tv_nsec = (long)(ns % 1000000000ull);
        pthread_cond_timedwait(&h->cond, &h->mutex, &until);
    }
    pthread_mutex_unlock(&h->mutex);
    return (load_acq(&h->uid) > have_uid || __atomic_load_n(&h->deleted, __ATOMIC_ACQUIRE)) ? 1 : 0;
}

int cmf_block_mapping(Block* block, void** base, uint64_t* bytes)
{
    if (!block || !block->shm || !base || !bytes) { set_err("null block or result pointer"); return CMF_ERR_INVALID; }
    *base = block->shm;
    *bytes = (uint64_t)block->mapped;
    return 0;
}

Frame* create_frame(void)
{
    PrivFrame* pf = static_cast<PrivFrame*>(calloc(1, sizeof(PrivFrame)));
    if (!pf) return nullptr;
    pf->pub.data = malloc(64);
    pf->capacity = pf->pub.data ? 64 : 0;
    return &pf->pub;
}

void delete_frame(Frame* frame)
{
    if (!frame) return;
    if (!reinterpret_cast<PrivFrame*>(frame)->foreign) free(frame->data);
    free(reinterpret_cast<PrivFrame*>(frame));
}

int cmf_frame_set_buffer(Frame* frame, void* buf, uint64_t capacity)
{
    if (!frame) { set_err("null frame"); return CMF_ERR_INVALID; }
    PrivFrame* pf = reinterpret_cast<PrivFrame*>(frame);
    if (!pf->foreign) free(frame->data);
    if (buf) {
        frame->data = buf;
        pf->capacity = (size_t)capacity;
        pf->foreign = true;
    } else {
        frame->data = malloc(64);
        pf->capacity = frame->data ? 64 : 0;
        pf->foreign = false;
    }
    return 0;
}

uint64_t cmf_block_entry_size(Block* block) { return (block && block->shm) ? (uint64_t)block->shm->max_entry_size_bytes : 0; }

uint64_t frame_size(Frame* frame) { return frame ? frame->total_size : 0; }

}  // extern "C"
