/*
 * camera_message_framework_c.h — C ABI of libcamera_message_framework.so, the shared-memory frame ring
 * ("CMF") between capture sources, vision modules and the GUI.
 *
 * Drop-in for the reference's lib/camera_message_framework_c.cpp:18-103 (cdef mirrored by
 * core/bindings/camera_message_framework.py:13-68): same 9 functions, same 4 data symbols, same
 * struct layouts, same file layout of /dev/shm/auv_visiond_<direction>
 * (lib/camera_message_framework.cpp:27-54, include/camera_message_framework.hpp:9-30), so old and new
 * processes can share a block.  Differences, all inside what the old ABI allowed:
 *   - nothing throws across the ABI: failures return NULL / a negative status and cmf_last_error()
 *     explains (the reference throws C++ exceptions out of `extern "C"`, lib/...cpp:261-304);
 *   - the writer bumps the slot's first sequence word *before* copying the payload (seqlock
 *     begin/end), closing the window in which a lapped reader could accept a half-new payload.
 */
#ifndef CAMERA_MESSAGE_FRAMEWORK_C_H
#define CAMERA_MESSAGE_FRAMEWORK_C_H
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMF_BUFFER_CNT 3          /* include/camera_message_framework.hpp:9  */
#define CMF_MAX_PLANE_CNT 4       /* :12 */
#define CMF_PLANE_NAME_MAX_LEN 32 /* :15 */

/* status codes (hpp:18-24) — also exported as data symbols, like the reference */
extern const char* BLOCK_STUB_CSTR; /* "/dev/shm/auv_visiond_" */
extern const int SUCCESS;           /* 0 */
extern const int NO_NEW_FRAME;      /* 1 */
extern const int FRAMEWORK_DELETED; /* 2 */
/* additional negative statuses of this implementation (the reference throws instead) */
#define CMF_ERR_INVALID (-1)   /* null pointer, plane count not in 1..4, type size not 1/4/8 */
#define CMF_ERR_TOO_LARGE (-2) /* payload larger than the block's max_entry_size_bytes */

typedef struct Block Block; /* opaque; owned by the library's process-global registry */

typedef struct FramePlane { /* 72 bytes (hpp:39-46) */
    size_t width, height, depth, type_size, offset;
    char name[CMF_PLANE_NAME_MAX_LEN];
} FramePlane;

typedef struct Frame { /* 360 bytes (hpp:48-85) */
    size_t width, height, depth, type_size;
    uint64_t acquisition_time;
    uint64_t uid;
    void* data; /* library-owned; valid until the next read_frame on this Frame */
    size_t total_size;
    size_t plane_count;
    FramePlane planes[CMF_MAX_PLANE_CNT];
} Frame;

typedef struct FramePlaneWrite { /* 48 bytes (hpp:91-98) */
    size_t width, height, depth, type_size;
    const void* data;
    const char* name; /* may be NULL */
} FramePlaneWrite;

/* cmf_c.cpp:23-41: create (or attach to) the block of `direction`; the creator unlinks it on
 * delete_block.  Same direction + same size returns the same pointer; a size mismatch returns NULL. */
Block* create_block(const char* direction, size_t max_entry_size_bytes);
/* cmf_c.cpp:43-60: attach to an existing block; NULL when the file does not exist. */
Block* open_block(const char* direction);
/* cmf_c.cpp:62-65 */
void delete_block(Block* block);
/* cmf_c.cpp:67-77 */
int write_frame(Block* block, uint64_t acquisition_time, size_t width, size_t height, size_t depth, size_t type_size,
                const unsigned char* data);
/* cmf_c.cpp:79-86 */
int write_frame_planes(Block* block, uint64_t acquisition_time, const FramePlaneWrite* planes, size_t plane_count);
/* cmf_c.cpp:88-90: latest-wins read; NO_NEW_FRAME when frame->uid is current; block_thread waits <= 1 s. */
int read_frame(Block* block, Frame* frame, bool block_thread);
/* cmf_c.cpp:92-102 */
Frame* create_frame(void);
void delete_frame(Frame* frame);
uint64_t frame_size(Frame* frame);

/* not in the reference: text of the last failure on this thread */
const char* cmf_last_error(void);

/* not in the reference: let read_frame copy straight into memory the caller owns (page-locked memory the module runtime hands to
 * process() as the frame's private copy - reference core/base.py:765-768 makes that copy with numpy after read_frame).
 * cmf_frame_set_buffer(frame, buf, capacity): from now on read_frame fills `buf`; it fails (negative status, frame untouched) when
 * capacity is below the block's entry size.  buf == NULL gives the frame a buffer of its own again.  The frame never frees `buf`.
 * cmf_block_entry_size(block): the block's max_entry_size_bytes (what `capacity` has to reach). */
int cmf_frame_set_buffer(Frame* frame, void* buf, uint64_t capacity);
uint64_t cmf_block_entry_size(Block* block);

/* not in the reference: a read that leaves moving the payload to the caller - a copy engine that takes the frame out of the ring slot
 * straight into device memory, instead of read_frame's memcpy (lib/camera_message_framework.cpp:421-452) followed by the runtime's
 * own copy (core/base.py:765-768) and an upload.
 *   cmf_peek_frame(block, frame, &payload, &ticket): read_frame without the payload copy.  Same statuses and the same uid rule
 *     (NO_NEW_FRAME while frame->uid is current).  On SUCCESS the metadata of `frame` (sizes, planes, acquisition_time, uid,
 *     total_size) describe the newest slot, frame->data is left alone, *payload points at the slot's bytes inside the block's mapping
 *     and *ticket is the slot's sequence number at that moment.
 *   cmf_peek_validate(block, frame->uid, ticket): call AFTER the bytes have been moved.  1: the writer has not touched the slot since
 *     the peek, what was copied is the frame the metadata describe.  0: the writer lapped the ring meanwhile; discard the copy and peek
 *     again (a newer frame exists by then: the slot of frame u is reused for frame u + 3).  Negative: invalid arguments.
 *   cmf_block_mapping(block, &base, &bytes): the block's mapping in this process, for page-locking it once (hipHostRegister). */
int cmf_peek_frame(Block* block, Frame* frame, const void** payload, uint64_t* ticket);
int cmf_peek_validate(Block* block, uint64_t uid, uint64_t ticket);
int cmf_block_mapping(Block* block, void** base, uint64_t* bytes);
/* cmf_wait_for_frame(block, have_uid, timeout_us): returns 1 as soon as the block holds a frame newer than `have_uid` (or has been
 * deleted), 0 after `timeout_us` without one; waits on the block's condition variable the way read_frame(block_thread = true) does
 * (lib/camera_message_framework.cpp:395-410), with a caller-chosen bound.  For a feeder thread that must not poll. */
int cmf_wait_for_frame(Block* block, uint64_t have_uid, uint32_t timeout_us);

/* not in the reference: a write whose payload is moved by the caller - a copy engine that puts a device image straight into the ring
 * slot, instead of a download into a host array followed by write_frame's memcpy (lib/camera_message_framework.cpp:306-374 does both
 * halves in one call; core/base.py:846-876 / :832-839 queue and flush a module's posts through it).  The slot protocol is the
 * reference's, cut in two at the payload copy:
 *   cmf_write_begin(block, entry_bytes, &payload, &ticket): picks the slot after the newest one ((uid + 1) % 3), bumps its first
 *     sequence number (a reader still copying the slot's old frame will retry) and returns the address of the slot's bytes inside the
 *     block's mapping.  Readers are NOT sent to this slot yet: uid has not moved, read_frame keeps serving the last complete frame.
 *     SUCCESS / FRAMEWORK_DELETED / negative (too large, a deferred write already open: one writer per block, one write at a time).
 *   cmf_write_commit(block, ticket, acquisition_time, planes, plane_count): call once the bytes are in the slot.  Writes the metadata
 *     (planes[i].data is ignored, sizes and names are used exactly as write_frame_planes uses them), the second sequence number, bumps
 *     uid and wakes waiting readers - the second half of write_frame_planes.
 *   cmf_write_abort(block, ticket): the copy failed or was given up: closes the slot without publishing it.
 * write_frame / write_frame_planes on a block with a deferred write open fail (negative status). */
int cmf_write_begin(Block* block, uint64_t entry_bytes, void** payload, uint64_t* ticket);
int cmf_write_commit(Block* block, uint64_t ticket, uint64_t acquisition_time, const FramePlaneWrite* planes, size_t plane_count);
int cmf_write_abort(Block* block, uint64_t ticket);

#ifdef __cplusplus
}
#endif
#endif
