// ThreadSanitizer harness for the shared-memory seqlock (csrc/cmf.cpp), built with -fsanitize=thread by tests/test_cmf.py:
// a writer thread (every fourth frame written in the deferred form: slot opened, payload moved by another thread, committed or given up) and three reader threads (one polling, one blocking on the condition variable, one that peeks and copies the payload itself) work on ONE mapping of a block, so
// the sanitizer sees every access of both sides.  Readers check what they accept: payload all one value that matches the frame's
// acquisition time, plane metadata intact, time never going backwards.  Exit code 0 = no torn frame accepted; the sanitizer adds
// its own verdict (TSAN_OPTIONS exitcode).  SURVEY section 5 asks for this build of the replacement; the reference has no such test.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
#include "../../include/camera_message_framework_c.h"

static std::atomic<int> g_bad{0};
static std::atomic<bool> g_done{false};
static long g_deferred = 0;

static void reader(Block* b, bool blocking, long* accepted)
{
    Frame* f = create_frame();
    uint64_t last_t = 0;
    while (!g_done.load(std::memory_order_acquire)) {
        const int st = read_frame(b, f, blocking);
        if (st == FRAMEWORK_DELETED) break;
        if (st != SUCCESS) continue;
        const unsigned char* d = static_cast<const unsigned char*>(f->data);
        const unsigned char want = (unsigned char)(f->acquisition_time % 251);
        bool ok = f->total_size == 2 * 4096 && f->plane_count == 2 && f->planes[1].offset == 4096 && strcmp(f->planes[1].name, "second") == 0 &&
                  f->acquisition_time >= last_t;
        for (size_t i = 0; ok && i < f->total_size; i++) ok = d[i] == want;
        if (!ok) { g_bad.fetch_add(1); fprintf(stderr, "torn or inconsistent frame accepted at t=%llu\n", (unsigned long long)f->acquisition_time); }
        last_t = f->acquisition_time;
        ++*accepted;
    }
    delete_frame(f);
}

// The consumer that moves the payload itself (cmf_peek_frame / cmf_peek_validate: the runtime's copy engine takes the frame out of the
// slot into device memory).  Here the "engine" is a slow byte-wise copy with relaxed atomic loads - slow on purpose, so that the writer
// laps the ring while it runs - and only copies that validate afterwards may be used.
static void peek_reader(Block* b, long* accepted, long* torn)
{
    Frame* f = create_frame();
    std::vector<unsigned char> mine(2 * 4096);
    uint64_t last_t = 0;
    while (!g_done.load(std::memory_order_acquire)) {
        const void* payload = nullptr;
        uint64_t ticket = 0;
        const int st = cmf_peek_frame(b, f, &payload, &ticket);
        if (st == FRAMEWORK_DELETED) break;
        if (st != SUCCESS) continue;
        const size_t n = f->total_size <= mine.size() ? f->total_size : mine.size();
        const unsigned char* src = static_cast<const unsigned char*>(payload);
        for (size_t i = 0; i < n; i++) mine[i] = __atomic_load_n(src + i, __ATOMIC_RELAXED);
        const int ok_copy = cmf_peek_validate(b, f->uid, ticket);
        if (ok_copy == 0) { ++*torn; continue; }           // lapped: the copy is discarded, the next peek finds a newer frame
        const unsigned char want = (unsigned char)(f->acquisition_time % 251);
        bool ok = ok_copy == 1 && f->total_size == 2 * 4096 && f->plane_count == 2 && f->planes[1].offset == 4096 &&
                  strcmp(f->planes[1].name, "second") == 0 && f->acquisition_time >= last_t;
        for (size_t i = 0; ok && i < n; i++) ok = mine[i] == want;
        if (!ok) { g_bad.fetch_add(1); fprintf(stderr, "peek reader: torn or inconsistent frame validated at t=%llu\n", (unsigned long long)f->acquisition_time); }
        last_t = f->acquisition_time;
        ++*accepted;
    }
    delete_frame(f);
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 20000;
    char name[64];
    snprintf(name, sizeof name, "tsan%d", (int)getpid());
    Block* w = create_block(name, 2 * 4096);
    if (!w) { fprintf(stderr, "create_block failed\n"); return 2; }
    Block* r = open_block(name);
    if (!r) { fprintf(stderr, "open_block failed\n"); return 2; }
    long a1 = 0, a2 = 0, a3 = 0, torn3 = 0;
    std::thread t1(reader, r, false, &a1), t2(reader, r, true, &a2), t3(peek_reader, r, &a3, &torn3);
    std::vector<unsigned char> p0(4096), p1(4096);
    for (int t = 1; t <= n; t++) {
        memset(p0.data(), t % 251, p0.size());
        memset(p1.data(), t % 251, p1.size());
        FramePlaneWrite planes[2] = {{64, 64, 1, 1, p0.data(), "first"}, {64, 64, 1, 1, p1.data(), "second"}};
        if (t % 4 == 0) {
            // The deferred form (posts by DMA): the slot is opened, ANOTHER thread - the stand-in for the copy engine - fills it while
            // the readers keep running, the writer commits once that thread is done.  Every 64th such write is given up instead
            // (cmf_write_abort): the next write must reuse the slot without a reader ever accepting the abandoned bytes.
            void* slot = nullptr;
            uint64_t ticket = 0;
            if (cmf_write_begin(w, 2 * 4096, &slot, &ticket) != SUCCESS) { fprintf(stderr, "begin failed\n"); g_bad.fetch_add(1); break; }
            if (write_frame_planes(w, (uint64_t)t, planes, 2) >= 0) { fprintf(stderr, "a plain write went through beside an open deferred write\n"); g_bad.fetch_add(1); }
            const bool give_up = t % 256 == 0;
            std::thread engine([slot, t, give_up] {
                unsigned char* dst = static_cast<unsigned char*>(slot);
                const unsigned char v = give_up ? (unsigned char)((t + 7) % 251) : (unsigned char)(t % 251);   // abandoned bytes are WRONG bytes
                for (size_t i = 0; i < 2 * 4096; i++) __atomic_store_n(dst + i, v, __ATOMIC_RELAXED);
            });
            engine.join();
            FramePlaneWrite meta[2] = {{64, 64, 1, 1, nullptr, "first"}, {64, 64, 1, 1, nullptr, "second"}};
            const int rc = give_up ? cmf_write_abort(w, ticket) : cmf_write_commit(w, ticket, (uint64_t)t, meta, 2);
            if (rc != SUCCESS) { fprintf(stderr, "commit / abort failed: %s\n", cmf_last_error()); g_bad.fetch_add(1); break; }
            g_deferred++;
        } else if (write_frame_planes(w, (uint64_t)t, planes, 2) != SUCCESS) { fprintf(stderr, "write failed\n"); g_bad.fetch_add(1); break; }
        if ((t & 1023) == 0) usleep(200);     // let the blocking reader through now and then
    }
    g_done.store(true, std::memory_order_release);
    // the blocking reader wakes by itself within a second (cond_timedwait) and then sees g_done
    t1.join();
    t2.join();
    t3.join();
    delete_block(r);
    delete_block(w);
    printf("deferred writes (begin / engine thread / commit or abort) %ld; ", g_deferred);
    printf("frames written %d, accepted by the polling reader %ld, by the blocking reader %ld, by the peek reader %ld (%ld copies discarded as lapped), bad %d\n",
           n, a1, a2, a3, torn3, g_bad.load());
    return (g_bad.load() == 0 && a1 + a2 > 0 && a3 > 0) ? 0 : 1;
}
