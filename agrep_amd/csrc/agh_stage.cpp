// agh_stage.cpp -- the host <-> HBM side of libagrep_hip.so: memory mode, files and pipes (the role of
// fill_buf / the residue carry, bitap.c:450-505, sgrep.c:465-471), record output while the input is still
// being read (asearch.c:66-324 prints from inside its block loop), record-aligned shards of a file.
// Kernel sequences over text that is already resident: agh_api.cpp.
#include <atomic>
#include <condition_variable>
#include <mutex>

#include <memory>

#include "agh_internal.h"

// Pinned host memory for what a scan hands back (match entries, record bytes): grown, never shrunk.
static int ensure_emit_pinned(agh_query *q, size_t bytes)
{
    if (q->h_emit_cap >= bytes) return 0;
    if (q->h_emit) (void)hipHostFree(q->h_emit);
    q->h_emit = nullptr;
    q->h_emit_cap = 0;
    const size_t want = bytes + bytes / 4 + 65536;
    HIP_TRY(hipHostMalloc((void **)&q->h_emit, want));
    q->h_emit_cap = want;
    return 0;
}

// Device arrays of a record list of `cap` entries (pos, rec, start, end).
static int ensure_list(agh_query *q, size_t cap, agh_list_out *list)
{
    if (q->match_pos.ensure(cap * sizeof(uint64_t)) || q->match_rec.ensure(cap * sizeof(uint32_t)) ||
        q->match_start.ensure(cap * sizeof(uint64_t)) || q->match_end.ensure(cap * sizeof(uint64_t)))
        return -1;
    list->pos = (uint64_t *)q->match_pos.p;
    list->rec = (uint32_t *)q->match_rec.p;
    list->start = (uint64_t *)q->match_start.p;
    list->end = (uint64_t *)q->match_end.p;
    list->cap = cap;
    list->rec_bytes = 0;
    return 0;
}

// The matches of the text staged in q->staging -> the caller's array: the list is already in file order on
// the device with its bounds (agh_records.hip); packed into agh_match entries there and copied back in one piece.
static int collect_matches(agh_query *q, const agh_result *res, agh_match *matches)
{
    const size_t ns = (size_t)res->n_stored;
    if (!ns) return 0;
    static_assert(sizeof(agh_match) == 3 * sizeof(uint64_t), "agh_match is three 64-bit words");
    if (q->match_out.ensure(ns * sizeof(agh_match))) return -1;
    if (q->match_off.ensure(((ns + 255) / 256 + 2) * sizeof(uint64_t))) return -1;
    HIP_TRY(hipMemsetAsync(q->match_off.p, 0, ((ns + 255) / 256 + 2) * sizeof(uint64_t), nullptr));
    agh_gather_shape g;
    memset(&g, 0, sizeof(g));
    agh_launch_gather_records(q->staging.p, q->staged_len, (const uint64_t *)q->match_start.p, (const uint64_t *)q->match_end.p,
                              (const uint32_t *)q->match_rec.p, (const uint64_t *)q->match_off.p, 0, (uint32_t)ns, g, 0,
                              nullptr, q->match_out.p, nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(matches, q->match_out.p, ns * sizeof(agh_match), hipMemcpyDeviceToHost));
    return 0;
}

static int scan_staged(agh_query *q, uint64_t len, unsigned flags, agh_result *res,
                       agh_match *matches, size_t cap, bool is_first = true, bool is_last = true)
{
    agh_list_out list;
    const bool want = matches && cap;
    if (want && ensure_list(q, cap, &list)) return -1;
    q->staged_len = len;
    q->staged_first = is_first;                 // (agh_rescan_staged scans the same shard again)
    q->staged_last = is_last;
    if (agh_scan_device_impl(q, q->staging.p, len, nullptr, flags, res, want ? &list : nullptr, is_first, is_last)) return -1;
    return want ? collect_matches(q, res, matches) : 0;
}

extern "C" int agh_scan_buffer(agh_query *q, const unsigned char *text, size_t len,
                               unsigned flags, agh_result *res, agh_match *matches, size_t cap)
{
    if (!q || !res || (!text && len)) return fail("null argument");
    agh_refresh_tuning(q);
    if (q->staging.ensure(((len + 15) & ~(size_t)15) + 16)) return -1;
    if (len) HIP_TRY(hipMemcpy(q->staging.p, text, len, hipMemcpyHostToDevice));
    return scan_staged(q, len, flags, res, matches, cap);
}


// File mode: read() lands directly in pinned memory and is copied to HBM asynchronously while the next
// chunks are being read (a ring of AGH_PIN_RING chunks) -- the staging role of fill_buf
// (bitap.c:450-477), without an intermediate pageable copy.
// bytes per pinned chunk of the ring (AGH_STAGE_CHUNK_MB: A/B switch, 1..256)
static size_t stage_chunk_bytes()
{
    static const size_t v = [] {
        const char *e = getenv("AGH_STAGE_CHUNK_MB");
        long mb = e && *e ? atol(e) : 16;       // (16: one process on a 4 GiB file 0.188-0.200 s, 32: 0.197-0.201, 64: 0.215-0.235;
        if (mb < 1 || mb > 256) mb = 16;        //  profiles/r05_startup_ab.log -- pinning costs 5.4 ms per 32 MiB)
        return (size_t)mb << 20;
    }();
    return v;
}
#define AGH_STAGE_CHUNK stage_chunk_bytes()
static const size_t AGH_SEG_PFX = 64;       // bytes in front of the text of a device segment of the stream (pipe_scan)

// Reader threads that live as long as their fd_reader: a chunk of the ring is 32 MiB, read in ~1 ms -- sixteen
// std::thread creations and joins per chunk were a third of that (4 GiB: 33 ms per GiB, 30 GB/s).
struct read_pool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    uint64_t gen = 0;
    unsigned pending = 0;
    bool quit = false;
    // the job: thread t reads [lo(t), hi(t)) of `take` bytes at file offset pos into dst
    int fd = -1;
    unsigned char *dst = nullptr;
    off_t pos = 0;
    size_t take = 0, piece = 0;
    std::vector<ssize_t> done;
    std::vector<int> err;

    void work(unsigned t)
    {
        const size_t lo = std::min(take, (size_t)t * piece), hi = std::min(take, lo + piece);
        size_t at = lo;
        while (at < hi) {
            ssize_t r = pread(fd, dst + at, hi - at, pos + (off_t)at);
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) { err[t] = errno; break; }       // EIO, ESTALE ...: not a truncation
            if (r == 0) break;                          // the file got shorter meanwhile
            at += (size_t)r;
        }
        done[t] = (ssize_t)(at - lo);
    }
    void run(unsigned t)
    {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_job.wait(lk, [&] { return gen != seen || quit; });
                if (quit) return;
                seen = gen;
            }
            work(t);
            std::unique_lock<std::mutex> lk(mu);
            if (--pending == 0) cv_done.notify_one();
        }
    }
    void start(unsigned n)
    {
        done.assign(n, 0);
        err.assign(n, 0);
        for (unsigned t = 0; t < n; ++t) th.emplace_back([this, t] { run(t); });
    }
    void dispatch()
    {
        std::unique_lock<std::mutex> lk(mu);
        std::fill(done.begin(), done.end(), 0);
        std::fill(err.begin(), err.end(), 0);
        pending = (unsigned)th.size();
        ++gen;
        cv_job.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
    ~read_pool()
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            quit = true;
            cv_job.notify_all();
        }
        for (auto &x : th) x.join();
    }
};

struct fd_reader {
    std::unique_ptr<read_pool> pool;
    int fd = -1;
    bool regular = false;
    off_t pos = 0;                  // regular files: next byte to read
    uint64_t left = 0;              // regular files: bytes still to read
    bool ranged = false;            // an explicit range: never read past it
    unsigned n_readers = 1;

    int open_fd(int fd_, bool with_range, uint64_t begin, uint64_t end, unsigned readers_override = 0)
    {
        fd = fd_;
        ranged = with_range;
        struct stat sb;
        if (fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode)) {
            off_t cur = with_range ? (off_t)begin : lseek(fd, 0, SEEK_CUR);
            if (cur < 0) cur = 0;
            uint64_t stop = with_range ? std::min<uint64_t>(end, (uint64_t)sb.st_size) : (uint64_t)sb.st_size;
            regular = true;
            pos = cur;
            left = (uint64_t)cur < stop ? stop - (uint64_t)cur : 0;
        } else if (with_range) {
            return fail("a byte range needs a seekable regular file");
        }
        n_readers = std::thread::hardware_concurrency();
        if (n_readers > 16) n_readers = 16;
        if (readers_override) n_readers = readers_override;      // AGH_READERS
        if (n_readers < 1) n_readers = 1;
        return 0;
    }
    uint64_t size_hint() const { return regular ? left : 0; }

    // up to `want` bytes into dst; 0 = end of input; -1 = error (message set)
    ssize_t fill(unsigned char *dst, size_t want)
    {
        size_t got = 0;
        if (regular && left >= want / 2 && n_readers > 1) {
            const size_t take = (size_t)std::min<uint64_t>(want, left);
            const size_t piece = ((take + n_readers - 1) / n_readers + 4095) & ~(size_t)4095;
            if (!pool) {
                pool.reset(new read_pool());
                pool->start(n_readers);
            }
            pool->fd = fd;
            pool->dst = dst;
            pool->pos = pos;
            pool->take = take;
            pool->piece = piece;
            pool->dispatch();
            for (unsigned t = 0; t < n_readers; ++t)
                if (pool->err[t]) return fail("read failed: %s", strerror(pool->err[t]));
            // contiguous prefix that really arrived (a file truncated meanwhile ends the scan)
            for (unsigned t = 0; t < n_readers; ++t) {
                const size_t lo = std::min(take, (size_t)t * piece), hi = std::min(take, lo + piece);
                got += (size_t)pool->done[t];
                if ((size_t)pool->done[t] < hi - lo) break;
            }
            pos += (off_t)got;
            left -= std::min<uint64_t>(left, got);
            if (got < take) left = 0;
            if (!ranged) (void)lseek(fd, pos, SEEK_SET);
            return (ssize_t)got;
        }
        if (regular) want = (size_t)std::min<uint64_t>(want, left);
        while (got < want) {
            ssize_t r = regular ? pread(fd, dst + got, want - got, pos + (off_t)got)
                                : read(fd, dst + got, want - got);
            if (r < 0) {
                if (errno == EINTR) continue;
                return fail("read failed: %s", strerror(errno));
            }
            if (r == 0) break;
            got += (size_t)r;
        }
        if (regular) {
            pos += (off_t)got;
            left -= std::min<uint64_t>(left, got);
            if (!ranged) (void)lseek(fd, pos, SEEK_SET);
        }
        return (ssize_t)got;
    }
};

// The copy stream (H2D of chunk i under the read of chunk i + 1) and the ring's events.  An input that fits one
// chunk has nothing to overlap: its copy goes to the default stream, and the 9 ms a stream costs to create (an HSA
// queue; profiles/r05_startup.log) are not spent on a 1 MiB file.
static int ensure_stage_resources(agh_query *q, bool need_stream)
{
    if (!q->pinned_ev[0])
        for (int b = 0; b < AGH_PIN_RING; ++b)
            HIP_TRY(hipEventCreateWithFlags(&q->pinned_ev[b], hipEventDisableTiming));
    if (need_stream && !q->stage_stream) HIP_TRY(hipStreamCreateWithFlags(&q->stage_stream, hipStreamNonBlocking));
    return 0;
}

// Pinned chunk b of the ring, at least `bytes` large.  Allocated on first use and no larger than the input
// needs: pinning 4 x 32 MiB costs tens of milliseconds, which a 1 MiB file should not pay.
static int ensure_pinned(agh_query *q, int b, size_t bytes, uint64_t size_hint)
{
    if (q->pinned_cap[b] >= bytes) return 0;
    size_t want = AGH_STAGE_CHUNK;
    if (size_hint && size_hint < AGH_STAGE_CHUNK) want = ((size_t)size_hint + 65535) & ~(size_t)65535;
    if (want < bytes) want = bytes;
    if (q->pinned[b]) (void)hipHostFree(q->pinned[b]);
    q->pinned[b] = nullptr;
    q->pinned_cap[b] = 0;
    HIP_TRY(hipHostMalloc((void **)&q->pinned[b], want));
    q->pinned_cap[b] = want;
    return 0;
}

// Round 5: the copy stream (its hardware queue: ~8 ms) is created by a helper thread while the main thread pins
// chunk 0 and reads the first bytes of the file into it.
struct stage_prep {
    agh_query *q = nullptr;
    int device = 0;
    std::atomic<int> stream_ready{0};           // 1: there, -1: failed (the main thread tries itself)
    std::thread th;
    void start()
    {
        th = std::thread([this] {
            (void)hipSetDevice(device);
            // (pinning chunks 1..3 of the ring here as well was tried: page pinning and the main thread's read() into
            // chunk 0 fight over the address space -- the first read took 14 ms instead of 1, profiles/r05_startup_ab.log)
            stream_ready.store((q->stage_stream || hipStreamCreateWithFlags(&q->stage_stream, hipStreamNonBlocking) == hipSuccess) ? 1 : -1,
                               std::memory_order_release);
        });
    }
    bool active() const { return th.joinable(); }
    void wait_stream() { while (!stream_ready.load(std::memory_order_acquire)) std::this_thread::yield(); }
    ~stage_prep() { if (th.joinable()) th.join(); }
};

// ---------------------------------------------------------------------------------------
// matched records of one resident segment -> the caller's emit function
// ---------------------------------------------------------------------------------------
struct rec_sink {
    agh_emit_fn emit;
    void *ctx;
    bool want_bytes;
    bool head_delim, tail_delim;    // AGH_EMIT_HEAD_DELIM / AGH_EMIT_TAIL_DELIM
};

static rec_sink make_sink(agh_emit_fn emit, void *ctx, unsigned flags)
{
    rec_sink s = {emit, ctx, !(flags & AGH_NO_BYTES), (flags & AGH_EMIT_HEAD_DELIM) != 0, (flags & AGH_EMIT_TAIL_DELIM) != 0};
    return s;
}

// Numbered scan of text[0, n) (any length: cut into kernel segments by agh_scan_device_impl) with the list of
// matched records left on the device IN FILE ORDER together with their bounds (agh_records.hip); then, per piece
// of the list, the record bytes are gathered back to back on the device, bytes and agh_match entries come back
// in ONE copy into pinned memory, and emit() is called.  Two host synchronisations for a list that fits one
// piece: the scan's counters, the copy.  Offsets and record numbers are shifted by base_off / rec_off (the
// segment's place in its file).
#define AGH_EMIT_PIECE_BYTES ((uint64_t)64 << 20)   // record bytes per emit() call (whole blocks of 256 records)
#define AGH_EMIT_PIECE_RECS ((size_t)1 << 20)

static int emit_records(agh_query *q, const void *d_text, uint64_t n, unsigned flags, bool first, bool last,
                        uint64_t base_off, uint64_t rec_off, const rec_sink &sink, agh_result *r, bool *stop)
{
    flags &= ~(AGH_COUNT | AGH_FILENAMEONLY | AGH_NO_BYTES | AGH_EMIT_HEAD_DELIM | AGH_EMIT_TAIL_DELIM);
    size_t mcap = std::max<size_t>(q->match_cap_hint, (size_t)1 << 16);
    agh_list_out list;
    for (;;) {
        if (ensure_list(q, mcap, &list)) return -1;
        if (agh_scan_device_impl(q, d_text, n, nullptr, flags, r, &list, first, last)) return -1;
        if (!r->truncated) break;
        mcap = std::max<size_t>(mcap * 4, (size_t)r->n_matched + 1024);     // the text is still resident: again
    }
    q->match_cap_hint = std::max<size_t>(q->match_cap_hint, (size_t)r->n_stored + (size_t)r->n_stored / 4);
    const size_t ns = (size_t)r->n_stored;
    if (!ns) return 0;
    // what a record occupies in the output (agh_gather_shape): with the delimiter in front of it / behind it on request
    agh_gather_shape g;
    memset(&g, 0, sizeof(g));
    g.base_off = base_off;
    g.pre_dlen = (sink.want_bytes && sink.head_delim) ? (uint32_t)q->dlen : 0u;
    g.post_dlen = (sink.want_bytes && sink.tail_delim) ? (uint32_t)q->dlen : 0u;
    memcpy(g.dbytes, q->delim, (size_t)q->dlen);
    // (an upper bound for one piece: only the first record of the input can have less than dlen bytes in front of it)
    const uint64_t total = sink.want_bytes ? list.rec_bytes + (uint64_t)ns * (g.pre_dlen + g.post_dlen) : 0;
    const size_t n_blocks = (ns + 255) / 256;
    if (q->match_off.ensure((n_blocks + 2) * sizeof(uint64_t))) return -1;
    uint64_t *d_blk = (uint64_t *)q->match_off.p;
    agh_launch_len_offsets(list.start, list.end, 0, (uint32_t)ns, g, d_blk, nullptr);
    HIP_TRY(hipGetLastError());
    const bool one_piece = ns <= AGH_EMIT_PIECE_RECS && total <= AGH_EMIT_PIECE_BYTES;
    std::vector<uint64_t> h_blk;
    if (!one_piece) {                           // the host cuts pieces at block boundaries: it needs the offsets
        h_blk.resize(n_blocks + 1);
        HIP_TRY(hipMemcpy(h_blk.data(), d_blk, (n_blocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    for (size_t b0 = 0; b0 < n_blocks && !*stop;) {
        size_t b1 = n_blocks;
        uint64_t piece_bytes = total;
        if (!one_piece) {
            b1 = b0 + 1;                        // at least one block, then as many as fit
            while (b1 < n_blocks && (b1 + 1 - b0) * 256 <= AGH_EMIT_PIECE_RECS &&
                   (!sink.want_bytes || h_blk[b1 + 1] - h_blk[b0] <= AGH_EMIT_PIECE_BYTES))
                ++b1;
            piece_bytes = sink.want_bytes ? h_blk[b1] - h_blk[b0] : 0;
        }
        const size_t r0 = b0 * 256, cnt = std::min(ns, b1 * 256) - r0;
        const size_t m_bytes = cnt * sizeof(agh_match);
        const size_t m_room = (m_bytes + 63) & ~(size_t)63;
        if (q->match_out.ensure(m_room + (size_t)piece_bytes + 64)) return -1;
        if (ensure_emit_pinned(q, m_room + (size_t)piece_bytes)) return -1;
        unsigned char *d_out = (unsigned char *)q->match_out.p;
        agh_launch_gather_records(d_text, n, list.start, list.end, list.rec, d_blk + b0, (uint32_t)r0, (uint32_t)cnt, g,
                                  rec_off, (sink.want_bytes && piece_bytes) ? d_out + m_room : nullptr, d_out, nullptr);
        HIP_TRY(hipGetLastError());
        // entries and bytes are neighbours on the device: one copy
        HIP_TRY(hipMemcpyAsync(q->h_emit, d_out, m_room + (size_t)piece_bytes, hipMemcpyDeviceToHost, nullptr));
        HIP_TRY(hipStreamSynchronize(nullptr));
        const agh_match *hm = (const agh_match *)q->h_emit;
        if (one_piece && g.pre_dlen && hm[0].start < g.pre_dlen) piece_bytes -= g.pre_dlen - hm[0].start;   // (the bound above)
        if (sink.emit(sink.ctx, hm, cnt, sink.want_bytes ? q->h_emit + m_room : nullptr,
                      sink.want_bytes ? (size_t)piece_bytes : 0))
            *stop = true;
        b0 = b1;
    }
    return 0;
}

static void add_result(agh_result *sum, const agh_result &r)
{
    sum->n_matched += r.n_matched;
    sum->n_records += r.n_records;
    sum->n_candidates += r.n_candidates;
    sum->n_bytes += r.n_bytes;
    sum->n_stored += r.n_stored;
    sum->device_ms += r.device_ms;
    sum->sweep_ms += r.sweep_ms;
    sum->sweep_launches += r.sweep_launches;
    sum->lean_reruns += r.lean_reruns;
    sum->n_segments += r.n_segments;
    sum->fused_segments += r.fused_segments;
    sum->copied_segments += r.copied_segments;
    if (r.engine) sum->engine = r.engine;
}

extern "C" int agh_scan_device_emit(agh_query *q, const void *dev_text, size_t len, unsigned flags,
                                    agh_result *res, agh_emit_fn emit, void *ctx)
{
    if (!q || !res || !emit) return fail("null argument");
    agh_refresh_tuning(q);
    memset(res, 0, sizeof(*res));
    res->n_bytes = len;
    if (!len) return 0;
    // one emit() per piece of at most AGH_EMIT_SEG_MB (the match arrays and the gathered bytes of a piece
    // are what the call holds at a time); pieces end where a record ends
    rec_sink sink = make_sink(emit, ctx, flags);
    bool stop = false;
    agh_result r;
    if (emit_records(q, dev_text, len, flags, true, true, 0, 0, sink, &r, &stop)) return -1;
    *res = r;
    if (stop) res->truncated = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------
// the streaming pipeline: read -> pinned ring -> one of two device segments | scan the other one
// ---------------------------------------------------------------------------------------
// The input passes through two device segments (AGH_STREAM_SEG_MB each, default 1 GiB).  While the host
// reads and copies segment i+1, a worker thread scans segment i -- count-only (-c, -l) or with its matched
// records going to emit() -- so HBM use is bounded whatever the input size, a pipe never needs a second
// copy, records come out while the input is still being read, and -l stops READING at the first segment
// with a match (asearch.c:130-161: print the name, return).  A segment is cut after the last delimiter
// that has arrived; the unfinished record is carried to the front of the next segment (fill_buf's
// residue carry, bitap.c:450-477, sgrep.c:465-471, at HBM scale).
struct pipe_worker {
    agh_query *q = nullptr;
    unsigned flags = 0;
    const rec_sink *sink = nullptr;
    int device = 0;
    std::mutex mu;
    std::condition_variable cv;
    bool has_job = false, busy = false, quit = false;
    const void *text = nullptr;
    uint64_t len = 0, base_off = 0;
    bool first = false, last = false;
    // owned by the worker while busy
    agh_result total;
    uint64_t rec_off = 0;
    std::atomic<int> rc{0};
    std::atomic<bool> stop{false};
    char err[512];
    std::thread th;

    void run()
    {
        (void)hipSetDevice(device);
        // (the code object of the library is loaded at the first launch -- ~7 ms; this thread has nothing to do until
        // the first segment is in HBM)
        agh_warm_core_module();
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return has_job || quit; });
            if (!has_job && quit) return;
            has_job = false;
            lk.unlock();
            agh_result r;
            memset(&r, 0, sizeof(r));
            int rc1;
            bool stop1 = false;
            agh_timeline("worker: scan of a segment starts");
            if (sink) rc1 = emit_records(q, text, len, flags, first, last, base_off, rec_off, *sink, &r, &stop1);
            else rc1 = agh_scan_device_impl(q, text, len, nullptr, flags, &r, nullptr, first, last);
            if (rc1) {
                rc = rc1;
                snprintf(err, sizeof(err), "%s", agh_last_error());
            } else {
                add_result(&total, r);
                rec_off += r.n_records;
                if (stop1 || (!sink && (flags & AGH_FILENAMEONLY) && total.n_matched)) stop = true;
            }
            agh_timeline("worker: scan of the segment done");
            lk.lock();
            busy = false;
            cv.notify_all();
        }
    }
    void wait_idle()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !busy; });
    }
    void submit(const void *t, uint64_t n, uint64_t off, bool f, bool l)
    {
        std::unique_lock<std::mutex> lk(mu);
        text = t; len = n; base_off = off; first = f; last = l;
        has_job = true;
        busy = true;
        cv.notify_all();
    }
    void finish()
    {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy; });
            quit = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
    }
};

// Where a segment may end inside the chunk that has just arrived: the end (exclusive) of the last
// delimiter occurrence in chunk[0, got) that is certainly a record end, 0 if there is none.  One byte:
// any occurrence.  Several bytes: occurrences are read leftmost non-overlapping (asearch.c:54-57), so one
// that no other occurrence overlaps from the left is selected whatever came before it.
static size_t last_record_end(const agh_query *q, const unsigned char *chunk, size_t got)
{
    const int dl = q->dlen;
    auto same = [&](unsigned char c, int j) -> bool {
        if (q->delim_fold && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        return c == q->delim[j];
    };
    if (dl == 1 && !q->delim_fold) {
        const unsigned char *hit = (const unsigned char *)memrchr(chunk, q->delim[0], got);
        return hit ? (size_t)(hit - chunk) + 1 : 0;
    }
    auto occurs = [&](size_t i) -> bool {      // the delimiter at chunk[i, i + dl)
        for (int j = 0; j < dl; ++j)
            if (!same(chunk[i + (size_t)j], j)) return false;
        return true;
    };
    if (got < (size_t)dl) return 0;
    for (size_t i = got - (size_t)dl + 1; i-- > 0;) {
        if (!occurs(i)) continue;
        bool clear = true;                      // no occurrence starts in (i - dl, i)
        for (int b = 1; b < dl && clear; ++b) {
            if (i < (size_t)b) { clear = false; break; }   // its left context is not in this chunk
            if (occurs(i - (size_t)b)) clear = false;
        }
        if (clear) return i + (size_t)dl;
    }
    return 0;
}

static int pipe_scan(agh_query *q, fd_reader &rd, unsigned flags, agh_result *res, bool is_first, bool is_last,
                     const rec_sink *sink)
{
    memset(res, 0, sizeof(*res));
    const uint64_t seg_cap = q->tune.stream_seg_mb << 20;
    const bool early = !sink && (flags & AGH_FILENAMEONLY) != 0;
    // -l: small first segments (1 MiB, x4 each time) so that a hit near the top of a file is reported after
    // the first megabyte has been read and scanned, whatever the engine costs
    uint64_t target = early ? std::min<uint64_t>(seg_cap, (uint64_t)1 << 20) : seg_cap;
    const uint64_t hint = rd.size_hint();
    // (a file smaller than a chunk is read -- and its pinned buffer sized -- as one piece of its own size)
    const uint64_t chunk_cap = (hint && hint < AGH_STAGE_CHUNK) ? ((hint + 65535) & ~(uint64_t)65535) : AGH_STAGE_CHUNK;
    dev_buf *seg[2] = {&q->staging, &q->staging_b};
    // every device segment keeps AGH_SEG_PFX bytes in front of its text: the last bytes of the segment before it,
    // i.e. the delimiter in front of its first record (AGH_EMIT_HEAD_DELIM reads it from there)
    const uint64_t want0 = std::min<uint64_t>(seg_cap, hint ? hint : seg_cap) + 2 * chunk_cap + 64 + AGH_SEG_PFX;
    if (seg[0]->ensure(want0)) return -1;
    if ((!hint || hint > seg_cap) && seg[1]->ensure(want0)) return -1;      // (a small file needs one segment)
    agh_timeline("pipe_scan: device segments allocated");
    q->staged_len = 0;                          // what stays in HBM is not the whole input
    stage_prep prep;                            // (joined on every way out, after the worker)
    // (a file of one chunk: its single copy goes through the null stream -- a stream of its own costs ~8 ms)
    const bool need_stream = !(hint && hint <= AGH_STAGE_CHUNK);
    if (need_stream && !early && hint > 2 * chunk_cap && !q->stage_stream) {
        prep.q = q;
        if (hipGetDevice(&prep.device) != hipSuccess) prep.device = 0;
        prep.start();
    } else if (need_stream && !q->stage_stream && ensure_stage_resources(q, true)) {
        return -1;
    }

    pipe_worker w;
    w.q = q;
    w.flags = flags;
    w.sink = sink;
    memset(&w.total, 0, sizeof(w.total));
    w.err[0] = 0;
    if (hipGetDevice(&w.device) != hipSuccess) w.device = 0;
    w.th = std::thread([&w] { w.run(); });

    int rc = 0;
    uint64_t used = 0, base_off = 0;            // bytes staged in the current segment / its offset in the input
    int cur = 0, b = 0;
    bool first = is_first, eof = false;
    bool busy[AGH_PIN_RING];
    for (int i = 0; i < AGH_PIN_RING; ++i) busy[i] = false;
    auto bail = [&](int code) { w.finish(); return code; };
#define PIPE_TRY(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            w.finish();                                                                                  \
            return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__);    \
        }                                                                                                \
    } while (0)
    while (!eof && !w.stop && !w.rc) {
        if (busy[b]) PIPE_TRY(hipEventSynchronize(q->pinned_ev[b]));
        busy[b] = false;
        // (-l reads no further ahead than its current segment target)
        // (chunks no larger than a segment: AGH_STREAM_SEG_MB below 32 exercises the residue carry in tests)
        const size_t ask = early ? (size_t)std::min<uint64_t>(AGH_STAGE_CHUNK, std::max<uint64_t>(target > used ? target - used : 0, 65536))
                                 : (size_t)std::min<uint64_t>(chunk_cap, seg_cap);
        if (ensure_pinned(q, b, ask, hint)) return bail(-1);
        if (!base_off && !used) agh_timeline("pipe_scan: first pinned chunk ready");
        const ssize_t got = rd.fill(q->pinned[b], ask);
        if (!base_off && used == 0) agh_timeline("pipe_scan: first chunk read");
        if (got < 0) return bail(-1);
        if (got == 0) eof = true;
        if (!base_off && !used) {               // agh_input_head(): known before the first emit()
            q->input_head_len = (uint32_t)std::min<size_t>((size_t)got, sizeof(q->input_head));
            memcpy(q->input_head, q->pinned[b], q->input_head_len);
        }
        // (the helper thread's stream: there by now, or the main thread makes it -- also when nothing was read, e.g. a
        // file truncated after fstat(): q->stage_stream is synchronised below either way and must not be read while
        // the helper is still writing it)
        if (prep.active()) prep.wait_stream();
        if (need_stream && !q->stage_stream && ensure_stage_resources(q, true)) return bail(-1);
        if (got > 0) {
            if (AGH_SEG_PFX + used + (uint64_t)got + 64 > seg[cur]->cap) {    // a record longer than the segment: grow
                dev_buf bigger;
                if (bigger.ensure((AGH_SEG_PFX + used + (uint64_t)got) * 2 + 64)) return bail(-1);
                PIPE_TRY(hipStreamSynchronize(q->stage_stream));
                PIPE_TRY(hipMemcpy(bigger.p, seg[cur]->p, AGH_SEG_PFX + used, hipMemcpyDeviceToDevice));
                seg[cur]->release();
                *seg[cur] = bigger;
            }
            PIPE_TRY(hipMemcpyAsync((unsigned char *)seg[cur]->p + AGH_SEG_PFX + used, q->pinned[b], (size_t)got,
                                    hipMemcpyHostToDevice, q->stage_stream));
            PIPE_TRY(hipEventRecord(q->pinned_ev[b], q->stage_stream));
            busy[b] = true;
            used += (uint64_t)got;
        }
        if (!eof && used < target) { b = (b + 1) % AGH_PIN_RING; continue; }
        // cut after the last record end of the chunk that has just arrived
        uint64_t cut = used;
        const unsigned char *tail_src = nullptr;
        uint64_t tail_len = 0;
        if (!eof) {
            const size_t idx = last_record_end(q, q->pinned[b], (size_t)got);
            if (!idx) { b = (b + 1) % AGH_PIN_RING; continue; }     // no record ends here yet: the segment grows
            cut = used - (uint64_t)got + idx;
            tail_src = q->pinned[b] + idx;
            tail_len = (uint64_t)got - idx;
        }
        PIPE_TRY(hipStreamSynchronize(q->stage_stream));
        w.wait_idle();                          // the scan of the segment before this one (the other buffer)
        if (w.rc || w.stop) break;
        if (cut) {
            w.submit((unsigned char *)seg[cur]->p + AGH_SEG_PFX, cut, base_off, first, eof && is_last);
            first = false;
        }
        base_off += cut;
        if (!eof) {
            // the unfinished record opens the next segment, in the buffer the worker has just left
            const int nxt = cur ^ 1;
            if (seg[nxt]->ensure(std::max<uint64_t>(want0, tail_len + 2 * chunk_cap + 64 + AGH_SEG_PFX))) return bail(-1);
            // ... behind the bytes that end this segment (its own prefix included when the segment is shorter)
            PIPE_TRY(hipMemcpyAsync(seg[nxt]->p, (unsigned char *)seg[cur]->p + cut, AGH_SEG_PFX, hipMemcpyDeviceToDevice,
                                    q->stage_stream));
            if (tail_len) {
                PIPE_TRY(hipMemcpyAsync((unsigned char *)seg[nxt]->p + AGH_SEG_PFX, tail_src, (size_t)tail_len, hipMemcpyHostToDevice, q->stage_stream));
                PIPE_TRY(hipEventRecord(q->pinned_ev[b], q->stage_stream));
                busy[b] = true;
            }
            cur = nxt;
        }
        used = tail_len;
        if (early && target < seg_cap) target = std::min<uint64_t>(seg_cap, target * 4);
        b = (b + 1) % AGH_PIN_RING;
    }
#undef PIPE_TRY
    agh_timeline("pipe_scan: input read, waiting for the last scan");
    if (prep.active()) prep.th.join();
    w.finish();
    (void)hipStreamSynchronize(q->stage_stream);
    agh_timeline("pipe_scan: done");
    if (w.rc) {
        agh_fail("%s", w.err);
        rc = -1;
    }
    *res = w.total;
    if (sink && w.stop) res->truncated = 1;     // the caller's emit() asked to stop
    return rc;
}

static int scan_fd_impl(agh_query *q, int fd, bool with_range, uint64_t begin, uint64_t end,
                        unsigned flags, agh_result *res, agh_match *matches, size_t cap, const rec_sink *sink)
{
    if (!q || !res) return fail("null argument");
    if (fd < 0) return fail("agh_scan_fd needs fd >= 0 (memory mode is agh_scan_buffer)");
    if (with_range && end < begin) return fail("empty byte range");
    agh_timeline("scan_fd: start");
    fd_reader rd;
    agh_refresh_tuning(q);
    if (rd.open_fd(fd, with_range, begin, end, q->tune.readers)) return -1;
    if (ensure_stage_resources(q, false)) return -1;       // (the copy stream: pipe_scan, next to its first read)
    agh_timeline("scan_fd: events");
    const bool count_only = (flags & (AGH_COUNT | AGH_FILENAMEONLY)) && !(matches && cap) && !sink;
    // one rank's shard of a file: the virtual head byte / the appended delimiter (asearch.c:69-91)
    // belong to the shards that hold the file's first / last byte
    bool is_first = true, is_last = true;
    if (with_range) {
        struct stat sb;
        is_first = begin == 0;
        is_last = !(fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && end < (uint64_t)sb.st_size);
    }
    if ((count_only || sink) && q->tune.stream)
        return pipe_scan(q, rd, flags, res, is_first, is_last, sink);

    // a match ARRAY wanted (agh_scan_fd with matches / cap; or AGH_STREAM=0): the whole input is staged,
    // then scanned -- agh_fetch_records and agh_rescan_staged work on that copy.  Bounded memory and
    // output while reading: agh_scan_fd_emit.
    if (ensure_stage_resources(q, !(rd.regular && rd.left <= AGH_STAGE_CHUNK))) return -1;
    size_t want = rd.regular ? (size_t)rd.left + 64 : AGH_STAGE_CHUNK * 2;
    if (q->staging.ensure(want + 32)) return -1;
    size_t used = 0;
    int b = 0;
    bool busy[AGH_PIN_RING];
    for (int i = 0; i < AGH_PIN_RING; ++i) busy[i] = false;
    for (;;) {
        if (busy[b]) HIP_TRY(hipEventSynchronize(q->pinned_ev[b]));   // its H2D copy finished
        busy[b] = false;
        if (ensure_pinned(q, b, AGH_STAGE_CHUNK, 0)) return -1;
        const ssize_t r = rd.fill(q->pinned[b], AGH_STAGE_CHUNK);
        if (r < 0) return -1;
        const size_t got = (size_t)r;
        if (got == 0) break;
        if (used + got + 32 > q->staging.cap) {     // unknown length (pipe): grow, keep contents
            dev_buf bigger;
            if (bigger.ensure((used + got) * 2 + 64)) return -1;
            HIP_TRY(hipStreamSynchronize(q->stage_stream));
            if (used) HIP_TRY(hipMemcpy(bigger.p, q->staging.p, used, hipMemcpyDeviceToDevice));
            q->staging.release();
            q->staging = bigger;
        }
        HIP_TRY(hipMemcpyAsync((unsigned char *)q->staging.p + used, q->pinned[b], got,
                               hipMemcpyHostToDevice, q->stage_stream));
        HIP_TRY(hipEventRecord(q->pinned_ev[b], q->stage_stream));
        busy[b] = true;
        used += got;
        b = (b + 1) % AGH_PIN_RING;
    }
    HIP_TRY(hipStreamSynchronize(q->stage_stream));
    if (sink) {                                 // (AGH_STREAM=0 with an emit function)
        bool stop = false;
        memset(res, 0, sizeof(*res));
        if (!used) return 0;
        if (emit_records(q, q->staging.p, used, flags, is_first, is_last, 0, 0, *sink, res, &stop)) return -1;
        if (stop) res->truncated = 1;
        return 0;
    }
    return scan_staged(q, used, flags, res, matches, cap, is_first, is_last);
}

extern "C" int agh_scan_fd(agh_query *q, int fd, unsigned flags, agh_result *res,
                           agh_match *matches, size_t cap)
{
    return scan_fd_impl(q, fd, false, 0, 0, flags, res, matches, cap, nullptr);
}

extern "C" int agh_scan_fd_range(agh_query *q, int fd, uint64_t begin, uint64_t end,
                                 unsigned flags, agh_result *res, agh_match *matches, size_t cap)
{
    return scan_fd_impl(q, fd, true, begin, end, flags, res, matches, cap, nullptr);
}

extern "C" size_t agh_input_head(const agh_query *q, unsigned char *out, size_t cap)
{
    if (!q || !out) return 0;
    const size_t n = std::min<size_t>(cap, q->input_head_len);
    memcpy(out, q->input_head, n);
    return n;
}

extern "C" int agh_scan_fd_emit(agh_query *q, int fd, unsigned flags, agh_result *res, agh_emit_fn emit, void *ctx)
{
    if (!emit) return fail("null argument");
    rec_sink sink = make_sink(emit, ctx, flags);
    return scan_fd_impl(q, fd, false, 0, 0, flags, res, nullptr, 0, &sink);
}

extern "C" int agh_scan_fd_range_emit(agh_query *q, int fd, uint64_t begin, uint64_t end, unsigned flags,
                                      agh_result *res, agh_emit_fn emit, void *ctx)
{
    if (!emit) return fail("null argument");
    rec_sink sink = make_sink(emit, ctx, flags);
    return scan_fd_impl(q, fd, true, begin, end, flags, res, nullptr, 0, &sink);
}

// cut a file into nranks record-aligned shards (a record belongs to the shard that
// holds its first byte).  Inner cut r = the nominal offset size * r / nranks if a record starts
// there, else just after the next delimiter -- the rule of agrep_amd/shard.py:record_cuts.
// Host-only code (pread); single-byte delimiters.
extern "C" int agh_shard_cuts_fd(int fd, const unsigned char *delim, int dlen, int nranks,
                                 uint64_t *cuts)
{
    if (fd < 0 || !delim || nranks < 1 || !cuts) return fail("agh_shard_cuts_fd: bad arguments");
    if (dlen < 1 || dlen > AGH_MAX_DELIM) return fail("delimiter length %d outside 1..%d", dlen, AGH_MAX_DELIM);
    // Several bytes: where a record ends must not depend on where the search starts.  That holds for
    // delimiters no proper prefix of which is also a suffix ("\r\n", "From ", "$$$x"): their
    // occurrences cannot overlap, every one is selected (asearch.c:54-57).  "\n\n" in "\n\n\n" is
    // selected by what came before -- such delimiters are not sharded here.
    for (int b = 1; b < dlen; ++b)
        if (memcmp(delim, delim + dlen - b, (size_t)b) == 0)
            return fail("sharding a file by a delimiter that can overlap itself is not supported");
    struct stat sb;
    if (fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode)) return fail("sharding needs a seekable regular file");
    const uint64_t size = (uint64_t)sb.st_size;
    cuts[0] = 0;
    std::vector<unsigned char> buf((1 << 16) + AGH_MAX_DELIM);
    const uint64_t dl = (uint64_t)dlen;
    for (int r = 1; r < nranks; ++r) {
        // size * r / nranks without overflow
        uint64_t nominal = size / (uint64_t)nranks * (uint64_t)r + size % (uint64_t)nranks * (uint64_t)r / (uint64_t)nranks;
        if (nominal < cuts[r - 1]) nominal = cuts[r - 1];
        if (nominal >= size) { cuts[r] = size; continue; }
        if (nominal == 0) { cuts[r] = 0; continue; }
        // the first delimiter occurrence that ENDS at or after the nominal offset: the cut is its end
        // (an occurrence that ends exactly at the nominal offset means a record starts right there)
        uint64_t cut = size;
        for (uint64_t off = nominal >= dl ? nominal - dl : 0; off < size && cut == size;) {
            ssize_t got = pread(fd, buf.data(), buf.size(), (off_t)off);
            if (got < 0 && errno == EINTR) continue;
            if (got < 0) return fail("read failed: %s", strerror(errno));
            if (got < (ssize_t)dl) break;
            for (size_t i = 0; i + dl <= (size_t)got; ++i) {
                if (off + i + dl < nominal) continue;
                if (buf[i] == delim[0] && memcmp(buf.data() + i, delim, (size_t)dl) == 0) { cut = off + i + dl; break; }
            }
            off += (uint64_t)got - (dl - 1);        // keep dlen-1 bytes: an occurrence may straddle the reads
        }
        cuts[r] = cut;
    }
    cuts[nranks] = size;
    return 0;
}

// Scan again what the last agh_scan_fd / agh_scan_buffer staged (e.g. with a larger match
// array after `truncated`, or with other flags) without touching the input again.
extern "C" int agh_rescan_staged(agh_query *q, unsigned flags, agh_result *res,
                                 agh_match *matches, size_t cap)
{
    if (!q || !res) return fail("null argument");
    agh_refresh_tuning(q);
    return scan_staged(q, q->staged_len, flags, res, matches, cap, q->staged_first, q->staged_last);
}

// Bytes of matched records of the most recent agh_scan_fd / agh_scan_buffer, concatenated in
// the order given (no delimiters in between): device-side gather + one D2H copy.
extern "C" int agh_fetch_records(agh_query *q, const agh_match *m, size_t n_matches,
                                 unsigned char *out, size_t out_cap, size_t *out_len)
{
    if (!q || (!m && n_matches) || (!out && out_cap)) return fail("null argument");
    std::vector<uint64_t> st(n_matches), en(n_matches);
    uint64_t total = 0;
    for (size_t i = 0; i < n_matches; ++i) {
        if (m[i].end < m[i].start || m[i].end > q->staged_len)
            return fail("match %zu lies outside the staged text", i);
        st[i] = m[i].start;
        en[i] = m[i].end;
        total += m[i].end - m[i].start;
    }
    if (out_len) *out_len = (size_t)total;
    if (total > out_cap) return fail("output buffer too small (%llu bytes needed)",
                                     (unsigned long long)total);
    if (!n_matches || !total) return 0;
    const size_t bytes = n_matches * sizeof(uint64_t);
    if (q->match_start.ensure(bytes) || q->match_end.ensure(bytes) ||
        q->match_off.ensure(((n_matches + 255) / 256 + 2) * sizeof(uint64_t)) || q->gather.ensure((size_t)total))
        return -1;
    HIP_TRY(hipMemcpy(q->match_start.p, st.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(q->match_end.p, en.data(), bytes, hipMemcpyHostToDevice));
    agh_gather_shape g;
    memset(&g, 0, sizeof(g));
    agh_launch_len_offsets((const uint64_t *)q->match_start.p, (const uint64_t *)q->match_end.p, 0, (uint32_t)n_matches, g,
                           (uint64_t *)q->match_off.p, nullptr);
    agh_launch_gather_records(q->staging.p, q->staged_len, (const uint64_t *)q->match_start.p, (const uint64_t *)q->match_end.p,
                              nullptr, (const uint64_t *)q->match_off.p, 0, (uint32_t)n_matches, g, 0, q->gather.p, nullptr,
                              nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, q->gather.p, (size_t)total, hipMemcpyDeviceToHost));
    return 0;
}

