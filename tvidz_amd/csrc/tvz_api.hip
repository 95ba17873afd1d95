// tvz_api.hip — version + thread-local error string of libtvz.so.
#include <algorithm>

#include "tvz_common.h"

namespace tvz {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace tvz

TVZ_EXPORT int tvz_version(void) { return TVZ_DIAGNOSTIC_BUILD ? -TVZ_VERSION : TVZ_VERSION; }
TVZ_EXPORT const char *tvz_last_error(void) { return tvz::err_buf(); }

// ------------------------------------------------------------------------------------------------
// Frame feeder I/O (replaces the pipe the reference reads at inspector/app.py:209-216):
// a whole micro-batch of luma planes goes from a file or a decoder pipe into the caller's (pinned)
// buffer in ONE call that holds no interpreter lock - the per-frame Python loop (readline + readinto
// + seek per frame, under the GIL with 16-64 reader threads) was 75-80 % of the driver's wall time.
#include <cerrno>
#include <sys/uio.h>
#include <unistd.h>

static int tvz_read_records_impl(int fd, int64_t file_offset, int64_t n_records, int64_t record_bytes,
                                 const void *magic, int64_t header_bytes, int64_t payload_bytes, void *dst,
                                 int64_t *n_done) {
    TVZ_REQUIRE(fd >= 0 && n_records >= 0 && payload_bytes > 0 && header_bytes >= 0 && header_bytes <= 64 &&
                    record_bytes >= header_bytes + payload_bytes && dst && n_done,
                "bad record layout");
    char hdr[64];
    int64_t done = 0;
    for (; done < n_records; ++done) {
        const int64_t at = file_offset + done * record_bytes;
        char *out = static_cast<char *>(dst) + done * payload_bytes;
        int64_t got = 0;
        const int64_t want = header_bytes + payload_bytes;
        while (got < want) {                               // header and plane in one positioned read
            iovec iov[2];
            int ni = 0;
            if (got < header_bytes) iov[ni++] = {hdr + got, (size_t)(header_bytes - got)};
            const int64_t pg = got > header_bytes ? got - header_bytes : 0;
            iov[ni++] = {out + pg, (size_t)(payload_bytes - pg)};
            const ssize_t r = preadv(fd, iov, ni, (off_t)(at + got));
            if (r < 0) {
                if (errno == EINTR) continue;
                return tvz::fail(TVZ_ERR_INVALID, "read of frame record %lld failed: %s", (long long)done, strerror(errno));
            }
            if (r == 0) break;                             // end of file
            got += r;
        }
        if (got < want) break;                             // a partial last record is not a frame
        if (magic && header_bytes && memcmp(hdr, magic, (size_t)header_bytes) != 0)
            return tvz::fail(TVZ_ERR_INVALID, "frame record %lld does not start with the expected header", (long long)done);
    }
    *n_done = done;
    return TVZ_OK;
}

static int tvz_read_stream_impl(int fd, int64_t n_records, int64_t payload_bytes, int64_t skip_bytes, void *dst,
                                int64_t *n_done) {
    TVZ_REQUIRE(fd >= 0 && n_records >= 0 && payload_bytes > 0 && skip_bytes >= 0 && dst && n_done, "bad record layout");
    static thread_local char sink[1 << 16];
    auto read_all = [&](char *p, int64_t n, bool discard) -> int64_t {   // bytes read (short only at end of stream)
        int64_t got = 0;
        while (got < n) {
            const size_t chunk = discard ? (size_t)std::min<int64_t>(n - got, (int64_t)sizeof sink) : (size_t)(n - got);
            const ssize_t r = read(fd, discard ? sink : p + got, chunk);
            if (r < 0) {
                if (errno == EINTR) continue;
                return -1;
            }
            if (r == 0) break;
            got += r;
        }
        return got;
    };
    int64_t done = 0;
    for (; done < n_records; ++done) {
        const int64_t a = read_all(static_cast<char *>(dst) + done * payload_bytes, payload_bytes, false);
        if (a < 0) return tvz::fail(TVZ_ERR_INVALID, "read from the decoder pipe failed: %s", strerror(errno));
        if (a < payload_bytes) break;
        if (skip_bytes) {
            const int64_t b = read_all(nullptr, skip_bytes, true);
            if (b < 0) return tvz::fail(TVZ_ERR_INVALID, "read from the decoder pipe failed: %s", strerror(errno));
            if (b < skip_bytes) { ++done; break; }         // the luma plane arrived whole: it is a frame
        }
    }
    *n_done = done;
    return TVZ_OK;
}

TVZ_EXPORT int tvz_read_records(int fd, int64_t file_offset, int64_t n_records, int64_t record_bytes,
                                const void *magic, int64_t header_bytes, int64_t payload_bytes, void *h_dst,
                                int64_t *n_done) {
    TVZ_GUARDED(tvz_read_records_impl(fd, file_offset, n_records, record_bytes, magic, header_bytes, payload_bytes, h_dst, n_done));
}

TVZ_EXPORT int tvz_read_stream(int fd, int64_t n_records, int64_t payload_bytes, int64_t skip_bytes, void *h_dst,
                               int64_t *n_done) {
    TVZ_GUARDED(tvz_read_stream_impl(fd, n_records, payload_bytes, skip_bytes, h_dst, n_done));
}
