// tvz_api.hip — version + thread-local error string of libtvz.so.
#include "tvz_common.h"

namespace tvz {
char *err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace tvz

TVZ_EXPORT int tvz_version(void) { return TVZ_VERSION; }
TVZ_EXPORT const char *tvz_last_error(void) { return tvz::err_buf(); }
