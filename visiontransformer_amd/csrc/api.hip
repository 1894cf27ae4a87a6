// Version and per-thread error text of libdswin_hip.so.
#include "common.hpp"

namespace dsw {
char *tls_error_buffer() {
    static thread_local char buf[512] = "ok";
    return buf;
}
}  // namespace dsw

extern "C" int dswin_version(void) { return 100; }

extern "C" const char *dswin_last_error_string(void) { return dsw::tls_error_buffer(); }
