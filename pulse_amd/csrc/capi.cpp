// ABI bookkeeping shared by every translation unit of libpulse_hip.so.
#include "common.h"

namespace pulse {
char* last_error_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace pulse

extern "C" {
int pulse_abi_version(void) { return PULSE_ABI_VERSION; }
const char* pulse_last_error(void) { return pulse::last_error_buf(); }
}
