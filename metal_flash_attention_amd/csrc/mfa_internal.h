// mfa_internal.h -- shared by the host translation units (not part of the ABI).
#pragma once
#include <string>

#include "../../include/mfa.h"

namespace mfa {
// records `message` as this thread's last error and returns `status`
mfa_status fail(mfa_status status, const std::string &message);
} // namespace mfa
