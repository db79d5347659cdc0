// Error transport of libk2hip (no HIP dependency, so the pure-host units -- the .k2w parser, the token -> text stage --
// also build with a plain C++ compiler under AddressSanitizer / UBSan: `make -C csrc san`).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "../../include/k2hip.h"

namespace k2hip {

// Internal exception; converted to (status, last_error) at the ABI boundary.
struct Error : std::runtime_error {
    int32_t code;
    Error(int32_t c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void failf(int32_t code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
[[noreturn]] inline void failf(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define K2_REQUIRE(cond, ...)                                        \
    do {                                                             \
        if (!(cond)) ::k2hip::failf(K2HIP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

}  // namespace k2hip
