// Shared helpers for the libisg HIP sources (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/isg.h"

namespace isg {

// Thread-local message behind isg_last_error().
void set_error(const char* fmt, ...);

// Records a failed HIP call / launch and maps it to ISG_ELAUNCH.
int check_launch(const char* what);

}  // namespace isg

#define ISG_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            isg::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)
