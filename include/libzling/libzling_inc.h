// libzling_inc.h -- common includes of the drop-in C++ API (counterpart of the reference's
// src/libzling_inc.h:35-59; the MSVC<1600 msinttypes switch is dropped: Linux/ROCm only).
#ifndef LIBZLING_AMD_INC_H
#define LIBZLING_AMD_INC_H

#include <stdint.h>
#include <inttypes.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <stdexcept>

#endif
