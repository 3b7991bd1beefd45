// libzling.h -- public entry points of the MI355X-native libzling replacement.
//
// Same two functions, same signatures and defaults as the reference (src/libzling.h:44-45):
//
//   int Encode(Inputter*, Outputter*, ActionHandler* = NULL, int level = 0);
//   int Decode(Inputter*, Outputter*, ActionHandler* = NULL);
//
// Return 0, or -1 when the inputter/outputter reports an error (src/libzling.cpp:290, 426).
// Decode throws std::runtime_error with the reference's messages on a corrupt stream
// (src/libzling.cpp:316, 327, 382, 392, 399, 407); both throw std::bad_alloc when working memory
// (here: HBM pools and pinned staging) cannot be obtained (src/libzling.cpp:121-127).
// Differences, all documented in INTEGRATION.md: a level outside 0..4 returns -1 instead of
// emitting a corrupt stream; a payload that would overrun the reference's obuf fails cleanly;
// std::runtime_error("zling: no gfx950 device") when no MI355X is visible -- there is no CPU path.
//
// Link with -lzling_amd (libzling_amd/libzling_amd.so), which sits on the C-ABI of zlng.h.
#ifndef LIBZLING_AMD_H
#define LIBZLING_AMD_H

#include "libzling_inc.h"
#include "libzling_utils.h"

namespace baidu {
namespace zling {

int Encode(Inputter* inputter, Outputter* outputter, ActionHandler* action_handler = NULL, int level = 0);
int Decode(Inputter* inputter, Outputter* outputter, ActionHandler* action_handler = NULL);

}  // namespace zling
}  // namespace baidu
#endif
