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

// Extension (no counterpart in the reference; callers that do not use it compile and behave as before): an ActionHandler that
// ALSO inherits this tag promises that its OnProcess never reads from the Inputter while decoding.  Decode() then need not keep
// the reference's exact pull order (src/libzling.cpp:306-336: nothing is read ahead of the block handed to OnProcess -- the
// Adler32 handler of demo/zling.cpp:124-132 relies on it) and reads ahead / decodes several blocks per GPU call.  A per-call
// trait of the handler object; the environment variable ZLNG_DECODE_READAHEAD=0|1 overrides it for a whole process.
struct DecodeReadAhead {};

// Extension, per call like the above: an ActionHandler that ALSO inherits this says where ONE Encode() call runs and how.
//   devices / ndevices   the HIP devices the stream is spread over, contiguous block ranges per batch (an index may repeat: several
//                        contexts on one GPU); overrides ZLNG_DEVICES / ZLNG_DEVICE for this call.  NULL / 0: the environment decides.
//   host_rank_contexts   k > 0: the k longest literal-rank chains of every range are walked by host threads while the device walks the
//                        others (zlng_group_set_host_rank_contexts; DESIGN.md section 7: the mode that pays on several GPUs); 0: all on
//                        the device (the default); -1: leave it to ZLNG_HOST_RANK_CONTEXTS.
// The bytes produced are the same in every setting.
struct EncodePlacement {
    const int* devices;
    int ndevices;
    int host_rank_contexts;
    EncodePlacement() : devices(NULL), ndevices(0), host_rank_contexts(-1) {}
};

int Encode(Inputter* inputter, Outputter* outputter, ActionHandler* action_handler = NULL, int level = 0);
int Decode(Inputter* inputter, Outputter* outputter, ActionHandler* action_handler = NULL);

}  // namespace zling
}  // namespace baidu
#endif
