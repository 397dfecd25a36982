// Recorded launch plans (round 6): the layer of the C ABI ABOVE single kernels.
//
// A training step of this library is ~1000 calls of the entry points of zs3hip.h, each made from Python through ctypes
// (10-30 us of host time per call: 29.7 ms per step against 43 ms of GPU time in fp32 storage, and the 2-byte mode's 28.7 ms step
// runs at the host's pace -- DESIGN.md section 7).  The calls of consecutive steps are identical up to a handful of scalars
// (learning rates, dropout seeds), so a step is recorded ONCE -- every launching entry point appends its argument block to the
// plan while it also executes -- and replayed from C: zs3_plan_replay walks the blocks and calls the same entry points with the
// same arguments on the same streams, ~2 us per launch and no interpreter in between.  This is not a hipGraph: nothing is
// captured or instantiated, the kernels are enqueued on the real streams exactly as the eager step enqueues them (the graph
// replay of this runtime serialises the weight-gradient side streams), cross-stream dependencies are the
// recorded zs3_stream_wait calls.
//
// The entry-point wrappers that do the recording are GENERATED from include/zs3hip.h (zs3_amd/build.py -> csrc/gen/
// plan_wrappers.hip): every function whose last parameter is `void* stream` gets a wrapper of its own name that pushes
// {arguments} to the recording plan, if any, and forwards to the implementation, which the kernel sources define under the name
// <entry>__impl (csrc/gen/plan_rename.h, force-included into them).
#pragma once
#include <cstddef>
#include <cstdint>

namespace zs3 {

struct ArgDesc {
  uint16_t offset;   // of the argument inside the entry point's argument block
  uint16_t bytes;    // sizeof the argument (inline host arrays: the whole array)
  char kind;         // 'p' device / opaque pointer, 'i' int, 'l' long, 'f' float, 'd' double, 'u' unsigned long long, 's' stream,
                     // 'h' host array copied into the block (zs3_sum_n's srcs, zs3_mmd_fwd's sigma, zs3_sgd_multi_g's group table)
};

struct FnDesc {
  const char* name;
  int (*call)(const void* block);   // calls <name>__impl with the block's arguments
  const ArgDesc* args;
  int nargs;
  uint32_t block_bytes;
};

extern const FnDesc plan_fns[];   // generated
extern const int plan_nfns;

struct Plan;
Plan* plan_recording();                                    // the plan that is recording, or nullptr (one at a time, process-wide)
void plan_push(Plan* plan, int fn, const void* block);     // append one call

}  // namespace zs3
