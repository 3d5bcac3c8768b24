// Optional per-kernel-class timing with HIP events on the launch stream (used by bench.py to report
// roofline.achieved live).  Disabled by default: launchers pay one branch.
#pragma once
#include "common.h"

namespace lavie {

enum KernelClass {
    KC_CONV3X3 = 0,     // igemm, gathered A operand (3x3 convs incl. fused shortcut / concat / up / down sampling)
    KC_LINEAR = 1,      // igemm, plain A rows (Linear, 1x1 conv, GEGLU)
    KC_ATTENTION = 2,   // fused spatial / text attention core
    KC_TEMPORAL = 3,    // temporal attention core
    KC_GROUPNORM = 4,
    KC_LAYERNORM = 5,
    KC_OTHER = 6,
    KC_CONV_PATCH = 7,  // igemm_patch_kernel alone (a subset of KC_CONV3X3: the scope brackets exactly that kernel launch)
    KC_FUSED_TEMPORAL = 8,   // rowfuse.hip: norm_temp + q|k|v + temporal attention + to_out + residual in one kernel
    KC_FUSED_FF = 9,         // rowfuse.hip: norm3 + GEGLU feed-forward + residual in one kernel
    KC_FUSED_CROSS = 10,     // rowfuse_cross.hip: attn1.to_out + residual + norm2 + text cross-attention + to_out + residual in one kernel
    KC_COUNT = 11
};

struct ProfileScope {
    // kernel_events = false: the scope brackets everything enqueued during its lifetime with two hipEventRecord calls
    // (~11 us of stream time per scope).  kernel_events = true: nothing is recorded here; the launcher passes start() /
    // stop() to hipExtLaunchKernelGGL, which stamps them with that ONE kernel's own begin / end (no extra packets).
    ProfileScope(int cls, hipStream_t s, double flops, double bytes, bool kernel_events = false);
    ~ProfileScope();
    bool active() const { return slot >= 0; }
    hipEvent_t start() const;
    hipEvent_t stop() const;
    int slot;
    hipStream_t stream;
    bool kernel_events;
};

// mask: bit i enables class i.  Returns 0.
int profile_begin(unsigned mask, int max_events);
// Synchronises `stream`, folds the recorded events; fills per class: launches, total ms, algorithmic flops, bytes.
int profile_end(hipStream_t stream, long long* launches, double* ms, double* flops, double* bytes);
bool profile_enabled(int cls);

}  // namespace lavie
