// gemm_queue.h — launch-time description of the per-XCD tile queues of the persistent product kernels (gemm.hip, update256.hip)
#pragma once
#include <stdint.h>

namespace gpmi {

struct QueueArgs {
    unsigned long long base[8];  // per-XCD value of the queue word at launch
    int64_t start[9];            // chunk x = tiles [start[x], start[x+1])
    int use_queue;
    // GEMM_PHASE_LOCK: value of the XCD's "tiles finished" word (queue word + 1) at launch; a tile of round r = (t - start) /
    // nloc starts once every tile of the earlier rounds has finished its K loop, so the ~64 tiles in flight on an XCD step
    // through K together and share their 16 operand panels in L2 slab by slab
    unsigned long long done_base[8];
    // batched launch (split-K with separate outputs): work item t is tile t % tiles_per of batch t / tiles_per,
    // whose operands and output start strideA / strideB / strideC elements further on
    int64_t tiles_per;
    int64_t strideA, strideB, strideC;
};

}  // namespace gpmi
