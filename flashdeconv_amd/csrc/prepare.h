// Queued form of a shard's "prepare" step (capi_dev.cpp): X_sketch / XtX on the library's side stream, sketch -> H of the own
// rows and the partial ||Y_s||^2 on the caller's stream - nothing waited for.  Internal.
#pragma once
#include <memory>

#include "fdx_internal.h"
#include "sketch_plan.h"

namespace fdx {

struct PrepareJob {
    DevBuf dX, dXs, dG, dYs, dRowSq, dSum;         // dG: XtX (K, K); dSum: the shard's partial YtY (one double, valid behind the caller's stream)
    std::shared_ptr<SketchPlan> plan_y, plan_x;
    hipStream_t side = nullptr;                    // nullptr: everything on the caller's stream
    hipEvent_t evX = nullptr;                      // X side done (XtX in dG, and on the host when asked for)
    hipEvent_t evSum = nullptr;                    // dSum written (on the side stream when there is one: consumers on another stream wait for it)
    ~PrepareJob() { if (evX) (void)hipEventDestroy(evX); if (evSum) (void)hipEventDestroy(evSum); }
};

// Y_dev: (n, G) rows of this shard in solver order (row_map_dev: optional gather).  XtX_host: pinned or pageable, K*K doubles or
// NULL - filled behind job->evX (pageable: before this returns).  H_out_dev (K, ldh): columns [0, n) written.
int prepare_queue(PrepareJob* job, const void* Y_dev, int y_dtype, long long n, int G, long long ldy, const int* row_map_dev,
                  const double* X, int K, const int* bucket, const double* weight_y, const double* weight_x, int d, int mode_y_in,
                  int mode_x, double* H_out_dev, long long ldh, double* XtX_host, hipStream_t st, const double* X_dev = nullptr);

}  // namespace fdx
