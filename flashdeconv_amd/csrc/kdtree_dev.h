// The restated cKDTree built on the device (kdtree_build_dev.cpp) in the layout the query kernel reads (kdtree_order.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include "fdx_internal.h"

namespace fdx {

struct KdDeviceTree {
    DevBuf meta;        // int4 per node: x = split dimension (-1: leaf), y / z = less / greater node, or first / past-last position of a leaf
    DevBuf split;       // double per node
    DevBuf idx;         // int per point: scipy's index array
    int n_nodes = 0;
    int levels = 0;
    bool overflow = false;   // a selection would have left introselect's partition loop (heap select), or the tree is deeper than the level cap: build on the host
    double mins[3] = {0, 0, 0}, maxes[3] = {0, 0, 0};   // of the whole set
};

// coords_dev: n x dim doubles (dim 1-3).  Queued on `st`; returns after the stream has drained (the node count and the flags are read).
int kd_build_device(const double* coords_dev, long long n, int dim, KdDeviceTree* out, hipStream_t st);

}  // namespace fdx
