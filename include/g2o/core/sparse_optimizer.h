// g2o/core/sparse_optimizer.h -- include-path shim: sparse-gslam's sources include this path; the whole mirrored g2o
// surface lives in g2o/sgo_g2o_compat.h (see the header comment there).
#pragma once
#include "../sgo_g2o_compat.h"
