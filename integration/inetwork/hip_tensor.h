// hip_tensor.h -- what the reference's high-level code (clstmhl.h, clstmocrtrain.cc, clstmocr.cc) sees when it includes
// "tensor.h" in the drop-in build (integration/Makefile: ref_drop_in): the Eigen-free Tensor2 / TensorMap2 / Batch /
// Sequence stand-ins of integration/mock/clstm_types.h (same member names and memory layout as tensor.h:176-330,
// batches.h:12-148) plus an empty `namespace Eigen` for the drivers' `using namespace Eigen;`.
#pragma once
#include <math.h>
#include <stddef.h>
#include <string.h>
#define CLSTM_TENSOR_HOST_MEMORY 1   // Sequences of the adapter stay on the host (see clstm_types.h)
#include "clstm_types.h"
namespace Eigen {}
