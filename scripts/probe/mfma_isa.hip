// ISA probe: compiles lstm_mfma.h alone (seconds) for register / schedule inspection
#include <type_traits>
#include "../../clstm_amd/csrc/lstm_mfma.h"
template __global__ void clstm::lstm_fwd_mfma_kernel<100, 48>(clstm::LstmMfmaArgs);
