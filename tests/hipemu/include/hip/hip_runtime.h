// TEST INFRASTRUCTURE ONLY: stands in for <hip/hip_runtime.h> when kernel sources are compiled by g++ for the host-side
// emulation (tests/hipemu/hipemu.h).  The product is compiled by hipcc against the real header.
#include "../../hipemu.h"
