#include "common.hpp"
extern "C" int cgc_abi_version(void) { return CGC_ABI_VERSION; }
