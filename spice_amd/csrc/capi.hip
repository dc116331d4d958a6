// Version and error-text entry points of libspice_amd.so.
#include "common.h"

extern "C" int spice_abi_version(void) { return SPICE_ABI_VERSION; }

extern "C" const char *spice_error_string(int code) {
    switch (code) {
        case SPICE_OK:
            return "ok";
        case SPICE_ERR_ARG:
            return "invalid argument (null pointer, negative size or misaligned base)";
        case SPICE_ERR_UNSUPPORTED:
            return "shape or option not supported by the gfx950 kernels";
        case SPICE_ERR_WORKSPACE:
            return "workspace smaller than the *_workspace_bytes() requirement";
        case SPICE_ERR_BLAS:
            return "a hipBLASLt call failed";
        case SPICE_ERR_STATIC_LIMIT:
            return "input exceeds the fixed scratch of the entry point that allocates nothing "
                   "(spice_confusion_update_static: at most 32 classes and 2^40 pixels)";
        default:
            return hipGetErrorString(static_cast<hipError_t>(code));
    }
}
