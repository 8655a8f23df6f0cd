#include "az_common.h"

extern "C" const char *az_strerror(int code) {
    switch (code) {
        case AZ_OK: return "AZ_OK";
        case AZ_EINVAL: return "AZ_EINVAL";
        case AZ_ENULL: return "AZ_ENULL";
        case AZ_ELAUNCH: return "AZ_ELAUNCH";
        case AZ_EUNSUPPORTED: return "AZ_EUNSUPPORTED";
        case AZ_EWORKSPACE: return "AZ_EWORKSPACE";
        default: return "AZ_E?";
    }
}

extern "C" int az_abi_version(void) { return 5; }
