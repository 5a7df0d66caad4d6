// placeholder, replaced below
#include "vit_internal.h"
bool vit_pk_supported(uint32_t) { return false; }
hipError_t vit_launch_pk(const uint8_t*, uint8_t*, const vit_frame_desc*, uint32_t, uint32_t, int64_t, hipStream_t) {
    return hipErrorNotSupported;
}
