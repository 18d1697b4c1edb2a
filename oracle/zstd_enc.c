/* placeholder until the level-3 encoder restatement lands (staged last) */
#include "oracle.h"
#include "../include/aircompressor_hip.h"
int64_t orc_zstd_compress(const uint8_t* in, int64_t in_len, uint8_t* out, int64_t out_cap)
{
    (void)in; (void)in_len; (void)out; (void)out_cap;
    return ACHIP_STATUS(ACHIP_CLASS_INVALID_ARGUMENT, ACHIP_D_UNSUPPORTED);
}
