// placeholder until the MFMA tower lands
#include "azr_internal.hpp"
namespace azr {
int net_bf16_alloc(azr_engine* h) { h->err = "bf16 net not built yet"; return AZR_E_STATE; }
void net_bf16_free(azr_engine*) {}
int net_bf16_upload(azr_engine*) { return AZR_OK; }
int net_bf16_forward(azr_engine* h, const uint8_t*, int, int, float*, float*) { h->err = "bf16 net not built yet"; return AZR_E_STATE; }
}
