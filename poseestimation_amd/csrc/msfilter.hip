// placeholder
#include "common.h"
int pd_launch_peaks_filtered_multi(pd_ctx* ctx, double thr) { (void)ctx; (void)thr; return PD_E_STATE; }
