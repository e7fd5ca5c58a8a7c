// oss_host.h -- host-side declarations shared by the kernel translation units and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/vmambair_oss.h"

namespace oss {
struct bf16_t;
struct f16_t;

template <typename T> int scan_fwd_dispatch(const oss_scan_fwd_params &p, int variant, hipStream_t stream);
// brackets the MAIN backward kernel only (not the two finishing kernels) for oss_prof_*
struct LaunchTimer {
    virtual void begin(hipStream_t) = 0;
    virtual void end(hipStream_t) = 0;
    virtual ~LaunchTimer() = default;
};
template <typename T>
int scan_bwd_dispatch(const oss_scan_bwd_params &p, int variant, hipStream_t stream, LaunchTimer *timer);

// number of row tiles the backward splits a group into for `variant` (workspace sizing)
int scan_bwd_rows_per_wg(int variant);
int scan_bwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups);
int scan_fwd_pick_variant(int batch, int dim, int seqlen, int dstate, int n_groups, int elem_bytes);
}  // namespace oss
