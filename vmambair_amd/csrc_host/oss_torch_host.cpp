// oss_torch_host.cpp -- compiled torch boundary of the selective scan: torch.ops.vmambair_host.scan_fwd / scan_bwd.
//
// The reference's boundary for this path is a pybind11 C++ module whose two functions check their arguments, allocate the
// outputs, fill a parameter struct and launch on the current stream
// (Mamba/kernels/selective_scan/csrc/selective_scan/cus/selective_scan.cpp:157-239 fwd, :241-349 bwd).  This file is the same
// layer over OUR C ABI (include/vmambair_oss.h): TORCH_CHECKs in the reference's order and wording, outputs allocated by the
// callee, oss_scan_fwd_params / oss_scan_bwd_params filled from sizes and ELEMENT strides, launch on the current HIP stream
// of u's device, no host synchronisation.  vmambair_amd/ops/scan.py calls these operators when this library is built and
// keeps its ctypes twin of the same logic as the test path (VMAMBAIR_HOST=ctypes); both produce bit-identical results because
// both only marshal arguments for libvmambair_oss.so.
//
// Differences from the reference's functions (all documented in include/vmambair_oss.h / DESIGN.md section 1): x holds one
// state every oss_scan_chunk() steps; bwd needs no zero-filled outputs and returns dB / dC already cast; the omni extensions
// (rev_group_start, u_row_mod, dout_row_mod, a_log_form, dbc_into, dt_weight) are extra trailing arguments.
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <cstring>
#include <vector>

#include "vmambair_oss.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

oss_dtype io_of(const Tensor &t) {
    switch (t.scalar_type()) {
        case at::ScalarType::Float: return OSS_F32;
        case at::ScalarType::Half: return OSS_F16;
        case at::ScalarType::BFloat16: return OSS_BF16;
        default: TORCH_CHECK(false, "u must be float32, float16 or bfloat16");
    }
    return OSS_F32;
}

const void *ptr(const OptTensor &t) { return (t.has_value() && t->defined() && t->numel() > 0) ? t->data_ptr() : nullptr; }

void check_rc(int rc, const char *what) {
    if (rc == 0) return;
    switch (rc) {
        case OSS_ERR_NULL: TORCH_CHECK(false, what, ": OSS_ERR_NULL: a required pointer is NULL");
        case OSS_ERR_SHAPE: TORCH_CHECK(false, what, ": OSS_ERR_SHAPE: invalid batch/dim/seqlen/dstate/n_groups");
        case OSS_ERR_DSTATE: TORCH_CHECK(false, what, ": OSS_ERR_DSTATE: selective_scan only supports state dimension <= 256");
        case OSS_ERR_WORKSPACE: TORCH_CHECK(false, what, ": OSS_ERR_WORKSPACE: workspace missing or too small");
        default: TORCH_CHECK(false, what, ": HIP error ", rc);
    }
}

// second line of defence inside the operators: the library the dynamic linker really bound (first scan call only)
void abi_check_once() {
    static const bool ok = [] {
        TORCH_CHECK(oss_abi_version() == OSS_ABI_VERSION && oss_abi_struct_bytes(0) == sizeof(oss_scan_fwd_params) &&
                        oss_abi_struct_bytes(1) == sizeof(oss_scan_bwd_params),
                    "libvmambair_torch.so was compiled against another revision of include/vmambair_oss.h than libvmambair_oss.so "
                    "(ABI ", OSS_ABI_VERSION, " vs ", oss_abi_version(), "): rebuild both with __graft_entry__.build()");
        return true;
    }();
    (void)ok;
}

struct Dims { int64_t batch, dim, seqlen, dstate, n_groups; };

// cus/selective_scan.cpp:165-215 (same order of checks)
Dims common_checks(const Tensor &u, const Tensor &delta, const Tensor &A, const Tensor &B, const Tensor &C, const OptTensor &D,
                   const OptTensor &delta_bias, int64_t u_row_mod, const OptTensor &dt_weight) {
    const auto it = u.scalar_type();
    TORCH_CHECK(it == at::ScalarType::Float || it == at::ScalarType::Half || it == at::ScalarType::BFloat16,
                "u must be float32, float16 or bfloat16");
    TORCH_CHECK(A.scalar_type() == at::ScalarType::Float, "A must be float32");
    TORCH_CHECK(delta.scalar_type() == it && B.scalar_type() == it && C.scalar_type() == it, "delta, B, C must have u's dtype");
    TORCH_CHECK(u.is_cuda(), "u must be a CUDA/HIP tensor");
    TORCH_CHECK(delta.is_cuda(), "delta must be a CUDA/HIP tensor");
    TORCH_CHECK(A.is_cuda(), "A must be a CUDA/HIP tensor");
    TORCH_CHECK(B.is_cuda(), "B must be a CUDA/HIP tensor");
    TORCH_CHECK(C.is_cuda(), "C must be a CUDA/HIP tensor");
    TORCH_CHECK(u.dim() == 3, "u must be (batch, dim, seqlen)");
    const int64_t batch = u.size(0), seqlen = u.size(2);
    int64_t dim = u.size(1);
    if (u_row_mod) {   // omni form: directions k and k + K/2 share the rows of u
        TORCH_CHECK(dim == u_row_mod && A.dim() == 2 && A.size(0) % u_row_mod == 0, "u must be (batch, u_row_mod, seqlen)");
        dim = A.size(0);
    }
    TORCH_CHECK(A.dim() == 2 && A.size(0) == dim, "A must be (dim, dstate)");
    const int64_t dstate = A.size(1);
    TORCH_CHECK(B.dim() == 4 && C.dim() == 4, "B and C must be (batch, n_groups, dstate, seqlen)");
    const int64_t n_groups = B.size(1);
    TORCH_CHECK(n_groups > 0 && dim % n_groups == 0, "dims should be dividable by n_groups");
    TORCH_CHECK(dstate <= 256, "selective_scan only supports state dimension <= 256");
    const bool fused = dt_weight.has_value() && dt_weight->defined();
    if (!fused) {
        TORCH_CHECK(delta.dim() == 3 && delta.size(0) == batch && delta.size(1) == dim && delta.size(2) == seqlen,
                    "delta must have u's shape");
    } else {
        const Tensor &w = *dt_weight;
        TORCH_CHECK(w.scalar_type() == at::ScalarType::Float && w.is_cuda() && w.dim() == 2 && w.size(0) == dim && w.is_contiguous() &&
                        w.size(1) >= 1 && w.size(1) <= 8, "dt_weight must be a contiguous (dim, R <= 8) float tensor");
        TORCH_CHECK(delta.dim() == 4 && delta.size(0) == batch && delta.size(1) == n_groups && delta.size(2) >= w.size(1) &&
                        delta.size(3) == seqlen, "with dt_weight, delta must be the (batch, n_groups, >= R, seqlen) factor");
    }
    TORCH_CHECK(B.size(0) == batch && B.size(2) == dstate && B.size(3) == seqlen, "B has the wrong shape");
    TORCH_CHECK(C.size(0) == batch && C.size(1) == n_groups && C.size(2) == dstate && C.size(3) == seqlen, "C has the wrong shape");
    auto last_contig = [](const Tensor &t) { return t.stride(-1) == 1 || t.size(-1) == 1; };
    TORCH_CHECK(last_contig(u), "u must be contiguous in its last dimension");
    TORCH_CHECK(last_contig(delta), "delta must be contiguous in its last dimension");
    TORCH_CHECK(last_contig(B), "B must be contiguous in its last dimension");
    TORCH_CHECK(last_contig(C), "C must be contiguous in its last dimension");
    TORCH_CHECK(last_contig(A), "A must be contiguous in its last dimension");
    auto opt_check = [&](const OptTensor &t, const char *name) {
        if (!(t.has_value() && t->defined())) return;
        TORCH_CHECK(t->scalar_type() == at::ScalarType::Float, name, " must be float32");
        TORCH_CHECK(t->is_cuda(), name, " must be a CUDA/HIP tensor");
        TORCH_CHECK(t->dim() == 1 && t->size(0) == dim, name, " must be (dim,)");
        TORCH_CHECK(t->stride(-1) == 1 || t->size(-1) == 1, name, " must be contiguous");
        TORCH_CHECK(t->device() == u.device(), "all tensors must be on the same device");
    };
    opt_check(D, "D");
    opt_check(delta_bias, "delta_bias");
    TORCH_CHECK(delta.device() == u.device() && A.device() == u.device() && B.device() == u.device() && C.device() == u.device(),
                "all tensors must be on the same device");
    return {batch, dim, seqlen, dstate, n_groups};
}

void fill_fwd(oss_scan_fwd_params &P, const Tensor &u, const Tensor &delta, const Tensor &A, const Tensor &B, const Tensor &C,
              const OptTensor &D, const OptTensor &delta_bias, const Tensor *out, const OptTensor &x, const Dims &d, bool softplus,
              int64_t rev_group_start, int64_t u_row_mod, bool a_log_form, const OptTensor &dt_weight, const OptTensor &hs = std::nullopt) {
    std::memset(&P, 0, sizeof(P));
    P.batch = (int)d.batch; P.dim = (int)d.dim; P.seqlen = (int)d.seqlen; P.dstate = (int)d.dstate; P.n_groups = (int)d.n_groups;
    P.delta_softplus = softplus ? 1 : 0;
    P.rev_group_start = rev_group_start < 0 ? (int)d.n_groups : (int)rev_group_start;
    P.u_row_mod = (int)u_row_mod;
    P.a_log_form = a_log_form ? 1 : 0;
    P.u_batch_stride = u.stride(0); P.u_d_stride = u.stride(1);
    P.delta_batch_stride = delta.stride(0); P.delta_d_stride = delta.stride(1);
    if (dt_weight.has_value() && dt_weight->defined()) {
        P.dt_weight = dt_weight->data_ptr<float>();
        P.dt_rank = (int)dt_weight->size(1);
        P.dt_group_stride = delta.stride(1); P.dt_rank_stride = delta.stride(2);
    }
    if (out) { P.out_batch_stride = out->stride(0); P.out_d_stride = out->stride(1); P.out = out->data_ptr(); }
    P.A_d_stride = A.stride(0);
    P.B_batch_stride = B.stride(0); P.B_group_stride = B.stride(1); P.B_dstate_stride = B.stride(2);
    P.C_batch_stride = C.stride(0); P.C_group_stride = C.stride(1); P.C_dstate_stride = C.stride(2);
    P.u = u.data_ptr(); P.delta = delta.data_ptr(); P.A = A.data_ptr<float>(); P.B = B.data_ptr(); P.C = C.data_ptr();
    P.D = reinterpret_cast<const float *>(ptr(D));
    P.delta_bias = reinterpret_cast<const float *>(ptr(delta_bias));
    P.x = reinterpret_cast<float *>(const_cast<void *>(ptr(x)));
    P.hs = reinterpret_cast<float *>(const_cast<void *>(ptr(hs)));
}

// cus/selective_scan.cpp:157-239
// want_hs: also return the lane states (include/vmambair_oss.h: hs) as a third tensor, for scan_bwd's `hs` argument
std::vector<Tensor> scan_fwd(const Tensor &u, const Tensor &delta, const Tensor &A, const Tensor &B, const Tensor &C,
                             const OptTensor &D, const OptTensor &delta_bias, bool delta_softplus, int64_t rev_group_start,
                             int64_t u_row_mod, bool a_log_form, const OptTensor &dt_weight, bool want_hs, int64_t tune_variant,
                             int64_t tune_segments, int64_t tune_carry_split) {
    const Dims d = common_checks(u, delta, A, B, C, D, delta_bias, u_row_mod, dt_weight);
    const at::hip::OptionalHIPGuardMasqueradingAsCUDA guard(u.device());
    const int n_chunks = oss_scan_num_chunks((int)d.seqlen);
    const bool fused = dt_weight.has_value() && dt_weight->defined();
    Tensor out;
    if (!fused) {
        out = at::empty_like(delta);
        if (out.dim() > 0 && out.stride(-1) != 1 && out.size(-1) != 1) out = at::empty(delta.sizes(), delta.options());
    } else {
        out = at::empty({d.batch, d.dim, d.seqlen}, u.options());
    }
    Tensor x = at::empty({d.batch, d.dim, (int64_t)n_chunks, 2 * d.dstate}, u.options().dtype(at::kFloat));
    OptTensor hs;
    TORCH_CHECK(!want_hs || (oss_scan_features() & OSS_FEATURE_LANE_STATES),
                "want_hs: libvmambair_oss.so was built without the lane-state scan form (-DOSS_WITHOUT_LANE_STATES)");
    if (want_hs)
        hs = at::empty({(int64_t)oss_scan_lane_state_floats((int)d.batch, (int)d.dim, (int)d.seqlen, (int)d.dstate)},
                       u.options().dtype(at::kFloat));
    if (d.batch == 0 || d.seqlen == 0) {   // nothing to launch
        if (want_hs) return {out, x, *hs};
        return {out, x};
    }
    oss_scan_fwd_params P;
    fill_fwd(P, u, delta, A, B, C, D, delta_bias, &out, x, d, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight, hs);
    // per-call launch tuning (0 = heuristic; include/vmambair_oss.h: oss_scan_fwd_params.tune_*)
    P.tune_variant = (int)tune_variant; P.tune_segments = (int)tune_segments; P.tune_carry_split = (int)tune_carry_split;
    Tensor ws;   // scratch of the time-segmented launch (under-filled grids); a few hundred KB
    const size_t ws_bytes = oss_scan_fwd_workspace_bytes((int)d.batch, (int)d.dim, (int)d.seqlen, (int)d.dstate, (int)d.n_groups);
    if (ws_bytes) {
        ws = at::empty({(int64_t)((ws_bytes + 3) / 4)}, u.options().dtype(at::kFloat));
        P.workspace = ws.data_ptr();
        P.workspace_bytes = (size_t)ws.numel() * 4;
    }
    hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    abi_check_once();
    check_rc(oss_scan_fwd(&P, io_of(u), reinterpret_cast<oss_stream_t>(stream)), "oss_scan_fwd");
    if (want_hs) return {out, x, *hs};
    return {out, x};
}

// cus/selective_scan.cpp:241-349.  Returns [du, ddelta, dA, dB, dC, dD, ddelta_bias, ddt_weight]; absent ones are empty tensors
// (the reference's undefined at::Tensor, :323-326).  With dbc_into (mutated: dB / dC -- and, with dt_weight, the gradient of
// the dt factor -- are written into its rows) dB and dC come back EMPTY: an operator must not return views of an argument.
std::vector<Tensor> scan_bwd(const Tensor &u, const Tensor &delta, const Tensor &A, const Tensor &B, const Tensor &C,
                             const OptTensor &D, const OptTensor &delta_bias, const Tensor &dout, const OptTensor &x,
                             bool delta_softplus, int64_t rev_group_start, int64_t u_row_mod, int64_t dout_row_mod, bool a_log_form,
                             const OptTensor &dbc_into, const OptTensor &dt_weight, const OptTensor &hs, int64_t tune_variant,
                             int64_t tune_segments, int64_t tune_carry_split, int64_t tune_partials, const OptTensor &finish_dt_weight) {
    const Dims d = common_checks(u, delta, A, B, C, D, delta_bias, u_row_mod, dt_weight);
    TORCH_CHECK(dout.scalar_type() == u.scalar_type() && dout.is_cuda(), "dout must be a CUDA/HIP tensor of u's dtype");
    TORCH_CHECK(dout.dim() == 3 && dout.size(0) == d.batch && dout.size(1) == (dout_row_mod ? dout_row_mod : d.dim) &&
                    dout.size(2) == d.seqlen, "dout must have u's shape");
    TORCH_CHECK(dout.stride(-1) == 1 || dout.size(-1) == 1, "dout must be contiguous in its last dimension");
    const at::hip::OptionalHIPGuardMasqueradingAsCUDA guard(u.device());
    const int n_chunks = oss_scan_num_chunks((int)d.seqlen);
    const bool has_x = x.has_value() && x->defined();
    if (n_chunks > 1) TORCH_CHECK(has_x, "x is required when the sequence spans several chunks");
    if (has_x) {
        TORCH_CHECK(x->scalar_type() == at::ScalarType::Float && x->is_cuda() && x->is_contiguous(), "x must be a contiguous float32 tensor");
        TORCH_CHECK(x->dim() == 4 && x->size(0) == d.batch && x->size(1) == d.dim && x->size(2) == n_chunks && x->size(3) == 2 * d.dstate,
                    "x has the wrong shape");
    }
    const bool fused = dt_weight.has_value() && dt_weight->defined();
    const bool into = dbc_into.has_value() && dbc_into->defined();
    TORCH_CHECK(!fused || into, "dt_weight needs dbc_into (the gradient of x_dbl the kernel fills)");
    if (hs.has_value() && hs->defined())
        TORCH_CHECK(hs->scalar_type() == at::ScalarType::Float && hs->is_cuda() && hs->is_contiguous() &&
                        (size_t)hs->numel() == oss_scan_lane_state_floats((int)d.batch, (int)d.dim, (int)d.seqlen, (int)d.dstate),
                    "hs must be the lane-state tensor the forward call returned");
    const auto io = u.options();
    const auto f32 = u.options().dtype(at::kFloat);
    Tensor du = at::empty({d.batch, d.dim, d.seqlen}, io);
    Tensor ddelta = fused ? at::empty({0}, io) : at::empty({d.batch, d.dim, d.seqlen}, io);
    Tensor ddtw = fused ? at::empty({d.dim, dt_weight->size(1)}, f32) : at::empty({0}, f32);
    Tensor dA = at::empty({d.dim, d.dstate}, f32);
    Tensor dB, dC;
    void *dB_ptr, *dC_ptr;
    int64_t rows = 0;
    if (into) {
        const Tensor &t = *dbc_into;
        rows = t.dim() == 4 ? t.size(2) : 0;
        TORCH_CHECK(t.is_contiguous() && t.scalar_type() == u.scalar_type() && t.dim() == 4 && t.size(0) == d.batch &&
                        t.size(1) == d.n_groups && t.size(3) == d.seqlen && rows > 2 * d.dstate, "dbc_into has the wrong layout");
        const size_t es = t.element_size();
        dB_ptr = static_cast<char *>(t.data_ptr()) + (size_t)(rows - 2 * d.dstate) * d.seqlen * es;
        dC_ptr = static_cast<char *>(t.data_ptr()) + (size_t)(rows - d.dstate) * d.seqlen * es;
        dB = at::empty({0}, io);
        dC = at::empty({0}, io);
    } else {
        dB = at::empty({d.batch, d.n_groups, d.dstate, d.seqlen}, io);
        dC = at::empty({d.batch, d.n_groups, d.dstate, d.seqlen}, io);
        dB_ptr = dB.data_ptr();
        dC_ptr = dC.data_ptr();
    }
    const bool has_D = D.has_value() && D->defined(), has_bias = delta_bias.has_value() && delta_bias->defined();
    Tensor dD = has_D ? at::empty({d.dim}, f32) : at::empty({0}, f32);
    Tensor dbias = has_bias ? at::empty({d.dim}, f32) : at::empty({0}, f32);
    if (d.batch == 0 || d.seqlen == 0) {
        dA.zero_();
        if (has_D) dD.zero_();
        if (has_bias) dbias.zero_();
        if (fused) ddtw.zero_();
        return {du, ddelta, dA, dB, dC, dD, dbias, ddtw};
    }
    const size_t ws_bytes = oss_scan_bwd_workspace_bytes((int)d.batch, (int)d.dim, (int)d.seqlen, (int)d.dstate, (int)d.n_groups);
    Tensor ws = at::empty({(int64_t)((std::max<size_t>(ws_bytes, 16) + 3) / 4)}, f32);
    oss_scan_bwd_params P;
    std::memset(&P, 0, sizeof(P));
    fill_fwd(P.f, u, delta, A, B, C, D, delta_bias, nullptr, x, d, delta_softplus, rev_group_start, u_row_mod, a_log_form, dt_weight, hs);
    P.dout_batch_stride = dout.stride(0); P.dout_d_stride = dout.stride(1);
    P.du_batch_stride = du.stride(0); P.du_d_stride = du.stride(1);
    if (fused) {
        P.ddt = dbc_into->data_ptr();
        P.ddt_weight = ddtw.data_ptr<float>();
        P.ddt_batch_stride = dbc_into->stride(0); P.ddt_group_stride = dbc_into->stride(1); P.ddt_rank_stride = dbc_into->stride(2);
    } else {
        P.ddelta_batch_stride = ddelta.stride(0); P.ddelta_d_stride = ddelta.stride(1);
        P.ddelta = ddelta.data_ptr();
    }
    P.dout = dout.data_ptr(); P.du = du.data_ptr(); P.dA = dA.data_ptr<float>();
    P.dB = dB_ptr; P.dC = dC_ptr;
    P.dD = has_D ? dD.data_ptr<float>() : nullptr;
    P.ddelta_bias = has_bias ? dbias.data_ptr<float>() : nullptr;
    P.workspace = ws.data_ptr(); P.workspace_bytes = (size_t)ws.numel() * 4;
    P.dout_row_mod = (int)dout_row_mod;
    P.dBC_group_stride = into ? dbc_into->stride(1) : 0;
    P.tune_variant = (int)tune_variant; P.tune_segments = (int)tune_segments; P.f.tune_carry_split = (int)tune_carry_split;
    P.tune_partials = (int)tune_partials;
    if (finish_dt_weight.has_value() && finish_dt_weight->defined()) {   // the dt-factor gradient in the finishing launch
        const Tensor &w = *finish_dt_weight;
        TORCH_CHECK(!fused && into, "finish_dt_weight needs dbc_into and the materialised-delta form");
        TORCH_CHECK(w.scalar_type() == at::kFloat && w.is_cuda() && w.is_contiguous() && w.dim() == 2 && w.size(0) == d.dim,
                    "finish_dt_weight must be a contiguous (dim, R) float32 tensor");
        TORCH_CHECK(oss_scan_bwd_finish_dt_ok((int)d.seqlen, (int)w.size(1)), "finish_dt_weight: rank <= 8 and seqlen % 4 == 0");
        P.finish_dt_weight = w.data_ptr<float>();
        P.finish_dt_rank = (int)w.size(1);
        P.ddt = dbc_into->data_ptr();
        P.ddt_batch_stride = dbc_into->stride(0); P.ddt_group_stride = dbc_into->stride(1); P.ddt_rank_stride = dbc_into->stride(2);
    }
    hipStream_t stream = at::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream();
    abi_check_once();
    check_rc(oss_scan_bwd(&P, io_of(u), reinterpret_cast<oss_stream_t>(stream)), "oss_scan_bwd");
    return {du, ddelta, dA, dB, dC, dD, dbias, ddtw};
}

}  // namespace

// what THIS translation unit was compiled against (vmambair_amd/_host.py compares with the C-ABI library's own values before
// it routes a single call through here: a host library left over from an older include/vmambair_oss.h fills structs the
// kernels would misread)
extern "C" int vmambair_host_abi_version(void) { return OSS_ABI_VERSION; }
extern "C" size_t vmambair_host_struct_bytes(int which) {
    return which == 0 ? sizeof(oss_scan_fwd_params) : which == 1 ? sizeof(oss_scan_bwd_params) : 0;
}

TORCH_LIBRARY(vmambair_host, m) {
    m.def("scan_fwd(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, bool delta_softplus, "
          "int rev_group_start, int u_row_mod, bool a_log_form, Tensor? dt_weight, bool want_hs, int tune_variant=0, "
          "int tune_segments=0, int tune_carry_split=0) -> Tensor[]");
    m.def("scan_bwd(Tensor u, Tensor delta, Tensor A, Tensor B, Tensor C, Tensor? D, Tensor? delta_bias, Tensor dout, Tensor? x, "
          "bool delta_softplus, int rev_group_start, int u_row_mod, int dout_row_mod, bool a_log_form, Tensor(a!)? dbc_into, "
          "Tensor? dt_weight, Tensor? hs, int tune_variant=0, int tune_segments=0, int tune_carry_split=0, int tune_partials=0, Tensor? finish_dt_weight=None) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(vmambair_host, CUDA, m) {
    m.impl("scan_fwd", &scan_fwd);
    m.impl("scan_bwd", &scan_bwd);
}
